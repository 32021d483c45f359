#!/bin/bash
# Ad-hoc (round 6): file -> file through lbzamd_compress -f/-o on tmpfs, 1 and 10 GB of wiki text; arguments: chunk_slabs:pipelines ...
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo")
import bench
d = bench.gen_input("wiki", 1_000_000_000, 2)
open("/dev/shm/w1.bin", "wb").write(d)
with open("/dev/shm/w10.bin", "wb") as f:
    for _ in range(10): f.write(d)
PY
EXE=lbzip2_amd/host/lbzamd_compress
for cfg in "$@"; do
  IFS=: read -r c p <<< "$cfg"
  for f in w1 w10; do
    $EXE -9 -f /dev/shm/$f.bin -o /dev/shm/$f.bz2 -c $c -p $p -t 2>&1 | grep "file splitter" | sed "s/^/$f c=$c p=$p: /" | cut -c1-330
  done
done
md5sum /dev/shm/w1.bz2
rm -f /dev/shm/w1.bin /dev/shm/w10.bin /dev/shm/w1.bz2 /dev/shm/w10.bz2
