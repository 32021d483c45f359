#!/bin/bash
# Ad-hoc: file -> file on tmpfs with and without the reader's MADV_POPULATE_READ of the mapped chunks (lbzamd_io.c), 10^9 and 3*10^9 bytes
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo")
import bench
d = bench.gen_input("wiki", 1_000_000_000, 2)
open("/dev/shm/w1.bin", "wb").write(d)
with open("/dev/shm/w3.bin", "wb") as f:
    for _ in range(3): f.write(d)
PY
EXE=lbzip2_amd/host/lbzamd_compress
for rep in 1 2 3 4; do
  for mode in populate nopopulate; do
    for f in w1 w3; do
      if [ $mode = nopopulate ]; then export LBZAMD_IO_NOPOPULATE=1; else unset LBZAMD_IO_NOPOPULATE; fi
      s=$(date +%s%N)
      $EXE -9 -f /dev/shm/$f.bin -o /dev/shm/$f.bz2 -t 2>&1 | grep "file splitter" | cut -c1-200 | sed "s/^/$mode $f: /"
      e=$(date +%s%N); echo "$mode $f: process $(( (e - s) / 1000000 )) ms"
    done
  done
done
unset LBZAMD_IO_NOPOPULATE
md5sum /dev/shm/w1.bz2
rm -f /dev/shm/w1.bin /dev/shm/w3.bin /dev/shm/w1.bz2 /dev/shm/w3.bz2
