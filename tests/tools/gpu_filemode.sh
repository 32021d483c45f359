#!/bin/bash
# Ad-hoc: the C file splitter/muxer (lbzamd_compress -f/-o) on a 1 GB and a 3 GB file of the enwik-like text (page cache),
# a few chunk sizes / pipeline counts; the 3 GB - 1 GB difference is the steady-state rate (context creation cancels).
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo")
import bench
d = bench.gen_input("wiki", 1_000_000_000, 2)
open("/tmp/w1.bin", "wb").write(d)
with open("/tmp/w3.bin", "wb") as f:
    for _ in range(3): f.write(d)
PY
EXE=lbzip2_amd/host/lbzamd_compress
for cfg in "256 2" "371 3" "556 2" "186 4" "128 6"; do
  set -- $cfg
  for f in w1 w3; do
    $EXE -9 -f /tmp/$f.bin -o /tmp/$f.bz2 -c $1 -p $2 -t 2>&1 | grep "file splitter" | sed "s/^/$f c=$1 p=$2: /"
  done
done
md5sum /tmp/w1.bz2
# the inverse path through the same driver: whole file in (pageable), every block decoded, whole file out (malloc'ed)
for f in w1 w3; do
  $EXE -d -f /tmp/$f.bz2 -o /tmp/$f.out -t 2>&1 | grep "decode:" | sed "s/^/$f -d: /"
done
cmp /tmp/w1.out /tmp/w1.bin && echo "w1 round trip ok"
