#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
T=/dev/shm
timeout 120 python - <<PY
import sys
sys.path.insert(0, "/root/repo")
import bench
d = bench.gen_input("wiki", 1_000_000_000, 2)
open("$T/w1.bin", "wb").write(d)
with open("$T/w10.bin", "wb") as f:
    for _ in range(10): f.write(d)
PY
EXE=lbzip2_amd/host/lbzamd_compress
{
for cfg in "256 2" "256 3" "256 4" "186 4" "371 2" "128 6"; do set -- $cfg
  for f in w1 w10; do timeout 90 $EXE -9 -f $T/$f.bin -o $T/$f.bz2 -c $1 -p $2 -t 2>&1 | grep "file splitter" | sed "s/^/$f c=$1 p=$2: /" | cut -c1-400; done
done
md5sum $T/w1.bz2
rm -f $T/w*.bin $T/w*.bz2
} 2>&1 | tee gpurun_out/r5r_filemode.txt
