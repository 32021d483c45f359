#!/bin/bash
# round 5: the input file mapped instead of read into page-locked buffers (LBZAMD_IO_NOMAP=1: as before), same box
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
with open("$T/w3.bin", "wb") as f:
    for _ in range(3): f.write(d)
PY
EXE=lbzip2_amd/host/lbzamd_compress
{
for rep in 1 2; do
for mode in map nomap; do
  if [ $mode = nomap ]; then export LBZAMD_IO_NOMAP=1; else unset LBZAMD_IO_NOMAP; fi
  for f in w1 w3; do
    timeout 60 $EXE -9 -f $T/$f.bin -o $T/$f.bz2 -t 2>&1 | grep "file splitter" | sed "s/^/$mode $f: /"
  done
done
done
unset LBZAMD_IO_NOMAP
md5sum $T/w1.bz2
for cfg in "128 3" "371 2" "556 1"; do set -- $cfg; timeout 60 $EXE -9 -f $T/w3.bin -o $T/w3.bz2 -c $1 -p $2 -t 2>&1 | grep "file splitter" | sed "s/^/map w3 c=$1 p=$2: /"; done
( time timeout 60 lbzip2_amd/host/lbzamd -k --report $T/w1.bin ) 2>&1 | grep -v "^$"; md5sum $T/w1.bin.bz2
rm -f $T/w*.bin $T/w*.bz2 $T/w1.bin.bz2
} 2>&1 | tee gpurun_out/r5o_filemode.txt
timeout 300 python -m pytest tests/test_cli.py tests/test_gpu_parity.py -m gpu -x -q -k "cli or splitter or c_side or n_devices" 2>&1 | tail -3
