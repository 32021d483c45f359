#!/bin/bash
# round 5, third GPU call: does the early rank chain really run beside the text launches (per-launch trace with queue ids),
# hardware queues, the counter passes with the L2 (TCC) counters, the splitter/muxer after the ring / context overlap.
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
bash tests/tools/gpu_trace.sh r5c_pysrc pysrc 371 2>&1 | tail -75
echo "== hardware queues"
for hq in 4 8; do
  GPU_MAX_HW_QUEUES=$hq timeout 600 python tests/tools/sweep_r5.py 1112 pysrc,realtar "LBZAMD_STREAMS=1,LBZAMD_SPLIT_CHAIN=0;LBZAMD_STREAMS=1,LBZAMD_SPLIT_CHAIN=1;LBZAMD_STREAMS=3,LBZAMD_SPLIT_CHAIN=0;LBZAMD_STREAMS=3,LBZAMD_SPLIT_CHAIN=1" 2>&1 | grep "MB/s" | sed "s/^/hwq=$hq /"
done | tee gpurun_out/r5c_hwq.txt
echo "== counters"
bash tests/tools/run_pmc.sh r05_a 556 wiki 2>&1 | tail -30
T=/dev/shm
python - <<PY
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
for cfg in "256 2 4 2" "128 3 4 2" "64 4 4 2" "556 1 4 2"; do
  set -- $cfg
  for f in w1 w3; do
    $EXE -9 -f $T/$f.bin -o $T/$f.bz2 -c $1 -p $2 -R $3 -W $4 -t 2>&1 | grep "file splitter" | sed "s/^/$f c=$1 p=$2 R=$3 W=$4: /"
  done
done
md5sum $T/w1.bz2
LBZAMD_FAKE_DEVICES=8 $EXE -9 -f $T/w3.bin -o $T/w3.bz2 -c 64 -p 1 -g 8 -R 8 -W 4 -t 2>&1 | grep "file splitter"
rm -f $T/w1.bz2 $T/w3.bz2 $T/w1.bin $T/w3.bin
} 2>&1 | tee gpurun_out/r5c_filemode.txt
