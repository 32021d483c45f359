#!/bin/bash
# round 5, first GPU call: the new tests (C4 at 10 GB, the command), the splitter/muxer with several readers / writers /
# devices on tmpfs, and this box's baseline for the sorter work that follows.  usage: tests/tools/gpu_r5a.sh [tests-k-expr]
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
K=${1:-"c4_ten or gpu_cli or native_library"}
timeout 900 python -m pytest tests -m gpu -x -q -k "$K" --durations=8 > gpurun_out/r5a_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r5a_pytest.log
T=/dev/shm; df -k $T | tail -1; free -g | head -2; nproc
avail=$(df -k $T | tail -1 | awk '{print $4}'); [ "$avail" -lt 6000000 ] && T=/tmp
echo "files on $T"
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
for cfg in "256 2 1 1" "256 2 4 2" "256 2 8 2" "371 3 4 2" "186 4 4 2" "128 4 6 3"; do
  set -- $cfg
  for f in w1 w3; do
    $EXE -9 -f $T/$f.bin -o $T/$f.bz2 -c $1 -p $2 -R $3 -W $4 -t 2>&1 | grep "file splitter" | sed "s/^/$f c=$1 p=$2 R=$3 W=$4: /"
  done
done
md5sum $T/w1.bz2
echo "== 8 logical devices on this GPU"
LBZAMD_FAKE_DEVICES=8 $EXE -9 -f $T/w3.bin -o $T/w3.bz2 -c 64 -p 1 -g 8 -R 8 -W 4 -t 2>&1 | grep "file splitter"
LBZAMD_FAKE_DEVICES=8 $EXE -9 -f $T/w3.bin -o $T/w3.bz2 -c 128 -p 1 -g 8 -R 4 -W 2 -t 2>&1 | grep "file splitter"
echo "== pipes (one reader, one writer, in order)"
cat $T/w3.bin | $EXE -9 -f - -o - -c 256 -p 2 -t 2>&1 >/dev/null | grep "file splitter"
echo "== the command"
cp $T/w3.bin $T/w3c.bin; ( time lbzip2_amd/host/lbzamd -k --report $T/w3c.bin ) 2>&1 | grep -v "^$"; md5sum $T/w3c.bin.bz2 $T/w3.bz2
( time lbzip2_amd/host/lbzamd -t --report $T/w3c.bin.bz2 ) 2>&1 | grep -v "^$"
rm -f $T/w3c.bin $T/w3c.bin.bz2 $T/w1.bz2 $T/w3.bz2 $T/w1.bin $T/w3.bin
} 2>&1 | tee gpurun_out/r5a_filemode.txt
echo "== baseline of this box"
for st in 1 3; do
  LBZAMD_STREAMS=$st LBZ_SLOTS=371 timeout 300 python tests/tools/quickperf.py 1112 wiki,tar,pysrc 2>&1 | grep "MB/s" | sed "s/^/streams=$st /"
done 2>&1 | tee gpurun_out/r5a_baseline.txt
