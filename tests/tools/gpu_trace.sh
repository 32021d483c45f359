#!/bin/bash
# usage: tests/tools/gpu_trace.sh tag kind [slabs] : per-launch timeline of one round of `slabs` blocks (one stream)
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests
TAG=$1; KIND=$2; SLABS=${3:-371}
OUT=/root/repo/gpurun_out/tr_$TAG
rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && LBZAMD_STREAMS=1 LBZ_SLOTS=$SLABS LBZ_ITERS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python /root/repo/tests/tools/quickperf.py $SLABS $KIND > $OUT/run.log 2>&1 )
grep "MB/s" $OUT/run.log
python tests/tools/trace_rounds.py $OUT > gpurun_out/tr_$TAG.txt 2>&1
cat gpurun_out/tr_$TAG.txt
find $OUT -name "*.csv" -size +2M -delete
