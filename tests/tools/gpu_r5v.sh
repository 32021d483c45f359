#!/bin/bash
# round 5: rocprofv3 kernel stats of the bench command for the round's last commit (default streams and LBZAMD_STREAMS=1)
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH TMPDIR=/tmp
TAG=${1:-r05_y3}
OUT=$PWD/gpurun_out
mkdir -p $OUT/prof_$TAG $OUT/prof_${TAG}_s1
B="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu --no-isolated --no-host --no-verify --no-decode --no-seq --no-legs"
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/stats -- $B > $OUT/${TAG}_rocprof_bench.json.log 2>&1 ); echo "rc=$?"
( cd /tmp && LBZAMD_STREAMS=1 timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_s1/stats -- $B > $OUT/${TAG}_s1_rocprof_bench.json.log 2>&1 ); echo "rc=$?"
python tests/tools/summarize_prof.py $OUT/prof_$TAG $OUT/$TAG 1112 3 "wiki -9"
python tests/tools/summarize_prof.py $OUT/prof_${TAG}_s1 $OUT/${TAG}_s1 1112 3 "wiki -9"
echo "== default streams"; cat $OUT/${TAG}_kernel_stats.csv
echo "== LBZAMD_STREAMS=1"; cat $OUT/${TAG}_s1_kernel_stats.csv
rm -rf $OUT/prof_$TAG $OUT/prof_${TAG}_s1
