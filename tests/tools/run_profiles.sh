#!/bin/bash
# Round evidence on the GPU box: rocprofv3 kernel stats + PMC passes of bench.py (same command), summaries -> profiles/.
#   <tag>_s1_*   LBZAMD_STREAMS=1: rounds one after the other, nothing overlaps -- per-kernel times add up to the step,
#                FETCH_SIZE / WRITE_SIZE passes (separate runs, --pmc only) attribute the traffic kernel by kernel
#   <tag>_*      the default configuration (rounds on three streams): kernel stats of the bench command as the driver runs it
# usage: tests/tools/run_profiles.sh <tag> [bench args]     (writes gpurun_out/<tag>_*; copy to profiles/ and commit)
set -u
TAG=${1:-r03}; shift
ARGS="$*"
REPO=$PWD
OUT=$PWD/gpurun_out
mkdir -p $OUT/prof_$TAG $OUT/prof_${TAG}_s1
export TMPDIR=/tmp
KIND=wiki; for a in $ARGS; do case $a in text|rand|mixed|tar|wiki) KIND=$a;; esac; done
B="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu --no-isolated --no-host --no-verify --no-decode --no-seq --no-legs $ARGS"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/stats -- $B > $OUT/${TAG}_rocprof_bench.json.log 2>&1 )
( cd /tmp && LBZAMD_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_s1/stats -- $B > $OUT/${TAG}_s1_rocprof_bench.json.log 2>&1 )
( cd /tmp && LBZAMD_STREAMS=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_s1/pmc_fetch -- $B > /dev/null 2>&1 )
( cd /tmp && LBZAMD_STREAMS=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_s1/pmc_write -- $B > /dev/null 2>&1 )
# the inverse path: kernel stats of one decode of the same workload's stream (3 timed passes + the size of the context)
( cd /tmp && LBZ_DEC_CASES=$KIND:1000000000 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/dec -- python $REPO/tests/tools/quickdec.py > $OUT/${TAG}_decode.log 2>&1 )
f=$(find $OUT/prof_$TAG/dec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|k_d" "$f" > $OUT/${TAG}_decode_kernel_stats.csv
rm -rf $OUT/prof_$TAG/dec
python tests/tools/summarize_prof.py $OUT/prof_$TAG $OUT/$TAG 1112 3 "$KIND -9"
python tests/tools/summarize_prof.py $OUT/prof_${TAG}_s1 $OUT/${TAG}_s1 1112 3 "$KIND -9"
ls $OUT | grep $TAG | head -30
echo "== default streams"; cat $OUT/${TAG}_kernel_stats.csv
echo "== LBZAMD_STREAMS=1"; cat $OUT/${TAG}_s1_kernel_stats.csv; cat $OUT/${TAG}_s1_pmc_traffic.json
tail -c 1500 $OUT/${TAG}_s1_rocprof_bench.json.log
