#!/bin/bash
# Round-end evidence on the GPU box: rocprofv3 kernel stats + PMC passes of bench.py, then the bench line.
# usage: tests/run_profiles.sh <tag>     (writes gpurun_out/<tag>_*)
set -u
TAG=${1:-r01_c}
OUT=$PWD/gpurun_out
mkdir -p $OUT/prof_$TAG
export TMPDIR=/tmp
B="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu --no-isolated"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/stats -- $B > $OUT/${TAG}_rocprof_bench.json.log 2>&1 )
( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_$TAG/pmc_fetch -- $B > /dev/null 2>&1 )
( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_$TAG/pmc_write -- $B > /dev/null 2>&1 )
python tests/summarize_prof.py $OUT/prof_$TAG $OUT/$TAG 1112 3 "text -9"
cp $OUT/${TAG}_pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
timeout 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
ls $OUT | head -30
