#!/bin/bash
# GPU box: rocprofv3 kernel stats of one single-stream pass over <slabs> slabs of <kind> (inputs.py kinds, the real-file ones included)
#   usage: tests/tools/prof_kind.sh <tag> <slabs> <kind>   -> gpurun_out/<tag>_<kind>_s1_kernel_stats.csv
TAG=$1; SLABS=$2; KIND=$3
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$REPO:$REPO/tests:$REPO/tests/tools
D=/tmp/prof_${TAG}_$KIND; rm -rf $D
( cd /tmp && LBZAMD_STREAMS=1 timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python $REPO/tests/tools/sweep_r5.py $SLABS $KIND "LBZAMD_STREAMS=1" > $OUT/${TAG}_${KIND}_prof.log 2>&1 )
f=$(find $D -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-5 "$f" | head -24 > $OUT/${TAG}_${KIND}_s1_kernel_stats.csv && cat $OUT/${TAG}_${KIND}_s1_kernel_stats.csv
t=$(find $D -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 - "$t" <<'PY' | tee $OUT/${TAG}_${KIND}_s1_launches.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last pass of the process: launches of k_bwt_* in order, with their durations
names = [r["Kernel_Name"].split("(")[0] for r in rows]
last = len(names) - 1 - names[::-1].index("k_collect") if "k_collect" in names else 0
# one pass = several rounds; print the last round's chain
for r in rows[last:]:
    print("%-14s %8.3f ms" % (r["Kernel_Name"].split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
rm -rf $D
