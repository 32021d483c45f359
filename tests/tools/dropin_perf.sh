#!/bin/bash
# Ad-hoc: throughput of the drop-in work-unit interface driven from C (lbzamd_compress -w N -t -r 3),
# codec phase only (no process start, no file I/O); the batch interface beside it.
# usage: tests/tools/dropin_perf.sh [MB] [threads...]
cd /root/repo
MB=${1:-450}; shift
python - <<PY
import sys; sys.path.insert(0, "/root/repo")
import bench
open("/tmp/in.txt", "wb").write(bench.gen_input("wiki", $MB * 1000000, 2))
PY
for w in ${@:-16 64 256 512}; do
  ./lbzip2_amd/host/lbzamd_compress -9 -w $w -t -r 3 < /tmp/in.txt 2>&1 > /tmp/out_$w.bz2 | tail -2
done
./lbzip2_amd/host/lbzamd_compress -9 -t -r 3 < /tmp/in.txt 2>&1 > /tmp/out_b.bz2 | tail -2
cmp /tmp/out_b.bz2 /tmp/out_${!#}.bz2 && echo identical
