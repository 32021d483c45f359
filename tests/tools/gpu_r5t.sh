#!/bin/bash
# round 5: per-kernel durations with and without k_bwt_long (text, 1112 slabs)
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
export TMPDIR=/tmp
for L in 0 2; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_long$L -o long$L -- python tests/tools/sweep_r5.py 1112 wiki "LBZAMD_LONG_ROUNDS=$L" > gpurun_out/r5t_long$L.log 2>&1; echo "rc=$?"
  f=$(find /tmp/prof_long$L -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/r5t_long${L}_kernel_stats.csv
  head -25 "$f"
done
