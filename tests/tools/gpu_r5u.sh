#!/bin/bash
# round 5: k_bwt_long with one-wave workgroups: A/B on three inputs; 16 instead of 8 of them per segment; per-kernel durations
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tests/tools/sweep_r5.py 1112 wiki,realtar,pysrc "LBZAMD_LONG_ROUNDS=0;LBZAMD_LONG_ROUNDS=2;LBZAMD_LONG_ROUNDS=1;LBZAMD_LONG_ROUNDS=0;LBZAMD_LONG_ROUNDS=2" > gpurun_out/r5u_long.txt 2>&1; echo "rc=$?"
cat gpurun_out/r5u_long.txt
LBZ_LIB=/root/repo/lbzip2_amd/csrc/alt_sub16/liblbzamd.so timeout 200 python tests/tools/sweep_r5.py 1112 wiki "LBZAMD_LONG_ROUNDS=2;LBZAMD_LONG_ROUNDS=0;LBZAMD_LONG_ROUNDS=2" > gpurun_out/r5u_sub16.txt 2>&1; echo "rc=$?"
cat gpurun_out/r5u_sub16.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_long2 -o long2 -- python tests/tools/sweep_r5.py 1112 wiki "LBZAMD_LONG_ROUNDS=2" > gpurun_out/r5u_trace.log 2>&1; echo "rc=$?"
f=$(find /tmp/prof_long2 -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r5u_long2_kernel_stats.csv
head -8 "$f" | cut -c1-30,300-420
