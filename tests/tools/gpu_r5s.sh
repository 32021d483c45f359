#!/bin/bash
# round 5: the long runs of the first text launches in a launch of their own (k_bwt_long): A/B on three inputs, one process
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout 420 python tests/tools/sweep_r5.py 1112 wiki,realtar,pysrc "LBZAMD_LONG_ROUNDS=0;LBZAMD_LONG_ROUNDS=1;LBZAMD_LONG_ROUNDS=2;LBZAMD_LONG_ROUNDS=0;LBZAMD_LONG_ROUNDS=2" > gpurun_out/r5s_long.txt 2>&1; echo "rc=$?"
cat gpurun_out/r5s_long.txt
