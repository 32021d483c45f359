#!/bin/bash
# round 5, second GPU call: the rank rounds of the early hand-overs on a stream of their own (LBZAMD_SPLIT_CHAIN), the
# hand-over thresholds, per-block tie statistics of real sources, and the parity suite on the new launch order.
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5b_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5b_pytest.log
S1="LBZAMD_STREAMS=1,LBZAMD_SPLIT_CHAIN=0;LBZAMD_STREAMS=1,LBZAMD_SPLIT_CHAIN=1"
S3="LBZAMD_STREAMS=3,LBZAMD_SPLIT_CHAIN=0;LBZAMD_STREAMS=3,LBZAMD_SPLIT_CHAIN=1;LBZAMD_STREAMS=3,LBZAMD_HANDOVER0=800;LBZAMD_STREAMS=3,LBZAMD_HANDOVER0=850;LBZAMD_STREAMS=3,LBZAMD_HANDOVER0=900;LBZAMD_STREAMS=3,LBZAMD_HANDOVER1=300;LBZAMD_STREAMS=3,LBZAMD_HANDOVER1=500;LBZAMD_STREAMS=4,LBZ_SLOTS=278;LBZAMD_STREAMS=2,LBZ_SLOTS=556"
timeout 900 python tests/tools/sweep_r5.py 1112 wiki,tar,pysrc,realtar "$S1;$S3" 2>&1 | grep -v "^$" | tee gpurun_out/r5b_sweep.txt
for k in pysrc realtar; do SHOW=64 timeout 300 python tests/tools/diag_rows.py 64 $k 2>&1 | tail -70 > gpurun_out/r5b_rows_$k.txt; tail -3 gpurun_out/r5b_rows_$k.txt; done
