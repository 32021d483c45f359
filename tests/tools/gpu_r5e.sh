#!/bin/bash
# round 5, fifth GPU call: a row's place counted among the keys of its own run (DEEP_NEAR) against the all-pairs count
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5e_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5e_pytest.log
for v in default allpairs near8; do
  if [ "$v" != "default" ]; then export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so; else unset LBZ_LIB; fi
  timeout 600 python tests/tools/sweep_r5.py 1112 wiki,tar,pysrc,realtar "LBZAMD_STREAMS=1;LBZAMD_STREAMS=3" 2>&1 | grep "MB/s" | sed "s/^/$v /"
done | tee gpurun_out/r5e_near.txt
