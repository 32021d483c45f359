#!/bin/bash
# round 5: pieces of long runs ordered in one pass (deep_mid_run) against the counting split, same box; parity first
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5m_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5m_pytest.log
for v in default nomid default; do
  if [ "$v" != "default" ]; then export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so; else unset LBZ_LIB; fi
  timeout 300 python tests/tools/sweep_r5.py 1112 wiki,tar,pysrc,mixed "LBZAMD_STREAMS=1;LBZAMD_STREAMS=3" 2>&1 | grep "MB/s" | sed "s/^/$v /"
done | tee gpurun_out/r5m_mid.txt
for k in wiki; do LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/ticks.so timeout 200 python tests/tools/diag_deep.py 371 $k 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r5m_deep_ticks.txt
