#!/bin/bash
# round-2 GPU call A: where does the time go on byte-alphabet text (current kernels)?
cd /root/repo
mkdir -p gpurun_out
export PYTHONPATH=/root/repo:/root/repo/tests
timeout 900 python tests/quickperf.py 512 wiki,tar,text > gpurun_out/r02a_perf.log 2>&1
LBZ_SLOTS=556 timeout 600 python tests/quickperf.py 1112 wiki >> gpurun_out/r02a_perf.log 2>&1
LBZ_LEVEL=1 LBZ_SEED=3 LBZ_SLOTS=1024 timeout 600 python tests/quickperf.py 2000 mixed >> gpurun_out/r02a_perf.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "baseline_configs" > gpurun_out/r02a_pytest.log 2>&1
tail -5 gpurun_out/r02a_pytest.log
cat gpurun_out/r02a_perf.log
