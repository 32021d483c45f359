#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests
LBZ_SLOTS=601 bash tests/gpu_var.sh wiki,tar,text 1112 default 2>&1 | grep -v amdgpu | cut -c1-130
LBZ_LEVEL=1 LBZ_SEED=3 LBZ_SLOTS=1024 timeout 200 python tests/quickperf.py 2000 mixed 2>&1 | grep "MB/s"
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or corpora or repetitive or small or ragged or round_schedule or periodic or literal or seeded or levels" 2>&1 | tail -1
