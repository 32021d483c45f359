#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests
for i in 1 2 3; do timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "corpora or periodic or fuzz" 2>&1 | tail -1; done
bash tests/gpu_var.sh wiki,tar,text 1112 default nopf 2>&1 | grep -v amdgpu
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "repetitive or small or ragged or round_schedule or literal or seeded" 2>&1 | tail -1
