#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests
LBZ_SLOTS=601 bash tests/gpu_var.sh wiki,tar,text 1112 default part2 2>&1 | grep -v amdgpu | cut -c1-130
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or corpora or repetitive or small or ragged or round_schedule or periodic or literal or seeded or levels" 2>&1 | tail -1
