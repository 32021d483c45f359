#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests
timeout 500 python -m pytest tests -x -q -m gpu -s -k "splitter or reference_cli or periodic_corpus or full_size or c_host" 2>&1 | grep -v amdgpu.ids | tail -12
for w in 64 256; do LD_LIBRARY_PATH=/opt/rocm/lib timeout 120 bash tests/dropin_perf.sh 450 $w 2>&1 | grep -v amdgpu | tail -3; done
