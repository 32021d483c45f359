#!/bin/bash
# round 5: confidence in the rebuilt text rounds -- repeated passes over the full-size fixtures (races between the segment
# workgroups of a block show as a rare mismatch) and the differential fuzzers with fresh seeds; every command under a timeout
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
{
timeout 300 python tests/tools/gpu_stress.py 6 2>&1 | grep -v amdgpu.ids
timeout 200 python tests/fuzz_gpu.py 5051 160 2>&1 | tail -3
timeout 200 python tests/fuzz_gpu.py 5052 16 big 2>&1 | tail -3
timeout 240 python tests/fuzz_corpora_gpu.py 5053 50 2>&1 | tail -3
timeout 200 python tests/fuzz_decode_gpu.py 5054 60 2>&1 | tail -3
} | tee gpurun_out/r5k_stress.txt
