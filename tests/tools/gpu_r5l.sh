#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
for k in wiki tar; do LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/ticks.so timeout 200 python tests/tools/diag_deep.py 371 $k 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r5l_deep_ticks.txt
