#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
for k in wiki tar; do LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/ticks.so timeout 200 python tests/tools/diag_deep.py 371 $k 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r5j_deep_ticks.txt
for v in default cg64 cg32; do
  if [ "$v" != "default" ]; then export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so; else unset LBZ_LIB; fi
  timeout 300 python tests/tools/sweep_r5.py 1112 wiki,pysrc "LBZAMD_STREAMS=1;LBZAMD_STREAMS=3" 2>&1 | grep "MB/s" | sed "s/^/$v /"
done | tee gpurun_out/r5j_cg.txt
