#!/bin/bash
# round 5, seventh GPU call: variants of the text rounds on one box (one stream: per-kernel sums are the measure)
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
for v in default scalar allpairs near8 bigregs default; do
  if [ "$v" != "default" ]; then export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so; else unset LBZ_LIB; fi
  timeout 600 python tests/tools/sweep_r5.py 1112 wiki,tar,pysrc "LBZAMD_STREAMS=1;LBZAMD_STREAMS=3" 2>&1 | grep "MB/s" | sed "s/^/$v /"
done | tee gpurun_out/r5g_variants.txt
