#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
for v in nomid mid_lb6 mid_lb5 default nomid; do
  if [ "$v" != "default" ]; then export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so; else unset LBZ_LIB; fi
  timeout 300 python tests/tools/sweep_r5.py 1112 wiki,tar,pysrc "LBZAMD_STREAMS=1;LBZAMD_STREAMS=3" 2>&1 | grep "MB/s" | sed "s/^/$v /"
done | tee gpurun_out/r5n_mid.txt
