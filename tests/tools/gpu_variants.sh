#!/bin/bash
# Ad-hoc: every library under lbzip2_amd/csrc/variants/ against the default build, same box: sweep_r5.py <slabs> <kinds>, one and three streams
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests:/root/repo/tests/tools
SLABS=${1:-1112}; KINDS=${2:-wiki,realtar}
for lib in default $(ls lbzip2_amd/csrc/variants/*.so 2>/dev/null); do
  if [ "$lib" = default ]; then unset LBZ_LIB; name=default; else export LBZ_LIB=/root/repo/$lib; name=$(basename $lib .so); fi
  timeout 600 python tests/tools/sweep_r5.py $SLABS $KINDS "LBZAMD_STREAMS=1;LBZAMD_STREAMS=3" 2>&1 | grep -E "MB/s|rror" | sed "s/^/$name /"
done
