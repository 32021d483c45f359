#!/bin/bash
# Ad-hoc: decoder variants (lbzip2_amd/csrc/variants/*.so) against the default build: quickdec.py on rand / mixed / wiki
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests
for lib in default $(ls lbzip2_amd/csrc/variants/*.so 2>/dev/null); do
  if [ "$lib" = default ]; then unset LBZ_LIB; name=default; else export LBZ_LIB=/root/repo/$lib; name=$(basename $lib .so); fi
  LBZ_DEC_CASES=${1:-rand:100000000,mixed:210000000,wiki:1000000000} timeout 300 python tests/tools/quickdec.py 2>&1 | grep -v amdgpu | grep "decoded\|strips" | sort | uniq | head -8 | sed "s/^/$name /" | cut -c1-230
done
