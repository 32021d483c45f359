#!/bin/bash
# decoder timing of the default library and of variants: tests/tools/gpu_dec.sh [variant ...]
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests
CASES=${LBZ_DEC_CASES:-wiki:1000000000,rand:100000000,mixed:210000000,tar:175000000}
for v in default "$@"; do
  echo "== $v"
  if [ $v = default ]; then unset LBZ_LIB; else export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so; fi
  LBZAMD_DTIMES=1 LBZ_DEC_CASES=$CASES timeout 250 python tests/tools/quickdec.py 2>&1 | grep -v amdgpu.ids | uniq -w 40
done
