#!/bin/bash
# usage: tests/tools/gpu_decvar.sh cases variant...
cd /root/repo
CASES=$1; shift
for v in "$@"; do
  echo "== ${v}"
  if [ "$v" != "default" ]; then export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so; else unset LBZ_LIB; fi
  LBZ_DEC_CASES=$CASES timeout 300 python tests/tools/quickdec.py 2>&1 | grep "MB/s"
done
