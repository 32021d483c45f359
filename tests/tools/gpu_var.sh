#!/bin/bash
# usage: tests/tools/gpu_var.sh kinds slabs variant...   ("" = default library)
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests
KINDS=$1; SLABS=$2; shift; shift
for v in "$@"; do
  echo "== ${v}"
  if [ "$v" != "default" ]; then export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so; else unset LBZ_LIB; fi
  LBZ_SLOTS=${LBZ_SLOTS:-556} timeout 200 python tests/tools/quickperf.py $SLABS $KINDS 2>&1 | grep "MB/s"
done
