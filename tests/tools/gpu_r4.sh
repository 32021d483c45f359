#!/bin/bash
# round 4: GPU suite, then new sorter against the round-3 library (variants/r3.so) on the same box
# usage: tests/tools/gpu_r4.sh [tests|notests] kinds slabs variant...
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests
mkdir -p gpurun_out
if [ "$1" = "tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4_pytest.log
fi
shift
KINDS=$1; SLABS=$2; shift; shift
for st in 1 3; do
for v in "$@"; do
  echo "== ${v} streams=$st"
  if [ "$v" != "default" ]; then export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so; else unset LBZ_LIB; fi
  LBZAMD_STREAMS=$st LBZ_SLOTS=${LBZ_SLOTS:-371} timeout 300 python tests/tools/quickperf.py $SLABS $KINDS 2>&1 | grep "MB/s"
done
done
