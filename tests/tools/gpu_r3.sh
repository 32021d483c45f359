#!/bin/bash
# Round-3 GPU session helper: suite, quick A/B of library variants, bench lines.  usage: tests/tools/gpu_r3.sh <tag> [steps...]
cd /root/repo
TAG=${1:-r03}; shift
export PYTHONPATH=/root/repo:/root/repo/tests
mkdir -p gpurun_out
for step in "$@"; do
  case $step in
    suite)
      timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; tail -5 gpurun_out/${TAG}_pytest.log;;
    smoke)
      timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1;;
    quick)
      for cfg in "2 601" "1 1112" "1 556" "2 1112"; do
        set -- $cfg
        echo "== default lib streams=$1 slots=$2"
        LBZAMD_STREAMS=$1 LBZ_SLOTS=$2 timeout 200 python tests/tools/quickperf.py 1112 wiki 2>&1 | grep -E "MB/s"
      done
      for v in ${VARIANTS:-}; do
        echo "== variant $v streams=2 slots=601"
        LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so LBZAMD_STREAMS=2 LBZ_SLOTS=601 timeout 200 python tests/tools/quickperf.py 1112 wiki 2>&1 | grep -E "MB/s"
      done
      echo "== 112 slabs (10^8 bytes)"
      LBZ_SLOTS=112 timeout 200 python tests/tools/quickperf.py 112 wiki 2>&1 | grep -E "MB/s"
      for v in ${VARIANTS:-}; do
        echo "== variant $v 112 slabs"
        LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so LBZ_SLOTS=112 timeout 200 python tests/tools/quickperf.py 112 wiki 2>&1 | grep -E "MB/s"
      done;;
    bench)
      timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 4000 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err;;
    bench8)
      timeout 300 python bench.py --bytes 100000000 --seed 1 --no-cpu > gpurun_out/${TAG}_bench_1e8.json 2>> gpurun_out/${TAG}_bench.err; head -c 1500 gpurun_out/${TAG}_bench_1e8.json; echo;;
    ab)   # A/B of library variants on the headline workload: VARIANTS="name ..." (default first), full quickperf lines for the default
      echo "== default"; LBZAMD_STREAMS=1 LBZ_SLOTS=1112 timeout 200 python tests/tools/quickperf.py 1112 wiki 2>&1 | grep -v "^$"
      for v in ${VARIANTS:-}; do
        echo "== variant $v (1 stream, 1112 slots)"
        LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so LBZAMD_STREAMS=1 LBZ_SLOTS=1112 timeout 200 python tests/tools/quickperf.py 1112 wiki 2>&1 | grep -E "MB/s|batch kernel"
        echo "== variant $v (2 streams, 601 slots)"
        LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so LBZAMD_STREAMS=2 LBZ_SLOTS=601 timeout 200 python tests/tools/quickperf.py 1112 wiki 2>&1 | grep -E "MB/s"
      done
      echo "== fix phases (default)"; LBZAMD_STREAMS=1 timeout 200 python tests/tools/diag_fix.py 2>&1 | tail -12;;
    sweep)   # two-stream headline rate of each variant, plus 3 streams for the first
      for v in ${VARIANTS:-}; do
        echo "== variant $v (2 streams, 601 slots)"
        LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so LBZAMD_STREAMS=2 LBZ_SLOTS=601 timeout 200 python tests/tools/quickperf.py 1112 wiki 2>&1 | grep -E "MB/s"
      done
      set -- ${VARIANTS:-}
      for cfg in "3 371" "2 556" "4 278"; do
        set -- $cfg
        echo "== first variant streams=$1 slots=$2"
        LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/${FIRSTV}.so LBZAMD_STREAMS=$1 LBZ_SLOTS=$2 timeout 200 python tests/tools/quickperf.py 1112 wiki 2>&1 | grep -E "MB/s"
      done;;
    sweep2)  # every variant at 2 streams / 601 slots and 3 streams / 371 slots, best of LBZ_ITERS passes
      for v in ${VARIANTS:-}; do
        for cfg in "2 601" "3 371"; do
          set -- $cfg
          echo -n "== $v streams=$1 slots=$2: "
          LBZ_ITERS=6 LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so LBZAMD_STREAMS=$1 LBZ_SLOTS=$2 timeout 200 python tests/tools/quickperf.py 1112 wiki 2>&1 | grep -E "MB/s" | cut -c1-150
        done
      done;;
    dropin)
      bash tests/tools/gpu_dropin.sh 2>&1 | grep -v "^$";;
    *) echo "unknown step $step";;
  esac
done
