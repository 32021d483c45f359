#!/bin/bash
# GPU box, one call: the steps named on the command line, each under its own timeout, logs under gpurun_out/<tag>_*.
#   usage: tests/tools/gpu_session.sh <tag> step [step ...]
#   steps: suite (pytest -m gpu + smoke) | bench (the driver's bench line) | sweep:<slabs>:<kinds>:<settings> (sweep_r5.py)
#          rows:<slabs>:<kind> (diag_rows.py) | small (small_rounds.py) | prof (run_profiles.sh) | pmc:<slabs>:<kind> (run_pmc.sh)
#          ab:<variant>:<slabs>:<kinds> (default library against a variant, same box)
#          wu (work-unit interface at 16/64/256 threads) | file (file -> file, hostpath_perf) | py:<script and args>
#          usepmc (the tag's PMC traffic becomes profiles/pmc_traffic.json for the bench steps that follow)
cd /root/repo
TAG=$1; shift
export PYTHONPATH=/root/repo:/root/repo/tests:/root/repo/tests/tools
mkdir -p gpurun_out
for step in "$@"; do
  IFS=: read -r what a b c <<< "$step"
  echo "=== $step"
  case $what in
    suite) timeout 900 python -m pytest tests -x -q -m gpu --durations=6 > gpurun_out/${TAG}_gpu_suite.log 2>&1; tail -12 gpurun_out/${TAG}_gpu_suite.log
           timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1 ;;
    bench) timeout 900 python bench.py $a $b $c > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 6000 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err ;;
    sweep) timeout 900 python tests/tools/sweep_r5.py $a $b "$c" 2>&1 | grep -E "MB/s|Error|error" | tee -a gpurun_out/${TAG}_sweep.txt ;;
    rows)  SHOW=${c:-24} timeout 600 python tests/tools/diag_rows.py $a $b 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_rows_$b.txt | tail -40 ;;
    small) timeout 600 python tests/tools/small_rounds.py 2>&1 | tee gpurun_out/${TAG}_small_rounds.txt | tail -30 ;;
    prof)  timeout 1500 bash tests/tools/run_profiles.sh $TAG 2>&1 | tail -60 ;;
    pmc)   timeout 1500 bash tests/tools/run_pmc.sh ${TAG}_pmc $a $b 2>&1 | tail -40 ;;
    wu)    # the work-unit interface driven by 16/64/256 pthreads (lbzamd_compress -w), 450 MB of wiki; wu:<lib> = with that library preloaded instead
           python - <<PY
import sys; sys.path.insert(0, "/root/repo")
import bench
open("/tmp/in.bin", "wb").write(bench.gen_input("wiki", 450 * 1000000, 2))
PY
           for w in 16 64 256; do
             ( [ -n "$a" ] && export LD_PRELOAD=/root/repo/lbzip2_amd/csrc/variants/$a.so; timeout 120 ./lbzip2_amd/host/lbzamd_compress -9 -w $w -t -r 3 < /tmp/in.bin 2>&1 > /tmp/out_$w.bz2 | grep -v amdgpu | tail -1 )
           done | sed "s/^/${a:-default} /" | tee -a gpurun_out/${TAG}_workunits.txt
           timeout 120 ./lbzip2_amd/host/lbzamd_compress -9 -t -r 2 < /tmp/in.bin 2>&1 > /tmp/out_b.bz2 | grep -v amdgpu | tail -1
           cmp /tmp/out_b.bz2 /tmp/out_256.bz2 && echo "work-unit stream identical to the batch call's" | tee -a gpurun_out/${TAG}_workunits.txt ;;
    file)  timeout 900 bash tests/tools/gpu_filemode.sh 2>&1 | tee gpurun_out/${TAG}_filemode.txt | tail -30 ;;
    ab)    # A/B on one box: sweep_r5.py with the default library, then with lbzip2_amd/csrc/variants/<a>.so    ab:<variant>:<slabs>:<kinds>
           for lib in default $a; do
             if [ "$lib" = default ]; then unset LBZ_LIB; else export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$lib.so; fi
             timeout 900 python tests/tools/sweep_r5.py $b $c "LBZAMD_STREAMS=1;LBZAMD_STREAMS=3" 2>&1 | grep -E "MB/s|rror" | sed "s/^/$lib /" | tee -a gpurun_out/${TAG}_ab.txt
           done; unset LBZ_LIB ;;
    usepmc) cp gpurun_out/${TAG}_s1_pmc_traffic.json profiles/pmc_traffic.json && echo "profiles/pmc_traffic.json <- ${TAG}_s1_pmc_traffic.json" ;;   # bench.py's roofline.traffic reads it
    py)    timeout 900 python $a $b $c 2>&1 | tail -40 ;;
  esac
done
