#!/bin/bash
# SQ / TCC counter passes (rocprofv3 --pmc, one pass per counter set, never combined with tracing) + a kernel-trace
# pass of the same command.  usage: tests/tools/run_pmc.sh <tag> [slabs] [kind]    -> gpurun_out/<tag>_pmc_summary.json etc.
set -u
TAG=${1:-r03_pmc}; SLABS=${2:-556}; KIND=${3:-wiki}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT/prof_$TAG
export TMPDIR=/tmp PYTHONPATH=$REPO:$REPO/tests
export LBZAMD_STREAMS=${LBZAMD_STREAMS:-1} LBZ_SLOTS=${LBZ_SLOTS:-$SLABS}
CMD="python $REPO/tests/tools/quickperf.py $SLABS $KIND"
run_pass() { # name counters...
  local name=$1; shift
  ( cd /tmp && timeout 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/prof_$TAG/$name -- $CMD > $OUT/prof_$TAG/$name.log 2>&1 ) || echo "pass $name failed"
}
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/stats -- $CMD > $OUT/prof_$TAG/stats.log 2>&1 )
grep "MB/s" $OUT/prof_$TAG/stats.log
run_pass sqA SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run_pass sqB SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run_pass sqC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
# the XCD's L2: hit rate of the sorting kernels' gathers (TCC has four slots a pass) and the fabric requests behind the misses
run_pass tccA TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run_pass tccB TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
python $REPO/tests/tools/pmc_summary.py $OUT/prof_$TAG $OUT/${TAG} $SLABS 2
