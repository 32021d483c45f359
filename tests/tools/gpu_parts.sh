# (tuning) A/B of the working tree's library against variants/h0.so (the last commit)
cd /root/repo; export PYTHONPATH=/root/repo:/root/repo/tests
for kind in ${KINDS:-wiki mixed}; do for v in default h0; do
  if [ "$v" != "default" ]; then export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so; else unset LBZ_LIB; fi
  echo "== $kind $v"; LBZAMD_STREAMS=1 LBZ_SLOTS=371 timeout 60 python tests/tools/quickperf.py 1112 $kind 2>&1 | grep "MB/s\|batch kernel"
done; done
