# (tuning) A/B of variants against the working tree's library
cd /root/repo; export PYTHONPATH=/root/repo:/root/repo/tests
for v in default ${VARIANTS:-ci}; do
  if [ "$v" != "default" ]; then export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so; else unset LBZ_LIB; fi
  echo "== $v one round"; LBZAMD_STREAMS=1 LBZ_SLOTS=1112 timeout 60 python tests/tools/quickperf.py 1112 wiki,mixed 2>&1 | grep "MB/s"
  echo "== $v three streams"; LBZAMD_STREAMS=3 LBZ_SLOTS=371 timeout 60 python tests/tools/quickperf.py 1112 wiki 2>&1 | grep "MB/s"
done
