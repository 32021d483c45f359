# (tuning) A/B of variants against the working tree's library
cd /root/repo; export PYTHONPATH=/root/repo:/root/repo/tests
for v in default ${VARIANTS:-ci}; do
  if [ "$v" != "default" ]; then export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so; else unset LBZ_LIB; fi
  echo "== $v"; LBZAMD_STREAMS=${ST:-1} LBZ_SLOTS=371 timeout 60 python tests/tools/quickperf.py 1112 ${KINDS:-wiki,pysrc} 2>&1 | grep "MB/s"
done
