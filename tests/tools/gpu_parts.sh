# (tuning) 32 against 16 segments per block, three streams
cd /root/repo; export PYTHONPATH=/root/repo:/root/repo/tests
for kind in wiki pysrc tar mixed; do for v in default s16; do
  if [ "$v" != "default" ]; then export LBZ_LIB=/root/repo/lbzip2_amd/csrc/variants/$v.so; else unset LBZ_LIB; fi
  echo "== $kind $v streams=3"; LBZAMD_STREAMS=3 LBZ_SLOTS=371 timeout 60 python tests/tools/quickperf.py 1112 $kind 2>&1 | grep "MB/s"
done; done
unset LBZ_LIB
for sl in 16 112; do echo "== wiki slabs=$sl"; LBZAMD_STREAMS=3 LBZ_SLOTS=$sl timeout 60 python tests/tools/quickperf.py $sl wiki 2>&1 | grep "MB/s"; done
