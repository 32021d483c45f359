cd /root/repo; export PYTHONPATH=/root/repo:/root/repo/tests
for kind in wiki mixed; do
  echo "== $kind one round of 1112"; LBZAMD_STREAMS=1 LBZ_SLOTS=1112 timeout 100 python tests/tools/quickperf.py 1112 $kind 2>&1 | grep "MB/s"
done
echo "== wiki 3 streams"; LBZAMD_STREAMS=3 LBZ_SLOTS=371 timeout 100 python tests/tools/quickperf.py 1112 wiki 2>&1 | grep "MB/s"
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or parity or golden" 2>&1 | tail -3
