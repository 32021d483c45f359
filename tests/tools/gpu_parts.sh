# (tuning) rounds per stream: slots x streams on wiki(10^9)
cd /root/repo; export PYTHONPATH=/root/repo:/root/repo/tests
for st in 2 3 4; do for sl in 140 186 278 371 556; do
  echo "== streams=$st slots=$sl"; LBZAMD_STREAMS=$st LBZ_SLOTS=$sl timeout 60 python tests/tools/quickperf.py 1112 ${KIND:-wiki} 2>&1 | grep "MB/s" | cut -c1-60
done; done
