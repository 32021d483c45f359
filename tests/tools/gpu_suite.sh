#!/bin/bash
# GPU box: the -m gpu suite, smoke, and the bench lines (all with hard timeouts).  usage: tests/tools/gpu_suite.sh <tag>
cd /root/repo
TAG=${1:-r02}
export PYTHONPATH=/root/repo:/root/repo/tests
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 3000 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --kind text --no-cpu --no-host > gpurun_out/${TAG}_bench_text.json 2>> gpurun_out/${TAG}_bench.err; head -c 700 gpurun_out/${TAG}_bench_text.json; echo
timeout 300 python bench.py --scaling strong --no-cpu --no-host --no-isolated > gpurun_out/${TAG}_bench_strong1.json 2>> gpurun_out/${TAG}_bench.err; head -c 900 gpurun_out/${TAG}_bench_strong1.json; echo
