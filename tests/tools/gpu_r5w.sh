#!/bin/bash
# round 5: the inverse path after the change of its diagnostics -- the decoder's and the command's GPU tests
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_decode.py tests/test_cli.py tests/test_dropin_link.py -m gpu -x -q --durations=4 > gpurun_out/r5w_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r5w_pytest.log
