#!/bin/bash
# decoder on the GPU: tests/tools/quickdec.py (sizes x kinds, stage timings), optionally the decode tests
mkdir -p gpurun_out
[ "$1" = "tests" ] && timeout 900 python -m pytest tests/test_decode.py -x -q -m gpu > gpurun_out/dec_tests.log 2>&1
timeout 600 python tests/tools/quickdec.py > gpurun_out/dec.log 2>&1
cat gpurun_out/dec_tests.log 2>/dev/null | tail -3
cat gpurun_out/dec.log
