#!/bin/bash
# round 5, fourth GPU call: rounds of few blocks with the ranks of a block over several workgroups; parity on the new path
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sequential.py -m gpu -x -q > gpurun_out/r5d_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5d_pytest.log
timeout 600 python tests/tools/small_rounds.py "LBZAMD_MTF_PARTS=1;LBZAMD_MTF_PARTS=2;LBZAMD_MTF_PARTS=4;" 2>&1 | grep slabs | tee gpurun_out/r5d_small.txt
bash tests/tools/gpu_trace.sh r5d_wiki112 wiki 112 2>&1 | grep -v copyBuffer | tail -50
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo")
import bench
open("/dev/shm/w450.bin", "wb").write(bench.gen_input("wiki", 450_000_000, 2))
PY
for n in 16 64 256; do lbzip2_amd/host/lbzamd_compress -9 -w $n -t -r 3 < /dev/shm/w450.bin 2>&1 >/dev/null | tail -1; done | tee gpurun_out/r5d_workunits.txt
rm -f /dev/shm/w450.bin
