#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_cli.py tests/test_gpu_parity.py tests/test_decode.py -m gpu -x -q -k "cli or splitter or c_side or n_devices or host_driver" 2>&1 | tail -3
timeout 500 python bench.py --steps 3 --no-cpu --no-decode --no-seq --no-legs > gpurun_out/r5q_bench.json 2> gpurun_out/r5q_bench.err; echo "bench rc=$?"
timeout 30 python - <<PY
import json
r = json.loads(open("gpurun_out/r5q_bench.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step", "verified")}, "host", r.get("value_host", {}).get("value"))
print("file", r.get("value_file"))
PY
T=/dev/shm
timeout 120 python - <<PY
import sys
sys.path.insert(0, "/root/repo")
import bench
d = bench.gen_input("wiki", 1_000_000_000, 2)
with open("$T/w10.bin", "wb") as f:
    for _ in range(10): f.write(d)
PY
timeout 60 lbzip2_amd/host/lbzamd_compress -9 -f $T/w10.bin -o $T/w10.bz2 -t 2>&1 | grep "file splitter" | cut -c1-330
LBZAMD_IO_NOMAP=1 timeout 60 lbzip2_amd/host/lbzamd_compress -9 -f $T/w10.bin -o $T/w10.bz2 -t 2>&1 | grep "file splitter" | cut -c1-330
LBZAMD_FAKE_DEVICES=8 timeout 60 lbzip2_amd/host/lbzamd_compress -9 -f $T/w10.bin -o $T/w10.bz2 -c 128 -p 1 -g 8 -t 2>&1 | grep "file splitter" | cut -c1-400
rm -f $T/w10.bin $T/w10.bz2
