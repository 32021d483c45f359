#!/bin/bash
cd /root/repo
export LD_LIBRARY_PATH=/opt/rocm/lib
python - <<PY
import sys; sys.path.insert(0, "/root/repo")
import bench
open("/tmp/in.bin", "wb").write(bench.gen_input("wiki", 450 * 1000000, 2))
PY
for w in 16 64 256 512; do
  timeout 120 ./lbzip2_amd/host/lbzamd_compress -9 -w $w -t -r 2 < /tmp/in.bin 2>&1 > /tmp/out_$w.bz2 | grep -v amdgpu | tail -1
done
timeout 120 ./lbzip2_amd/host/lbzamd_compress -9 -t -r 2 < /tmp/in.bin 2>&1 > /tmp/out_b.bz2 | grep -v amdgpu | tail -1
cmp /tmp/out_b.bz2 /tmp/out_256.bz2 && echo identical
for n in 16 64 256; do s=$(date +%s.%N); timeout 120 oracle/_ref/lbzip2_dropin_gpu -9 -n $n < /tmp/in.bin > /tmp/out_cli.bz2 2>/dev/null; e=$(date +%s.%N); echo "reference CLI (process.c) + GPU library, -n $n: $(python3 -c "print(round($e - $s, 3))") s wall (450 MB, process start and pool creation included)"; done
cmp /tmp/out_b.bz2 /tmp/out_cli.bz2 && echo cli identical
