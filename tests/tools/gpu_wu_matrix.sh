#!/bin/bash
# Ad-hoc: the work-unit interface (lbzamd_compress -w N) under the pool's tuning knobs.
cd /root/repo
python - <<PY
import sys; sys.path.insert(0, "/root/repo")
import bench
open("/tmp/in2.bin", "wb").write(bench.gen_input("wiki", 1800 * 1000000, 3))
PY
run() { for w in 16 64 256; do echo -n "$1 | "; env $1 ./lbzip2_amd/host/lbzamd_compress -9 -w $w -t -r 4 < /tmp/in2.bin 2>&1 > /dev/null | grep -v "amdgpu\|^round" | tail -1; done; }
run "X=0"
run "LBZAMD_POOL_KIN=0"
run "LBZAMD_POOL_KOUT=0"
run "LBZAMD_POOL_GATHER=0"
run "LBZAMD_POOL_GATHER=300"
run "LBZAMD_POOL_SLABS=512"
