#!/bin/bash
# round 5: the splitter/muxer after the ring-order fix (every command under a timeout: a stuck run cost 25 GPU-minutes), the
# configuration that hung, and the bench line's legs with ONE rank chain per round again
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
mkdir -p gpurun_out
T=/dev/shm
timeout 120 python - <<PY
import sys
sys.path.insert(0, "/root/repo")
import bench
d = bench.gen_input("wiki", 1_000_000_000, 2)
open("$T/w1.bin", "wb").write(d)
with open("$T/w3.bin", "wb") as f:
    for _ in range(3): f.write(d)
PY
EXE=lbzip2_amd/host/lbzamd_compress
export LBZAMD_IO_DEBUG=10
{
for f in w1 w3 w1 w3; do
  timeout 60 $EXE -9 -f $T/$f.bin -o $T/$f.bz2 -t 2>&1 | grep "file splitter\|lbzamd_io" | sed "s/^/$f defaults: /"
done
for cfg in "64 6 4 2" "64 6 4 2" "64 4 4 2" "128 3 4 2"; do
  set -- $cfg
  for f in w3; do
    timeout 60 $EXE -9 -f $T/$f.bin -o $T/$f.bz2 -c $1 -p $2 -R $3 -W $4 -t 2>&1 | grep "file splitter\|lbzamd_io" | sed "s/^/$f c=$1 p=$2 R=$3 W=$4: /"; echo "rc=$?"
  done
done
md5sum $T/w1.bz2
( time timeout 60 lbzip2_amd/host/lbzamd -k --report $T/w1.bin ) 2>&1 | grep -v "^$"; md5sum $T/w1.bin.bz2
( time timeout 60 lbzip2_amd/host/lbzamd -dc --report $T/w1.bin.bz2 | md5sum ) 2>&1 | grep -v "^$"; md5sum $T/w1.bin
rm -f $T/w*.bin $T/w*.bz2 $T/w1.bin.bz2
} 2>&1 | tee gpurun_out/r5h_filemode.txt
unset LBZAMD_IO_DEBUG
timeout 400 python bench.py --steps 3 --no-cpu --no-decode --no-seq > gpurun_out/r5h_bench.json 2> gpurun_out/r5h_bench.err; echo "bench rc=$?"
timeout 30 python - <<PY
import json
r = json.loads(open("gpurun_out/r5h_bench.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step", "verified")}, "host", r.get("value_host", {}).get("value"))
vf = r.get("value_file") or {}
print("file", vf.get("value"), vf.get("seconds"), vf.get("first_context_s"), vf.get("behind_first_context_MBps"), vf.get("verified"))
for c in r.get("configs", []): print(" ", c["config"], c.get("value"), c.get("ms_per_step"), c.get("verified"))
PY
