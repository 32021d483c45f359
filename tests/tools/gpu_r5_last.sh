#!/bin/bash
# round 5: GPU suite and the bench line of the round's last commit (kernels as in r05_z: its rocprofv3 / PMC passes stand)
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
TAG=${1:-r05_zz}
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q --durations=4 > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/${TAG}_bench.err
timeout 30 python - <<PY
import json
r = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step", "verified", "ratio")})
vf = r.get("value_file") or {}
print("host", r.get("value_host", {}).get("value"), "file", vf.get("value"), vf.get("seconds"), vf.get("verified"))
print("roofline", {k: r["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "traffic_over_alg", "avg_launch_ms")})
print("isolated", {k: v["ms_per_step"] for k, v in r["roofline"]["isolated"]["per_kernel"].items()})
print("decode", r["decode"]["value"], [o["value"] for o in r["decode"].get("others", [])], "seq", r["sequential"]["value"], r["sequential"]["verified"])
for c in r.get("configs", []): print(" ", c["config"], c.get("value"), c.get("verified"))
print("cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"], r["cpu_baseline"]["MBps_by_threads"])
PY
