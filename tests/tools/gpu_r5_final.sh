#!/bin/bash
# round 5: the evidence of the round's last build -- GPU suite, rocprofv3 kernel stats + PMC passes of the bench command,
# the counter summary, then `python bench.py` with this run's traffic figures in place.
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
TAG=${1:-r05_z}
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/${TAG}_pytest.log
timeout 600 bash tests/tools/run_profiles.sh $TAG > gpurun_out/${TAG}_profiles.log 2>&1; tail -5 gpurun_out/${TAG}_profiles.log
cp gpurun_out/${TAG}_s1_pmc_traffic.json profiles/pmc_traffic.json
timeout 600 bash tests/tools/run_pmc.sh ${TAG}p 556 wiki > gpurun_out/${TAG}_pmc.log 2>&1; grep -E "^k_bwt_deep |^k_bwt_deepr|^k_bwt_batch|^k_mtf" gpurun_out/${TAG}_pmc.log | cut -c1-400
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/${TAG}_bench.err
python - <<PY
import json
r = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step", "verified", "ratio")})
print("host", r.get("value_host", {}).get("value"), "file", r.get("value_file"))
print("roofline", {k: r["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "traffic_over_alg", "avg_launch_ms")})
print("isolated", {k: v["ms_per_step"] for k, v in r["roofline"]["isolated"]["per_kernel"].items()})
print("decode", r["decode"]["value"], [o["value"] for o in r["decode"].get("others", [])], "seq", r["sequential"]["value"], r["sequential"]["verified"])
for c in r.get("configs", []): print(" ", c["config"], c.get("value"), c.get("verified"))
print("cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"], r["cpu_baseline"]["MBps_by_threads"])
PY
