"""Ad-hoc: rocprofv3 csv output of tests/tools/run_pmc.sh -> one JSON table per kernel (counter sums over all dispatches,
per-row figures, stall split).  usage: pmc_summary.py <prof dir> <out prefix> <slabs> <iterations>"""
import collections
import csv
import glob
import json
import os
import sys

d, out = sys.argv[1], sys.argv[2]
slabs, iters = int(sys.argv[3]), int(sys.argv[4])
K = collections.defaultdict(dict)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    tot = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        tot[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, c), v in tot.items():
        K[k][c] = v
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0]
        K[k]["calls"] = int(r["Calls"]); K[k]["total_ms"] = float(r["TotalDurationNs"]) / 1e6
rows = slabs * iters * 900000.0
res = {}
for k, c in sorted(K.items(), key=lambda kv: -kv[1].get("total_ms", 0)):
    if not k.startswith("k_"):
        continue
    e = {kk: (round(v, 3) if isinstance(v, float) and v < 1e6 else v) for kk, v in c.items()}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        e["frac_wait_any"] = round(c.get("SQ_WAIT_ANY", 0) / wc, 3)
        e["frac_wait_inst"] = round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
        e["frac_active"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
        e["frac_active_valu"] = round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3)
        e["frac_active_lds"] = round(c.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3)
    for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM",
              "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS_ATOMIC", "SQ_INSTS_BRANCH"):
        if n in c:
            e[n + "_per_row"] = round(c[n] / rows, 4)
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        e["l2_requests_per_row"] = round((c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) / rows, 3)
    if c.get("TCC_EA0_RDREQ_sum"):
        e["fabric_rdreq_per_row"] = round(c["TCC_EA0_RDREQ_sum"] / rows, 3)
        e["fabric_rdreq_32B_share"] = round(c.get("TCC_EA0_RDREQ_32B_sum", 0.0) / c["TCC_EA0_RDREQ_sum"], 3)
    if c.get("TCC_EA0_WRREQ_sum"):
        e["fabric_wrreq_per_row"] = round(c["TCC_EA0_WRREQ_sum"] / rows, 3)
        e["fabric_wrreq_64B_share"] = round(c.get("TCC_EA0_WRREQ_64B_sum", 0.0) / c["TCC_EA0_WRREQ_sum"], 3)
    if "FETCH_SIZE" in c:
        e["fetch_kb_per_slab_raw"] = round(c["FETCH_SIZE"] / (slabs * iters), 1)
    if "WRITE_SIZE" in c:
        e["write_kb_per_slab_raw"] = round(c["WRITE_SIZE"] / (slabs * iters), 1)
    if "total_ms" in c:
        e["ms_per_iter"] = round(c["total_ms"] / iters, 3)
    res[k] = e
json.dump({"slabs": slabs, "iterations": iters, "note": "wave-instruction counts per block row (900000 rows per slab); SQ cycle counters in quad-cycles",
           "kernels": res}, open(out + "_pmc_summary.json", "w"), indent=1)
for k, e in res.items():
    print(k, {kk: e[kk] for kk in e if kk.startswith("frac") or kk.endswith("per_row") or kk.endswith("share") or kk in ("ms_per_iter", "fetch_kb_per_slab_raw", "write_kb_per_slab_raw", "l2_hit_rate")})
