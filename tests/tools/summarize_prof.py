"""Ad-hoc: turn rocprofv3 csv output into the small summaries kept under profiles/."""
import csv, collections, glob, sys, os
d = sys.argv[1]; out = sys.argv[2]
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(out + "_kernel_stats.csv", "w") as o:
        o.write("kernel,calls,total_ns,avg_ns,percent\n")
        for r in rows:
            o.write('"%s",%s,%s,%s,%s\n' % (r["Name"].split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    tag = os.path.relpath(f, d).split(os.sep)[0]          # pmc_fetch / pmc_write
    with open(out + "_pmc_" + tag + ".csv", "w") as o:
        o.write("kernel,counter,dispatches,sum,mean_per_dispatch\n")
        for k in agg:
            for c in agg[k]:
                o.write('"%s",%s,%d,%.6g,%.6g\n' % (k, c, n[k][c], agg[k][c], agg[k][c] / n[k][c]))
print("done")

# KB per slab of the benchmark workload for bench.py's roofline.traffic (argv[3] = slabs per step, argv[4] = steps profiled)
if len(sys.argv) > 4:
    import json
    nblocks, steps = int(sys.argv[3]), int(sys.argv[4])
    kern = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        tot = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                tot[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
        for (k, c), v in tot.items():
            kern[k]["fetch_kb_per_slab" if c == "FETCH_SIZE" else "write_kb_per_slab"] = v / (nblocks * steps)
    kern = {k: v for k, v in kern.items() if len(v) == 2}
    json.dump({"workload": sys.argv[5] if len(sys.argv) > 5 else "text -9", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; raw counter KB, uncorrected",
               "kernels": kern}, open(out + "_pmc_traffic.json", "w"), indent=1)
