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
    tag = os.path.basename(os.path.dirname(f))
    with open(out + "_pmc_" + tag + ".csv", "w") as o:
        o.write("kernel,counter,dispatches,sum,mean_per_dispatch\n")
        for k in agg:
            for c in agg[k]:
                o.write('"%s",%s,%d,%.6g,%.6g\n' % (k, c, n[k][c], agg[k][c], agg[k][c] / n[k][c]))
print("done")
