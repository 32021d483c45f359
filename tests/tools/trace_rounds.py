"""Ad-hoc: per-launch durations of the LAST compress call in a rocprofv3 --kernel-trace capture (single stream, one round):
the chain k_collect, k_bwt_part, k_bwt_batch, k_bwt_deep x10, k_bwt_fix0, k_bwt_fixr x17, k_bwt_fixend, k_mtf, k_encode."""
import csv, glob, os, sys
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
ev.sort()
last = max(i for i, e in enumerate(ev) if e[2].startswith("k_collect"))
ev = ev[last:]
t0 = ev[0][0]
cnt = {}
tot = {}
for s, e, name, qid, sid in ev:
    k = cnt.get(name, 0); cnt[name] = k + 1
    tot[name] = tot.get(name, 0) + (e - s)
    if (e - s) > 30000 or not (name.startswith("k_bwt_fixr") or name.startswith("k_bwt_deep")):
        print("%9.3f ms  %8.3f ms  %s#%d  (queue %s stream %s)" % ((s - t0) / 1e6, (e - s) / 1e6, name, k, qid, sid))
print("span %.3f ms" % ((ev[-1][1] - t0) / 1e6))
for name in tot:
    print("  %-14s %8.3f ms in %d launches" % (name, tot[name] / 1e6, cnt[name]))
