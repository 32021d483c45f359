"""Ad-hoc: per-stream timeline of the LAST call in a rocprofv3 --kernel-trace capture (calls start with k_collect)."""
import csv, glob, os, sys
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r["Stream_Id"], r["Grid_Size_X"], r["Workgroup_Size_X"]))
ev.sort()
ncall = int(sys.argv[2]) if len(sys.argv) > 2 else 2            # k_collect launches per call
cols = [i for i, e in enumerate(ev) if e[2] == "k_collect"]
i0 = cols[-ncall]
base = ev[i0][0]
for s, e, name, st, g, wg in ev[i0:]:
    if (e - s) > 20000 or not name.startswith("k_bwt_fixr"):
        print("%8.3f +%7.3f ms  %-14s stream %s grid %s wg %s" % ((s - base) / 1e6, (e - s) / 1e6, name, st, g, wg))
print("span %.3f ms" % ((max(e for s, e, *_ in ev[i0:]) - base) / 1e6))
