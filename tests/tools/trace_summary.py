"""Ad-hoc: timeline of the LAST host-buffer call in a rocprofv3 --kernel-trace --memory-copy-trace capture."""
import csv, glob, os, sys
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), ""))
ev.sort()
# calls are separated by the k_collect of a short first round: find the starts of the three calls by big H2D copies
starts = [e[0] for e in ev if e[2] == "C" and "HOST_TO_DEVICE" in e[3].upper().replace("MEMORY_COPY_", "") and int(e[3].split()[-1]) > 50_000_000]
if not starts:
    starts = [ev[0][0]]
# last call: from the 4th-last big H2D on (4 rounds per call)
t0 = starts[-4] if len(starts) >= 4 else starts[0]
last = [e for e in ev if e[0] >= t0 - 3_000_000]
base = last[0][0]
print("events in the last call:", len(last))
for s, e, k, name, q in last:
    if k == "C" or name in ("k_collect", "k_bwt_part", "k_bwt_part_w", "k_bwt_batch", "k_bwt_fix0", "k_mtf", "k_encode", "k_offsets", "k_gather", "k_bwt_fixend"):
        print("%9.3f ms  +%8.3f ms  %s %s %s" % ((s - base) / 1e6, (e - s) / 1e6, k, name, q))
# busy union
iv = sorted((s, e) for s, e, k, name, q in last if k == "K")
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((cur_e, s)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.2f ms, kernels busy (union) %.2f ms, idle gaps > 0.2 ms:" % ((cur_e - base) / 1e6, busy / 1e6))
for a, b in gaps:
    if b - a > 200_000:
        print("   gap at %.2f ms: %.2f ms" % ((a - base) / 1e6, (b - a) / 1e6))
