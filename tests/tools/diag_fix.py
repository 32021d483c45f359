"""Ad-hoc: how much work does the doubling fix do on source-like text?"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, lbzip2_amd
import glob
def pysrc(n):
    out = bytearray()
    for f in sorted(glob.glob("/usr/lib/python3*/**/*.py", recursive=True)) + sorted(glob.glob("/usr/local/lib/python3*/dist-packages/**/*.py", recursive=True)):
        try: out += open(f, "rb").read()
        except Exception: pass
        if len(out) >= n: break
    while len(out) < n: out += out[:n - len(out)]
    return out[:n]
lib = lbzip2_amd.library()
data = bytes(pysrc(64 * 900000))
with lib.context(9, 64) as ctx:
    ctx.run_stages(data, 1)
    for b in range(0, 16, 2):
        bi = ctx.block_info(b)
        t = list(bi.fticks)
        print("blk", b, "n", bi.n, "rounds", bi.rounds, "tied after batch %d (%.1f%%)" % (t[1], 100.0 * t[1] / bi.n),
              "doubling rows / n = %.2f" % (bi.sort_elems / bi.n - 1.0),
              "| fix ms: regroup %.2f total %.2f | %d batches: load %.2f runs %.2f sort %.2f write %.2f" % (t[6] / 1e5, t[7] / 1e5, t[0], t[2] / 1e5, t[3] / 1e5, t[4] / 1e5, t[5] / 1e5))
