"""Ad-hoc: per block, the rows the text rounds still hold after each launch and what was handed to the rank rounds."""
import sys, os
os.environ["LBZAMD_DIAG_DEEP"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, lbzip2_amd
if os.environ.get('LBZ_LIB'): lbzip2_amd.LIB_PATH = os.environ['LBZ_LIB']
lib = lbzip2_amd.library()
from bench import gen_input
slabs = int(sys.argv[1]); kind = sys.argv[2]
n = slabs * 900000
sys.path.insert(0, "/root/repo/tests/tools")
import inputs
data = inputs.get(kind, n, 2)
src = torch.frombuffer(data, dtype=torch.uint8).cuda()
dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
ctx = lib.context(9, slabs, slabs)
for _ in range(2): ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
s = ctx.stats()
print(f"{kind}: bwt={s.ms_bwt:.1f} (part={s.ms_bwt_part:.1f} batch={s.ms_bwt_batch:.1f} fix={s.ms_bwt_fix:.1f})")
rows = []
for b in range(0, 2 * slabs, 2):
    bi = ctx.block_info(b)
    f = list(bi.fticks)
    rows.append((bi.n, bi.periodic, bi.rounds, f))
nrank = sum(1 for r in rows if r[1] or r[2])
print("blocks %d, of which periodic flag / rank rounds: %d" % (len(rows), nrank))
import collections
print("blk      n per rnd | tied rows after batch, after each launch ... | h0 skip long rows hmin1 hminN")
for i, (n_, per, rnd, f) in enumerate(rows[: int(os.environ.get("SHOW", "40"))]):
    print("%3d %7d %d %3d | %s | %d %d %d %d %d %d" % (i, n_, per, rnd, " ".join("%7d" % x for x in f[:9]), f[9], f[10], f[11], f[12], f[13], f[14]))
tot = [sum(r[3][k] for r in rows) / len(rows) for k in range(9)]
print("mean tied rows:", " ".join("%.0f" % x for x in tot))
print("mean rank rounds per block: %.2f; skip=%d; long>n/2: %d" % (sum(r[2] for r in rows) / len(rows), sum(1 for r in rows if r[3][10]), sum(1 for r in rows if 2 * r[3][11] > r[0])))
