"""Ad-hoc: where round 0 of the text rounds (k_bwt_deep, -DDEEP_TICKS build) spends its waves' time, per block."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, lbzip2_amd, ctypes as C
if os.environ.get('LBZ_LIB'): lbzip2_amd.LIB_PATH = os.environ['LBZ_LIB']
lib = lbzip2_amd.library()
sys.path.insert(0, "/root/repo/tests/tools"); import inputs
slabs = int(sys.argv[1]); kind = sys.argv[2]
n = slabs * 900000
data = inputs.get(kind, n, 2)
src = torch.frombuffer(data, dtype=torch.uint8).cuda()
dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
ctx = lib.context(9, slabs, slabs)
for _ in range(2): ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
s = ctx.stats()
print(f"{kind}: bwt={s.ms_bwt:.1f} (part={s.ms_bwt_part:.1f} batch={s.ms_bwt_batch:.1f} fix={s.ms_bwt_fix:.1f})")
t = [0] * 16; tk = [0] * 8; cnt = 0
for b in range(0, 2 * slabs, 2):
    bi = ctx.block_info(b)
    for i in range(16): t[i] += bi.fticks[i]
    for i in range(8): tk[i] += bi.ticks[i]
    cnt += 1
print("round 0, wave-ms per block (100 MHz ticks): long runs %.2f, strip set-up %.2f, steps %.2f, output %.2f; strips %d; kernel wall per round (sum over segments, ms): %s"
      % (t[0] / cnt / 1e5, t[1] / cnt / 1e5, t[2] / cnt / 1e5, t[3] / cnt / 1e5, t[4] / cnt, " ".join("%.2f" % (t[8 + i] / cnt / 1e5) for i in range(8))))
print("   long runs, all launches: pieces %d per block, rows %d per block; wave-ms per block: common prefix %.2f, count %.2f, scan %.2f, placement %.2f"
      % (tk[6] / cnt, tk[7] / cnt, t[5] / cnt / 1e5, t[6] / cnt / 1e5, t[7] / cnt / 1e5, tk[5] / cnt / 1e5))
