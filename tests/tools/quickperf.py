"""Ad-hoc GPU timing helper (not a test): per-kernel ms and BWT phase ticks."""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, lbzip2_amd
import ctypes as C
import os
if os.environ.get('LBZ_LIB'): lbzip2_amd.LIB_PATH = os.environ['LBZ_LIB']
lib = lbzip2_amd.library()
g = C.CDLL("/root/repo/lbzip2_amd/host/libgen_inputs.so")
sys.path.insert(0, "/root/repo/tests/tools")
import inputs
def gen(kind, n, seed): return inputs.get(kind, n, seed)
slabs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
LEVEL = int(os.environ.get("LBZ_LEVEL", "9"))
for kind in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("text", "rand")):
    n = slabs * LEVEL * 100000
    data = gen(kind, n, int(os.environ.get('LBZ_SEED', '2')))
    src = torch.frombuffer(data, dtype=torch.uint8).cuda()
    dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    for slots in ([int(x) for x in os.environ.get('LBZ_SLOTS', '256').split(',')]):
        ctx = lib.context(LEVEL, slabs, slots)
        best = None
        for it in range(int(os.environ.get("LBZ_ITERS", "2"))):
            t = time.time()
            m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
            dt = time.time() - t
            best = dt if best is None or (it and dt < best) else (dt if it == 1 else best)   # the first pass warms up
            s = ctx.stats()
        dt = best
        print(f"{kind} slabs={slabs} slots={slots}: {n/dt/1e6:.1f} MB/s ms: collect={s.ms_collect:.1f} bwt={s.ms_bwt:.1f} "
              f"(part={s.ms_bwt_part:.1f} batch={s.ms_bwt_batch:.1f} fix={s.ms_bwt_fix:.1f}) mtf={s.ms_mtf:.1f} enc={s.ms_encode:.1f} fin={s.ms_finish:.1f}", flush=True)
        tk = [0] * 8; cnt = 0
        for b in range(0, min(2 * slabs, 64), 2):
            bi = ctx.block_info(b)
            for i in range(8): tk[i] += bi.ticks[i]
            cnt += 1
        nfix = sum(1 for b in range(0, 2 * slabs, 2) if ctx.block_info(b).rounds > 0)
        print("   batch kernel ms/blk: total=%.2f load=%.2f groupscan=%.2f waves=%.2f | blocks needing the doubling fix: %d of %d, ratio %.3f"
              % (tk[0] / cnt / 1e5, tk[3] / cnt / 1e5, tk[4] / cnt / 1e5, tk[5] / cnt / 1e5, nfix, slabs, n / max(1, m)), flush=True)
        print("   waves phase: busy/16 = %.2f ms (utilisation %.0f%%), first sort = %.2f ms/16"
              % (tk[6] / cnt / 1e5 / 16, 100.0 * tk[6] / 16 / max(1, tk[5]), tk[7] / cnt / 1e5 / 16), flush=True)
        print("   sum over batches of the longest chunk: first sort %.2f ms, sort+refine+emit %.2f ms" % (tk[1] / cnt / 1e5, tk[2] / cnt / 1e5), flush=True)
        if os.environ.get("BATCH_TICKS"):
            ft = [0] * 16
            for b in range(0, min(2 * slabs, 64), 2):
                bi = ctx.block_info(b)
                for i in range(16): ft[i] += bi.fticks[i]
            f = [x / cnt / 1e5 for x in ft]
            print("   BATCH_TICKS per block: wave-ms short groups %.2f long groups %.2f runs %.2f closed %.2f emit %.2f | WG-ms batch_runs %.2f sort/split %.2f plan %.2f | batches %d rows %d: alone %d counted %d long %d"
                  % (f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], ft[9] // cnt, ft[10] // cnt, ft[11] // cnt, ft[12] // cnt, ft[13] // cnt), flush=True)
        if os.environ.get("MTF_TICKS"):
            print("   mtf kernel ms/blk: prelude %.2f ranks %.2f zrle %.2f" % (tk[3] / cnt / 1e5, tk[4] / cnt / 1e5, tk[5] / cnt / 1e5), flush=True)
        if os.environ.get("SORT_TICKS"):
            print("   first sort: long-group radix %.2f ms/16, run detection %.2f ms/16 of %.2f" % (tk[0] / cnt / 1e5 / 16, tk[2] / cnt / 1e5 / 16, tk[7] / cnt / 1e5 / 16), flush=True)
        if os.environ.get("COL_TICKS"):
            print("   collect kernel ms/blk (primary block): rle pass %.3f crc %.3f" % (tk[6] / cnt / 1e5, tk[7] / cnt / 1e5), flush=True)
        if os.environ.get("ENC_TICKS"):
            print("   encode kernel ms/blk: E-steps %.2f M-steps %.2f (of which the sorts %.2f) limited codes %.2f selectors+size %.2f packing %.2f"
                  % tuple(tk[i] / cnt / 1e5 for i in (1, 2, 7, 3, 4, 5)), flush=True)
        ctx.close()
