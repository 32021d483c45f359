import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, lbzip2_amd
import oracle_lib as L
lib = lbzip2_amd.library()
n = 1_000_000_000
data = L.gen_kind("wiki", n, 2)
src = torch.frombuffer(data, dtype=torch.uint8).cuda()
dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
for seq in (False, True):
    with lib.context(9, 1112) as ctx:
        ctx.set_sequential(seq)
        for it in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        s = ctx.stats()
        print("sequential" if seq else "default", m, "%.1f MB/s" % (n / dt / 1e6), "ms: collect %.1f bwt %.1f mtf %.1f enc %.1f total %.1f" % (s.ms_collect, s.ms_bwt, s.ms_mtf, s.ms_encode, s.ms_total), "blocks", s.nblocks, flush=True)
