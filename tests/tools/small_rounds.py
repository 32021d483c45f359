"""Ad-hoc: device-resident rate of small inputs (rounds of few blocks): 16, 64 and 112 slabs of the enwik-like text."""
import sys, time, hashlib
sys.path.insert(0, "/root/repo")
import torch, lbzip2_amd, os
import bench
lib = lbzip2_amd.Library(os.environ["LBZ_LIB"]) if os.environ.get("LBZ_LIB") else lbzip2_amd.library()
for n in [int(a) for a in sys.argv[1:]] or (14_400_000, 57_600_000, 100_000_000):
    data = bench.gen_input("wiki", n, 1)
    src = torch.frombuffer(data, dtype=torch.uint8).cuda()
    dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    with lib.context(9, (n + 899999) // 900000) as ctx:
        best = None
        for it in range(6):
            torch.cuda.synchronize(); t = time.time()
            m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
            torch.cuda.synchronize(); dt = time.time() - t
            best = dt if best is None or dt < best else best
        st = ctx.stats()
    print("%d B (%d slabs): %.2f ms = %.0f MB/s  md5 %s  part %.2f batch %.2f fix %.2f" % (n, (n + 899999) // 900000, best * 1e3, n / best / 1e6, hashlib.md5(dst[:m].cpu().numpy().tobytes()).hexdigest()[:8], st.ms_bwt_part, st.ms_bwt_batch, st.ms_bwt_fix), flush=True)
