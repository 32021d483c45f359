"""decoder timing on the GPU: compress a generated input, decode it, compare, print the stage times"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lbzip2_amd
import oracle_lib as L

lib = lbzip2_amd.Library(os.environ["LBZ_LIB"]) if os.environ.get("LBZ_LIB") else lbzip2_amd.library()
cases = os.environ.get("LBZ_DEC_CASES", "wiki:100000000,wiki:1000000000,rand:100000000,tar:175000000,text:1000000000")
with lib.decoder(2400) as d:
    for case in cases.split(","):
        kind, n = case.split(":")
        n = int(n)
        data = L.gen_kind(kind, n, 2)
        src = torch.frombuffer(data, dtype=torch.uint8).cuda()
        del data
        z = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
        with lib.context(9, min(1200, (n + 899999) // 900000)) as ctx:
            m = ctx.compress_device(src.data_ptr(), n, z.data_ptr(), z.numel())
        out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        best = None
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            try:
                k = d.decompress_device(z.data_ptr(), m, out.data_ptr(), out.numel())
            except lbzip2_amd.LbzError as ex:                      # timing experiments with crippled kernels
                k = -1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        st = d.stats()
        ok = k == n and bool(torch.equal(out[:n], src))
        print(f"{kind} {n} -> {m} decoded ok: {ok} {n / best / 1e6:.1f} MB/s (wall {best * 1e3:.1f} ms) stages ms: scan {st.ms_scan:.1f} blocks {st.ms_blocks:.1f} [slowest block: huff {st.ms_huff:.1f} "
              f"sort {st.ms_sort:.1f} walk {st.ms_walk:.1f}] emit {st.ms_emit:.1f} blocks {st.nblocks}", flush=True)
        del src, z, out
