import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, lbzip2_amd
import oracle_lib as L
lib = lbzip2_amd.library()
for kind, n in (("text", 115_200_000), ("rand", 57_600_000)):
    data = (L.gen_text if kind == "text" else L.gen_rand)(n, 2)
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    for slots in (256, 512):
        try:
            ctx = lib.context(9, 128, slots)
        except Exception as e:
            print("slots", slots, e); continue
        for it in range(2):
            t = time.time()
            m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
            dt = time.time() - t
            s = ctx.stats()
            print(f"{kind} n={n} slots={slots} it={it}: {n/dt/1e6:.1f} MB/s out={m} blocks={s.nblocks} "
                  f"ms: collect={s.ms_collect:.1f} bwt={s.ms_bwt:.1f} mtf={s.ms_mtf:.1f} enc={s.ms_encode:.1f} fin={s.ms_finish:.1f} "
                  f"sort_elems/n_rle={s.sort_elems/max(1,s.n_rle):.2f}", flush=True)
        ctx.close()
