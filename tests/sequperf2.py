import sys, time, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, lbzip2_amd
import oracle_lib as L
lib = lbzip2_amd.Library(os.environ["LBZ_LIB"])
n = 300_000_000
data = L.gen_kind("wiki", n, 2)
src = torch.frombuffer(data, dtype=torch.uint8).cuda()
dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
with lib.context(9, 400) as ctx:
    ctx.set_sequential(True)
    for it in range(2):
        m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
    s = ctx.stats()
    print("collect ms %.1f blocks %d" % (s.ms_collect, s.nblocks))
    tk = [ctx.block_info(2 * b).fticks for b in range(s.nblocks)]
    for i, name in ((8, "wait"), (9, "skip"), (10, "cut"), (11, "rest"), (14, "skip:load+heads"), (15, "skip:maxscan"), (7, "skip:count+sum")):
        v = [t[i] for t in tk]
        print(name, "avg %.1f us  max %.1f us" % (sum(v) / len(v) / 100.0, max(v) / 100.0))
    links = [(tk[b + 1][13] - tk[b][13]) & 0xFFFFFFFF for b in range(len(tk) - 1)]
    print("link (cut to cut) avg %.1f us" % (sum(links) / len(links) / 100.0))
