"""Ad-hoc: a few device-resident calls on a small input for a timeline trace (rocprofv3 --kernel-trace)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch, lbzip2_amd
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
data = bench.gen_input("wiki", n, 1)
lib = lbzip2_amd.library()
src = torch.frombuffer(data, dtype=torch.uint8).cuda()
dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
with lib.context(9, (n + 899999) // 900000) as ctx:
    for it in range(3):
        torch.cuda.synchronize(); t = time.time()
        ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
        torch.cuda.synchronize(); print("call %d: %.2f ms" % (it, (time.time() - t) * 1e3), flush=True)
