"""Ad-hoc: the host-buffer path (H2D + compress + stream back in host memory) through the C ABI: page-locked buffers with
the device writing the output itself, the same with the staging copy (LBZAMD_NO_DIRECT_OUT), pageable buffers; raw PCIe
rates beside them.  usage: hostpath_perf.py [bytes] [kind]"""
import os, sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch, lbzip2_amd
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "wiki"
data = np.frombuffer(bench.gen_input(kind, n, 2), dtype=np.uint8)
lib = lbzip2_amd.library()
bound = lib.bound(n)
L = lib.lib
L.lbzamd_compress_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
hin = torch.from_numpy(data).pin_memory()
hout = torch.empty(bound, dtype=torch.uint8).pin_memory()
pout = np.empty(bound, dtype=np.uint8); pout[:] = 0
src = hin.cuda()
dst = torch.empty(bound, dtype=torch.uint8, device="cuda")
with lib.context(9, (n + 899999) // 900000) as ctx:
    def run(name, i_ptr, o_ptr, reps=4):
        best = None
        for it in range(reps):
            got = C.c_size_t(0)
            t = time.time(); rc = L.lbzamd_compress_host(ctx.h, i_ptr, n, o_ptr, bound, C.byref(got)); dt = time.time() - t
            best = dt if best is None or (it and dt < best) else (dt if it == 1 else best)
        print("%-58s rc %d %8.1f MB/s (%.1f ms), out %d" % (name, rc, n / best / 1e6, best * 1e3, got.value), flush=True)
    best = None
    for it in range(4):
        torch.cuda.synchronize(); t = time.time(); ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), bound); torch.cuda.synchronize(); dt = time.time() - t
        best = dt if best is None or (it and dt < best) else (dt if it == 1 else best)
    print("%-58s      %8.1f MB/s (%.1f ms)" % ("device-resident", n / best / 1e6, best * 1e3), flush=True)
    run("pinned in, pinned out (device writes the stream)", hin.data_ptr(), hout.data_ptr())
    os.environ["LBZAMD_NO_DIRECT_OUT"] = "1"
    run("pinned in, pinned out, staging copy (LBZAMD_NO_DIRECT_OUT)", hin.data_ptr(), hout.data_ptr())
    del os.environ["LBZAMD_NO_DIRECT_OUT"]
    run("pageable in, pageable out", data.ctypes.data, pout.ctypes.data, 3)
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, h in (("pageable", torch.from_numpy(data)), ("pinned", hin)):
        torch.cuda.synchronize(); t = time.time(); d.copy_(h); torch.cuda.synchronize(); dt = time.time() - t
        print("H2D %s: %.1f GB/s" % (name, n / dt / 1e9), flush=True)
    torch.cuda.synchronize(); t = time.time(); hout[:250_000_000].copy_(d[:250_000_000]); torch.cuda.synchronize(); dt = time.time() - t
    print("D2H pinned: %.1f GB/s" % (0.25 / dt), flush=True)
