"""Ad-hoc: throughput of the host-buffer path (H2D + compress + D2H) through the C ABI, no Python copies."""
import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch, lbzip2_amd
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
data = np.frombuffer(bench.gen_input("text", n, 2), dtype=np.uint8)
lib = lbzip2_amd.library()
bound = lib.bound(n)
out = np.empty(bound, dtype=np.uint8); out[:] = 0
L = lib.lib
L.lbzamd_compress_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
with lib.context(9, (n + 899999) // 900000) as ctx:
    for it in range(3):
        got = C.c_size_t(0)
        t = time.time(); rc = L.lbzamd_compress_host(ctx.h, data.ctypes.data, n, out.ctypes.data, bound, C.byref(got)); dt = time.time() - t
        print("C ABI host path (pageable): rc %d %.1f MB/s (%.1f ms), out %d" % (rc, n / dt / 1e6, dt * 1e3, got.value), flush=True)
    # raw copies for reference
    src = torch.from_numpy(data)
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, h in (("pageable", src), ("pinned", src.pin_memory())):
        torch.cuda.synchronize(); t = time.time(); d.copy_(h); torch.cuda.synchronize(); dt = time.time() - t
        print("H2D %s: %.1f GB/s" % (name, n / dt / 1e9), flush=True)
    t = time.time(); p = src.pin_memory(); print("pin_memory copy of 1 GB: %.1f ms" % ((time.time() - t) * 1e3))
    raw = data.tobytes()
    for it in range(2):
        t = time.time(); o = ctx.compress(raw); dt = time.time() - t
        print("Python Context.compress(bytes): %.1f MB/s (%.1f ms)" % (n / dt / 1e6, dt * 1e3), flush=True)
