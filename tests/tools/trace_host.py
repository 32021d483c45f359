"""Ad-hoc: a few host-buffer calls for a timeline trace (rocprofv3 --kernel-trace --memory-copy-trace)."""
import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch, lbzip2_amd
import bench
n = 1_000_000_000
data = np.frombuffer(bench.gen_input("wiki", n, 2), dtype=np.uint8)
lib = lbzip2_amd.library()
bound = lib.bound(n)
L = lib.lib
L.lbzamd_compress_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
hin = torch.from_numpy(data).pin_memory()
hout = torch.empty(bound, dtype=torch.uint8).pin_memory()
with lib.context(9, 1112) as ctx:
    for it in range(3):
        got = C.c_size_t(0)
        t = time.time(); rc = L.lbzamd_compress_host(ctx.h, hin.data_ptr(), n, hout.data_ptr(), bound, C.byref(got)); dt = time.time() - t
        print("call %d: %.1f ms rc %d" % (it, dt * 1e3, rc), flush=True)
