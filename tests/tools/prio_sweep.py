"""Ad-hoc: the host-buffer path (pinned in, stream in pinned memory out) under the stream-priority variants of
LBZAMD_PRIO (lbz_api.hip): 1 = copy stream on its own (high-priority) hardware queue, 2 = head round high, 4 = last lane low."""
import os, sys, time, ctypes as C
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
for prio in (sys.argv[1:] or ["0", "1", "3", "7", "0"]):
    os.environ["LBZAMD_PRIO"] = prio
    with lib.context(9, 1112) as ctx:
        ts = []
        for it in range(6):
            got = C.c_size_t(0)
            t = time.time(); rc = L.lbzamd_compress_host(ctx.h, hin.data_ptr(), n, hout.data_ptr(), bound, C.byref(got)); ts.append(time.time() - t)
        ts = sorted(ts[1:])
        print("LBZAMD_PRIO=%s: best %.1f ms (%.0f MB/s), median %.1f ms, rc %d out %d" % (prio, ts[0] * 1e3, n / ts[0] / 1e6, ts[len(ts) // 2] * 1e3, rc, got.value), flush=True)
