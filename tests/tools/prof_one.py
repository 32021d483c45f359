"""Ad-hoc: one compress call of N slabs of text for profiling."""
import sys
sys.path.insert(0, "/root/repo")
import torch, lbzip2_amd, ctypes as C
lib = lbzip2_amd.library()
g = C.CDLL("/root/repo/lbzip2_amd/host/libgen_inputs.so")
slabs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
n = slabs * 900000
buf = bytearray(n); cb = (C.c_uint8 * n).from_buffer(buf)
f = g.lbzgen_text if kind == "text" else g.lbzgen_rand
f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]; f(cb, n, 2); del cb
src = torch.frombuffer(buf, dtype=torch.uint8).cuda()
dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
ctx = lib.context(9, slabs, 0)
for _ in range(2):
    ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
s = ctx.stats(); print("ms", s.ms_collect, s.ms_bwt, s.ms_mtf, s.ms_encode)
