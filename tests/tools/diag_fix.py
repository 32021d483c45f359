"""Ad-hoc: phases of the deep-tie rounds (k_bwt_fix0 / k_bwt_fixr), summed over a block's segments and rounds."""
import ctypes as C
import os
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, lbzip2_amd  # noqa: E401,F401
if os.environ.get('LBZ_LIB'): lbzip2_amd.LIB_PATH = os.environ['LBZ_LIB']
lib = lbzip2_amd.library()
g = C.CDLL("/root/repo/lbzip2_amd/host/libgen_inputs.so")
kind = sys.argv[1] if len(sys.argv) > 1 else "wiki"
slabs = 256
n = slabs * 900000
buf = bytearray(n); cb = (C.c_uint8 * n).from_buffer(buf)
f = getattr(g, "lbzgen_" + kind); f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]; f(cb, n, 2); del cb
with lib.context(9, slabs) as ctx:
    ctx.run_stages(bytes(buf), 1)
    for b in range(0, 12, 2):
        bi = ctx.block_info(b)
        t = list(bi.fticks)
        print("blk", b, "n", bi.n, "rounds", bi.rounds, "tied after batch %d (%.1f%%)" % (t[1], 100.0 * t[1] / bi.n),
              "doubling rows / n = %.2f" % (bi.sort_elems / bi.n - 1.0),
              "| ms summed over segment workgroups: tie lists %.2f rounds %.2f | %d batches: load %.2f runs %.2f sort %.2f write %.2f"
              % (t[6] / 1e5, t[7] / 1e5, t[0], t[2] / 1e5, t[3] / 1e5, t[4] / 1e5, t[5] / 1e5))
