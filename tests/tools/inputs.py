"""Ad-hoc inputs of the tuning scripts (not a test): the generators of gen_inputs.c by name, `pysrc` = Python sources of the
image, `realtar` = bench.py's tar of real headers and sources."""
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
_g = None


def pysrc(n):
    out = bytearray()
    for f in sorted(glob.glob("/usr/lib/python3*/**/*.py", recursive=True)) + sorted(glob.glob("/usr/local/lib/python3*/dist-packages/**/*.py", recursive=True)):
        try:
            out += open(f, "rb").read()
        except Exception:
            pass
        if len(out) >= n:
            break
    while len(out) < n:
        out += out[:n - len(out)]
    return out[:n]


def get(kind, n, seed=2):
    global _g
    if kind == "pysrc":
        return pysrc(n)
    if kind == "realtar":
        import bench
        made = bench.real_tar(n)
        data = made[0]
        while len(data) < n:
            data += data[:n - len(data)]
        return data[:n]
    if _g is None:
        _g = C.CDLL(os.path.join(ROOT, "lbzip2_amd", "host", "libgen_inputs.so"))
    buf = bytearray(n)
    cb = (C.c_uint8 * n).from_buffer(buf)
    f = getattr(_g, "lbzgen_" + kind)
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
    f(cb, n, seed)
    del cb
    return buf
