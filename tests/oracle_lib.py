"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when present, the
compiled reference (oracle/_ref/libref.so).  TEST INFRASTRUCTURE ONLY: imported by
tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke(); never by lbzip2_amd.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref.so")

MAX_SEL = 18002
MAX_ALPHA = 258


class CollectT(C.Structure):
    _fields_ = [("nblock", C.c_uint32), ("crc", C.c_uint32),
                ("inuse", C.c_uint8 * 256), ("consumed", C.c_size_t)]


class CodeT(C.Structure):
    _fields_ = [("num_trees", C.c_uint32), ("num_selectors", C.c_uint32),
                ("selector", C.c_uint8 * MAX_SEL),
                ("length", (C.c_uint8 * (MAX_ALPHA + 1)) * 6),
                ("code", (C.c_uint32 * (MAX_ALPHA + 1)) * 6),
                ("old2new", C.c_uint32 * 6), ("new2old", C.c_uint32 * 6),
                ("cost", C.c_uint32)]


class BlockT(C.Structure):
    _fields_ = [("nblock", C.c_uint32), ("crc", C.c_uint32), ("bwt_idx", C.c_uint32),
                ("nmtf", C.c_uint32), ("alpha", C.c_uint32),
                ("inuse", C.c_uint8 * 256), ("pc", CodeT),
                ("selector_mtf", C.c_uint8 * MAX_SEL),
                ("num_selectors_tx", C.c_uint32), ("tree_pad", C.c_uint32),
                ("out_len", C.c_uint32)]


def build_oracle():
    """(Re)build liboracle.so (and _ref when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        O = C.CDLL(ORACLE_SO)
        O.orc_collect.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_void_p, C.POINTER(CollectT)]
        O.orc_collect.restype = None
        O.orc_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        O.orc_crc32.restype = C.c_uint32
        O.orc_bwt.argtypes = [C.c_char_p, C.c_int32, C.c_void_p]
        O.orc_bwt.restype = C.c_int32
        O.orc_is_periodic.argtypes = [C.c_char_p, C.c_int32]
        O.orc_is_periodic.restype = C.c_int
        O.orc_mtf.argtypes = [C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        O.orc_mtf.restype = C.c_uint32
        O.orc_prefix_code.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint, C.POINTER(CodeT)]
        O.orc_prefix_code.restype = None
        O.orc_encode_block.argtypes = [C.c_void_p, C.POINTER(CollectT), C.c_uint, C.c_void_p, C.POINTER(BlockT)]
        O.orc_encode_block.restype = None
        O.orc_transmit.argtypes = [C.POINTER(BlockT), C.c_void_p, C.c_void_p]
        O.orc_transmit.restype = None
        O.orc_compress_stream.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
        O.orc_compress_stream.restype = C.c_size_t
        O.orc_gen_rand.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        O.orc_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        _oracle = O
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(REF_SO)
        R.ref_compress_stream.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.c_void_p, C.c_size_t]
        R.ref_compress_stream.restype = C.c_size_t
        R.encoder_alloc_size.argtypes = [C.c_ulong]
        R.encoder_alloc_size.restype = C.c_size_t
        R.encoder_init.argtypes = [C.c_void_p, C.c_ulong, C.c_uint]
        R.encoder_init.restype = None
        R.collect.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t)]
        R.collect.restype = C.c_int
        R.encode.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        R.encode.restype = C.c_size_t
        R.transmit.argtypes = [C.c_void_p, C.c_void_p]
        R.transmit.restype = C.c_void_p
        R.ref_bwt.argtypes = [C.c_char_p, C.c_int32, C.c_void_p]
        R.ref_bwt.restype = C.c_int32
        for name, rt in [("ref_nblock", C.c_uint32), ("ref_block_crc", C.c_uint32),
                         ("ref_bwt_idx", C.c_uint32), ("ref_nmtf", C.c_uint32),
                         ("ref_num_selectors", C.c_uint32), ("ref_num_trees", C.c_uint32),
                         ("ref_tree_pad", C.c_uint), ("ref_rle_state", C.c_int)]:
            f = getattr(R, name); f.argtypes = [C.c_void_p]; f.restype = rt
        for name in ["ref_block", "ref_inuse", "ref_mtfv", "ref_selector", "ref_selector_mtf",
                     "ref_tmap_new2old", "ref_tmap_old2new"]:
            f = getattr(R, name); f.argtypes = [C.c_void_p]; f.restype = C.c_void_p
        for name in ["ref_length", "ref_code"]:
            f = getattr(R, name); f.argtypes = [C.c_void_p, C.c_uint]; f.restype = C.c_void_p
        _ref = R
    return _ref


# ---------------------------------------------------------------- helpers
def _cap(n):
    return n + n // 20 + 100000


def orc_compress(data: bytes, level: int = 9) -> bytes:
    O = oracle()
    cap = _cap(len(data))
    out = C.create_string_buffer(cap)
    n = O.orc_compress_stream(data, len(data), level, out, cap, None)
    assert n > 0
    return out.raw[:n]


def ref_compress_seq(data: bytes, level: int = 9) -> bytes:
    """-u / --sequential blocking through the compiled reference (oracle/ref_probe.c: ref_compress_seq)"""
    R = ref()
    R.ref_compress_seq.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.c_void_p, C.c_size_t]
    R.ref_compress_seq.restype = C.c_size_t
    cap = _cap(len(data))
    out = C.create_string_buffer(cap)
    n = R.ref_compress_seq(bytes(data), len(data), level, out, cap)
    assert n > 0
    return out.raw[:n]


def orc_compress_seq(data: bytes, level: int = 9) -> bytes:
    O = oracle()
    O.orc_compress_seq.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p]
    O.orc_compress_seq.restype = C.c_size_t
    cap = _cap(len(data))
    out = C.create_string_buffer(cap)
    n = O.orc_compress_seq(bytes(data), len(data), level, out, cap, None)
    assert n > 0
    return out.raw[:n]


def ref_compress(data: bytes, level: int = 9) -> bytes:
    R = ref()
    cap = _cap(len(data))
    out = C.create_string_buffer(cap)
    n = R.ref_compress_stream(data, len(data), level, out, cap)
    assert n > 0
    return out.raw[:n]


class MtBlk(C.Structure):
    _fields_ = [("out_len", C.c_uint32), ("crc", C.c_uint32), ("bwt_idx", C.c_uint32), ("copies", C.c_uint32), ("slab", C.c_uint32), ("pad_", C.c_uint32)]


def _compress_mt(fn, data, level, nthreads):
    """The pthreads driver of oracle/cpu_mt.h: (stream bytes, [(out_len, crc)], seconds)."""
    M = level * 100000
    n = len(data)
    cap = _cap(n) + (n // M + 2) * 64
    out = C.create_string_buffer(cap)
    blocks = (MtBlk * (2 * (n // M + 2)))()
    nb = C.c_uint32()
    sec = C.c_double()
    if isinstance(data, (bytes, bytearray)) and not isinstance(data, bytes):
        src = (C.c_char * n).from_buffer(data)
    else:
        src = data
    fn.argtypes = [C.c_void_p if not isinstance(src, bytes) else C.c_char_p, C.c_size_t, C.c_uint, C.c_void_p, C.c_size_t,
                   C.c_uint, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    fn.restype = C.c_size_t
    m = fn(src, n, level, out, cap, nthreads, blocks, C.byref(nb), C.byref(sec))
    assert m > 0
    return out.raw[:m], [(blocks[i].out_len, blocks[i].crc, blocks[i].bwt_idx, blocks[i].copies, blocks[i].slab) for i in range(nb.value)], sec.value


def ref_compress_mt(data, level=9, nthreads=1, canon=False):
    """canon: origin pointers of exactly periodic blocks rewritten to the smallest equal row."""
    C.c_int.in_dll(ref(), "ref_canon").value = 1 if canon else 0
    try:
        return _compress_mt(ref().ref_compress_mt, data, level, nthreads)
    finally:
        C.c_int.in_dll(ref(), "ref_canon").value = 0


def orc_compress_mt(data, level=9, nthreads=1):
    return _compress_mt(oracle().orc_compress_mt, data, level, nthreads)


_gen = None


def gen_kind(kind, n, seed):
    """Workload generators of lbzip2_amd/host/gen_inputs.c (plain C, built with gcc): wiki, mixed, tar, text, rand."""
    global _gen
    if _gen is None:
        path = os.path.join(ROOT, "lbzip2_amd", "host", "libgen_inputs.so")
        if not os.path.exists(path):
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", path,
                                   os.path.join(ROOT, "lbzip2_amd", "host", "gen_inputs.c")])
        _gen = C.CDLL(path)
    buf = bytearray(n)
    if n:
        cb = (C.c_uint8 * n).from_buffer(buf)
        f = getattr(_gen, "lbzgen_" + kind)
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        f.restype = None
        f(cb, n, seed)
        del cb
    return buf


def gen_rand(n, seed):
    b = C.create_string_buffer(n)
    oracle().orc_gen_rand(b, n, seed)
    return b.raw


def gen_text(n, seed):
    b = C.create_string_buffer(n)
    oracle().orc_gen_text(b, n, seed)
    return b.raw


def orc_blocks(data: bytes, level: int = 9):
    """Per-block stage records from the oracle: list of dicts."""
    O = oracle()
    M = level * 100000
    out = []
    for off in range(0, len(data), M):
        slab = data[off:off + M]
        pos = 0
        while pos < len(slab):
            rest = slab[pos:]
            blk = C.create_string_buffer(M)
            c = CollectT()
            O.orc_collect(rest, len(rest), M, blk, C.byref(c))
            n = c.nblock
            T = blk.raw[:n]
            bwt = C.create_string_buffer(n)
            idx = O.orc_bwt(T, n, bwt)
            mtfv = (C.c_uint16 * (n + 1 + 50))()
            b = BlockT()
            O.orc_encode_block(blk, C.byref(c), 8, mtfv, C.byref(b))
            outb = C.create_string_buffer(b.out_len + 8)
            O.orc_transmit(C.byref(b), mtfv, outb)
            out.append(dict(consumed=c.consumed, nblock=n, crc=c.crc, inuse=bytes(c.inuse),
                            block=T, bwt=bwt.raw[:n], bwt_idx=idx, nmtf=b.nmtf,
                            mtfv=bytes(memoryview(mtfv).cast("B")[:2 * b.nmtf]),
                            alpha=b.alpha, num_trees=b.pc.num_trees,
                            num_selectors=b.num_selectors_tx, tree_pad=b.tree_pad,
                            selector=bytes(b.pc.selector[:b.pc.num_selectors]),
                            lengths=[bytes(b.pc.length[b.pc.new2old[t]][:b.alpha]) for t in range(b.pc.num_trees)],
                            out_len=b.out_len, out=outb.raw[:b.out_len],
                            periodic=bool(O.orc_is_periodic(T, n))))
            pos += c.consumed
    return out


def ref_blocks(data: bytes, level: int = 9):
    """Per-block stage records from the compiled reference."""
    R = ref()
    M = level * 100000
    out = []
    for off in range(0, len(data), M):
        slab = data[off:off + M]
        pos = 0
        while pos < len(slab):
            rest = slab[pos:]
            e = C.create_string_buffer(R.encoder_alloc_size(M))
            R.encoder_init(e, M, 8)
            left = C.c_size_t(len(rest))
            R.collect(e, rest, C.byref(left))
            consumed = len(rest) - left.value
            crc = C.c_uint32()
            size = R.encode(e, C.byref(crc))           # closes the open run first
            n = R.ref_nblock(e)
            T = C.string_at(R.ref_block(e), n)
            bwt = C.create_string_buffer(n)
            idx = R.ref_bwt(T, n, bwt)
            nm = R.ref_nmtf(e)
            mtfv = C.string_at(R.ref_mtfv(e), 2 * nm)
            alpha = int.from_bytes(mtfv[-2:], "little") + 1
            nt = R.ref_num_trees(e)
            ns = R.ref_num_selectors(e)
            n2o = (C.c_uint * 6).from_address(R.ref_tmap_new2old(e))
            lengths = [C.string_at(R.ref_length(e, n2o[t]), alpha) for t in range(nt)]
            sel = C.string_at(R.ref_selector(e), (nm + 49) // 50)
            buf = C.create_string_buffer((size + 3) // 4 * 4 + 8)
            R.transmit(e, buf)
            out.append(dict(consumed=consumed, nblock=n, crc=crc.value,
                            inuse=C.string_at(R.ref_inuse(e), 256), block=T,
                            bwt=bwt.raw[:n], bwt_idx=R.ref_bwt_idx(e), nmtf=nm, mtfv=mtfv,
                            alpha=alpha, num_trees=nt, num_selectors=ns,
                            tree_pad=R.ref_tree_pad(e), selector=sel, lengths=lengths,
                            out_len=size, out=buf.raw[:size]))
            pos += consumed
    return out
