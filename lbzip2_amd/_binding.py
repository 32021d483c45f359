"""ctypes binding of the C ABI declared in include/lbzip2_amd.h.

Host-side mirror of the reference's work-unit interface (src/encode.h:22-38) and of its
compression pipeline (src/compress.c): same names, same argument meaning, same error
behaviour (the hot-path functions cannot fail; fatal problems abort, as lbzip2's fail() does).
No codec arithmetic happens in Python: every byte of the stream comes from the HIP kernels.
"""
import ctypes as C
import os

HEADER_SIZE = 4       # encode.h:23
TRAILER_SIZE = 10     # encode.h:24
CLUSTER_FACTOR = 8    # encode.h:22

STAGE_RLE, STAGE_BWT, STAGE_MTFV, STAGE_OUT = 0, 1, 2, 3

EXPORTS = [
    "encoder_alloc_size", "encoder_init", "collect", "encode", "transmit",
    "lbzamd_encoder_alloc_size", "lbzamd_encoder_init", "lbzamd_collect", "lbzamd_encode",
    "lbzamd_transmit", "lbzamd_encoder_abandon",
    "lbzamd_create", "lbzamd_destroy", "lbzamd_last_error", "lbzamd_compress_device",
    "lbzamd_compress_host", "lbzamd_bound", "lbzamd_get_stats", "lbzamd_stream", "lbzamd_slots", "lbzamd_round_shape", "lbzamd_set_sequential",
    "lbzamd_block_slots", "lbzamd_block_info_get", "lbzamd_read_stage", "lbzamd_run_stages",
    "lbzamd_compress_device_body", "lbzamd_compress_host_body", "lbzamd_fold_parts",
    "lbzamd_pinned_alloc", "lbzamd_pinned_free", "lbzamd_device_count",
    "lbzamd_dcreate", "lbzamd_ddestroy", "lbzamd_decompress_device", "lbzamd_decompress_host", "lbzamd_dget_stats",
    "lbzamd_decompress_alloc", "lbzamd_free", "lbzamd_last_error_code", "lbzamd_decompress_window",
]


def combine_crc(cc, c):
    """encode.h:38 -- cc' = rotl32(cc, 1) ^ ~c on 32-bit values."""
    return (((cc << 1) | (cc >> 31)) ^ c ^ 0xFFFFFFFF) & 0xFFFFFFFF


class Stats(C.Structure):
    _fields_ = [("n_in", C.c_uint64), ("n_rle", C.c_uint64), ("n_mtf", C.c_uint64),
                ("n_out", C.c_uint64), ("sort_elems", C.c_uint64),
                ("nblocks", C.c_uint32), ("nperiodic", C.c_uint32),
                ("ms_collect", C.c_float), ("ms_bwt", C.c_float), ("ms_mtf", C.c_float),
                ("ms_encode", C.c_float), ("ms_finish", C.c_float), ("ms_total", C.c_float),
                ("ms_bwt_part", C.c_float), ("ms_bwt_batch", C.c_float), ("ms_bwt_fix", C.c_float),
                ("seq_fast_links", C.c_uint32)]


class Part(C.Structure):
    """lbzamd_part: what the muxer of a multi-GPU job needs from a range (include/lbzip2_amd.h)."""
    _fields_ = [("bytes", C.c_uint64), ("nblocks", C.c_uint32), ("crc_fold", C.c_uint32)]


def fold_parts(cc, parts):
    """cc = rotl32(cc, nblocks mod 32) ^ crc_fold, range by range (parts: (nblocks, crc_fold) pairs)."""
    for nblocks, fold in parts:
        r = nblocks & 31
        cc = (((cc << r) | (cc >> (32 - r))) & 0xFFFFFFFF if r else cc) ^ fold
    return cc


class DResume(C.Structure):
    _fields_ = [("consumed_bit", C.c_uint64), ("started", C.c_uint32), ("in_stream", C.c_uint32), ("level", C.c_uint32), ("cc", C.c_uint32),
                ("stream_blocks", C.c_uint32), ("finished", C.c_uint32), ("nblocks_total", C.c_uint32), ("nstreams_total", C.c_uint32),
                ("base_bytes", C.c_uint64)]


class DStats(C.Structure):
    _fields_ = [("n_in", C.c_uint64), ("n_out", C.c_uint64), ("nblocks", C.c_uint32), ("nstreams", C.c_uint32),
                ("ms_scan", C.c_float), ("ms_huff", C.c_float), ("ms_sort", C.c_float), ("ms_walk", C.c_float),
                ("ms_emit", C.c_float), ("ms_total", C.c_float), ("ms_blocks", C.c_float)]


class BlockInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n", "crc", "consumed", "bwt_idx", "periodic", "nmtf",
                                           "alpha", "num_trees", "num_sel", "out_len", "err",
                                           "rounds", "sort_elems")] + [("ticks", C.c_uint32 * 8), ("fticks", C.c_uint32 * 16),
                                                                        ("inuse", C.c_uint8 * 256)]


class LbzError(RuntimeError):
    pass


class Library:
    """One loaded build of the C ABI."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise LbzError(
                f"{path} not found: the HIP extension is not built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C lbzip2_amd/csrc`). There is no CPU fallback.")
        self.path = path
        lib = C.CDLL(path)
        self.lib = lib
        vp, sz, u32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)
        szp = C.POINTER(C.c_size_t)
        for pre in ("", "lbzamd_"):
            getattr(lib, pre + "encoder_alloc_size").argtypes = [C.c_ulong]
            getattr(lib, pre + "encoder_alloc_size").restype = sz
            getattr(lib, pre + "encoder_init").argtypes = [vp, C.c_ulong, C.c_uint]
            getattr(lib, pre + "encoder_init").restype = None
            getattr(lib, pre + "collect").argtypes = [vp, vp, szp]
            getattr(lib, pre + "collect").restype = C.c_int
            getattr(lib, pre + "encode").argtypes = [vp, u32p]
            getattr(lib, pre + "encode").restype = sz
            getattr(lib, pre + "transmit").argtypes = [vp, vp]
            getattr(lib, pre + "transmit").restype = vp
        lib.lbzamd_encoder_abandon.argtypes = [vp]
        lib.lbzamd_encoder_abandon.restype = None
        lib.lbzamd_create.argtypes = [C.POINTER(vp), C.c_int, C.c_uint, C.c_uint, C.c_uint]
        lib.lbzamd_create.restype = C.c_int
        lib.lbzamd_destroy.argtypes = [vp]
        lib.lbzamd_destroy.restype = None
        lib.lbzamd_last_error.restype = C.c_char_p
        lib.lbzamd_compress_device.argtypes = [vp, vp, sz, vp, sz, szp]
        lib.lbzamd_compress_device.restype = C.c_int
        lib.lbzamd_compress_host.argtypes = [vp, vp, sz, vp, sz, szp]
        lib.lbzamd_compress_host.restype = C.c_int
        lib.lbzamd_compress_device_body.argtypes = [vp, vp, sz, vp, sz, szp, C.POINTER(Part)]
        lib.lbzamd_compress_device_body.restype = C.c_int
        lib.lbzamd_compress_host_body.argtypes = [vp, vp, sz, vp, sz, szp, C.POINTER(Part)]
        lib.lbzamd_compress_host_body.restype = C.c_int
        lib.lbzamd_fold_parts.argtypes = [C.c_uint32, C.POINTER(Part), sz]
        lib.lbzamd_fold_parts.restype = C.c_uint32
        lib.lbzamd_dcreate.argtypes = [C.POINTER(vp), C.c_int, C.c_uint]
        lib.lbzamd_dcreate.restype = C.c_int
        lib.lbzamd_ddestroy.argtypes = [vp]
        lib.lbzamd_ddestroy.restype = None
        lib.lbzamd_decompress_device.argtypes = [vp, vp, sz, vp, sz, szp]
        lib.lbzamd_decompress_device.restype = C.c_int
        lib.lbzamd_decompress_host.argtypes = [vp, vp, sz, vp, sz, szp]
        lib.lbzamd_decompress_host.restype = C.c_int
        lib.lbzamd_decompress_alloc.argtypes = [vp, vp, sz, C.POINTER(C.POINTER(C.c_uint8)), szp]
        lib.lbzamd_decompress_alloc.restype = C.c_int
        lib.lbzamd_free.argtypes = [vp]
        lib.lbzamd_free.restype = None
        lib.lbzamd_decompress_window.argtypes = [vp, vp, sz, C.c_int, C.POINTER(DResume), C.POINTER(C.POINTER(C.c_uint8)), szp]
        lib.lbzamd_decompress_window.restype = C.c_int
        lib.lbzamd_dget_stats.argtypes = [vp, C.POINTER(DStats)]
        lib.lbzamd_dget_stats.restype = C.c_int
        lib.lbzamd_bound.argtypes = [sz]
        lib.lbzamd_bound.restype = sz
        lib.lbzamd_get_stats.argtypes = [vp, C.POINTER(Stats)]
        lib.lbzamd_get_stats.restype = C.c_int
        lib.lbzamd_stream.argtypes = [vp]
        lib.lbzamd_stream.restype = vp
        lib.lbzamd_set_sequential.argtypes = [vp, C.c_int]
        lib.lbzamd_set_sequential.restype = C.c_int
        lib.lbzamd_slots.argtypes = [vp]
        lib.lbzamd_slots.restype = C.c_uint32
        lib.lbzamd_round_shape.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.lbzamd_round_shape.restype = None
        lib.lbzamd_block_slots.argtypes = [vp]
        lib.lbzamd_block_slots.restype = C.c_uint32
        lib.lbzamd_block_info_get.argtypes = [vp, C.c_uint32, C.POINTER(BlockInfo)]
        lib.lbzamd_block_info_get.restype = C.c_int
        lib.lbzamd_read_stage.argtypes = [vp, C.c_uint32, C.c_int, vp, sz]
        lib.lbzamd_read_stage.restype = C.c_long
        lib.lbzamd_run_stages.argtypes = [vp, vp, sz, C.c_int]
        lib.lbzamd_run_stages.restype = C.c_int

    def error(self):
        return (self.lib.lbzamd_last_error() or b"").decode()

    def bound(self, n):
        return self.lib.lbzamd_bound(n)

    def context(self, level=9, max_slabs=64, nslots=0, device=-1):
        return Context(self, level, max_slabs, nslots, device)

    def encoder(self, max_block_size):
        return Encoder(self, max_block_size)

    def decoder(self, max_blocks=64, device=-1):
        return Decoder(self, max_blocks, device)

    def decompress(self, data, max_blocks=None):
        """.bz2 bytes (one or several streams) -> bytes, block-parallel on the device."""
        with self.decoder(max_blocks or max(1, min(4096, len(data) // 20000 + 8))) as d:
            return d.decompress(data)

    # ---- whole-stream helpers -------------------------------------------------
    def compress(self, data, level=9, max_slabs=None, sequential=False):
        """bytes -> .bz2 bytes through the batch interface (sequential: the reference's -u blocking)."""
        M = level * 100000
        if max_slabs is None:
            max_slabs = max(1, min(1200, (len(data) + M - 1) // M))
        with self.context(level, max_slabs) as ctx:
            if sequential:
                ctx.set_sequential(True)
            return ctx.compress(data)

    def compress_workunits(self, data, level=9):
        """bytes -> .bz2 bytes through the drop-in work-unit interface, reproducing the call
        sequence of the reference's compress.c (slab split process.c:631, do_collect :73-118,
        do_transmit :210-228, do_reorder :238-250, header/trailer :291-321)."""
        M = level * 100000
        out = bytearray(b"BZh" + bytes([0x30 + level]))
        combined = 0
        for off in range(0, len(data), M):
            slab = data[off:off + M]
            pos = 0
            while pos < len(slab):
                enc = self.encoder(M)
                pos += enc.collect(slab[pos:])
                size, crc = enc.encode()
                out += enc.transmit()[:size]
                combined = combine_crc(combined, crc)
        out += bytes([0x17, 0x72, 0x45, 0x38, 0x50, 0x90]) + combined.to_bytes(4, "big")
        return bytes(out)


class Context:
    """Batch interface: a device context holding max_slabs resident slabs."""

    def __init__(self, library, level=9, max_slabs=64, nslots=0, device=-1):
        self.L = library
        self.level = level
        self.h = C.c_void_p()
        if library.lib.lbzamd_create(C.byref(self.h), device, level, max_slabs, nslots):
            raise LbzError("lbzamd_create: " + library.error())

    def close(self):
        if self.h:
            self.L.lib.lbzamd_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compress_body(self, data):
        """bytes-like -> (body bytes, nblocks, crc_fold): the blocks of the range only, for a muxer."""
        cap = self.L.bound(len(data))
        out = C.create_string_buffer(cap)
        n = C.c_size_t()
        part = Part()
        buf = (C.c_char * len(data)).from_buffer_copy(bytes(data)) if len(data) else None
        if self.L.lib.lbzamd_compress_host_body(self.h, buf, len(data), out, cap, C.byref(n), C.byref(part)):
            raise LbzError("lbzamd_compress_host_body: " + self.L.error())
        return out.raw[:n.value], part.nblocks, part.crc_fold

    def compress_device_body(self, d_in, length, d_out, out_cap):
        """Device-resident range -> (bytes written, nblocks, crc_fold)."""
        n = C.c_size_t()
        part = Part()
        if self.L.lib.lbzamd_compress_device_body(self.h, C.c_void_p(d_in), length, C.c_void_p(d_out),
                                                  out_cap, C.byref(n), C.byref(part)):
            raise LbzError("lbzamd_compress_device_body: " + self.L.error())
        return n.value, part.nblocks, part.crc_fold

    def compress(self, data):
        """bytes-like -> .bz2 bytes.  No copy on the way in (the C ABI reads the object's own buffer);
        the output buffer is kept with the context and one copy of the result is returned."""
        cap = self.L.bound(len(data))
        if getattr(self, "_out_cap", 0) < cap:
            self._out = C.create_string_buffer(cap)
            self._out_cap = cap
        n = C.c_size_t()
        if not isinstance(data, bytes):
            try:
                keep = (C.c_char * len(data)).from_buffer(data) if len(data) else None   # writable buffers (bytearray, numpy)
            except TypeError:
                data = bytes(data)                                                  # read-only views: one copy
        if isinstance(data, bytes):
            keep = data
            buf = C.cast(C.c_char_p(data), C.c_void_p) if data else None      # pointer into the bytes object
        else:
            buf = C.cast(keep, C.c_void_p) if keep is not None else None
        if self.L.lib.lbzamd_compress_host(self.h, buf, len(data), self._out, cap, C.byref(n)):
            raise LbzError("lbzamd_compress_host: " + self.L.error())
        return C.string_at(self._out, n.value)

    def compress_host_ptr(self, h_in, length, h_out, out_cap):
        """Host buffers by address (e.g. pinned torch tensors' .data_ptr()): H2D per round on the round's
        stream, kernels, one D2H of the stream.  Returns the stream length."""
        n = C.c_size_t()
        if self.L.lib.lbzamd_compress_host(self.h, C.c_void_p(h_in), length, C.c_void_p(h_out), out_cap, C.byref(n)):
            raise LbzError("lbzamd_compress_host: " + self.L.error())
        return n.value

    def set_sequential(self, on=True):
        """the reference's -u / --sequential: blocks are cut where they are full (compress.c:129-198)"""
        if self.L.lib.lbzamd_set_sequential(self.h, 1 if on else 0):
            raise LbzError("lbzamd_set_sequential: " + self.L.error())

    def compress_device(self, d_in, length, d_out, out_cap):
        """d_in/d_out: integer device addresses (e.g. torch tensor .data_ptr())."""
        n = C.c_size_t()
        if self.L.lib.lbzamd_compress_device(self.h, C.c_void_p(d_in), length, C.c_void_p(d_out),
                                             out_cap, C.byref(n)):
            raise LbzError("lbzamd_compress_device: " + self.L.error())
        return n.value

    def stats(self):
        s = Stats()
        self.L.lib.lbzamd_get_stats(self.h, C.byref(s))
        return s

    @property
    def nslots(self):
        return self.L.lib.lbzamd_slots(self.h)

    def round_shape(self, blocks, overlapped):
        """(segment workgroups per block in the sorting kernels, workgroups per block in the partition) of such a round."""
        a, b = C.c_uint32(0), C.c_uint32(0)
        self.L.lib.lbzamd_round_shape(self.h, blocks, 1 if overlapped else 0, C.byref(a), C.byref(b))
        return a.value, b.value

    @property
    def stream(self):
        return self.L.lib.lbzamd_stream(self.h)

    # ---- stage access (parity tests) ----
    def run_stages(self, data, upto=3):
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        if self.L.lib.lbzamd_run_stages(self.h, buf, len(data), upto):
            raise LbzError("lbzamd_run_stages: " + self.L.error())

    def block_slots(self):
        return self.L.lib.lbzamd_block_slots(self.h)

    def block_info(self, blk):
        bi = BlockInfo()
        if self.L.lib.lbzamd_block_info_get(self.h, blk, C.byref(bi)):
            raise LbzError("lbzamd_block_info_get: " + self.L.error())
        return bi

    def read_stage(self, blk, stage, cap):
        buf = C.create_string_buffer(max(cap, 1))
        n = self.L.lib.lbzamd_read_stage(self.h, blk, stage, buf, cap)
        if n < 0:
            raise LbzError("lbzamd_read_stage: " + self.L.error())
        return buf.raw[:n]

    def blocks(self, data, upto=3):
        """Per-block stage records of one chunk of input (list of dicts, stream order)."""
        self.run_stages(data, upto)
        out = []
        for blk in range(self.block_slots()):
            bi = self.block_info(blk)
            if bi.n == 0:
                continue
            rec = dict(blk=blk, nblock=bi.n, crc=bi.crc, consumed=bi.consumed, inuse=bytes(bi.inuse),
                       block=self.read_stage(blk, STAGE_RLE, bi.n), err=bi.err)
            if upto >= 1:
                rec.update(bwt=self.read_stage(blk, STAGE_BWT, bi.n), bwt_idx=bi.bwt_idx,
                           periodic=bool(bi.periodic), rounds=bi.rounds)
            if upto >= 2:
                rec.update(nmtf=bi.nmtf, alpha=bi.alpha,
                           mtfv=self.read_stage(blk, STAGE_MTFV, 2 * bi.nmtf))
            if upto >= 3:
                rec.update(num_trees=bi.num_trees, num_selectors=bi.num_sel, out_len=bi.out_len,
                           out=self.read_stage(blk, STAGE_OUT, bi.out_len))
            out.append(rec)
        return out


class Decoder:
    """The inverse path: every block of a stream decoded at once, one block per workgroup."""

    def __init__(self, library, max_blocks=64, device=-1):
        self.L = library
        self.h = C.c_void_p()
        if library.lib.lbzamd_dcreate(C.byref(self.h), device, max_blocks):
            raise LbzError("lbzamd_dcreate: " + library.error())

    def close(self):
        if self.h:
            self.L.lib.lbzamd_ddestroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decompress(self, data, out_cap=None):
        data = bytes(data)
        n = C.c_size_t()
        if out_cap is None:                                   # size unknown: one pass, the library allocates
            p = C.POINTER(C.c_uint8)()
            rc = self.L.lib.lbzamd_decompress_alloc(self.h, data, len(data), C.byref(p), C.byref(n))
            if rc:
                why = self.L.error()
                e = LbzError("lbzamd_decompress_alloc: " + why)
                e.decoded_in_front = b""                      # -3: the whole blocks in front of the one that was refused
                if p:
                    e.decoded_in_front = C.string_at(p, n.value)
                    self.L.lib.lbzamd_free(p)
                raise e
            try:
                return C.string_at(p, n.value)
            finally:
                self.L.lib.lbzamd_free(p)
        out = C.create_string_buffer(max(1, out_cap))
        rc = self.L.lib.lbzamd_decompress_host(self.h, data, len(data), out, out_cap, C.byref(n))
        if rc:
            raise LbzError("lbzamd_decompress_host: " + self.L.error())
        return out.raw[:n.value]

    def decompress_windows(self, data, window):
        """The input taken `window` bytes at a time (lbzamd_decompress_window), as lbzamd_io_decompress does: the caller's loop of
        include/lbzip2_amd.h -- keep the bytes from consumed_bit / 8 on, read more behind them, grow a window that held no
        whole block.  Returns (bytes, windows taken); raises LbzError with .decoded_in_front like decompress()."""
        data = bytes(data)
        rs = DResume()
        out = bytearray()
        pos = 0                       # bytes of `data` "read" so far
        buf = b""
        calls = 0
        while True:
            want = max(window, 1)
            take = data[pos:pos + max(0, want - len(buf))]
            pos += len(take)
            buf += take
            final = pos >= len(data)
            p = C.POINTER(C.c_uint8)(); n = C.c_size_t()
            rc = self.L.lib.lbzamd_decompress_window(self.h, buf, len(buf), 1 if final else 0, C.byref(rs), C.byref(p), C.byref(n))
            calls += 1
            got = C.string_at(p, n.value) if p else b""
            if p:
                self.L.lib.lbzamd_free(p)
            if rc:
                e = LbzError("lbzamd_decompress_window: " + self.L.error())
                e.decoded_in_front = bytes(out) + got
                raise e
            out += got
            if final:
                return bytes(out), calls
            keep = rs.consumed_bit // 8
            if keep == 0:
                window *= 2
            buf = buf[keep:]

    def decompress_device(self, d_in, length, d_out, out_cap):
        n = C.c_size_t()
        rc = self.L.lib.lbzamd_decompress_device(self.h, C.c_void_p(d_in), length, C.c_void_p(d_out), out_cap, C.byref(n))
        if rc:
            raise LbzError("lbzamd_decompress_device: " + self.L.error())
        return n.value

    def stats(self):
        s = DStats()
        self.L.lib.lbzamd_dget_stats(self.h, C.byref(s))
        return s


def compress_workunits_seq(library, data, level=9):
    """bytes -> .bz2 bytes through the drop-in symbols, called as the reference's -u mode calls them
    (compress.c:129-198 do_collect_seq): ONE encoder keeps collecting slab after slab until its block is full."""
    M = level * 100000
    out = bytearray(b"BZh" + bytes([0x30 + level]))
    combined = 0
    slabs = [data[off:off + M] for off in range(0, len(data), M)]
    k, pos = 0, 0                       # current slab, position inside it
    while k < len(slabs):
        enc = library.encoder(M)
        full = False
        while not full and k < len(slabs):
            took, full = enc.collect_full(slabs[k][pos:])
            pos += took
            if pos == len(slabs[k]):
                k, pos = k + 1, 0
        size, crc = enc.encode()
        out += enc.transmit()[:size]
        combined = combine_crc(combined, crc)
    out += bytes([0x17, 0x72, 0x45, 0x38, 0x50, 0x90]) + combined.to_bytes(4, "big")
    return bytes(out)


class Encoder:
    """The reference's work unit (encode.h:27-33): caller-allocated opaque state."""

    def __init__(self, library, max_block_size, cluster_factor=CLUSTER_FACTOR):
        self.L = library
        self.mbs = max_block_size
        self.state = C.create_string_buffer(library.lib.encoder_alloc_size(max_block_size))
        library.lib.encoder_init(self.state, max_block_size, cluster_factor)
        self.size = 0

    def collect(self, buf):
        """Returns the number of bytes consumed from buf."""
        return self.collect_full(buf)[0]

    def collect_full(self, buf):
        """(bytes consumed, the reference's return value: block full).  May be called again on the same state
        with further input until the block is full (compress.c:160-170, the -u mode)."""
        left = C.c_size_t(len(buf))
        b = (C.c_char * max(1, len(buf))).from_buffer_copy(bytes(buf) or b"\0")
        full = self.L.lib.collect(self.state, b, C.byref(left))
        return len(buf) - left.value, bool(full)

    def encode(self):
        crc = C.c_uint32()
        self.size = self.L.lib.encode(self.state, C.byref(crc))
        return self.size, crc.value

    def transmit(self):
        buf = C.create_string_buffer((self.size + 3) // 4 * 4)
        self.L.lib.transmit(self.state, buf)
        return buf.raw
