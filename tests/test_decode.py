"""The inverse path: block-parallel decoding of .bz2 streams through the C ABI
(lbzamd_decompress_*).  Oracles: the input itself (round trips of this library's, the reference's and Python
bz2's compressors -- the latter two give bit-aligned blocks and multi-stream files) and error behaviour on
damaged streams.  CPU: the kernel sources under the emulator; GPU: full-size configurations (C5's round trip)."""
import bz2
import os
import subprocess

import pytest

import oracle_lib as L
from golden_util import gen
from lbzip2_amd._binding import LbzError, Library

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "WG=1024"])
    return Library(os.path.join(EMU_DIR, "_build", "liblbzamd_emu_1024.so"))


CASES = [("empty", b"", 9), ("one", b"a", 9), ("banana", b"banana", 9), ("aaaa", b"a" * 4, 9), ("run259", b"a" * 259 + b"b", 9),
         ("runs", b"ab" * 3 + b"c" * 300 + b"d" + b"e" * 4, 1), ("zeros", bytes(10000), 9), ("all256", bytes(range(256)) * 3, 9),
         ("text3k", bytes(gen("text", 3000, 5)), 1), ("rand30k", bytes(gen("rand", 30000, 3)), 9),
         ("wiki260k", bytes(gen("wiki", 260000, 3)), 1), ("runs150k", bytes(gen("runs", 150000, 4)), 1),
         ("abab", b"ab" * 5000, 9)]


@pytest.mark.parametrize("name,data,level", CASES, ids=[c[0] for c in CASES])
def test_round_trip_of_oracle_streams(emu, name, data, level):
    assert emu.decompress(L.orc_compress(data, level)) == data


def test_fuzz_slice(emu):
    """A slice of tests/fuzz_gpu.py::run_decode small enough for the emulator: structured random inputs, streams of
    Python's bz2 and of this library, one or two streams per file."""
    import fuzz_gpu
    assert fuzz_gpu.run_decode(emu, 31, 40, small=True) == []


def test_unaligned_stream_pointer(emu):
    """lbzamd_decompress_device on a stream that starts at any byte of an 8-byte word (the magic scan reads aligned words
    and must neither see the bytes in front of the stream -- here they spell the start of a block magic -- nor miss its end)."""
    import ctypes as C
    data = bytes(gen("text", 5000, 3))
    z = bz2.compress(data, 1) + bz2.compress(b"xyz" * 100, 9)
    want = data + b"xyz" * 100
    f = emu.lib.lbzamd_decompress_device
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    for off in range(9):
        raw = C.create_string_buffer((b"\x31\x41\x59" * 3)[:off] + z + b"\x55" * 16)
        out = C.create_string_buffer(len(want) + 64)
        n = C.c_size_t()
        with emu.decoder(8) as d:
            assert f(d.h, C.addressof(raw) + off, len(z), C.addressof(out), len(want) + 64, C.byref(n)) == 0, off
        assert out.raw[:n.value] == want, off


def test_wide_workgroups(emu, monkeypatch):
    """Files of few blocks are decoded by 1024-thread workgroups (k_dblock_w; lbz_api.hip picks by the block count,
    LBZAMD_DWIDE forces either): same bytes.  The rest of this file runs the 256-thread kernel under the emulator (its
    1024 fibers per block are slow) -- on the GPU both are taken as the block counts have it."""
    d2 = bytes(gen("wiki", 120000, 4))
    for width in ("1", "2"):                                          # 1024 and 512 threads (k_dblock_w, k_dblock_m)
        monkeypatch.setenv("LBZAMD_DWIDE", width)
        for name, data, level in CASES:
            if name in ("empty", "banana", "run259", "zeros", "all256", "text3k", "rand30k", "abab"):
                assert emu.decompress(L.orc_compress(data, level)) == data, (name, width)
        assert emu.decompress(bz2.compress(d2, 1) + L.orc_compress(b"tail" * 100, 9)) == d2 + b"tail" * 100, width


def test_reference_and_python_streams(emu):
    """bit-aligned blocks (bzip2's own compressor), several streams in one file, the reference's stream"""
    d1, d2 = bytes(gen("text", 250000, 9)), bytes(gen("wiki", 120000, 4))
    z = bz2.compress(d1, 1) + bz2.compress(b"tail" * 100, 9) + bz2.compress(b"", 5) + L.orc_compress(d2, 2)
    assert emu.decompress(z) == d1 + b"tail" * 100 + d2
    if L.have_ref():
        assert emu.decompress(L.ref_compress(d2, 1)) == d2
    assert emu.decompress(bz2.compress(d2, 9) + b"\0" * 7) == d2            # trailing garbage is ignored


def expand_cases():
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "expand_cases.json")
    return json.load(open(path))["cases"]


def damaged_cases():
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "damaged_cases.json")
    return json.load(open(path))["cases"]


def check_expand_case(lib, c):
    """one case of the reference's decompressor suite (its tests/README: 32767 selectors, 20-bit codes, a zip bomb,
    randomised and cyclic blocks, the largest origin pointer, concatenated streams, gaps and trailing garbage, block
    overruns, the CVE-2010-0405 stream, block and stream CRC errors, truncated and empty files): accepted or refused
    as the compiled reference does, same bytes out"""
    import hashlib
    z = bytes.fromhex(c["bz2_hex"])
    if c["ok"]:
        out = lib.decompress(z)
        assert len(out) == c["out_len"] and hashlib.md5(out).hexdigest() == c["out_md5"], c["name"]
    else:
        with pytest.raises(LbzError):
            lib.decompress(z)


@pytest.mark.parametrize("c", [c for c in expand_cases() if c["out_len"] < 10**6], ids=lambda c: c["name"])
def test_reference_expand_suite(emu, c):
    check_expand_case(emu, c)


def test_damaged_streams_are_refused_with_the_reference_s_error(emu):
    """tests/golden/damaged_cases.json through the library call: refused (or, the one good stream, decoded), and
    lbzamd_last_error_code() is the reference's `enum error` value of the diagnostic the compiled reference printed
    (common.h:54-76; tests/test_cli.py compares the command's words)."""
    names = ["bad stream header magic", "bad block header magic", "empty source alphabet", "bad number of trees", "no coding groups",
             "invalid selector", "invalid delta code", "invalid prefix code", "incomplete prefix code", "empty block", "unterminated block",
             "missing run length", "block CRC mismatch", "stream CRC mismatch", "block overflow", "primary index too large", "unexpected end of file"]
    for c in damaged_cases():
        check_expand_case(emu, c)
        if not c["ok"] and "compressed data error: " in c["ref_message"]:
            want = [3 + names.index(m.split("compressed data error: ")[1]) for m in [c["ref_message"]] + c.get("also", [])]
            assert emu.lib.lbzamd_last_error_code() in want, (c["name"], emu.lib.lbzamd_last_error_code(), want)


def _bytes_in_front_of_the_damage(lib):
    """A stream of four blocks, a bit flipped inside each block's payload in turn (and in the first block's header, and the
    stream cut inside the last block): refused, and the bytes handed back with the error are the whole blocks in front of
    the one that was refused -- a prefix of the data, cut at a block boundary."""
    data = bytes(gen("text", 350000, 5))
    z = bz2.compress(data, 1)
    for where, nblocks in ((0.1, 0), (0.4, 1), (0.7, 2), (0.95, 3)):
        b = bytearray(z)
        b[int(len(b) * where)] ^= 0x20
        with pytest.raises(LbzError) as e:
            lib.decompress(bytes(b))
        got = e.value.decoded_in_front
        assert data.startswith(got) and (len(got) == 0) == (nblocks == 0), (where, len(got))
        if nblocks:
            assert abs(len(got) - nblocks * 100000) < 200, (where, len(got))          # whole blocks (a block's decoded size: 100 000 less what its runs save)
    with pytest.raises(LbzError) as e:
        lib.decompress(z[:len(z) * 9 // 10])
    assert data.startswith(e.value.decoded_in_front) and abs(len(e.value.decoded_in_front) - 300000) < 200
    with pytest.raises(LbzError) as e:                        # a good stream, then one whose STREAM CRC is wrong: its blocks are good, all of it comes back
        lib.decompress(z + bz2.compress(b"tail", 9)[:-4] + b"\0\0\0\0")
    assert e.value.decoded_in_front == data + b"tail"
    bad = bytearray(bz2.compress(b"tail" * 100, 9)); bad[12] ^= 1          # ... and one whose BLOCK CRC is wrong: the first stream's bytes
    with pytest.raises(LbzError) as e:
        lib.decompress(z + bytes(bad))
    assert e.value.decoded_in_front == data


def test_bytes_in_front_of_the_damage(emu):
    _bytes_in_front_of_the_damage(emu)


def test_hand_made_blocks_with_random_tables(emu):
    """tests/craft_bz2.py: random_block_streams -- valid streams that no encoder writes (any alphabet, up to six tables of random
    shape with codes of up to 20 bits, a random table per group, runs and counts of every kind): decoded to the bytes they were
    made from, and as Python's bz2 decodes them.  (A differential run of 2 000 such streams, some of them made invalid, against
    the compiled reference program found no difference: DESIGN 10.)"""
    import craft_bz2
    with emu.decoder(4) as d:
        for i, (z, data) in enumerate(craft_bz2.random_block_streams(5, 80)):
            assert bz2.decompress(z) == data, i
            assert d.decompress(z) == data, i


def suite_streams(step):
    import tarfile
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "suite_inputs.tar")
    with tarfile.open(path) as t:
        members = [m for m in t.getmembers() if m.isfile()]
        for m in members[::step]:
            yield m.name, t.extractfile(m).read()


def test_reference_compress_suite_streams_decode(emu):
    """the .bz2 files of the reference's compressor suites (fuzz-divbwt, fuzz-collect, manual-compress) are streams
    too: every 12th of the 1093 here (all of them on the GPU), decoded and compared with Python's bz2"""
    with emu.decoder(64) as d:
        for name, z in suite_streams(12):
            assert d.decompress(z) == bz2.decompress(z), name


def test_randomisation_table_is_libbz2s():
    import ctypes
    import ctypes.util
    import re
    name = ctypes.util.find_library("bz2")
    if not name:
        pytest.skip("no libbz2 here")
    tab = list((ctypes.c_int * 512).in_dll(ctypes.CDLL(name), "BZ2_rNums"))
    src = open(os.path.join(os.path.dirname(EMU_DIR), "..", "lbzip2_amd", "csrc", "lbz_rand.h")).read()
    mine = [int(x) for x in re.findall(r"\d+", src[src.index("LBZ_RNUMS[512] = {") + 18:])][:512]
    assert mine == tab


def test_more_blocks_than_the_context_holds(emu):
    """max_blocks = 2: seven blocks are taken in four passes, offsets carry over"""
    d = bytes(gen("wiki", 650000, 8))
    z = L.orc_compress(d, 1)
    with emu.decoder(2) as dec:
        assert dec.decompress(z) == d
        st = dec.stats()
    assert st.nblocks == 7 and st.nstreams == 1 and st.n_out == len(d)


def test_one_pass_when_the_size_is_unknown(emu):
    """lbzamd_decompress_alloc (what Decoder.decompress(bytes) calls): the device output buffer grows between the block
    passes -- also when the blocks come in several passes and the first guess (4 x the compressed size) is far too small"""
    d = bytes(100_000) * 9 + bytes(gen("wiki", 120000, 4))              # ratio > 1000 on the zeros
    z = L.orc_compress(d, 1)
    for max_blocks in (64, 3):
        with emu.decoder(max_blocks) as dec:
            assert dec.decompress(z) == d
            assert dec.decompress(L.orc_compress(b"", 9)) == b""
            assert dec.decompress(z, out_cap=len(d)) == d                    # the caller's buffer, exact size
            with pytest.raises(LbzError):
                dec.decompress(z, out_cap=len(d) - 1)


def test_damaged_streams_are_refused(emu):
    z = bytearray(L.orc_compress(bytes(gen("text", 120000, 6)), 1))
    for mutate in (lambda b: b.__setitem__(len(b) // 2, b[len(b) // 2] ^ 0x10),       # payload bit: block CRC or code error
                   lambda b: b.__setitem__(len(b) - 1, b[-1] ^ 1),                      # stream CRC
                   lambda b: b.__delitem__(slice(len(b) - 20, len(b))),                  # truncated
                   lambda b: b.__setitem__(0, ord("X"))):                                # no header
        bad = bytearray(z)
        mutate(bad)
        with pytest.raises(LbzError):
            emu.decompress(bytes(bad))


def check_planted_magics(lib):
    """A 48-bit block magic or end-of-stream magic INSIDE a block's payload (tests/craft_bz2.py; libbzip2 decodes
    these files): the scan reports it, the chain header -> block -> where the block ended -> next magic must not
    follow it.  Also between other blocks and streams, and with real garbage between two blocks (refused)."""
    import craft_bz2 as cb
    for magic in (cb.BLOCK_MAGIC, cb.END_MAGIC):
        z, want = cb.crafted_stream(magic)
        assert bz2.decompress(z) == want
        assert lib.decompress(z) == want, hex(magic)
        d1 = bytes(gen("text", 150000, 12))
        multi = bz2.compress(d1, 1) + z + L.orc_compress(d1[:50000], 1) + b"garbage" + z
        assert lib.decompress(multi) == d1 + want + d1[:50000]
    # two byte-aligned blocks with four stray bytes between them: no magic where block 1 ends
    d = bytes(gen("text", 230000, 3))
    good = L.orc_compress(d, 1)
    cut = good.index(bytes.fromhex("314159265359"), 20)
    with pytest.raises(LbzError):
        lib.decompress(good[:cut] + b"\0\0\0\0" + good[cut:])


def test_magic_inside_a_payload(emu):
    check_planted_magics(emu)


@pytest.mark.gpu
def test_magic_inside_a_payload_on_the_gpu():
    import lbzip2_amd
    check_planted_magics(lbzip2_amd.library())


def test_cli_decompress(emu):
    """lbzip2_amd/host/lbzamd_compress.c -d: file in, every block at once, file out"""
    d = bytes(gen("wiki", 200000, 5))
    z = bz2.compress(d, 1) + L.orc_compress(b"x" * 1000, 9)
    exe = os.path.join(EMU_DIR, "_build", "lbzamd_compress_emu")
    r = subprocess.run([exe, "-d"], input=z, capture_output=True, timeout=600)
    assert r.returncode == 0 and r.stdout == d + b"x" * 1000, r.stderr[-500:]
    r = subprocess.run([exe, "-d"], input=z[:-3], capture_output=True, timeout=600)
    assert r.returncode != 0 and b"lbzamd" in r.stderr


@pytest.mark.gpu
def test_cli_decompress_on_the_gpu(tmp_path):
    """the host driver's two directions back to back: file -> .bz2 -> file"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "lbzip2_amd", "host", "lbzamd_compress")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "lbzip2_amd", "csrc")])
    data = L.gen_kind("wiki", 300_000_000, 6)
    src, z, back = tmp_path / "in.bin", tmp_path / "in.bz2", tmp_path / "back.bin"
    src.write_bytes(data)
    subprocess.check_call([exe, "-9", "-f", str(src), "-o", str(z)], timeout=600)
    subprocess.check_call([exe, "-d", "-f", str(z), "-o", str(back)], timeout=600)
    assert hashlib.md5(back.read_bytes()).digest() == hashlib.md5(data).digest()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,seed,level", [("wiki", 100_000_000, 1, 9), ("rand", 30_000_000, 4, 9), ("mixed", 120_000_000, 3, 1),
                                               ("tar", 175_000_000, 5, 9), ("tar", 1_400_000_000, 5, 9), ("text", 50_000_000, 2, 5)])
def test_round_trip_full_size(kind, n, seed, level):
    """compress on the device, decode on the device, compare on the device (C5: tar-like stream, round trip)"""
    import torch
    import lbzip2_amd
    lib = lbzip2_amd.library()
    data = L.gen_kind(kind, n, seed)
    src = torch.frombuffer(data, dtype=torch.uint8).cuda()
    z = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    M = level * 100000
    with lib.context(level, min(2400, (n + M - 1) // M)) as ctx:
        m = ctx.compress_device(src.data_ptr(), n, z.data_ptr(), z.numel())
    out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    with lib.decoder(2400) as d:
        k = d.decompress_device(z.data_ptr(), m, out.data_ptr(), out.numel())
        st = d.stats()
    assert k == n and bool(torch.equal(out[:n], src))
    assert st.nblocks >= (n + M - 1) // M and st.nstreams == 1


@pytest.mark.gpu
@pytest.mark.parametrize("c", expand_cases(), ids=lambda c: c["name"])
def test_reference_expand_suite_on_the_gpu(c):
    import lbzip2_amd
    check_expand_case(lbzip2_amd.library(), c)


@pytest.mark.gpu
def test_reference_compress_suite_streams_decode_on_the_gpu():
    import lbzip2_amd
    with lbzip2_amd.library().decoder(64) as d:
        for name, z in suite_streams(1):
            assert d.decompress(z) == bz2.decompress(z), name


@pytest.mark.gpu
def test_bytes_in_front_of_the_damage_on_the_gpu():
    import lbzip2_amd
    _bytes_in_front_of_the_damage(lbzip2_amd.library())


@pytest.mark.gpu
def test_hand_made_blocks_with_random_tables_on_the_gpu():
    import craft_bz2
    import lbzip2_amd
    with lbzip2_amd.library().decoder(4) as d:
        for i, (z, data) in enumerate(craft_bz2.random_block_streams(6, 300)):
            assert d.decompress(z) == data, i


@pytest.mark.gpu
def test_fuzz_on_the_gpu():
    """tests/fuzz_gpu.py::run_decode: structured random inputs up to 400 000 bytes, streams of Python's bz2 and of this
    library, one or two streams per file, through both workgroup sizes of k_dblock."""
    import fuzz_gpu
    import lbzip2_amd
    assert fuzz_gpu.run_decode(lbzip2_amd.library(), 41, 120, both_widths=True) == []


@pytest.mark.gpu
def test_foreign_streams_on_the_gpu():
    """the reference's and Python bz2's streams (bit-aligned blocks, several streams), and a damaged one"""
    import lbzip2_amd
    lib = lbzip2_amd.library()
    d1, d2 = bytes(gen("wiki", 3_000_000, 7)), bytes(gen("text", 1_000_000, 8))
    z = bz2.compress(d1, 9) + bz2.compress(d2, 1)
    if L.have_ref():
        z += L.ref_compress(d2, 3)
        d2x = d2 + d2
    else:
        d2x = d2
    assert lib.decompress(z) == d1 + d2x
    bad = bytearray(z); bad[len(bad) // 3] ^= 4
    with pytest.raises(lbzip2_amd.LbzError):
        lib.decompress(bytes(bad))


def test_retrieve_error_just_in_front_of_a_chunk_boundary():
    """decode.h's retrieve() (lbz_api.hip section D) fed in two pieces, with a malformed field (7 coding tables: ERR_TREES,
    decode.c) a few bits in front of the first piece's end.  The kernel reads the 15-bit selector count behind the field
    before it looks at either, so a first piece that ends inside that count is "ran past the bits there are": MORE; the
    second call sees the same error at a bit position that now lies in front of its own input -- it must come back as the
    reference's error code with the caller's bitstream left consumed, not as a position computed from a negative offset
    (round-4 review: wild pointer).  A first piece that holds both fields gets the error at once, the stream left where
    the walk stopped (as the reference's retrieve() leaves it)."""
    import ctypes as C
    import struct
    import subprocess
    emu_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
    subprocess.check_call(["make", "-s", "-C", emu_dir, "WG=1024"])
    lib = C.CDLL(os.path.join(emu_dir, "_build", "liblbzamd_emu_1024.so"))

    class BitStream(C.Structure):           # decode.h:38-45
        _fields_ = [("live", C.c_uint), ("buff", C.c_uint64), ("block", C.c_void_p), ("data", C.POINTER(C.c_uint32)),
                    ("limit", C.POINTER(C.c_uint32)), ("eof", C.c_bool)]

    class DecoderState(C.Structure):        # decode.h:48-66
        _fields_ = [("internal_state", C.c_void_p), ("rand", C.c_bool), ("bwt_idx", C.c_uint), ("block_size", C.c_uint),
                    ("crc", C.c_uint32), ("ftab", C.c_uint32 * 256), ("tt", C.c_void_p), ("rle_state", C.c_int),
                    ("rle_crc", C.c_uint32), ("rle_index", C.c_uint32), ("rle_avail", C.c_uint32), ("rle_char", C.c_uint8),
                    ("rle_prev", C.c_uint8)]

    lib.lbzamd_retrieve.argtypes = [C.POINTER(DecoderState), C.POINTER(BitStream)]
    lib.lbzamd_decoder_init.argtypes = [C.POINTER(DecoderState)]
    lib.lbzamd_decoder_free.argtypes = [C.POINTER(DecoderState)]
    ERR_TREES, MORE, OK = 6, 1, 0           # common.h:54-76
    for extra in (b"", b"Q", b"Q~", b"Q~\x01"):   # (more 16-byte ranges in use: the fields move by 16 bits)
        data = bytes(gen("text", 20000, 3)) + extra
        z = L.orc_compress(data, 1)
        body = bytearray(z[14:]) + bytes(16)    # behind "BZh1", the block magic and the stored CRC: what retrieve() reads
        # randomised (1) + origin pointer (24) + 16-bit map of ranges + 16 bits per used range, then the 3-bit table count
        nranges = bin(int.from_bytes(body[3:6], "big") >> 7 & 0xFFFF).count("1")
        pos = 1 + 24 + 16 + 16 * nranges
        if 32 * ((pos + 3 + 31) // 32) < pos + 18:      # a whole number of words ends inside the selector count
            break
    else:
        raise AssertionError("no variant puts a word boundary inside the selector count")
    def words(b):
        b = bytes(b) + bytes(-len(b) % 4)
        return (C.c_uint32 * (len(b) // 4)).from_buffer_copy(b)     # big-endian words as they lie in the file (decode.c:404)

    def call(ds, w, lo, hi, eof):
        bs = BitStream()
        bs.live, bs.buff, bs.eof = 0, 0, eof
        base = C.cast(w, C.POINTER(C.c_uint32))
        bs.data = C.cast(C.addressof(w) + 4 * lo, C.POINTER(C.c_uint32))
        bs.limit = C.cast(C.addressof(w) + 4 * hi, C.POINTER(C.c_uint32))
        rc = lib.lbzamd_retrieve(C.byref(ds), C.byref(bs))
        return rc, (C.addressof(bs.data.contents) - C.addressof(w)) // 4, bs.live

    # the intact block first, in two pieces: MORE, then OK with the stream left behind the block's last code
    ds = DecoderState()
    lib.lbzamd_decoder_init(C.byref(ds))
    w = words(body)
    cut = (pos + 3 + 31) // 32 + 1
    rc, at, live = call(ds, w, 0, cut, False)
    assert rc == MORE and at == cut
    rc, at, live = call(ds, w, cut, len(w), True)
    assert rc == OK and ds.block_size > 0 and cut <= at <= len(w)
    lib.lbzamd_decoder_free(C.byref(ds))
    # now 7 tables, and a first piece that ends less than a word behind the field
    bad = bytearray(body)
    for k in range(3):
        bad[(pos + k) // 8] |= 0x80 >> ((pos + k) % 8)
    w = words(bad)
    ds = DecoderState()
    lib.lbzamd_decoder_init(C.byref(ds))
    cut = (pos + 3 + 31) // 32
    rc, at, live = call(ds, w, 0, cut, False)
    assert rc == MORE and at == cut and live == 0
    rc, at, live = call(ds, w, cut, len(w), False)
    assert rc == ERR_TREES, rc
    assert 32 * at - live == pos + 18, (at, live, pos)          # where the walk stopped: behind the selector count
    lib.lbzamd_decoder_free(C.byref(ds))
    # a first piece that holds the selector count too: the error at once, the stream where the walk stopped
    ds = DecoderState()
    lib.lbzamd_decoder_init(C.byref(ds))
    rc, at, live = call(ds, w, 0, cut + 1, False)
    assert rc == ERR_TREES and 32 * at - live == pos + 18, (rc, at, live, pos)
    lib.lbzamd_decoder_free(C.byref(ds))
    # the same damage with nothing behind it and end of file: the error itself, not "unexpected end of file"
    ds = DecoderState()
    lib.lbzamd_decoder_init(C.byref(ds))
    rc, at, live = call(ds, w, 0, cut, True)
    assert rc == ERR_TREES, rc
    lib.lbzamd_decoder_free(C.byref(ds))


def _window_cases():
    text = bytes(gen("text", 230000, 3))
    z1 = L.orc_compress(text, 1)                                          # three blocks of 100 000
    other = b"second stream " * 3000
    multi = z1 + bz2.compress(other, 2) + bz2.compress(b"") + L.orc_compress(b"z", 9)
    return [("one stream", z1, text), ("four streams", multi, text + other + b"z"),
            ("trailing garbage", multi + b"not a stream header " * 400, text + other + b"z"),
            ("empty stream first", bz2.compress(b"") + z1, text),
            ("zeros", L.orc_compress(bytes(1_000_000), 1), bytes(1_000_000))]


def test_windows_decode_what_the_whole_input_decodes(emu):
    """lbzamd_decompress_window (bounded memory: what lbzamd -d / bzcat read with): the input taken 40 000 and 2 500 bytes
    at a time -- blocks begin at any bit, a window ends inside a block, inside a magic, between a marker and its CRC, between
    two streams -- gives the bytes of the one-call form; a window that holds no whole block grows."""
    with emu.decoder(4) as dec:
        for name, z, want in _window_cases():
            assert dec.decompress(z) == want, name
            for w in (len(z) + 1, 40000, 2500):
                got, calls = dec.decompress_windows(z, w)
                assert got == want, (name, w)
                assert calls == 1 if w > len(z) else calls > 1, (name, w, calls)


def test_windows_refuse_what_the_whole_input_refuses(emu):
    """damage in the middle, at the end, a file cut inside a block / inside the trailer: refused window by window too, with
    the reference's error code where the parser finds it (a missing magic, the end of the file, the stream CRC), and the whole
    blocks in front of the damage delivered"""
    text = bytes(gen("text", 230000, 5))
    z = L.orc_compress(text, 1)
    cases = []
    b = bytearray(z); b[len(b) // 2] ^= 0x10; cases.append(("payload bit", bytes(b)))
    b = bytearray(z); b[-1] ^= 1; cases.append(("stream CRC", bytes(b)))
    cases.append(("cut in a block", z[:len(z) * 3 // 4]))
    cases.append(("cut in the trailer", z[:-3]))
    with emu.decoder(4) as dec:
        for name, bad in cases:
            with pytest.raises(LbzError) as whole:
                dec.decompress(bad)
            code = emu.lib.lbzamd_last_error_code()
            for w in (len(bad) + 1, 20000):
                with pytest.raises(LbzError) as e:
                    dec.decompress_windows(bad, w)
                if name != "payload bit":                          # (a block's own error is reported with its window: lbzip2_amd.h)
                    assert emu.lib.lbzamd_last_error_code() == code, (name, w)
                assert text.startswith(e.value.decoded_in_front), (name, w)
                if name == "stream CRC":
                    assert len(e.value.decoded_in_front) >= len(whole.value.decoded_in_front), (name, w)


def test_command_decodes_in_windows(emu, tmp_path):
    """lbzamd -dc with LBZAMD_IO_DWINDOW=20000: the same bytes as with the default window, a pipe that closes early ends the
    program (bzcat big | head), -t reads to the end; a file that is no bzip2 file is copied through with -f, window by window"""
    exe = os.path.join(EMU_DIR, "_build", "lbzamd_emu")
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "WG=1024"])
    text = bytes(gen("text", 210000, 9))
    z = L.orc_compress(text, 1) + bz2.compress(b"tail" * 5000, 1) + b"garbage" * 100
    env = dict(os.environ, LBZAMD_IO_DWINDOW="20000")
    for e in (os.environ, env):
        r = subprocess.run([exe, "-dc"], input=z, capture_output=True, timeout=900, env=e)
        assert r.returncode == 0 and r.stdout == text + b"tail" * 5000, r.stderr[-300:]
    r = subprocess.run([exe, "-t"], input=z, capture_output=True, timeout=900, env=env)
    assert r.returncode == 0 and r.stdout == b""
    r = subprocess.run([exe, "-t"], input=z[:50000], capture_output=True, timeout=900, env=env)
    assert r.returncode == 1 and b"lbzamd" in r.stderr
    plain = bytes(gen("rand", 70000, 2))
    r = subprocess.run([exe, "-dcf"], input=plain, capture_output=True, timeout=900, env=env)
    assert r.returncode == 0 and r.stdout == plain
    # a reader that leaves early: the writer gets EPIPE/SIGPIPE after the first windows, not after the whole input
    p = subprocess.Popen(f"{exe} -dc | head -c 1000 | wc -c", shell=True, stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
    out, _ = p.communicate(z, timeout=900)
    assert out.strip() == b"1000"
