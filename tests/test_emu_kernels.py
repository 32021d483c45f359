"""CPU: kernel LOGIC under the fiber emulator (tests/emu) against the oracle, small blocks.
The emulator build is test infrastructure; the product library is hipcc/gfx950 only."""
import bz2
import os
import subprocess

import pytest

import oracle_lib as L
from golden_util import gen, suite_inputs
from lbzip2_amd._binding import Library

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "WG=1024"])
    return Library(os.path.join(EMU_DIR, "_build", "liblbzamd_emu_1024.so"))


KEYS = ["consumed", "nblock", "crc", "inuse", "block", "bwt", "bwt_idx", "nmtf", "alpha", "mtfv",
        "num_trees", "num_selectors", "out_len", "out"]


def _stages(emu, data, level):
    M = level * 100000
    with emu.context(level, max(1, (len(data) + M - 1) // M), 8) as ctx:
        gb = ctx.blocks(data, 3)
    ob = L.orc_blocks(data, level)
    assert len(gb) == len(ob)
    for g, o in zip(gb, ob):
        assert g["err"] == 0
        for k in KEYS:
            assert g[k] == o[k], k
        assert g["periodic"] == o["periodic"]


@pytest.mark.parametrize("name,data", [
    ("banana", b"banana"), ("one", b"a"), ("two", b"ab"), ("same", b"\0" * 7),
    ("text3k", gen("text", 3000, 5)), ("rand5k", gen("rand", 5000, 6)),
    ("zeros", bytes(10000)), ("abab", b"ab" * 500), ("all256", bytes(range(256)) * 3),
    ("text30k", gen("text", 30000, 7)), ("lines40k", gen("lines", 40000, 3)),
    ("alpha100", bytes((i * 37 + (i >> 3) + (i >> 9)) % 100 for i in range(12000))),
])
def test_stages_small(emu, name, data):
    _stages(emu, data, 1)


def test_blocks_of_about_one_batch(emu):
    """Blocks of at most one batch of k_bwt_batch (BATCH_ROWS = 832 rows since round 6, 1024 before) are sorted whole in LDS,
    without a partition; one row more and the block is partitioned and cut into batches.  Sizes on both sides of either limit
    and of the 64-row strips the per-wave loops take two at a time, on text (ties), few symbols (long groups) and random bytes."""
    for n in (63, 64, 65, 127, 128, 129, 191, 193, 767, 831, 832, 833, 895, 897, 1023, 1024, 1025, 1663, 1664, 1665, 2500):
        for data in (gen("text", n, n), bytes((i * i + (i >> 2)) % 3 + 65 for i in range(n)), gen("rand", n, n + 1), (b"abcab" * n)[:n]):
            _stages(emu, data, 1)


def test_streams_multi_slab_spill_chunked(emu):
    for kind, n, seed, slabs in [("text", 230000, 3, 2), ("runs", 150000, 4, 2)]:
        data = gen(kind, n, seed)
        with emu.context(1, slabs, 4) as ctx:
            got = ctx.compress(data)
        assert got == L.orc_compress(data, 1)
        assert bz2.decompress(got) == data
    assert emu.compress(b"", 9) == L.orc_compress(b"", 9)


def test_rounds_of_many_and_of_few_blocks(emu):
    """The partition runs a launch per pass with 8-16 workgroups per block (k_bwt_hist / k_bwt_scat / k_bwt_segs), or --
    rounds of more blocks than the (emulated, 8-CU) device has CUs side by side on several streams -- as one workgroup per
    block (k_bwt_part) (lbz_api.hip: launch_sort; lbzamd_round_shape says which).  Same stream."""
    data = bytes(gen("wiki", 1_130_000, 6) + gen("text", 520_000, 8) + gen("rand", 400_000, 9))       # 21 slabs at -1
    want = L.orc_compress(data, 1)
    for max_slabs, nslots in ((21, 21), (21, 4), (21, 10)):
        with emu.context(1, max_slabs, nslots) as ctx:
            assert ctx.compress(data) == want, (max_slabs, nslots)
            segs, parts = ctx.round_shape(nslots, True)
            assert segs == 32 and parts == (1 if nslots > 8 else 16 if nslots <= 4 else 8), (nslots, segs, parts)
            assert ctx.round_shape(nslots, False)[1] in (8, 16)


def test_round_schedule(emu, monkeypatch):
    """Rounds of nslots slabs (short last round issued first, slot sets per stream, spill blocks
    riding with their primaries): the stream must not depend on the schedule."""
    data = gen("text", 110000, 11) + gen("runs", 90000, 5) + gen("rand", 70000, 6)   # 3 slabs at -1, one with a big spill
    want = L.orc_compress(data, 1)
    for streams, nslots, max_slabs in (("2", 2, 3), ("3", 1, 3), ("2", 1, 2)):           # the last one: two chunks
        monkeypatch.setenv("LBZAMD_STREAMS", streams)
        with emu.context(1, max_slabs, nslots) as ctx:
            assert ctx.compress(data) == want, (streams, nslots, max_slabs)


def test_host_output_paths(emu, monkeypatch):
    """lbzamd_compress_host: a page-locked output buffer is written by the device round by round (k_gather straight into
    host memory); any other buffer goes through the device staging copy.  Same stream either way, also when the call is
    chunked and when the buffer is too small."""
    data = bytes(gen("wiki", 430000, 9) + gen("runs", 60000, 2))
    want = L.orc_compress(data, 1)
    for no_direct in ("", "1"):
        if no_direct:
            monkeypatch.setenv("LBZ_EMU_NO_HOSTPTR", "1")
        for max_slabs, nslots in ((5, 2), (2, 1)):
            with emu.context(1, max_slabs, nslots) as ctx:
                assert ctx.compress(data) == want, (no_direct, max_slabs, nslots)
    monkeypatch.delenv("LBZ_EMU_NO_HOSTPTR")
    # pageable input: a round's copy is issued next to the launches of the round before it (lbz_api.hip: run_chunk)
    monkeypatch.setenv("LBZ_EMU_PAGEABLE", "1")
    for max_slabs, nslots in ((5, 2), (5, 1)):
        with emu.context(1, max_slabs, nslots) as ctx:
            assert ctx.compress(data) == want, ("pageable", max_slabs, nslots)
    monkeypatch.delenv("LBZ_EMU_PAGEABLE")
    import ctypes as C
    with emu.context(1, 5, 2) as ctx:
        small = C.create_string_buffer(len(want) - 7)
        n = C.c_size_t()
        rc = emu.lib.lbzamd_compress_host(ctx.h, C.cast(C.c_char_p(data), C.c_void_p), len(data), small, len(want) - 7, C.byref(n))
        assert rc == -2 and b"too small" in emu.lib.lbzamd_last_error()


def test_fuzz_slice(emu):
    """A slice of tests/fuzz_gpu.py small enough for the emulator (blocks of <= 9000 bytes)."""
    import fuzz_gpu
    assert fuzz_gpu.run(emu, L.orc_compress, 21, 120, small=True) == []


def test_workunit_interface(emu):
    data = gen("runs", 110000, 2) + gen("text", 20000, 9)
    assert emu.compress_workunits(data, 1) == L.orc_compress(data, 1)


def test_workunit_interface_threads(emu, monkeypatch):
    """The combining of work-unit requests across caller threads (lbz_api.hip, pool_submit): more
    threads than pool slabs, so states also wait for slabs."""
    from concurrent.futures import ThreadPoolExecutor
    monkeypatch.setenv("LBZAMD_POOL_SLABS", "2")
    datas = [gen("text", 30000 + 3000 * i, 40 + i) + gen("runs", 20000, i) for i in range(5)]
    with ThreadPoolExecutor(5) as ex:
        outs = list(ex.map(lambda d: emu.compress_workunits(d, 2), datas))      # -2: a pool no other test has created
    for d, o in zip(datas, outs):
        assert o == L.orc_compress(d, 2)


def test_reference_suite_sample(emu):
    """A slice of the reference's own compress corpora (small members only: emulation is slow)."""
    inputs = suite_inputs()
    names = [n for n in sorted(inputs) if len(inputs[n]) <= 6000][::12]
    assert len(names) > 30
    for n in names:
        raw = inputs[n]
        if raw:
            assert emu.compress(raw, 9) == L.orc_compress(raw, 9), n


def test_file_splitter_muxer(emu, tmp_path):
    """lbzamd_compress -f/-o: reader thread -> ring of chunk buffers -> pipeline threads with
    their own contexts (body-only slab ranges) -> writer thread in order, CRC folded from the 12-byte
    partials.  Sizes: several chunks with a ragged tail, an exact multiple of the chunk, one slab, empty."""
    exe = os.path.join(EMU_DIR, "_build", "lbzamd_compress_emu")
    for n, chunk, pipes in ((527000, 2, 2), (400000, 2, 3), (99000, 4, 1), (0, 2, 2)):
        data = bytes(gen("wiki", n, 3)) if n else b""
        src, dst = tmp_path / "in.bin", tmp_path / "out.bz2"
        src.write_bytes(data)
        subprocess.check_call([exe, "-1", "-f", str(src), "-o", str(dst), "-c", str(chunk), "-p", str(pipes)], timeout=600)
        assert dst.read_bytes() == L.orc_compress(data, 1), (n, chunk, pipes)


def test_c_side_multi_device(emu, tmp_path):
    """The C host side on several devices (north star: lbzip2's splitter/muxer feeding N GPUs; process.c:515-548):
    `lbzamd_compress -f/-o -g N` puts pipeline i's context on device i mod N, and LBZAMD_DEVICES=N keeps one work-unit
    pool per device behind the drop-in symbols.  The emulator shows N fake devices (LBZ_EMU_DEVICES): the per-device
    code paths run, the stream must be the single-device one."""
    exe = os.path.join(EMU_DIR, "_build", "lbzamd_compress_emu")
    data = bytes(gen("wiki", 730000, 4))
    want = L.orc_compress(data, 1)
    src, dst = tmp_path / "in.bin", tmp_path / "out.bz2"
    src.write_bytes(data)
    env = dict(os.environ, LBZ_EMU_DEVICES="3")
    for g, p in (("3", "1"), ("0", "2"), ("2", "1")):
        r = subprocess.run([exe, "-1", "-f", str(src), "-o", str(dst), "-c", "2", "-p", p, "-g", g, "-t"], env=env,
                           capture_output=True, timeout=900)
        assert r.returncode == 0, r.stderr[-400:]
        assert dst.read_bytes() == want, (g, p)
        assert (b"on %d device(s)" % (3 if g in ("0", "3") else 2)) in r.stderr, r.stderr[-300:]
    # the reference's work-unit calls from 4 threads, the states' slabs leased round-robin from 2 and 3 pools
    for ndev in ("2", "all"):
        r = subprocess.run([exe, "-1", "-w", "4"], input=data, env=dict(env, LBZAMD_DEVICES=ndev, LBZAMD_POOL_SLABS="3"),
                           capture_output=True, timeout=900)
        assert r.returncode == 0 and r.stdout == want, (ndev, r.stderr[-400:])
    r = subprocess.run([exe, "-1", "-f", str(src), "-o", str(dst), "-g", "2"], env=dict(os.environ, LBZ_EMU_DEVICES="1"),
                       capture_output=True, timeout=900)
    assert r.returncode == 0 and dst.read_bytes() == want           # more devices asked for than there are: clamped
    # LBZAMD_FAKE_DEVICES: two logical devices on the ONE device present (what the GPU suite uses on its one-GPU box)
    one = dict(os.environ, LBZ_EMU_DEVICES="1", LBZAMD_FAKE_DEVICES="2")
    r = subprocess.run([exe, "-1", "-f", str(src), "-o", str(dst), "-c", "2", "-p", "1", "-g", "2", "-t"], env=one, capture_output=True, timeout=900)
    assert r.returncode == 0 and dst.read_bytes() == want and b"on 2 device(s)" in r.stderr, r.stderr[-300:]
    r = subprocess.run([exe, "-1", "-w", "4"], input=data, env=dict(one, LBZAMD_DEVICES="2", LBZAMD_POOL_SLABS="3"), capture_output=True, timeout=900)
    assert r.returncode == 0 and r.stdout == want, r.stderr[-400:]


def test_long_repeats_step_by_ranks(emu):
    """Passages of thousands of symbols that occur two and three times (a source tree's licence headers, copied files).
    First input: a third of the rows are tied after the first text launch, so the block stays with the text rounds; the rows
    of the repeats outlast them launch after launch (tied rows per launch 27059, 15803, 12828, 12204, 7112, 338, 28; least
    depth 9, 10, 36, 116, 810, 836, 1285: a -DDEEP_DEBUG build prints them), get rank entries and step through the repeats by
    ranks -- with the depth a run had when the launch began written into its rows' entries first (k_bwt_deep).  Second
    input: four fifths tied after the first launch: handed to the rank rounds.  All stages against the oracle."""
    P, Q, base = gen("wiki", 4000, 32), gen("text", 2500, 33), gen("wiki", 40000, 34)
    _stages(emu, bytes(base[:15000] + P + base[15000:30000] + P + base[30000:] + Q + Q), 1)
    a, b, c = gen("wiki", 9000, 21), gen("text", 6000, 22), gen("wiki", 2500, 23)
    data = gen("wiki", 8000, 24) + a + gen("text", 3000, 25) + b + a[:7000] + gen("rand", 500, 26) + b + c + a + c[:2000] + b[1000:]
    _stages(emu, bytes(data), 1)


def test_oversized_runs_at_segment_bounds(emu):
    """Case 10 of the decoder fuzz (seed 41: tests/test_decode.py::test_fuzz_on_the_gpu): three byte values, nearly
    periodic, so that groups of more than a batch of EQUAL keys lie on both sides of a segment bound.  Such a group trades
    its keys for later symbols (k_bwt.hip, big_group) while the neighbouring segment workgroup is still looking at the key
    column: nothing of a segment's may be read from outside it (round 4: the group-end search ran to the block's end)."""
    import random

    import fuzz_gpu
    rng = random.Random(41)
    fuzz_gpu.BIG = fuzz_gpu.SMALL = False
    for i in range(11):
        data = fuzz_gpu.make(rng)
        level = rng.choice([1, 1, 2, 9])
        rng.random()
        if rng.random() < 0.3:
            fuzz_gpu.make(rng)
            rng.choice([1, 9])
    assert (len(data), level) == (79835, 1)
    assert emu.compress(data, level) == L.orc_compress(data, level)


def test_runs_of_every_length_in_the_text_rounds(emu):
    """The strips of the text rounds count a row's place among the rows of its own run when the strip's longest run has at
    most DEEP_NEAR rows (one pass, 14 symbols: deep_stage_near) and against all 64 keys otherwise (two passes of 52 bits).
    Passages of 12 to 120 bytes occurring 2, 3, ... 40 times with different bytes behind them give runs of every length from
    2 to 40 that tie for 9 to 100 and more symbols and split at different depths, side by side in the same strips; runs that
    tie on whole passages and differ only behind them exercise the second word of the key.  Every stage against the oracle."""
    import random
    rng = random.Random(5)
    base = bytes(gen("wiki", 30000, 51))
    out = bytearray()
    for mult in list(range(2, 41)) + [2, 3, 5, 9, 17, 33, 63, 64, 65]:
        plen = rng.choice([12, 14, 15, 16, 22, 23, 27, 29, 40, 77, 120])
        at = rng.randrange(0, len(base) - plen)
        passage = base[at:at + plen]
        for k in range(mult):
            out += passage
            out += bytes([65 + (k * 7 + mult) % 26]) * rng.choice([1, 1, 2]) + bytes(gen("text", rng.choice([3, 9, 30]), 1000 * mult + k))
    _stages(emu, bytes(out[:99000]), 1)
    _stages(emu, bytes(out[40000:] + out[:30000]), 2)


def test_file_splitter_muxer_more_readers_than_ring_positions(emu, tmp_path):
    """lbzamd_io.c under thread counts that make several readers wait for ONE ring position (8 readers, 3 pipelines: 8
    positions; chunks of one slab): the position goes to the earliest chunk, or the chunks behind it -- compressed in order
    -- would wait for a position that only their own output can free.  (Round 5: that deadlock showed on the GPU box as a
    command that never ended; `LBZAMD_IO_DEBUG=seconds` prints what every part of the engine waits for.)"""
    exe = os.path.join(EMU_DIR, "_build", "lbzamd_compress_emu")
    data = bytes(gen("text", 2_600_000, 4))
    want = L.orc_compress(data, 1)
    src = tmp_path / "in.bin"
    src.write_bytes(data)
    env = dict(os.environ, LBZ_EMU_THREADS="2", LBZ_EMU_DEVICES="2")
    for rep in range(2):
        for c, p, r, w in ((1, 3, 8, 4), (1, 2, 16, 8), (1, 6, 4, 2)):
            out = tmp_path / "out.bz2"
            subprocess.run([exe, "-1", "-f", str(src), "-o", str(out), "-c", str(c), "-p", str(p), "-R", str(r), "-W", str(w)],
                           env=env, check=True, timeout=120)
            assert out.read_bytes() == want, (c, p, r, w)
    z = subprocess.run([exe, "-1", "-f", "-", "-o", "-", "-c", "1", "-p", "5"], input=data, env=env, capture_output=True, timeout=120)
    assert z.returncode == 0 and z.stdout == want


def test_long_runs_of_every_size(emu):
    """Runs of 64 rows and more go to k_bwt_deep's long-run code (deep_big_run: common prefix, then a counting split on the first
    symbol the rows do not all share, pieces of 64 rows and more split again).  Passages of 12 .. 200 bytes that occur 64 ..
    300 times, with one to three different bytes and then different text behind them: pieces of every size, shared prefixes
    shorter and longer than the 16 bytes a step loads (and longer than the 64 a launch follows), sub-runs that stay tied."""
    import random
    rng = random.Random(11)
    base = bytes(gen("wiki", 20000, 77))
    out = bytearray()
    for mult, plen in ((64, 12), (70, 29), (130, 16), (200, 40), (256, 23), (257, 15), (300, 120), (90, 200), (65, 77)):
        at = rng.randrange(0, len(base) - plen)
        passage = base[at:at + plen]
        for k in range(mult):
            out += passage + bytes([97 + k % 3]) * (1 + k % 2) + bytes([65 + (k * 5) % 23]) + bytes(gen("text", rng.choice([2, 5, 11]), 3000 + 7 * k + mult))
    data = bytes(out)
    assert len(data) > 60000
    _stages(emu, data[:99000], 1)
    _stages(emu, data[50000:50000 + 180000], 2)


def test_long_runs_of_every_size(emu, monkeypatch):
    """Runs of 64 tied rows and more (deep_big_run: counting splits, pieces on a stack, closed sub-runs) of every size, two of
    them next to each other in a list, in rounds of many blocks and of few -- and the same inputs with every block handed to
    the rank rounds at once / after the first text launch (LBZAMD_HANDOVER0 / 1 = 1 thousandth), so that the fall-back meets
    the closed runs the batches and the first launch left tied in the suffix array."""
    import random
    rng = random.Random(12)
    base = bytes(gen("wiki", 20000, 78))
    out = bytearray()
    for mult, plen in ((64, 12), (65, 13), (70, 29), (77, 9), (130, 16), (256, 23), (257, 15), (300, 120), (700, 8), (90, 200)):
        at = rng.randrange(0, len(base) - plen)
        passage = base[at:at + plen]
        for k in range(mult):
            out += passage + bytes([97 + k % 3]) * (1 + k % 2) + bytes([65 + (k * 5) % 23]) + bytes(gen("text", rng.choice([2, 5, 11]), 4000 + 7 * k + mult))
    many = bytes(gen("wiki", 1_130_000, 6) + gen("text", 520_000, 8) + gen("rand", 400_000, 9))       # 21 slabs at -1
    cases = ((bytes(out)[:99000], 1, 1, 1), (bytes(out), 2, 2, 2), (many, 1, 21, 21), (many, 1, 21, 4))
    for data, level, max_slabs, nslots in cases:
        want = L.orc_compress(data, level)
        for knob in (None, "LBZAMD_HANDOVER0", "LBZAMD_HANDOVER1"):
            if knob:
                monkeypatch.setenv(knob, "1")
            with emu.context(level, max_slabs, nslots) as ctx:
                assert ctx.compress(data) == want, (len(data), level, nslots, knob)
            if knob:
                monkeypatch.delenv(knob)


def test_long_duplicates_and_huge_runs_of_one_key(emu, monkeypatch):
    """Round 6: (1) passages of thousands of bytes that occur twice and thrice with different bytes in front of them -- their
    first rows are the only OPEN ones (the rows inside the copies are closed runs without rank entries), and the late text
    launches walk them 512 bytes a trip (deep_wide_skip); (2) more than LONG_RUN_MAX rows with ONE 64-bit key -- records of
    equal fields -- which the batch kernel re-keys on the text behind the key, level by level (big_group's frames).  Also with
    every block handed to the rank rounds after the first launch: the fall-back must give the same stream."""
    import random
    rng = random.Random(5)
    src = bytes(gen("wiki", 60000, 11))
    out = bytearray()
    for ln in (3000, 9000, 20000, 40000):
        at = rng.randrange(0, len(src) - ln)
        out += b"<" + src[at:at + ln] + b">" + bytes(gen("text", 500, ln)) + b"[" + src[at:at + ln] + b"]"
        out += b"{" + src[at:at + ln // 2] + b"}"
    dups = bytes(out)
    out = bytearray()
    for i in range(7000):                       # "field=AbCdEfGh;" x 7000: one key 7000 times over, then the same again one, two ... fields deeper
        out += b"AbCdEfGhIjKlMnOp" * rng.choice([1, 1, 2, 3, 5]) + bytes(gen("text", rng.choice([3, 7, 12]), i)) + rng.choice([b"\n", b";\n", b",\n"])
    keys = bytes(out)
    for data, level in ((dups, 2), (keys, 2), (keys[:99000] + dups[:99000], 9)):
        want = L.orc_compress(data, level)
        M = level * 100000
        for knob in (None, "LBZAMD_HANDOVER1"):
            if knob:
                monkeypatch.setenv(knob, "1")
            with emu.context(level, (len(data) + M - 1) // M, 4) as ctx:
                assert ctx.compress(data) == want, (len(data), level, knob)
            if knob:
                monkeypatch.delenv(knob)
