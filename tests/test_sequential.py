"""-u / --sequential (reference compress.c:129-198): blocks are cut where they are FULL, not at
every bs100k * 100000 input bytes.  Fixtures: tests/golden/seq_fixtures.json (the compiled reference driven as
do_collect_seq drives it, cross-checked against `lbzip2 -u` by make_seq_fixtures.py).  CPU: oracle vs reference and
fixtures, the kernels (k_collect_seq's block chain) under the emulator; GPU: the fixtures at full size."""
import bz2
import hashlib
import json
import os
import subprocess

import pytest

import oracle_lib as L
from golden_util import gen
from lbzip2_amd._binding import LbzError, Library

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")
FIX = json.load(open(os.path.join(HERE, "golden", "seq_fixtures.json")))["records"]


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "WG=1024"])
    return Library(os.path.join(EMU_DIR, "_build", "liblbzamd_emu_1024.so"))


def small(rec):
    return rec["n"] <= 500_000


@pytest.mark.parametrize("rec", [r for r in FIX if small(r)], ids=lambda r: f"{r['kind']}-{r['n']}")
def test_oracle_matches_the_reference_fixture(rec):
    z = L.orc_compress_seq(bytes(gen(rec["kind"], rec["n"], rec["seed"])), rec["level"])
    assert len(z) == rec["out_len"] and hashlib.md5(z).hexdigest() == rec["ref_md5"]


@pytest.mark.skipif(not L.have_ref(), reason="compiled reference not present")
def test_oracle_matches_the_compiled_reference():
    for kind, n, level in (("wiki", 1_300_000, 1), ("runs", 2_500_000, 3), ("rand", 450_000, 1), ("text", 2_000_000, 9)):
        d = bytes(gen(kind, n, 11))
        assert L.orc_compress_seq(d, level) == L.ref_compress_seq(d, level)
        if kind != "rand":
            assert L.ref_compress_seq(d, level) != L.ref_compress(d, level)      # it IS a different blocking


@pytest.mark.parametrize("rec", [r for r in FIX if small(r)], ids=lambda r: f"{r['kind']}-{r['n']}")
def test_kernels_under_the_emulator(emu, rec):
    d = bytes(gen(rec["kind"], rec["n"], rec["seed"]))
    z = emu.compress(d, rec["level"], sequential=True)
    assert len(z) == rec["out_len"] and hashlib.md5(z).hexdigest() == rec["ref_md5"]
    assert emu.compress(d, rec["level"], max_slabs=2, sequential=True) == z       # chunks of two blocks: the chain carries over


def test_edges_under_the_emulator(emu):
    for d in (b"", b"a", bytes(700_000), b"ab" * 60_000 + bytes(300) + b"c" * 259 * 3):
        z = emu.compress(d, 1, sequential=True)
        assert z == L.orc_compress_seq(d, 1) and bz2.decompress(z) == d
    with emu.context(1, 4) as ctx:                                                  # a slab range of such a stream is refused
        ctx.set_sequential(True)
        with pytest.raises(LbzError):
            ctx.compress_body(b"x" * 1000)
        ctx.set_sequential(False)
        assert ctx.compress(b"x" * 1000) == L.orc_compress(b"x" * 1000, 1)


def test_step_tables_of_the_block_chain(emu, monkeypatch):
    """The chain link through the step tables (k_seq_tiles / k_seq_prefix: what every 32 KB step of the input emits) and
    the walk it replaces give the same cuts: text (table path), runs that cross step boundaries and block starts inside
    long runs (the walk), a run head exactly on a step boundary, inputs shorter than a step."""
    import random
    rng = random.Random(7)
    cases = [("wiki", bytes(gen("wiki", 500000, 3))), ("runs", bytes(gen("runs", 300000, 4))), ("mixed", bytes(gen("mixed", 400000, 5))),
             ("zeros", bytes(350000)),
             ("longruns", b"".join(bytes([rng.randrange(3)]) * rng.randrange(1, 70000) for _ in range(20))),
             ("text+run", bytes(gen("text", 99990, 2)) + b"x" * 40000 + bytes(gen("text", 150000, 3))),
             ("boundary", bytes(gen("text", 32768, 4)) + b"q" * 32768 + bytes(gen("text", 131072 + 5, 5))),
             ("rand", bytes(gen("rand", 250000, 6))), ("short", b"abc" * 1000)]
    for name, data in cases:
        want = L.orc_compress_seq(data, 1)
        for slabs in (2, 64):
            with emu.context(1, slabs, 2) as ctx:
                ctx.set_sequential(True)
                assert ctx.compress(data) == want, (name, slabs)
                st = ctx.stats()
            if name in ("wiki", "mixed", "rand"):
                assert st.seq_fast_links >= st.nblocks - 2, (name, st.seq_fast_links, st.nblocks)     # ordinary data takes the tables
            if name == "zeros":
                assert st.seq_fast_links <= 1          # only a block that starts exactly where the run does
    monkeypatch.setenv("LBZAMD_SEQ_NO_TABLES", "1")
    with emu.context(1, 8, 2) as ctx:
        ctx.set_sequential(True)
        assert ctx.compress(cases[0][1]) == L.orc_compress_seq(cases[0][1], 1)
        assert ctx.stats().seq_fast_links == 0


def test_drop_in_symbols_called_as_the_u_mode_calls_them(emu):
    """collect() re-entered on one state until the block is full (compress.c:160-170), then encode / transmit"""
    from lbzip2_amd._binding import compress_workunits_seq
    for kind, n in (("wiki", 350000), ("runs", 450000), ("zeros", 700000), ("one", 1), ("empty", 0)):
        d = bytes(gen(kind, n, 7)) if kind in ("wiki", "runs") else (bytes(n) if kind == "zeros" else b"a" * n)
        assert compress_workunits_seq(emu, d, 1) == L.orc_compress_seq(d, 1), kind
    assert emu.compress_workunits(bytes(gen("wiki", 250000, 3)), 1) == L.orc_compress(bytes(gen("wiki", 250000, 3)), 1)


@pytest.mark.gpu
def test_drop_in_symbols_u_mode_on_the_gpu():
    import lbzip2_amd
    from lbzip2_amd._binding import compress_workunits_seq
    rec = [r for r in FIX if r["kind"] == "runs" and r["n"] == 20_000_000][0]
    d = bytes(gen("runs", rec["n"], rec["seed"]))
    z = compress_workunits_seq(lbzip2_amd.library(), d, rec["level"])
    assert len(z) == rec["out_len"] and hashlib.md5(z).hexdigest() == rec["ref_md5"]


@pytest.mark.gpu
@pytest.mark.parametrize("rec", FIX, ids=lambda r: f"{r['kind']}-{r['n']}-{r['level']}")
def test_fixtures_on_the_gpu(rec):
    import torch
    import lbzip2_amd
    lib = lbzip2_amd.library()
    n, M = rec["n"], rec["level"] * 100000
    data = L.gen_kind(rec["kind"], n, rec["seed"]) if rec["kind"] != "runs" else bytes(gen("runs", n, rec["seed"]))
    assert hashlib.md5(data).hexdigest() == rec["in_md5"]
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    with lib.context(rec["level"], min(1200, (n + M - 1) // M + 1)) as ctx:
        ctx.set_sequential(True)
        m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
        st = ctx.stats()
    z = dst[:m].cpu().numpy().tobytes()
    assert m == rec["out_len"] and hashlib.md5(z).hexdigest() == rec["ref_md5"] and st.nblocks == rec["nblocks"]
    if n <= 100_000_000:
        assert lib.decompress(z) == bytes(data)


@pytest.mark.gpu
def test_host_call_and_driver_on_the_gpu(tmp_path):
    import lbzip2_amd
    lib = lbzip2_amd.library()
    rec = [r for r in FIX if r["kind"] == "tar"][0]
    data = bytes(L.gen_kind(rec["kind"], rec["n"], rec["seed"]))
    z = lib.compress(data, rec["level"], max_slabs=20, sequential=True)            # host buffers, three chunks
    assert hashlib.md5(z).hexdigest() == rec["ref_md5"]
    root = os.path.dirname(HERE)
    exe = os.path.join(root, "lbzip2_amd", "host", "lbzamd_compress")
    r = subprocess.run([exe, "-u", "-%d" % rec["level"]], input=data, capture_output=True, timeout=600)
    assert r.returncode == 0 and hashlib.md5(r.stdout).hexdigest() == rec["ref_md5"], r.stderr[-300:]
