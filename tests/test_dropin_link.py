"""The drop-in boundary, proven with the reference's own program: every source file of lbzip2 except
encode.c and divbwt.c -- main.c, process.c (splitter / muxer threads), compress.c (work units),
signals.c, timespec.c, expand.c, decode.c, parse.c, crctab.c -- compiled UNCHANGED where it lies under
/root/reference/src (oracle/Makefile: `stock`, `dropin`) and linked against this repository's library
in place of the block codec.  compress.c then drives GPU blocks through encode.h
(encoder_alloc_size / encoder_init / collect / encode / transmit, compress.c:89-94,113,220-223).

CPU (here): linked against the emulator build of the kernels; output must equal stock lbzip2's byte for
byte at several levels and worker counts, and the reference fixture.
GPU: the same program linked against the product library (built here by hipcc, travels with the
repository under oracle/_ref/) compresses the enwik8-sized stand-in on the MI355X."""
import hashlib
import os
import subprocess

import pytest

from golden_util import bench_fixtures, gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
HAVE_SRC = os.path.exists("/root/reference/src/compress.c")


def _make(*args):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")] + list(args))


@pytest.fixture(scope="module")
def programs():
    if not HAVE_SRC:
        pytest.skip("reference sources absent (GPU box): the CPU link test runs in the build container")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "WG=1024"])
    _make("stock")
    _make("dropin", "DROPIN=../tests/emu/_build/liblbzamd_emu_1024.so", "OUT=lbzip2_dropin_emu")
    _make("dropin_d", "DROPIN=../tests/emu/_build/liblbzamd_emu_1024.so", "OUT=lbzip2_dropin_d_emu")
    return os.path.join(REFDIR, "lbzip2_stock"), os.path.join(REFDIR, "lbzip2_dropin_emu")


def _run(exe, args, data, env=None, timeout=900):
    p = subprocess.run([exe] + args, input=data, capture_output=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr[-500:]
    return p.stdout


@pytest.mark.parametrize("level,workers,kind,n,seed", [(1, 1, "wiki", 350000, 2), (1, 4, "wiki", 350000, 2),
                                                       (2, 3, "runs", 410000, 9), (1, 2, "text", 99999, 4)])
def test_unmodified_reference_cli_drives_the_library(programs, level, workers, kind, n, seed):
    stock, dropin = programs
    data = bytes(gen(kind, n, seed))
    args = [f"-{level}", "-n", str(workers)]
    want = _run(stock, args, data)
    got = _run(dropin, args, data, env={"LBZAMD_POOL_SLABS": "8", "LBZ_EMU_THREADS": "2"})
    assert got == want
    for r in bench_fixtures(max_n=400000):
        if (r["kind"], r["n"], r["seed"], r["level"]) == (kind, n, seed, level):
            assert hashlib.md5(got).hexdigest() == r["ref_md5"]
    assert _run(stock, ["-d"], got) == data                      # and the reference's decompressor takes it


@pytest.mark.parametrize("level,workers,kind,n,seed", [(1, 1, "wiki", 350000, 7), (1, 3, "runs", 450000, 7), (1, 2, "rand", 250000, 7)])
def test_reference_cli_sequential_mode(programs, level, workers, kind, n, seed):
    """lbzip2 -u: compress.c:129-198 re-enters collect() on one encoder until its block is full; the drop-in
    collect() serves that too (the block is re-tokenised from its start with the appended input)"""
    import json
    stock, dropin = programs
    data = bytes(gen(kind, n, seed))
    args = ["-u", f"-{level}", "-n", str(workers)]
    want = _run(stock, args, data)
    got = _run(dropin, args, data, env={"LBZAMD_POOL_SLABS": "8", "LBZ_EMU_THREADS": "2"})
    assert got == want
    recs = json.load(open(os.path.join(ROOT, "tests", "golden", "seq_fixtures.json")))["records"]
    rec = [r for r in recs if (r["kind"], r["n"], r["seed"], r["level"]) == (kind, n, seed, level)]
    assert rec and hashlib.md5(got).hexdigest() == rec[0]["ref_md5"]


def test_empty_input_through_the_cli(programs):
    stock, dropin = programs
    assert _run(dropin, ["-9"], b"", env={"LBZAMD_POOL_SLABS": "2"}) == _run(stock, ["-9"], b"")


def test_reference_cli_decompresses_through_the_library(programs):
    """The decoder's work-unit boundary (decode.h:77-81): the reference's program WITHOUT decode.c -- expand.c's scheduler,
    parse.c's header parser and bit stream -- linked against the library (oracle/Makefile: dropin_d).  `lbzip2 -d` then
    takes its blocks through decoder_init / retrieve / decode / emit / decoder_free on the (here: emulated) device: same
    bytes as the stock program for its own streams at several levels and worker counts, for bzip2's (bit-aligned blocks,
    several streams in a file), for trailing garbage; the same refusal for a damaged block and a truncated file."""
    import bz2
    stock = programs[0]
    dd = os.path.join(REFDIR, "lbzip2_dropin_d_emu")
    env = {"LBZ_EMU_THREADS": "2"}
    cases = []
    for kind, n, seed, lvl in (("wiki", 250000, 2, 1), ("runs", 210000, 9, 2), ("rand", 120000, 3, 9)):
        data = bytes(gen(kind, n, seed))
        cases.append((data, _run(stock, [f"-{lvl}", "-n", "2"], data)))
    d2 = bytes(gen("wiki", 130000, 5))
    cases.append((d2 + d2[:70000], bz2.compress(d2, 1) + bz2.compress(d2[:70000], 9)))          # bzip2's own encoder, two streams
    cases.append((d2, bz2.compress(d2, 1) + b"\0garbage behind the stream"))
    cases.append((b"", _run(stock, ["-9"], b"")))
    for workers in ("2",):
        for want, z in cases:
            p = subprocess.run([dd, "-d", "-n", workers], input=z, capture_output=True, timeout=900, env=dict(os.environ, **env))
            q = subprocess.run([stock, "-d", "-n", workers], input=z, capture_output=True, timeout=900)
            assert p.returncode == q.returncode and p.stdout == q.stdout, (workers, len(z), p.stderr[-300:])
            assert q.stdout == want or q.returncode != 0
    # damage: a flipped payload bit (block CRC), a truncated file
    z = bytearray(cases[0][1]); z[len(z) // 2] ^= 0x10
    for bad in (bytes(z), cases[0][1][:len(cases[0][1]) * 2 // 3]):
        p = subprocess.run([dd, "-d"], input=bad, capture_output=True, timeout=900, env=dict(os.environ, **env))
        q = subprocess.run([stock, "-d"], input=bad, capture_output=True, timeout=900)
        assert q.returncode != 0 and p.returncode == q.returncode, (p.returncode, q.returncode, p.stderr[-300:], q.stderr[-300:])


def _damaged_cases_through(exe, env):
    import json
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "damaged_cases.json")))["cases"]
    name = os.path.basename(exe)
    for c in cases:
        p = subprocess.run([exe, "-dc"], input=bytes.fromhex(c["bz2_hex"]), capture_output=True, timeout=300, env=dict(os.environ, **env))
        msg = p.stderr.decode(errors="replace").strip().replace(name + ":", "lbzip2_stock:")
        assert p.returncode == c["ref_exit"], (c["name"], p.returncode, msg)
        if c["ok"]:
            assert len(p.stdout) == c["out_len"] and hashlib.md5(p.stdout).hexdigest() == c["out_md5"], c["name"]
        else:
            assert msg in [c["ref_message"]] + c.get("also", []), (c["name"], msg, c["ref_message"])


def test_damaged_streams_through_the_decoder_s_work_unit_boundary(programs):
    """tests/golden/damaged_cases.json (103 hand-made and damaged streams with the compiled reference's verdict) through the
    reference's OWN expand.c and parse.c over this library's retrieve / decode / emit: the diagnostic is then the reference's
    scheduler's, made from what retrieve() and emit() return and from where retrieve() leaves the bit stream (at the point it
    stopped, as the reference's does -- the parser goes on from there)."""
    _damaged_cases_through(os.path.join(REFDIR, "lbzip2_dropin_d_emu"), {"LBZ_EMU_THREADS": "2", "LBZAMD_POOL_SLABS": "4"})


@pytest.mark.gpu
def test_damaged_streams_through_the_decoder_s_work_unit_boundary_on_the_gpu():
    exe = os.path.join(REFDIR, "lbzip2_dropin_d_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/lbzip2_dropin_d_gpu not built (needs the reference sources at build time)")
    _damaged_cases_through(exe, {})


@pytest.mark.gpu
def test_reference_cli_decompresses_on_the_gpu():
    """oracle/_ref/lbzip2_dropin_d_gpu = the reference's program without encode.c, divbwt.c AND decode.c, linked against
    lbzip2_amd/csrc/liblbzamd.so: `lbzip2 -d` with 16 worker threads decodes the enwik8-sized stand-in's stream (the
    reference fixture, written by the same program's compressor side) block by block on the MI355X."""
    exe = os.path.join(REFDIR, "lbzip2_dropin_d_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/lbzip2_dropin_d_gpu not built (needs the reference sources at build time)")
    rec = [r for r in bench_fixtures() if r["kind"] == "wiki" and r["n"] == 100_000_000][0]
    data = bytes(gen(rec["kind"], rec["n"], rec["seed"]))
    z = _run(exe, ["-9", "-n", "64"], data, timeout=300)
    assert len(z) == rec["out_len"] and hashlib.md5(z).hexdigest() == rec["ref_md5"]
    back = _run(exe, ["-d", "-n", "16"], z, timeout=600)
    assert back == data


@pytest.mark.gpu
def test_reference_cli_on_the_gpu():
    """oracle/_ref/lbzip2_dropin_gpu = the reference's unmodified CLI + process.c splitter/muxer, linked
    against lbzip2_amd/csrc/liblbzamd.so: 64 worker threads feed GPU blocks; stream == reference fixture."""
    exe = os.path.join(REFDIR, "lbzip2_dropin_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/lbzip2_dropin_gpu not built (needs the reference sources at build time)")
    rec = [r for r in bench_fixtures() if r["kind"] == "wiki" and r["n"] == 100_000_000][0]
    data = bytes(gen(rec["kind"], rec["n"], rec["seed"]))
    out = _run(exe, ["-9", "-n", "64"], data, timeout=300)
    assert len(out) == rec["out_len"] and hashlib.md5(out).hexdigest() == rec["ref_md5"]


@pytest.mark.gpu
def test_reference_cli_sequential_mode_on_the_gpu():
    """the same program with -u: the stream of tests/golden/seq_fixtures.json (tar-like, 5 * 10^7 bytes, -9)"""
    import json
    exe = os.path.join(REFDIR, "lbzip2_dropin_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/lbzip2_dropin_gpu not built (needs the reference sources at build time)")
    import oracle_lib as L
    rec = [r for r in json.load(open(os.path.join(ROOT, "tests", "golden", "seq_fixtures.json")))["records"] if r["kind"] == "tar"][0]
    data = bytes(L.gen_kind(rec["kind"], rec["n"], rec["seed"]))
    out = _run(exe, ["-u", "-%d" % rec["level"], "-n", "16"], data, timeout=300)
    assert len(out) == rec["out_len"] and hashlib.md5(out).hexdigest() == rec["ref_md5"]
