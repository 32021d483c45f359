"""GPU (MI355X): parity of the HIP path, called through the C ABI, against the golden vectors
generated from the compiled reference and against the CPU oracle on the same seeded inputs.
Bit-exact everywhere; the single documented exception is the 24-bit origin pointer of exactly
periodic blocks (T = u^k), where the reference's choice is an artefact of its quicksort."""
import bz2
import os
from concurrent.futures import ThreadPoolExecutor

import pytest

import oracle_lib as L
from golden_util import bench_fixtures, gen, load, md5, suite_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import lbzip2_amd
    return lbzip2_amd.library()          # raises if the HIP extension is missing


def cpu_reference(data, level):
    """The real reference when its build travelled with the repo, else the pinned oracle."""
    if L.have_ref():
        return L.ref_compress(data, level), True
    return L.orc_compress(data, level), False


def test_native_library_loaded(lib):
    import lbzip2_amd
    assert lib.path == lbzip2_amd.LIB_PATH and os.path.exists(lib.path)
    assert b"liblbzamd.so" in open("/proc/self/maps", "rb").read()


def test_literal_streams(lib):
    for hx, by_level in load("streams.json")["literals"].items():
        data = bytes.fromhex(hx)
        for lvl, want in by_level.items():
            got = lib.compress(data, int(lvl))
            if data == b"abababab":
                assert got == L.orc_compress(data, int(lvl))
                continue
            assert got.hex() == want, (hx, lvl)


@pytest.mark.parametrize("rec", load("streams.json")["seeded"],
                         ids=lambda r: f"{r['kind']}-{r['n']}-{r['seed']}-L{r['level']}")
def test_seeded_streams_vs_reference_md5(lib, rec):
    data = gen(rec["kind"], rec["n"], rec["seed"])
    out = lib.compress(data, rec["level"])
    assert len(out) == rec["out_len"]
    assert md5(out) == rec["canon_md5"]
    assert bz2.decompress(out) == data


def _fixture_id(r):
    return f"{r['kind']}-{r['n']}-s{r['seed']}-L{r['level']}"


@pytest.mark.parametrize("rec", [r for r in bench_fixtures(max_n=2_000_000_000)
                                 if r["seed"] <= 5 and not (r["kind"] == "wiki" and r["seed"] not in (1, 2))],
                         ids=_fixture_id)
def test_baseline_configs_vs_reference(lib, rec):
    """Every BASELINE.json configuration that fits one GPU, at full size, against the stream of the
    compiled reference (tests/golden/bench_fixtures.json, make_bench_fixtures.py): C1 enwik8-sized
    and C2 enwik9-sized enwik-like text, the round-1 word soup, C3 mixed entropy at -1 and -9,
    C4 random bytes (112 slabs and one GPU's eighth of 10 GB), C5 tar-like source tree (whole and
    one GPU's eighth).  Device-resident path; the stream is hashed on the host."""
    import torch
    data = gen(rec["kind"], rec["n"], rec["seed"])
    assert md5(data) == rec["in_md5"]
    n, lvl = rec["n"], rec["level"]
    M = lvl * 100000
    src = torch.frombuffer(data, dtype=torch.uint8).cuda() if isinstance(data, bytearray) else torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    with lib.context(lvl, min((n + M - 1) // M, 2400)) as ctx:
        m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
        st = ctx.stats()
    out = dst[:m].cpu().numpy().tobytes()
    assert m == rec["out_len"] and st.nblocks == rec["blocks"]
    if md5(out) != rec["canon_md5"]:
        # localise: the body cut at every 100th slab is in the fixture
        assert out[:4] == b"BZh" + bytes([48 + lvl])
        pytest.fail(f"stream md5 differs from the reference's ({rec['config']}); combined CRC "
                    f"{int.from_bytes(out[-4:], 'big'):#x} vs {rec['combined_crc']:#x}")
    assert st.nperiodic == rec["periodic_blocks"]
    if rec["periodic_blocks"] == 0:
        assert rec["canon_md5"] == rec["ref_md5"]
    if n <= 200_000_000:
        assert bz2.decompress(out) == bytes(data)


def test_c4_ten_gigabytes_as_specified(lib):
    """BASELINE.json configs[3] at the size it names: 10^10 random bytes at -9 -- `rand(10^10, seed 4)` -- through ONE context
    in ONE call.  The context holds 1112 slabs, so the 11 112 slabs stream through it in ten chunks (stream position and CRC
    fold carried from chunk to chunk in lbz_stream_state); input and stream are resident in HBM (20 GB of the 288), the host
    only ever holds a piece: the input is generated piece by piece (lbzgen_rand_from continues the sequence) and the stream is
    hashed piece by piece.  Fixture: the compiled reference on the same pieces (make_bench_fixtures.py c4full)."""
    import ctypes as C
    import hashlib
    import torch
    rec = [r for r in bench_fixtures(min_n=10_000_000_000) if r["kind"] == "rand"]
    assert rec, "tests/golden/bench_fixtures.json has no 10^10-byte record (make_bench_fixtures.py c4full)"
    rec = rec[0]
    n, lvl = rec["n"], rec["level"]
    g = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lbzip2_amd", "host", "libgen_inputs.so"))
    g.lbzgen_rand_from.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    g.lbzgen_rand_from.restype = None
    src = torch.empty(n, dtype=torch.uint8, device="cuda")
    dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    piece = 500_000_000
    bufs = [torch.empty(piece, dtype=torch.uint8).pin_memory() for _ in range(2)]
    state = C.c_uint32(rec["seed"])
    hin = hashlib.md5()
    with ThreadPoolExecutor(1) as ex:
        pending = None
        for i, off in enumerate(range(0, n, piece)):
            ln = min(piece, n - off)
            b = bufs[i & 1]
            g.lbzgen_rand_from(b.data_ptr(), ln, C.byref(state))          # (the hash of the piece before runs beside it)
            if pending is not None:
                pending.result()
            src[off:off + ln].copy_(b[:ln], non_blocking=False)
            pending = ex.submit(hin.update, memoryview(b.numpy())[:ln])
        pending.result()
    assert hin.hexdigest() == rec["in_md5"]
    with lib.context(lvl, 1112) as ctx:
        m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
        st = ctx.stats()
    assert m == rec["out_len"] and st.nblocks == rec["blocks"] and st.nperiodic == 0
    hout = hashlib.md5()
    for off in range(0, m, piece):
        ln = min(piece, m - off)
        bufs[0][:ln].copy_(dst[off:off + ln])
        hout.update(memoryview(bufs[0].numpy())[:ln])
    tail = dst[m - 4:m].cpu().numpy().tobytes()
    assert int.from_bytes(tail, "big") == rec["combined_crc"]
    assert hout.hexdigest() == rec["ref_md5"]


@pytest.mark.parametrize("rec", load("stages.json"),
                         ids=lambda r: f"{r['name']}-L{r['level']}-b{r['block']}")
def test_stage_goldens(lib, rec):
    """Every intermediate of a block against the values dumped from the reference."""
    data = gen(rec["kind"], rec["n"], rec["seed"])
    M = rec["level"] * 100000
    with lib.context(rec["level"], (len(data) + M - 1) // M) as ctx:
        b = ctx.blocks(data, 3)[rec["block"]]
    assert b["err"] == 0
    for k in ("consumed", "nblock", "crc", "bwt_idx", "nmtf", "alpha", "num_trees", "num_selectors", "out_len"):
        assert b[k] == rec[k], k
    assert md5(b["inuse"]) == rec["inuse_md5"]
    assert md5(b["block"]) == rec["block_md5"]
    assert md5(b["bwt"]) == rec["bwt_md5"]
    assert md5(b["mtfv"]) == rec["mtfv_md5"]
    assert md5(b["out"]) == rec["out_md5"]


@pytest.mark.parametrize("level", [9, 1])
def test_reference_suite_corpora(lib, level):
    """All 1093 inputs of the reference's compress suites (manual-compress, fuzz-collect,
    fuzz-divbwt), byte-identical to reference lbzip2's output (md5 fixtures)."""
    inputs = suite_inputs()
    exp = load("suite_expected.json")
    bad = []
    with lib.context(level, 10) as ctx:
        for name in sorted(inputs):
            out = ctx.compress(inputs[name])
            e = exp[name][str(level)]
            if len(out) != e["len"] or md5(out) != e["canon_md5"]:
                bad.append(name)
            if e["periodic_blocks"] == 0 and e["canon_md5"] != e["ref_md5"]:
                bad.append(name + ":fixture")
    assert not bad, bad[:10]


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 9, 50, 51, 255, 256, 4097, 99999, 100000, 100001,
                               899999, 900000, 900001, 1800017])
def test_ragged_sizes_vs_oracle(lib, n):
    for kind, lvl in (("text", 9), ("rand", 1)):
        data = gen(kind, n, n % 97 + 1)
        assert lib.compress(data, lvl) == L.orc_compress(data, lvl), (kind, n)


def test_repetitive_source_like_text(lib):
    """Long exact repeats and indentation runs (what source code and markup look like): large
    groups, deep ties, the prefix-doubling finish."""
    for n, seed in ((3_000_000, 4), (900_000, 5)):
        data = gen("lines", n, seed)
        for lvl in (9, 1):
            want, _ = cpu_reference(data, lvl)
            assert lib.compress(data, lvl) == want, (n, lvl)


def test_levels_1_to_9(lib):
    data = gen("text", 1234567, 77)
    for lvl in range(1, 10):
        assert lib.compress(data, lvl) == L.orc_compress(data, lvl), lvl


def test_rle_boundary_cases(lib):
    """Runs landing on the block limit (the cases of tests/suite/manual-compress), at -1."""
    M = 100000
    filler = bytes((i * 7 + 1) % 251 for i in range(M))
    with lib.context(1, 2) as ctx:
        for run in list(range(1, 9)) + [258, 259, 260, 263, 264, 518, 519]:
            for before in range(0, 9):
                data = (filler[:M - before] + bytes([0xAA]) * run + b"xyz" + filler[:300])[:2 * M]
                assert ctx.compress(data) == L.orc_compress(data, 1), (run, before)


def test_blocks_of_about_one_batch(lib):
    """Block sizes on both sides of a batch of k_bwt_batch (832 rows: sorted whole in LDS up to there, partitioned beyond), of the
    old limit (1024) and of the 64-row strips, on text, three symbols, random bytes and a short period (tests/test_emu_kernels.py
    holds the same under the emulator, stage by stage)."""
    with lib.context(1, 1) as ctx:
        for n in (63, 64, 65, 127, 128, 129, 191, 193, 767, 831, 832, 833, 895, 897, 1023, 1024, 1025, 1663, 1664, 1665, 2500, 4159, 4161):
            for data in (bytes(gen("text", n, n)), bytes((i * i + (i >> 2)) % 3 + 65 for i in range(n)), bytes(gen("rand", n, n + 1)), (b"abcab" * n)[:n]):
                assert ctx.compress(data) == L.orc_compress(data, 1), n


def test_periodic_blocks_documented_divergence(lib):
    """T = u^k: identical stream except the origin pointer, which is the smallest equal row."""
    for data in (b"ab" * 450000, b"abc" * 1000, b"\x01" * 3, b"xy" * 50000):
        got = lib.compress(data, 9)
        assert got == L.orc_compress(data, 9)
        assert bz2.decompress(got) == data
        if L.have_ref():
            r = bytearray(L.ref_compress(data, 9)); g = bytearray(got)
            assert len(r) == len(g)
            r[14:18] = g[14:18] = b"\0" * 4       # stream header 4 + block bytes 10..13
            assert r == g


def test_periodic_corpus_blocks_enumerated(lib):
    """Every exactly periodic block of the reference's corpora (tests/golden/periodic_blocks.json: the
    reference's origin pointer, k, the smallest equal row): the GPU emits the smallest equal row and
    flags the block periodic; all other blocks are bit-exact (test_reference_suite_corpora)."""
    pb = load("periodic_blocks.json")
    inputs = suite_inputs()
    by_input = {}
    for e in pb:
        by_input.setdefault((e["input"], e["level"]), []).append(e)
    for (name, lvl), entries in sorted(by_input.items()):
        raw = inputs[name]
        M = lvl * 100000
        with lib.context(lvl, max(1, (len(raw) + M - 1) // M)) as ctx:
            blocks = ctx.blocks(raw, 1)
        for e in entries:
            b = blocks[e["block"]]
            assert b["periodic"] and b["bwt_idx"] == e["canon_bwt_idx"], e
            assert e["canon_bwt_idx"] == e["ref_bwt_idx"] - e["ref_bwt_idx"] % e["copies"]


def test_workunit_interface(lib):
    """The drop-in encode.h functions driven with compress.c's call sequence."""
    data = gen("text", 2500000, 5) + gen("runs", 950000, 6)
    want, _ = cpu_reference(data, 9)
    assert lib.compress_workunits(data, 9) == want


def test_workunit_interface_threads(lib):
    """Encoders are independent: drive them from several host threads (compress.c:81-115)."""
    datas = [gen("text", 400000 + 1000 * i, 100 + i) for i in range(8)]
    with ThreadPoolExecutor(8) as ex:
        outs = list(ex.map(lambda d: lib.compress_workunits(d, 5), datas))
    for d, o in zip(datas, outs):
        assert o == L.orc_compress(d, 5)


def test_workunit_pool_smaller_than_threads(lib, monkeypatch):
    """More caller threads than the shared pool has slabs: states wait for a slab, rounds stay
    small, nothing deadlocks (the pool of a block size is created once per process: -2 is used
    by no other test, so the environment still applies)."""
    monkeypatch.setenv("LBZAMD_POOL_SLABS", "3")
    datas = [gen("text", 150000 + 7000 * i, 200 + i) + gen("runs", 90000, i) for i in range(12)]
    with ThreadPoolExecutor(12) as ex:
        outs = list(ex.map(lambda d: lib.compress_workunits(d, 2), datas))
    for d, o in zip(datas, outs):
        assert o == L.orc_compress(d, 2)


def test_c_host_driver(lib, tmp_path):
    """The C program that drives the work-unit interface from pthreads the way compress.c does
    (lbzip2_amd/host/lbzamd_compress.c), and its batch mode."""
    import subprocess
    exe = os.path.join(os.path.dirname(lib.path), "..", "host", "lbzamd_compress")
    if not os.path.exists(exe):
        pytest.skip("C driver not built")
    data = gen("text", 3_000_000, 21) + gen("runs", 1_000_000, 22)
    src = tmp_path / "in.bin"
    src.write_bytes(data)
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    want, _ = cpu_reference(data, 9)
    for extra in ([], ["-w", "6"]):
        out = subprocess.run([exe, "-9"] + extra, stdin=open(src, "rb"), capture_output=True, env=env, timeout=300)
        assert out.returncode == 0, out.stderr[-500:]
        assert out.stdout == want, extra


def test_file_splitter_muxer_full_size(lib, tmp_path):
    """SURVEY 8f-1 on the device: the enwik9-sized stand-in goes file -> reader thread -> pinned ring ->
    two pipelines (own contexts, body-only slab ranges of 256 slabs) -> writer thread -> file, never resident
    as a whole; the file must be the reference's stream (fixture md5)."""
    import hashlib
    import subprocess
    exe = os.path.join(os.path.dirname(lib.path), "..", "host", "lbzamd_compress")
    if not os.path.exists(exe):
        pytest.skip("C driver not built")
    rec = [r for r in bench_fixtures() if r["kind"] == "wiki" and r["n"] == 1_000_000_000 and r["seed"] == 2][0]
    src, dst = tmp_path / "wiki.bin", tmp_path / "wiki.bz2"
    src.write_bytes(gen(rec["kind"], rec["n"], rec["seed"]))
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([exe, "-9", "-f", str(src), "-o", str(dst), "-c", "256", "-p", "2", "-t"], capture_output=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-500:]
    print(p.stderr.decode()[-300:])
    out = dst.read_bytes()
    assert len(out) == rec["out_len"] and hashlib.md5(out).hexdigest() == rec["ref_md5"]


def test_c_side_multi_gpu(lib, tmp_path):
    """The C host side over every device of the box (one on the test box -- the per-device code path still runs):
    `lbzamd_compress -f/-o -g 0` (pipelines' contexts dealt over the devices, one reader, one writer) and the
    reference's work-unit calls with LBZAMD_DEVICES=all (one pool per device, slabs leased round-robin) must both
    write the reference's stream.  process.c:515-548, compress.c:73-118,238-250."""
    import hashlib
    import subprocess
    exe = os.path.join(os.path.dirname(lib.path), "..", "host", "lbzamd_compress")
    if not os.path.exists(exe):
        pytest.skip("C driver not built")
    rec = [r for r in bench_fixtures() if r["kind"] == "tar" and r["n"] == 175_000_000][0]
    data = bytes(gen(rec["kind"], rec["n"], rec["seed"]))
    src, dst = tmp_path / "tar.bin", tmp_path / "tar.bz2"
    src.write_bytes(data)
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([exe, "-9", "-f", str(src), "-o", str(dst), "-c", "64", "-p", "2", "-g", "0", "-t"], capture_output=True, env=env, timeout=300)
    assert p.returncode == 0 and b"device(s)" in p.stderr, p.stderr[-500:]
    out = dst.read_bytes()
    assert len(out) == rec["out_len"] and hashlib.md5(out).hexdigest() == rec["ref_md5"]
    p = subprocess.run([exe, "-9", "-w", "64"], input=data, capture_output=True, env=dict(env, LBZAMD_DEVICES="all"), timeout=300)
    assert p.returncode == 0, p.stderr[-500:]
    assert len(p.stdout) == rec["out_len"] and hashlib.md5(p.stdout).hexdigest() == rec["ref_md5"]


def test_n_devices_on_one_gpu(lib, tmp_path):
    """N > 1 on the real kernels of a ONE-GPU box: LBZAMD_FAKE_DEVICES=2 shows two logical devices (both on the device
    present), so that `lbzamd_compress -g 2 -p 2` deals its pipelines' contexts over two devices and LBZAMD_DEVICES=2
    keeps two work-unit pools with their own leaders, streams and staging behind the reference's five symbols.  Both
    must write the reference's stream.  (With the emulator the same branches run on fake devices: test_emu_kernels.py.)"""
    import hashlib
    import subprocess
    exe = os.path.join(os.path.dirname(lib.path), "..", "host", "lbzamd_compress")
    if not os.path.exists(exe):
        pytest.skip("C driver not built")
    assert lib.lib.lbzamd_device_count() >= 1
    rec = [r for r in bench_fixtures() if r["kind"] == "tar" and r["n"] == 175_000_000][0]
    data = bytes(gen(rec["kind"], rec["n"], rec["seed"]))
    src, dst = tmp_path / "tar.bin", tmp_path / "tar.bz2"
    src.write_bytes(data)
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""), LBZAMD_FAKE_DEVICES="2")
    p = subprocess.run([exe, "-9", "-f", str(src), "-o", str(dst), "-c", "48", "-p", "2", "-g", "2", "-t"], capture_output=True, env=env, timeout=300)
    assert p.returncode == 0 and b"2 device(s)" in p.stderr, p.stderr[-500:]
    out = dst.read_bytes()
    assert len(out) == rec["out_len"] and hashlib.md5(out).hexdigest() == rec["ref_md5"]
    p = subprocess.run([exe, "-9", "-w", "64"], input=data, capture_output=True,
                       env=dict(env, LBZAMD_DEVICES="2", LBZAMD_POOL_SLABS="48"), timeout=300)
    assert p.returncode == 0, p.stderr[-500:]
    assert len(p.stdout) == rec["out_len"] and hashlib.md5(p.stdout).hexdigest() == rec["ref_md5"]


GLOO_WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
import lbzip2_amd
from lbzip2_amd.shard import compress_sharded
from golden_util import gen, bench_fixtures
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
lib = lbzip2_amd.library()                      # the product library: HIP kernels on cuda:0 in every rank
rec = [r for r in bench_fixtures() if r["kind"] == "wiki" and r["n"] == 100_000_000][0]
data = bytes(gen(rec["kind"], rec["n"], rec["seed"]))
whole = compress_sharded(lib, data, rec["level"], dist, "cpu")      # bodies leave the device, gloo carries them
if rank == 0:
    assert len(whole) == rec["out_len"] and hashlib.md5(whole).hexdigest() == rec["ref_md5"], (len(whole), rec["out_len"])
    print("GLOO_MUX_OK", world, len(whole))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_stream_mux_over_gloo_on_the_real_kernels(tmp_path, world):
    """lbzip2_amd.shard.StreamMux with `world` ranks whose compressors are the HIP kernels (all on the one GPU of the test
    box): slab ranges of wiki(10^8) -> body-only bytes + 12-byte CRC partials -> ONE stream on rank 0, which must be the
    reference's.  The transport is gloo (RCCL does not take two ranks on one device: test_rccl_leg_of_the_stream_mux);
    everything else -- shard_plan, lbzamd_compress_*_body on the GPU, the fold of the partials, header and trailer -- is
    the code bench.py --scaling strong runs."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER.format(root=root))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(29641 + world), str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GLOO_MUX_OK" in out.stdout


def test_rccl_leg_of_the_stream_mux(tmp_path):
    """lbzip2_amd.shard.StreamMux over RCCL (all_gather of the partials, grouped send/recv of the bodies between device
    buffers): two ranks of `bench.py --scaling strong`.  The test box has ONE GPU; RCCL may refuse two ranks on one
    device ("Duplicate GPU detected") -- then the leg can only run in the driver's multi-GPU bench and this test says so."""
    import json
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ndev = torch.cuda.device_count()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if ndev < 2:
        env["LBZ_BENCH_ONE_DEVICE"] = "1"            # both ranks on cuda:0
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--scaling", "strong",
           "--bytes", "100000000", "--seed", "1", "--no-cpu", "--no-host", "--no-isolated", "--no-decode", "--no-seq", "--no-legs"]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=240)
    except subprocess.TimeoutExpired:
        pytest.skip("two RCCL ranks on one device did not finish in 240 s: the RCCL leg needs two GPUs (driver's SCALE run)")
    if p.returncode != 0:
        text = p.stdout + p.stderr
        why = [l for l in text.splitlines() if any(k in l for k in ("Duplicate GPU", "NCCL", "RCCL", "nccl", "Error", "error"))]
        if ndev < 2:          # one GPU: whatever RCCL makes of two ranks on it, the leg belongs to the driver's multi-GPU bench
            pytest.skip("two RCCL ranks on ONE device do not run here (the leg runs in the driver's multi-GPU bench): " + " | ".join(why[:4])[:600])
        assert False, text[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["verified"] is True, line[:600]


def test_repeated_passes_are_identical(lib):
    """The segment workgroups of a block share its rank table while they refine it (k_bwt.hip, ISA_ENTRY): a race between
    them would show up as a rare mismatch.  Eight passes over the 10^8-byte fixture, every stream hashed."""
    import hashlib
    import torch
    rec = [r for r in bench_fixtures() if r["kind"] == "wiki" and r["n"] == 100_000_000][0]
    data = gen(rec["kind"], rec["n"], rec["seed"])
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    dst = torch.empty(lib.bound(rec["n"]), dtype=torch.uint8, device="cuda")
    with lib.context(9, 112) as ctx:
        for p in range(8):
            m = ctx.compress_device(src.data_ptr(), rec["n"], dst.data_ptr(), dst.numel())
            assert m == rec["out_len"] and hashlib.md5(dst[:m].cpu().numpy().tobytes()).hexdigest() == rec["ref_md5"], p


def test_full_size_property(lib):
    """BASELINE-sized behaviour by properties: 60 MB of text at -9 with chunked streaming
    (resident capacity smaller than the input) round-trips through an independent decoder,
    equals the CPU reference, and the stream CRC is the fold of the block CRCs (refolded here)."""
    data = gen("text", 60_000_000, 2)
    with lib.context(9, 24) as ctx:
        out = ctx.compress(data)
        st = ctx.stats()
    assert st.n_in == len(data) and st.n_out == len(out) and st.nblocks >= 67
    assert bz2.decompress(out) == data
    if L.have_ref():
        # the reference on all host cores (oracle/cpu_mt.h): same bytes, and the stream CRC is the fold of
        # the block CRCs in order (encode.h:38), block by block and range by range
        want, blocks, _ = L.ref_compress_mt(data, 9, os.cpu_count() or 1)
        assert out == want
        import lbzip2_amd
        cc = 0
        for _olen, crc, _idx, _k, _slab in blocks:
            cc = lbzip2_amd.combine_crc(cc, crc)
        assert cc == int.from_bytes(out[-4:], "big")
        half = len(blocks) // 2
        parts = []
        for rng in (blocks[:half], blocks[half:]):
            f = 0
            for _olen, crc, _idx, _k, _slab in rng:
                f = lbzip2_amd.combine_crc(f, crc)
            parts.append((len(rng), f))
        assert lbzip2_amd.fold_parts(0, parts) == cc
    else:
        assert out == L.orc_compress_mt(data, 9, os.cpu_count() or 1)[0]
    assert out[:4] == b"BZh9" and out[-10:-4] == bytes([0x17, 0x72, 0x45, 0x38, 0x50, 0x90])


@pytest.mark.parametrize("streams,nslots,max_slabs", [("1", 3, 14), ("2", 3, 14), ("2", 4, 14), ("3", 2, 14), ("2", 5, 9)])
def test_round_schedule(lib, monkeypatch, streams, nslots, max_slabs):
    """Multi-round, multi-stream scheduling (what the 1 GB benchmark runs with 2 x 556 slots) at a
    size the CPU reference checks in a second: rounds of nslots slabs on `streams` streams, short
    last round first, spill blocks (large ones: the `runs` part expands under RLE1) riding with
    their primaries, and a second chunk when max_slabs < 14."""
    data = gen("text", 6_000_000, 21) + gen("runs", 3_000_000, 22) + gen("rand", 2_000_000, 23) + gen("lines", 1_500_000, 24)
    want = cpu_reference(data, 9)[0]
    monkeypatch.setenv("LBZAMD_STREAMS", streams)
    with lib.context(9, max_slabs, nslots) as ctx:
        for _ in range(2):                                    # the second call reuses slots and events
            assert ctx.compress(data) == want


@pytest.mark.parametrize("knob,value", [("LBZAMD_HANDOVER0", "1"), ("LBZAMD_HANDOVER1", "1"), ("LBZAMD_HANDOVER0", "0")])
def test_hand_over_rules_forced(lib, monkeypatch, knob, value):
    """The sorter's fall-back with the hand-over rules forced (lbz_api.hip: launch_sort): every block to the rank rounds before
    its first text launch / after it -- they then meet the CLOSED runs (k_bwt.hip) that the batches and the first launch left
    tied in the suffix array -- and no block handed over on what the batches leave tied.  The reference's stream each time:
    text (long runs in every block), sources and a tar of this image's files, rounds of 5 slabs."""
    import bench
    made = bench.real_tar(4_000_000)
    data = bytes(gen("wiki", 9_000_000, 31)) + bytes(gen("lines", 2_000_000, 32)) + (bytes(made[0]) if made else b"")
    want = cpu_reference(data, 9)[0]
    monkeypatch.setenv(knob, value)
    with lib.context(9, 17, 5) as ctx:
        for _ in range(2):
            assert ctx.compress(data) == want


def test_fuzz_vs_oracle(lib):
    """Structured random inputs (tests/fuzz_gpu.py): tiny alphabets (oversized groups, pre-split,
    trimmed batches), runs at the RLE1 limits, near-periodic and periodic text, word soups."""
    import fuzz_gpu
    assert fuzz_gpu.run(lib, L.orc_compress, 11, 150) == []
    assert fuzz_gpu.run(lib, L.orc_compress, 12, 4, big=True) == []


def test_fuzz_workload_slices_vs_reference(lib):
    """Random slices and splices of the workload generators (deep ties, oversized groups, verbatim
    duplicates) against the compiled reference (tests/fuzz_corpora_gpu.py)."""
    import fuzz_corpora_gpu
    assert fuzz_corpora_gpu.run(lib, L, 31, 10) == []


def test_device_resident_api(lib):
    import torch
    data = gen("text", 3_000_000, 8)
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    dst = torch.empty(lib.bound(len(data)), dtype=torch.uint8, device="cuda")
    with lib.context(9, 4) as ctx:
        n = ctx.compress_device(src.data_ptr(), len(data), dst.data_ptr(), dst.numel())
        torch.cuda.synchronize()
        assert bytes(dst[:n].cpu().numpy()) == L.orc_compress(data, 9)
        # too-small output buffer must be reported, not overrun
        import lbzip2_amd
        with pytest.raises(lbzip2_amd.LbzError):
            ctx.compress_device(src.data_ptr(), len(data), dst.data_ptr(), 1000)
