"""CPU: the parts of bench.py that do not need a GPU -- the input generators (the BASELINE.json
stand-ins of lbzip2_amd/host/gen_inputs.c), the fixture lookup and the cpu_baseline leg (the only
place outside tests/ and smoke() that may call into oracle/)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
import oracle_lib as L  # noqa: E402
from golden_util import bench_fixtures  # noqa: E402


def test_generator_matches_oracle_generator():
    n = 300000
    assert bytes(bench.gen_input("text", n, 2)) == L.gen_text(n, 2)
    assert bytes(bench.gen_input("rand", n, 7)) == L.gen_rand(n, 7)
    # known answers of SURVEY.md App. B
    assert L.gen_rand(8, 1).hex() == "00049d128e2c2519"


def test_generators_match_fixture_inputs():
    """The small fixtures pin the generators: same bytes here as when the reference compressed them;
    a prefix of a longer run of the same seed is the same text (strong-scaling slices are prefixes)."""
    for r in bench_fixtures(max_n=3_000_000):
        data = bench.gen_input(r["kind"], r["n"], r["seed"])
        assert hashlib.md5(data).hexdigest() == r["in_md5"], (r["kind"], r["n"])
        assert bench.find_fixture(r["kind"], r["n"], r["seed"], r["level"])["ref_md5"] == r["ref_md5"]
    big = bench.gen_input("wiki", 2_000_000, 2)
    assert bytes(big[:350000]) == bytes(bench.gen_input("wiki", 350000, 2))
    assert len(set(big)) >= 190                      # byte alphabet: 8-bit sort symbols
    assert bench.find_fixture("wiki", 1_000_000_000, 2, 9)["out_len"] > 2 * 10**8


def test_cpu_baseline_leg():
    data = bench.gen_input("wiki", 24 * 900000, 2)
    r = bench.cpu_baseline(data, 9, seconds_budget=3.0)
    assert r["kind"] in ("reference", "port") and r["unit"] == "MB/s"
    assert r["value"] > 0 and 1 <= r["cores"] <= r["usable_cpus"]
    assert "slabs of 900000 B" in r["sample"] and "1" in r["MBps_by_threads"]
    if r["usable_cpus"] >= 4 and r["cgroup_cpu_quota"] is None:
        assert r["value"] > 1.5 * r["MBps_by_threads"]["1"], r    # the pthreads driver scales


def test_reference_driver_equals_single_thread_driver():
    """oracle/cpu_mt.h on N threads == the single-thread reference stream == the oracle."""
    data = bytes(bench.gen_input("mixed", 1_300_000, 3))
    if L.have_ref():
        a = L.ref_compress_mt(data, 1, 4)[0]
        assert a == L.ref_compress(data, 1)
        assert a == L.orc_compress_mt(data, 1, 3)[0]
    assert L.orc_compress_mt(data, 1, 2)[0] == L.orc_compress(data, 1)


def test_real_file_leg_input():
    """bench.py's real-data legs compress a tar built from files of the image: the same bytes every time it is built
    (sorted members, zeroed owner and time), a valid archive, cut at the size asked for, no member twice."""
    import io
    import tarfile
    roots, suf = ["/usr/lib/python3.10", "/usr/lib/python3"], (".py",)
    a = bench.real_tar(3_000_000, roots, suf)
    if a is None:
        import pytest
        pytest.skip("the file set is not in this container")
    b = bench.real_tar(3_000_000, roots, suf)
    assert a[1:] == b[1:] and bytes(a[0]) == bytes(b[0]) and len(a[0]) == 3_000_000
    names = []
    with tarfile.open(fileobj=io.BytesIO(bytes(a[0]) + bytes(1024)), mode="r:") as tf:      # (cut mid-member: pad so that the reader ends cleanly)
        try:
            for m in tf:
                names.append(m.name)
        except (tarfile.ReadError, EOFError):
            pass
    assert len(names) >= a[1] - 1 and len(set(names)) == len(names) and names == sorted(names)


def test_bare_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` typed bare, the way the driver types `--gpus 1` (no torchrun around it): bench.py starts its own
    two ranks, and the ONE line rank 0 prints carries the weak figure, the strong leg (one input -> one stream over both ranks,
    verified against the reference fixture), the host -> host leg of both ranks, the command over two devices, the roofline and
    the backend.  LBZ_BENCH_EMU (a test hook, see bench.py) puts the emulated kernel build and gloo where the GPU box has the
    HIP library and RCCL; everything else -- launcher, ranks, collectives, the line -- is the code the driver runs."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tests", "emu"), "WG=1024"])
    env = dict(os.environ, LBZ_BENCH_EMU="1", LBZ_EMU_THREADS="2", LBZ_EMU_DEVICES="2",
               LBZ_BENCH_LBZAMD=os.path.join(root, "tests", "emu", "_build", "lbzamd_emu"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--bytes", "350000",
                          "--level", "1", "--seed", "2", "--no-cpu"], env=env, capture_output=True, text=True, timeout=1500, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["backend"].startswith("gloo") and r["emulated"] is True
    assert r["scaling"] == "weak" and r["value"] > 0 and r["verified"] is True
    assert r["strong"]["verified"] is True and r["strong"]["value"] > 0 and r["strong"]["value_from_rank0"] > 0
    assert r["strong"]["slabs_per_rank"] == [2, 2]
    assert r["value_host"]["ranks"] == 2 and r["value_host"]["same_stream"] is True
    assert r["value_node_file"]["verified"] is True and r["value_node_file"]["devices"] == 2
    assert r["roofline"]["bound"] == "hbm" and "frac" in r["roofline"] and "isolated" in r["roofline"]      # (the emulator's events time nothing)


def test_a_hanging_extra_does_not_cost_the_line(tmp_path):
    """N > 1: the untimed extras (host buffers on all ranks, the strong leg's send/recv, the node's file leg) hold collectives that
    have never run on two devices; if one of them does not come back, rank 0 prints the weak-scaling result -- measured by then
    -- and every rank leaves with status 0 (bench.py: the timer behind the timed region).  Here the timer is set to fire at once."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LBZ_BENCH_EMU="1", LBZ_EMU_THREADS="2", LBZ_EMU_DEVICES="2", LBZ_BENCH_EXTRAS_TIMEOUT="0.001")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--bytes", "350000",
                          "--level", "1", "--seed", "2", "--no-cpu"], env=env, capture_output=True, text=True, timeout=1500, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 0 and r["verified"] is True and "extras" in r
