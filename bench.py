#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: bzip2 -9 compress throughput.

    python bench.py --gpus N --steps K --warmup W           (N > 1 without WORLD_SIZE: starts its own N ranks, below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (RLE1+CRC -> BWT -> MTF/ZRLE -> prefix codes -> bit
packing -> stream assembly) over the workload, input already resident in HBM, the .bz2 bytes
left in HBM.  Workload (BASELINE.json configs[1]): enwik9-sized text, 10^9 bytes, level -9, per
GPU.  The real enwik9 is used if $LBZ_ENWIK9 points at it ($LBZ_ENWIK8, $LBZ_SILESIA, $LBZ_LINUX_TAR add legs on the other
real corpora of BASELINE.json; each is checked on the box against the compiled reference); otherwise the seeded ENWIK-LIKE
stand-in `wiki(10^9, seed 2+rank)` of lbzip2_amd/host/gen_inputs.c: XML page wrappers, wiki markup,
Zipf words and phrases, UTF-8 interwiki text (191 distinct bytes -> 8-bit sort symbols) and verbatim
passage repeats (deep ties) -- the profile of real English text (tied rows by depth within a few
points of a GNU-manual corpus, bzip2 ratio 4.3 vs enwik9's 3.9).  `--kind text` is round 1's
28-letter word soup (kept as a second workload: it is the sorter's best case).

--scaling weak (default)   every rank compresses its own 10^9-byte input into its own stream.
--scaling strong           ONE 10^9-byte input, slab ranges dealt over the ranks, bodies gathered
                           to rank 0 into ONE stream that is byte-identical to the single-GPU
                           (and the reference's) stream (lbzip2_amd/shard.py, RCCL send/recv).

N > 1: the line's `value` is the weak-scaling figure (per-GPU work fixed) and the same run adds `strong`: rank 0's 10^9-byte
input dealt over the N ranks as slab ranges and gathered into ONE stream on rank 0, md5 against the reference fixture
(`strong.value`: ranges resident per GPU; `strong.value_from_rank0`: scatter from rank 0 over xGMI included), `value_host`
(every rank host buffer -> host buffer at once) and `value_node_file` (the command with --devices=N, file -> file on tmpfs).

After the timed region the stream is copied to the host, hashed and compared with the fixture
generated from the compiled reference (tests/golden/bench_fixtures.json): "verified": true/false,
null if no fixture exists for the workload.  Nothing in the timed region touches the host.

Extra objects in the JSON line:
  roofline      the kernel with the largest share of device time, timed live in the timed region
                with HIP events on the stream each launch goes to.  SURVEY 8(d) prices the path at
                N_in + 13 N_rle + 20 N_mtf + N_out algorithmic bytes; per kernel:
                collect N_in+N_rle | bwt_part 5 N_rle | bwt_batch+fix 6 N_rle | mtf N_rle+2 N_mtf |
                encode 18 N_mtf+N_out.  peak 8000 GB/s (MI355X HBM3E).  roofline.isolated: the same
                table from one extra untimed single-stream pass (no overlap between rounds).
  value_host    host buffer in -> .bz2 bytes in host memory (pinned, PCIe-inclusive), same input, timed the
                same way as `value`: K steps between barriers, wall clock (SURVEY 8(d)'s definition of the
                metric; `value` itself is the device-resident rate the bench contract asks for).
  configs       the other single-GPU BASELINE.json configurations (C1 10^8 text, C3 mixed -1/-9, C4's and
                C5's per-GPU share) at full size, device-resident MB/s, each verified against its fixture.
  value_file    the command `lbzamd FILE` on tmpfs (file -> file, process start to exit), stream checked against the fixture.
  decode        the inverse path on the stream just written (untimed by the driver): decoded MB/s, round trip checked.
  cpu_baseline  reference lbzip2's block codec (oracle/_ref: "reference") or the restatement
                (oracle/: "port") on the box's host cores through the pthreads driver of
                oracle/cpu_mt.h, over a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E
KINDS = ["wiki", "text", "rand", "mixed", "tar"]
KERNELS = ["k_collect", "k_bwt_part", "k_bwt_batch", "k_bwt_fix", "k_mtf", "k_encode"]
# the other single-GPU configurations of BASELINE.json (stand-ins of tests/golden/bench_fixtures.json)
LEGS = [("C1 enwik8-like", "wiki", 100_000_000, 1, 9), ("C3 silesia-like -1", "mixed", 211_938_580, 3, 1),
        ("C3 silesia-like -9", "mixed", 211_938_580, 3, 9), ("C4 random, one GPU's eighth", "rand", 1_250_000_000, 4, 9),
        ("C5 tarball-like", "tar", 1_400_000_000, 5, 9)]


def gen_input(kind, n, seed):
    g = C.CDLL(os.path.join(ROOT, "lbzip2_amd", "host", "libgen_inputs.so"))
    buf = bytearray(n)
    cbuf = (C.c_uint8 * n).from_buffer(buf)
    fn = getattr(g, "lbzgen_" + kind)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
    fn.restype = None
    fn(cbuf, n, seed)
    del cbuf
    return buf


def usable_cpus():
    """CPUs this process may run on, and the cgroup CPU quota if there is one."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(p).read().split()
            if p.endswith("cpu.max"):
                quota = None if t[0] == "max" else float(t[0]) / float(t[1])
            else:
                q = float(t[0])
                quota = None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    return aff, quota


def cpu_baseline(data, level, seconds_budget=12.0):
    """The CPU codec on the host cores through the C/pthreads driver (one reusable encoder per thread,
    slabs handed out from a shared counter, in-order mux), on a bounded sample of the same workload:
    1, 16, 64 and all usable threads, each at least two slabs per thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as L   # test infrastructure: only used here as the timed CPU baseline
    kind = "reference" if L.have_ref() else "port"
    fn = L.ref_compress_mt if kind == "reference" else L.orc_compress_mt
    M = level * 100000
    aff, quota = usable_cpus()
    nslabs = max(1, len(data) // M)
    t1 = min(fn(data[:min(3, nslabs) * M], level, 1)[2] for _ in range(2))
    rate1 = min(3, nslabs) * M / t1
    table = {"1": round(rate1 / 1e6, 2)}
    best, best_nt, best_sample = rate1, 1, min(3, nslabs) * M
    for nt in sorted({min(16, aff), min(64, aff), aff}):
        if nt <= 1:
            continue
        # ~seconds_budget/3 seconds of wall time per point at perfect scaling, >= 2 slabs per thread
        per = max(2, int(seconds_budget / 3 * rate1 / M))
        ns = min(nslabs, nt * per)
        t = min(fn(data[:ns * M], level, nt)[2] for _ in range(2))     # best of two: the first call warms pages and threads
        r = ns * M / t
        table[str(nt)] = round(r / 1e6, 2)
        if r > best:
            best, best_nt, best_sample = r, nt, ns * M
    note = None
    if best < 0.5 * rate1 * best_nt:
        note = ("throughput does not scale with threads: the lease's CPU quota"
                + (f" (cgroup cpu.max = {quota:.1f} CPUs)" if quota else " or shared cores") + " caps it")
    return {"value": round(best / 1e6, 2), "unit": "MB/s", "cores": best_nt, "threads": best_nt, "kind": kind,
            "sample": f"{best_sample} B ({best_sample // M} slabs of {M} B) of the same workload, level -{level}, "
                      f"{best_nt} pthreads over oracle/cpu_mt.h",
            "MBps_by_threads": table, "usable_cpus": aff, "cgroup_cpu_quota": quota, "note": note}


REAL_ROOTS = ["/opt/rocm/include", "/usr/include", "/usr/lib/python3", "/usr/lib/python3.10",
              "/usr/local/lib/python3.10/dist-packages"]
REAL_SUFFIXES = (".py", ".pyi", ".h", ".hpp", ".hh", ".c", ".cc", ".cpp", ".cu", ".cuh", ".hip", ".cl", ".inc", ".txt", ".md",
                 ".rst", ".json", ".yaml", ".yml", ".cmake", ".cfg", ".toml", ".html", ".js", ".css", ".xml")


def real_tar(target, roots=REAL_ROOTS, suffixes=REAL_SUFFIXES):
    """REAL data that every box of this image holds: a tar (GNU format, sorted paths, zeroed owner and time: the same bytes
    on every box) of the headers and sources under `roots`, members added until the archive reaches `target` bytes -- no
    member twice, nothing repeated to pad.  Returns (bytearray, number of members, md5) or None if the files are missing."""
    import io
    import tarfile
    paths = []
    for r in roots:
        for d, dirs, files in os.walk(r):
            dirs.sort()
            for f in sorted(files):
                if f.endswith(suffixes):
                    paths.append(os.path.join(d, f))
    paths.sort()
    buf = io.BytesIO()
    count = 0
    with tarfile.open(fileobj=buf, mode="w", format=tarfile.GNU_FORMAT) as tf:
        for pth in paths:
            try:
                if os.path.islink(pth) or not os.path.isfile(pth):
                    continue
                with open(pth, "rb") as fh:
                    body = fh.read()
            except OSError:
                continue
            ti = tarfile.TarInfo(pth.lstrip("/"))
            ti.size = len(body)
            ti.mtime = 0
            ti.mode = 0o644
            ti.uid = ti.gid = 0
            ti.uname = ti.gname = ""
            tf.addfile(ti, io.BytesIO(body))
            count += 1
            if buf.tell() >= target:
                break
    if buf.tell() < target // 2:
        return None
    data = bytearray(buf.getbuffer()[:min(buf.tell(), target)])
    return data, count, hashlib.md5(data).hexdigest()


def real_leg(lib, torch, name, target, level, local, roots=REAL_ROOTS, suffixes=REAL_SUFFIXES, steps=3):
    """BASELINE.json configs[4]'s kind of input (a source tree as a tar) made of REAL files instead of a generator: compressed
    device-resident like the other legs; the stream is compared with the one the compiled reference (oracle/_ref, the
    checker -- untimed) writes for the same bytes, or, where that library is missing, taken through Python's bz2 and back."""
    made = real_tar(target, roots, suffixes)
    if made is None:
        return {"config": name, "skipped": "the file set is not on this box"}
    data, count, md5 = made
    what = f"tar of {count} real files of this image ({', '.join(roots)}; sorted, no repeats), {len(data)} B, md5 {md5}, -{level}"
    return data_leg(lib, torch, name, what, data, level, local, steps)


# the real corpora BASELINE.json names, where a box holds them: environment variable -> (config, levels)
CORPORA = [("LBZ_ENWIK8", "C1 enwik8 (real file)", (9,)), ("LBZ_SILESIA", "C3 Silesia corpus, concatenated (real file)", (1, 9)),
           ("LBZ_LINUX_TAR", "C5 Linux kernel tarball (real file)", (9,))]


def corpus_legs(lib, torch, local):
    """$LBZ_ENWIK8 / $LBZ_SILESIA / $LBZ_LINUX_TAR (and $LBZ_ENWIK9 through --kind wiki): the real corpora of BASELINE.json
    where a box has them.  No fixture can exist for a file this repository has never seen, so the stream is checked on the
    box against the compiled reference (oracle/_ref), as the real-file legs are."""
    out = []
    for var, name, levels in CORPORA:
        path = os.environ.get(var)
        if not path or not os.path.isfile(path):
            continue
        data = bytearray(open(path, "rb").read())
        for level in levels:
            what = f"{path}, {len(data)} B, md5 {hashlib.md5(data).hexdigest()}, -{level}"
            out.append(data_leg(lib, torch, name, what, data, level, local))
    return out


def check_against_reference(z, data, level):
    """Untimed checker: the stream against the compiled reference's for the same bytes (oracle/_ref; origin pointers of
    exactly periodic blocks canonical), or, where that library is missing, through Python's bz2 and back."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as L   # test infrastructure: the checker, after the timed region
    try:
        if L.have_ref():
            aff, _ = usable_cpus()
            want = L.ref_compress_mt(data, level, max(1, aff), canon=True)[0]
            return z == want, "byte-identical to the compiled reference's stream (oracle/_ref, origin pointers of periodic blocks canonical)"
    except Exception:                                   # noqa: BLE001 -- fall through to the independent decoder
        pass
    import bz2
    return bz2.decompress(z) == bytes(data), "round trip through Python's bz2"


def data_leg(lib, torch, name, what, data, level, local, steps=3):
    """One device-resident leg over bytes that have no fixture: timed like the others, checked by check_against_reference."""
    n = len(data)
    M = level * 100000
    nslabs = (n + M - 1) // M
    src = torch.frombuffer(data, dtype=torch.uint8).cuda()
    dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    with lib.context(level, nslabs, 0, local) as ctx:
        m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        st = ctx.stats()
        kms = {"k_collect": st.ms_collect, "k_bwt_part": st.ms_bwt_part, "k_bwt_batch": st.ms_bwt_batch, "k_bwt_deep+fix": st.ms_bwt_fix,
               "k_mtf": st.ms_mtf, "k_encode": st.ms_encode}
    z = dst[:m].cpu().numpy().tobytes()
    del src, dst
    torch.cuda.empty_cache()
    ok, how = check_against_reference(z, data, level)
    return {"config": name, "workload": what,
            "value": round(n / dt / 1e6, 1), "unit": "MB/s", "ms_per_step": round(dt * 1e3, 2), "steps": steps, "blocks": st.nblocks,
            "ratio": round(n / max(1, m), 4), "verified": bool(ok), "verified_against": how,
            "kernel_ms_sum_over_streams": {k: round(v, 2) for k, v in kms.items()}}


def run_leg(lib, torch, name, kind, n, seed, level, local, steps=3):
    """One more BASELINE configuration, device-resident, `steps` passes between synchronisations; stream checked
    against the fixture generated from the compiled reference."""
    data = gen_input(kind, n, seed)
    M = level * 100000
    nslabs = (n + M - 1) // M
    src = torch.frombuffer(data, dtype=torch.uint8).cuda()
    dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    with lib.context(level, nslabs, 0, local) as ctx:
        m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        st = ctx.stats()
    fx = find_fixture(kind, n, seed, level)
    ok = None
    if fx is not None:
        z = dst[:m].cpu().numpy().tobytes()
        ok = len(z) == fx["out_len"] and hashlib.md5(z).hexdigest() == fx["canon_md5"]
    del src, dst
    torch.cuda.empty_cache()
    return {"config": name, "workload": f"{kind}({n}, seed {seed}) -{level}", "value": round(n / dt / 1e6, 1), "unit": "MB/s",
            "ms_per_step": round(dt * 1e3, 2), "steps": steps, "blocks": st.nblocks, "ratio": round(n / max(1, m), 4), "verified": ok}


def decode_leg(lib, torch, kind, n, seed, level, local):
    """The inverse path on one more kind of data (device-resident; the stream is this library's, the round trip is checked)."""
    data = gen_input(kind, n, seed)
    M = level * 100000
    nslabs = (n + M - 1) // M
    src = torch.frombuffer(data, dtype=torch.uint8).cuda()
    z = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    with lib.context(level, nslabs, 0, local) as ctx:
        m = ctx.compress_device(src.data_ptr(), n, z.data_ptr(), z.numel())
    back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    with lib.decoder(2 * nslabs + 8) as dec:
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            td = time.perf_counter()
            k = dec.decompress_device(z.data_ptr(), m, back.data_ptr(), back.numel())
            torch.cuda.synchronize()
            dt = time.perf_counter() - td
            best = dt if best is None or dt < best else best
        ds = dec.stats()
    ok = bool(k == n and torch.equal(back[:n], src))
    del src, z, back
    torch.cuda.empty_cache()
    return {"workload": f"{kind}({n}, seed {seed}) -{level}", "value": round(n / best / 1e6, 1), "unit": "MB/s", "ms_total": round(best * 1e3, 2),
            "round_trip": ok, "blocks": ds.nblocks,
            "slowest_block_ms": {"codes": round(ds.ms_huff, 2), "sort": round(ds.ms_sort, 2), "walk": round(ds.ms_walk, 2)}}


def file_leg(data, level, fixture, devices=1):
    """the command (lbzip2_amd/host/lbzamd: lbzip2's options over the batch path, lbzamd_io.c's readers,
    pipelines and writers) file -> file on tmpfs, contexts and page-locked buffers included -- what a user of `lbzip2 FILE`
    waits for.  The stream is checked against the reference fixture; the program's own report line gives the rate behind the
    first context and how busy its reader and writer threads were."""
    import re
    import shutil
    import subprocess
    import tempfile
    exe = os.environ.get("LBZ_BENCH_LBZAMD") or os.path.join(ROOT, "lbzip2_amd", "host", "lbzamd")
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 3 * len(data) else tempfile.gettempdir()
    if not os.path.exists(exe):
        return {"skipped": "lbzip2_amd/host/lbzamd is not built"}
    d = tempfile.mkdtemp(prefix="lbzamd_bench_", dir=base)
    try:
        path = os.path.join(d, "input")
        with open(path, "wb") as f:
            f.write(data)
        best, line = None, ""
        for _ in range(2):
            if os.path.exists(path + ".bz2"):
                os.unlink(path + ".bz2")
            t0 = time.perf_counter()
            p = subprocess.run([exe, "-k", "-%d" % level, "--report"] + (["--devices=%d" % devices] if devices > 1 else []) + [path],
                               capture_output=True, timeout=600)
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                return {"error": p.stderr.decode(errors="replace")[-300:]}
            if best is None or dt < best:
                best, line = dt, p.stderr.decode(errors="replace").strip().splitlines()[-1]
        h = hashlib.md5()
        with open(path + ".bz2", "rb") as f:
            for piece in iter(lambda: f.read(1 << 24), b""):
                h.update(piece)
        m = re.search(r"= (\d+) MB/s \(contexts included: first one ready after ([0-9.]+) s.*?; (\d+) MB/s behind the first context\)", line)
        return {"value": round(len(data) / best / 1e6, 1), "unit": "MB/s", "seconds": round(best, 3), "where": base,
                "verified": (h.hexdigest() == fixture["canon_md5"]) if fixture else None,
                "first_context_s": float(m.group(2)) if m else None, "behind_first_context_MBps": int(m.group(3)) if m else None,
                "report": line,
                "devices": devices,
                "what": "`lbzamd -k -9 --report%s FILE` on tmpfs, process start to exit (HIP initialisation, contexts, page-locked ring included); "
                        "best of two" % (" --devices=%d" % devices if devices > 1 else "")}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def find_fixture(kind, n, seed, level):
    try:
        for r in json.load(open(os.path.join(ROOT, "tests", "golden", "bench_fixtures.json"))):
            if (r["kind"], r["n"], r["seed"], r["level"]) == (kind, n, seed, level):
                return r
    except (OSError, ValueError):
        pass
    return None


def relaunch(gpus):
    """`python bench.py --gpus N` as the driver types it for N = 1, with N > 1 and no launcher around it: start the N ranks
    ourselves (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1) with the same arguments.  The line rank 0
    prints is this process's output."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # RCCL between processes needs dmabuf IPC on this pool
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def strong_leg(args, torch, dist, shard, ctx, src_full, nbytes, dst, rank, world, dev, sync, fixture):
    """ONE input -> ONE stream over all ranks (north_star; process.c:515-548: N workers behind one splitter and one muxer).
    Rank 0 holds the input in HBM; the slab ranges of shard_plan go to their ranks (send/recv between device buffers: RCCL over
    xGMI), every rank compresses its range body-only, the bodies and the 12-byte CRC partials come back to rank 0 (StreamMux).
    Timed twice, K steps between barriers, max over ranks: with the ranges already resident per GPU (`value`), and with the
    scatter from rank 0 inside the step (`value_from_rank0`).  The stream is hashed against the reference fixture."""
    plan = shard.shard_plan(nbytes, world, args.level)
    off, ln = plan[rank]
    part = src_full[off:off + ln] if rank == 0 else torch.empty(max(1, ln), dtype=torch.uint8, device=dev)
    mux = shard.StreamMux(dist, args.level, lib_bound(nbytes), dev)

    def scatter():
        if rank == 0:
            ops = [dist.P2POp(dist.isend, src_full[plan[r][0]:plan[r][0] + plan[r][1]], r) for r in range(1, world) if plan[r][1]]
        else:
            ops = [dist.P2POp(dist.irecv, part[:ln], 0)] if ln else []
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def step(with_scatter):
        if with_scatter:
            scatter()
        m, nb, fold = ctx.compress_device_body(part.data_ptr(), ln, dst.data_ptr(), dst.numel())
        return mux.gather(dst, m, nb, fold)

    def barrier():
        sync()
        dist.barrier()
        sync()

    out = {}
    scatter()                                       # the ranges are resident from here on
    for name, ws in (("resident", False), ("from_rank0", True)):
        for _ in range(max(1, args.warmup)):
            total = step(ws)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            total = step(ws)
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out[name] = float(t.item()) / args.steps
    res = None
    if rank == 0:
        z = mux.out[:total].cpu().numpy().tobytes()
        ok = None
        if fixture is not None and not args.no_verify:
            ok = len(z) == fixture["out_len"] and hashlib.md5(z).hexdigest() == fixture["canon_md5"]
        elif not args.no_verify and nbytes <= 300_000_000:
            import bz2
            ok = bz2.decompress(z) == bytes(src_full.cpu().numpy().tobytes())
        res = {"value": round(nbytes / out["resident"] / 1e6, 1), "unit": "MB/s", "ms_per_step": round(out["resident"] * 1e3, 2),
               "value_from_rank0": round(nbytes / out["from_rank0"] / 1e6, 1), "ms_per_step_from_rank0": round(out["from_rank0"] * 1e3, 2),
               "steps": args.steps, "scaling": "strong", "out_bytes": total, "verified": ok,
               "slabs_per_rank": [(l + args.level * 100000 - 1) // (args.level * 100000) for _, l in plan],
               "what": "ONE %d-byte input -> ONE .bz2 stream on rank 0 over %d ranks: slab ranges by send/recv between device buffers, "
                       "bodies + 12-byte CRC partials gathered by StreamMux (lbzip2_amd/shard.py); `value`: ranges resident per GPU when "
                       "the step starts, `value_from_rank0`: the scatter from rank 0 inside the step" % (nbytes, world)}
    del mux
    return res


lib_bound = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bytes", type=int, default=1_000_000_000, help="input bytes per GPU (weak) or in all (strong)")
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--kind", default="wiki", choices=KINDS)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--slabs", type=int, default=0, help="resident slabs per chunk (0 = all)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed md5 check against the reference fixture")
    ap.add_argument("--no-host", action="store_true", help="skip the host-to-host leg (value_host)")
    ap.add_argument("--no-isolated", action="store_true", help="skip the extra single-stream pass (profiling runs)")
    ap.add_argument("--no-decode", action="store_true", help="skip the inverse-path leg (decode)")
    ap.add_argument("--no-seq", action="store_true", help="skip the -u / --sequential leg")
    ap.add_argument("--no-legs", action="store_true", help="skip the other BASELINE configurations (configs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch(args.gpus)                          # does not return

    import datetime

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # TEST HOOK (tests/test_bench_cpu.py, no GPU in the build container): LBZ_BENCH_EMU=1 runs this file's control flow -- the
    # launcher, the ranks, the collectives (gloo), the JSON line -- against the emulated kernel build of tests/emu on host
    # memory.  Never a measurement: the line says "emulated": true and the driver's runs do not set it.
    emu = bool(os.environ.get("LBZ_BENCH_EMU"))
    dev = "cpu" if emu else "cuda"
    if not emu:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path exists)"
        if os.environ.get("LBZ_BENCH_ONE_DEVICE"):      # tests: several ranks on one device (exercises the collective code path)
            local = 0
        torch.cuda.set_device(local)

    def sync():
        if not emu:
            torch.cuda.synchronize()

    def free_cache():
        if not emu:
            torch.cuda.empty_cache()

    if world > 1:
        if emu:
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=30))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(minutes=30))

    import lbzip2_amd
    from lbzip2_amd import shard
    if emu:
        lib = lbzip2_amd.Library(os.path.join(ROOT, "tests", "emu", "_build", "liblbzamd_emu_1024.so"))
    else:
        lib = lbzip2_amd.library()
    global lib_bound
    lib_bound = lib.bound

    M = args.level * 100000
    strong = args.scaling == "strong"
    seed = args.seed if strong else args.seed + rank
    path = os.environ.get("LBZ_ENWIK9")
    if path and os.path.exists(path) and args.kind == "wiki":
        full = bytearray(open(path, "rb").read()[:args.bytes])
        source, fixture = "enwik9 file", None
    else:
        full = gen_input(args.kind, args.bytes, seed)
        source = f"synthetic {args.kind}({args.bytes}, seed {seed})"
        fixture = find_fixture(args.kind, args.bytes, seed, args.level)
    if strong:
        off, ln = shard.shard_plan(len(full), world, args.level)[rank]
        data = full[off:off + ln]
    else:
        data = full
    n = len(data)
    nslabs = max(1, (n + M - 1) // M)
    slabs = args.slabs or nslabs

    src = torch.frombuffer(data, dtype=torch.uint8).to(dev) if n else torch.empty(0, dtype=torch.uint8, device=dev)
    dst = torch.empty(lib.bound(n), dtype=torch.uint8, device=dev)
    ctx = lib.context(args.level, slabs, 0, local)
    mux = shard.StreamMux(dist if world > 1 else None, args.level, lib.bound(len(full)), dev) if strong else None

    def step():
        if strong:
            m, nb, fold = ctx.compress_device_body(src.data_ptr(), n, dst.data_ptr(), dst.numel())
            return mux.gather(dst, m, nb, fold)        # rank 0: length of the single stream (in mux.out)
        return ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    out_len = 0
    for _ in range(args.warmup):
        out_len = step()
    barrier()
    t0 = time.perf_counter()
    kms = {k: 0.0 for k in KERNELS + ["finish"]}
    tot_ms = 0.0
    st = None
    for _ in range(args.steps):
        out_len = step()
        st = ctx.stats()
        for k, v in (("k_collect", st.ms_collect), ("k_bwt_part", st.ms_bwt_part), ("k_bwt_batch", st.ms_bwt_batch),
                     ("k_bwt_fix", st.ms_bwt_fix), ("k_mtf", st.ms_mtf), ("k_encode", st.ms_encode),
                     ("finish", st.ms_finish)):
            kms[k] += v
        tot_ms += st.ms_total
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        o = torch.tensor([0 if strong and rank else out_len, n], dtype=torch.int64, device=dev)
        dist.all_reduce(o)
        total_out, total_in = int(o[0].item()), int(o[1].item())
    else:
        total_out, total_in = out_len, n

    # ---- untimed: is the stream the reference's? ----
    verified, verified_against = None, None
    if not args.no_verify and (rank == 0 or not strong):
        stream = (mux.out if strong else dst)[:out_len].cpu().numpy().tobytes() if (rank == 0 or not strong) else b""
        ok = None
        if fixture is not None:
            ok = len(stream) == fixture["out_len"] and hashlib.md5(stream).hexdigest() == fixture["canon_md5"]
            verified_against = (f"tests/golden/bench_fixtures.json {args.kind}({args.bytes}, seed {seed}) -{args.level}: "
                                f"reference stream md5 {fixture['canon_md5']}, {fixture['out_len']} B")
        elif source == "enwik9 file":
            ok, verified_against = check_against_reference(stream, full, args.level)
        elif len(full) <= 300_000_000:
            import bz2
            ok = bz2.decompress(stream) == bytes(full)
            verified_against = "round trip through Python's bz2 (no reference fixture for this workload)"
        verified = ok
    if world > 1 and not strong:
        v = torch.tensor([1 if verified else 0, 1 if verified is None else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(v)
        verified = None if int(v[1].item()) == world else int(v[0].item()) + int(v[1].item()) == world

    # ---- N > 1: the extras below (host buffers on every rank at once, the strong leg's send/recv, the node's file leg) hold
    # collectives that have never run on two devices.  None of them may cost the run its line: from here on a timer stands by
    # -- if an extra hangs, rank 0 prints the WEAK result (all of it is measured by now) and every rank leaves with status 0 --
    # and an extra that raises is recorded and the remaining collectives are skipped on that rank (the others meet the timer).
    watchdog = None
    extras_broken = []
    if world > 1:
        import threading

        def minimal_line():
            return {"metric": "compress MB/s (whole node) + ratio, enwik9 -9, at 1/2/4/8 MI355X",
                    "value": round((len(full) if strong else total_in) * args.steps / elapsed / 1e6, 1), "unit": "MB/s",
                    "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
                    "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8",
                    "data": "synthetic" if "synthetic" in source else "enwik9",
                    "config": {"workload": f"{source}, level -{args.level}, {nslabs} slabs of {M} B per GPU", "bytes_per_gpu": n, "level": args.level,
                               "parallelism": f"{world} independent shard(s)"},
                    "ratio": round((len(full) if strong else total_in) / max(1, total_out), 4), "out_bytes": total_out, "verified": verified,
                    "rccl_ranks": world, "roofline": None, "cpu_baseline": None,
                    "extras": "an untimed extra leg did not come back within LBZ_BENCH_EXTRAS_TIMEOUT: this is the weak-scaling result alone "
                              "(roofline and cpu_baseline are in the N = 1 line)", "extras_broken": extras_broken}

        def give_up():
            if rank == 0:
                print(json.dumps(minimal_line()), flush=True)
            os._exit(0)

        watchdog = threading.Timer(float(os.environ.get("LBZ_BENCH_EXTRAS_TIMEOUT", "420")) + (0 if rank == 0 else 15), give_up)
        watchdog.daemon = True
        watchdog.start()

    # ---- untimed extras on rank 0 ----
    iso = None
    value_host = None
    if rank == 0 and not args.no_isolated:
        os.environ["LBZAMD_STREAMS"] = "1"
        try:
            with lib.context(args.level, slabs, 0, local) as c1:
                for _ in range(2):
                    c1.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
                s1 = c1.stats()
                iso = {"slots": c1.nslots, "ms_total": s1.ms_total,
                       "ms": {"k_collect": s1.ms_collect, "k_bwt_part": s1.ms_bwt_part, "k_bwt_batch": s1.ms_bwt_batch,
                              "k_bwt_fix": s1.ms_bwt_fix, "k_mtf": s1.ms_mtf, "k_encode": s1.ms_encode}}
        finally:
            del os.environ["LBZAMD_STREAMS"]
    if not args.no_host and not strong:
      try:
        # every rank at once, each its own host buffers: the node's host -> host rate is the sum
        hin = torch.frombuffer(data, dtype=torch.uint8)
        hout = torch.empty(lib.bound(n), dtype=torch.uint8)
        if not emu:
            hin, hout = hin.pin_memory(), hout.pin_memory()
        for _ in range(max(1, args.warmup)):
            m = ctx.compress_host_ptr(hin.data_ptr(), n, hout.data_ptr(), hout.numel())
        barrier()
        th = time.perf_counter()
        for _ in range(args.steps):                          # the protocol of `value`: K steps between barriers, wall clock
            m = ctx.compress_host_ptr(hin.data_ptr(), n, hout.data_ptr(), hout.numel())
        barrier()
        dth = max_over_ranks(time.perf_counter() - th) / args.steps
        ok = m == out_len and (fixture is None or args.no_verify
                               or hashlib.md5(hout[:m].numpy().tobytes()).hexdigest() == fixture["canon_md5"])
        if world > 1:
            v = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
            dist.all_reduce(v)
            ok = int(v.item()) == world
        value_host = {"value": round(total_in / dth / 1e6, 1), "unit": "MB/s", "ms_per_step": round(dth * 1e3, 2), "steps": args.steps,
                      "same_stream": bool(ok), "ranks": world,
                      "what": "SURVEY 8(d)'s end-to-end form of the metric: pinned host buffer in -> complete .bz2 stream in pinned host "
                              "memory; the input crosses PCIe round by round on one copy stream (issued up front), the stream leaves through the "
                              "page-locked output buffer as k_gather writes it; timed like `value` (K steps, wall clock"
                              + ("; all %d ranks at once, each its own buffers: the sum)" % world if world > 1 else ")")}
        del hin, hout
      except Exception as e:                                     # noqa: BLE001 -- an extra: never the line's failure
        if world == 1:
            raise
        extras_broken.append("value_host: " + repr(e)[:200])

    strong_res = None
    if world > 1 and not strong and not os.environ.get("LBZ_NO_STRONG") and not extras_broken:
        # the same run, the other scaling: rank 0's input over all ranks into ONE stream
        sfix = find_fixture(args.kind, args.bytes, args.seed, args.level) if "synthetic" in source else None
        try:
            strong_res = strong_leg(args, torch, dist, shard, ctx, src if rank == 0 else None, len(full), dst, rank, world,
                                    dev, sync, sfix)
        except Exception as e:                                   # noqa: BLE001
            extras_broken.append("strong: " + repr(e)[:200])
            strong_res = {"error": repr(e)[:200]} if rank == 0 else None

    decode = None
    if rank == 0 and not args.no_decode and world == 1 and not strong:     # (strong: dst holds the body only)
        # the inverse path on the stream just written: every block decoded at once, compared with the input
        back = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        with lib.decoder(max(8, min(4096, 2 * nslabs + 8))) as dec:
            best = None
            for _ in range(3):
                sync()
                td = time.perf_counter()
                k = dec.decompress_device(dst.data_ptr(), out_len, back.data_ptr(), back.numel())
                sync()
                dt = time.perf_counter() - td
                best = dt if best is None or dt < best else best
            ds = dec.stats()
        decode = {"value": round(n / best / 1e6, 1), "unit": "MB/s", "ms_total": round(best * 1e3, 2),
                  "round_trip": bool(k == n and torch.equal(back[:n], src)), "blocks": ds.nblocks,
                  "kernel_ms": {"k_dscan": round(ds.ms_scan, 2), "k_dblock": round(ds.ms_blocks, 2), "k_demit": round(ds.ms_emit, 2)},
                  "slowest_block_ms": {"codes": round(ds.ms_huff, 2), "sort": round(ds.ms_sort, 2), "walk": round(ds.ms_walk, 2)},
                  "what": "decoded bytes per second, .bz2 stream and output both resident in HBM: magic scan, then one workgroup "
                          "per block (prefix codes on one wave, inverse MTF by chunks on the others, counting sort, list ranking walk, "
                          "CRC), then inverse RLE1 into place; `others`: the same on high-entropy inputs (few blocks: 1024-thread workgroups)"}
        del back
        free_cache()
        if not args.no_legs:
            decode["others"] = [decode_leg(lib, torch, "rand", 100_000_000, 2, 9, local), decode_leg(lib, torch, "mixed", 210_000_000, 2, 9, local)]

    sequential = None
    if rank == 0 and not args.no_seq and world == 1 and not strong:
        # the reference's -u blocking of the same input, checked against its own fixture
        with lib.context(args.level, slabs, 0, local) as cs:
            cs.set_sequential(True)
            best = None
            for _ in range(2):
                sync()
                tq = time.perf_counter()
                mq = cs.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
                sync()
                dt = time.perf_counter() - tq
                best = dt if best is None or dt < best else best
            sq = cs.stats()
        okq = None
        try:
            recs = json.load(open(os.path.join(ROOT, "tests", "golden", "seq_fixtures.json")))["records"]
            rec = [r for r in recs if (r["kind"], r["n"], r["seed"], r["level"]) == (args.kind, args.bytes, seed, args.level)]
            if rec and not args.no_verify:
                zq = dst[:mq].cpu().numpy().tobytes()
                okq = mq == rec[0]["out_len"] and hashlib.md5(zq).hexdigest() == rec[0]["ref_md5"]
        except (OSError, ValueError, KeyError):
            okq = None
        sequential = {"value": round(n / best / 1e6, 1), "unit": "MB/s", "ms_total": round(best * 1e3, 2), "out_bytes": mq,
                      "blocks": sq.nblocks, "verified": okq, "ms_block_chain": round(sq.ms_collect, 2),
                      "what": "lbzip2 -u blocking (blocks cut where they are full): the blocks' first pass is a chain on the device"}

    nslots = ctx.nslots
    nstreams = int(os.environ.get("LBZAMD_STREAMS", "3"))
    per_round = min(nslots, nslabs)
    segs, parts = ctx.round_shape(per_round, nstreams > 1 and nslabs > per_round)     # lbz_api.hip: launch_sort's choice, asked of the library
    value_file = None
    if rank == 0 and not args.no_host and world == 1 and not strong and not os.environ.get("LBZ_NO_FILE_LEG"):
        try:
            value_file = file_leg(data, args.level, fixture)
        except Exception as e:                                   # noqa: BLE001 -- a leg of its own: never the bench line's failure
            value_file = {"error": repr(e)[:200]}
    value_node_file = None
    if world > 1 and not args.no_host and not strong and not os.environ.get("LBZ_NO_FILE_LEG") and not extras_broken:
        # the command over all N devices, file -> file: the ranks give their device memory back first (the command's own
        # contexts take up to half of what is free), rank 0 runs it, the others wait
        try:
            ctx.close()
            ctx = None
            del src, dst
            free_cache()
            barrier()
            if rank == 0:
                try:
                    value_node_file = file_leg(data, args.level, fixture, devices=world)
                except Exception as e:                           # noqa: BLE001
                    value_node_file = {"error": repr(e)[:200]}
            barrier()
        except Exception as e:                                   # noqa: BLE001
            extras_broken.append("value_node_file: " + repr(e)[:200])

    legs = None
    if rank == 0 and not args.no_legs and world == 1 and not strong and args.kind == "wiki" and args.bytes == 1_000_000_000:
        del src, dst
        free_cache()
        legs = [run_leg(lib, torch, *leg, local) for leg in LEGS]
        if not os.environ.get("LBZ_NO_REAL"):
            legs.append(real_leg(lib, torch, "C5 real files: tar of headers and sources", 1_000_000_000, args.level, local))
            legs.append(real_leg(lib, torch, "real files: Python sources only", 300_000_000, args.level, local,
                                 ["/usr/lib/python3", "/usr/lib/python3.10", "/usr/local/lib/python3.10/dist-packages"], (".py",)))
        legs += corpus_legs(lib, torch, local)

    if rank == 0:
        nchunks = (nslabs + slabs - 1) // slabs
        rounds = sum(-(-min(slabs, nslabs - i * slabs) // nslots) for i in range(nchunks))
        alg = {"k_collect": st.n_in + st.n_rle, "k_bwt_part": 5.0 * st.n_rle, "k_bwt_batch": 6.0 * st.n_rle, "k_bwt_fix": 6.0 * st.n_rle,
               "k_mtf": st.n_rle + 2.0 * st.n_mtf, "k_encode": 18.0 * st.n_mtf + st.n_out}     # bytes per step
        # k_bwt_batch and k_bwt_fix share the 6 N_rle of "read rows, gather the preceding byte, write BWT": priced on their sum
        kms_sort = dict(kms)
        kms_sort["k_bwt_batch"] = kms["k_bwt_batch"] + kms["k_bwt_fix"]
        names = ["k_collect", "k_bwt_part", "k_bwt_batch", "k_mtf", "k_encode"]
        per_kernel = {k: {"ms_per_step": round(kms_sort[k] / args.steps, 3), "launches_per_step": rounds,
                          "alg_GB_per_step": round(alg[k] / 1e9, 3),
                          "achieved_GBps": round(alg[k] * args.steps / (kms_sort[k] * 1e-3) / 1e9, 2) if kms_sort[k] > 0 else 0.0}
                      for k in names}
        per_kernel["k_bwt_batch"]["includes"] = "k_bwt_deep (text rounds) + k_bwt_fix* (rank rounds): %.3f ms per step" % (kms["k_bwt_fix"] / args.steps)
        iso_tab = None
        if iso:
            im = dict(iso["ms"])
            im["k_bwt_batch"] += im.pop("k_bwt_fix")
            iso_rounds = sum(-(-min(slabs, nslabs - i * slabs) // iso["slots"]) for i in range(nchunks))
            iso_tab = {k: {"ms_per_step": round(im[k], 3), "launches_per_step": iso_rounds,
                           "achieved_GBps": round(alg[k] / (im[k] * 1e-3) / 1e9, 2) if im[k] > 0 else 0.0,
                           "frac": round(alg[k] / (im[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if im[k] > 0 else 0.0}
                       for k in names}
        dom = max(names, key=lambda k: kms_sort[k])
        launches = rounds * args.steps
        # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, separate runs of this command; profiles/pmc_traffic.json holds KB per slab).
        # FETCH_SIZE is doubled (gfx950 tallies wide reads at half, MI355X_MICROARCH.md, HBM).
        traffic = None
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pt.get("workload") == f"{args.kind} -{args.level}":
                tk = pt["kernels"]
                cand = {"k_bwt_part": ["k_bwt_part2", "k_bwt_part"], "k_bwt_batch": ["k_bwt_batch", "k_bwt_long", "k_bwt_deep", "k_bwt_deepr", "k_bwt_fix", "k_bwt_fix0", "k_bwt_fixr", "k_bwt_fixend"]}.get(dom, [dom])
                kb = sum(2.0 * tk[c]["fetch_kb_per_slab"] + tk[c]["write_kb_per_slab"] for c in cand if c in tk)
                if kb > 0:
                    traffic = round(kb * 1024.0 * nslabs * args.steps / launches)
        except (OSError, ValueError, KeyError):
            traffic = None
        # The headline fraction is the ISOLATED one: one stream, nothing overlaps, per-kernel times add up to the pass and
        # reproduce from profiles/*_s1_kernel_stats.csv.  The live sums over three overlapping streams (a launch's events
        # also bracket the time it shares the device with the other streams' kernels) are kept beside it, labelled.
        if iso_tab:
            dom = max(names, key=lambda k: iso_tab[k]["ms_per_step"])
            achieved = iso_tab[dom]["achieved_GBps"]
            dom_ms, dom_launches = iso_tab[dom]["ms_per_step"], iso_tab[dom]["launches_per_step"]
        else:
            achieved = per_kernel[dom]["achieved_GBps"]
            dom_ms, dom_launches = kms_sort[dom] / args.steps, rounds
        if traffic is not None:
            traffic = round(traffic * launches / max(1, dom_launches * args.steps))      # per launch of the table the fraction comes from
        pipe_alg = st.n_in + 13.0 * st.n_rle + 20.0 * st.n_mtf + st.n_out  # SURVEY 8(d), per step
        res = {
            "metric": "compress MB/s (whole node) + ratio, enwik9 -9, at 1/2/4/8 MI355X",
            "value": round((len(full) if strong else total_in) * args.steps / elapsed / 1e6, 1),
            "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "u8",
            "data": "synthetic" if "synthetic" in source else "enwik9",
            "config": {"workload": f"{source}, level -{args.level}, {nslabs} slabs of {M} B per GPU, "
                                   f"{slabs} resident per chunk, rounds of <= {nslots} slabs on {nstreams} streams; one workgroup per block (collect, "
                                   + ("partition, " if parts == 1 else "") + f"MTF, coding), "
                                   + (f"{parts} per block in the partition's launch-per-pass kernels, " if parts != 1 else "")
                                   + f"{segs} segment workgroups per block in the sorting kernels"
                                   + ("; ONE stream gathered on rank 0 (RCCL send/recv of block bytes + 12-byte CRC partials)" if strong else ""),
                       "bytes_per_gpu": n, "level": args.level,
                       "parallelism": f"{world} slab range(s) of one input -> one stream" if strong else f"{world} independent shard(s)"},
            "ratio": round((len(full) if strong else total_in) / max(1, total_out), 4), "out_bytes": total_out,
            "verified": verified, "verified_against": verified_against,
            "value_is": "device-resident rate (input in HBM when the timed region starts, stream left in HBM), as the bench contract "
                        "prescribes; `value_host.value` is the same job host buffer -> host buffer (PCIe included), the form BASELINE.md 3 words",
            "roofline": {"bound": "hbm", "kernel": dom + (" (+k_bwt_deep, k_bwt_fix*)" if dom == "k_bwt_batch" else ""),
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "measured": ("one single-stream pass (LBZAMD_STREAMS=1): HIP events around every launch, nothing overlaps"
                                      if iso_tab else "live sums over the overlapping streams"),
                         "alg_bytes_per_launch": round(alg[dom] / max(1, dom_launches)), "launches_per_step": dom_launches,
                         "traffic_over_alg": (round(traffic / max(1.0, alg[dom] / max(1, dom_launches)), 2) if traffic else None),
                         "avg_launch_ms": round(dom_ms / max(1, dom_launches), 3),
                         "overlapped": {"note": "HIP-event sums of the timed region's %s streams: a launch's time includes what it shares "
                                                "with the other streams' kernels, so a kernel's sum can exceed ms_per_step"
                                                % os.environ.get("LBZAMD_STREAMS", "3"), "per_kernel": per_kernel},
                         "pipeline_alg_bytes_per_step": round(pipe_alg),
                         "pipeline_achieved_GBps": round(pipe_alg * args.steps / (tot_ms * 1e-3) / 1e9, 2),
                         "pipeline_frac": round(pipe_alg * args.steps / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "isolated": ({"note": "one untimed single-stream pass, no overlap: the table `frac` is taken from", "slots": iso["slots"],
                                       "ms_total": round(iso["ms_total"], 2), "per_kernel": iso_tab} if iso else None)},
            "kernel_ms_per_step": {k: round(v / args.steps, 2) for k, v in kms.items()},
            "sorter": {"blocks": st.nblocks, "periodic_blocks": st.nperiodic},
        }
        if world > 1:
            res["rccl_ranks"] = dist.get_world_size()
            res["backend"] = dist.get_backend() + (" (RCCL: one process per GPU, torch.distributed)" if not emu else " (emulated kernels, host memory)")
        if emu:
            res["emulated"] = True
        if extras_broken:
            res["extras_broken"] = extras_broken
        if strong_res:
            res["strong"] = strong_res
        if value_host:
            res["value_host"] = value_host
        if value_file:
            res["value_file"] = value_file
        if value_node_file:
            res["value_node_file"] = value_node_file
        if decode:
            res["decode"] = decode
        if sequential:
            res["sequential"] = sequential
        if legs:
            res["configs"] = legs
        if not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(full, args.level)
        if watchdog:
            watchdog.cancel()
        print(json.dumps(res), flush=True)
    if watchdog:
        watchdog.cancel()
    if extras_broken:
        os._exit(0)                                              # (ranks that wait in a collective this one skipped leave by their timer)
    if ctx is not None:
        ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
