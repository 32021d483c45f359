#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: bzip2 -9 compress throughput.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (RLE1+CRC -> BWT -> MTF/ZRLE -> prefix codes -> bit
packing -> stream assembly) over the workload, input already resident in HBM, the complete
.bz2 stream left in HBM.  Workload (BASELINE.json configs[1]): enwik9-sized text, 10^9 bytes,
level -9, per GPU; the real enwik9 is used if $LBZ_ENWIK9 points at it, else the seeded
stand-in text(10^9, seed 2+rank) of SURVEY.md 8d.  With N GPUs every rank compresses its own
10^9-byte shard into its own complete stream (independent slabs, no data-path collective;
concatenated streams are a valid .bz2 file) -> weak scaling; value = all ranks' input bytes
over the max-over-ranks time.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      the kernel with the largest share of device time, timed live in the timed region
                (HIP events on the stream each launch goes to; rounds of blocks run on two streams
                and their launches overlap, so per-launch time includes sharing the device).
                roofline.isolated repeats the per-kernel table from one extra untimed pass on a
                single stream (nothing overlaps).  SURVEY 8(d) prices the path at
                N_in + 13 N_rle + 20 N_mtf + N_out algorithmic bytes; per kernel that is
                collect N_in+N_rle | bwt_part 5 N_rle | bwt_batch 6 N_rle | mtf N_rle+2 N_mtf |
                encode 18 N_mtf+N_out.  achieved = bytes per launch / mean launch time from HIP
                events recorded on the library's own stream; peak 8000 GB/s (MI355X HBM3E)
  cpu_baseline  reference lbzip2's block codec (oracle/_ref, "reference") or the bit-exact
                restatement (oracle/, "port") on the box's host cores over a bounded sample
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E


def gen_input(kind, n, seed):
    g = C.CDLL(os.path.join(ROOT, "lbzip2_amd", "host", "libgen_inputs.so"))
    buf = bytearray(n)
    cbuf = (C.c_uint8 * n).from_buffer(buf)
    fn = g.lbzgen_text if kind == "text" else g.lbzgen_rand
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
    fn(cbuf, n, seed)
    del cbuf
    return buf


def cpu_baseline(data, level, seconds_budget=1.0):
    """Time the CPU codec on all host cores over a bounded sample of the same workload
    (at most seconds_budget seconds of wall time at the single-thread rate, i.e. ~cores x that of CPU work)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as L   # test infrastructure: only used here as the timed CPU baseline
    kind = "reference" if L.have_ref() else "port"
    fn = L.ref_compress if kind == "reference" else L.orc_compress
    M = level * 100000
    cores = os.cpu_count() or 1
    # single thread: ~20 MB/s -> 3 slabs ~ 0.15 s; probe the rate first
    t0 = time.perf_counter()
    fn(bytes(data[:3 * M]), level)
    t1 = time.perf_counter() - t0
    rate1 = 3 * M / t1
    # every host core gets the same number of whole slabs; the sample is bounded by the budget of CPU seconds
    nslabs = max(1, len(data) // M)
    nthreads = min(cores, nslabs)
    per_thread = max(1, min(nslabs // nthreads, int(seconds_budget * cores * rate1 / M / nthreads)))
    pieces = [bytes(data[i * per_thread * M:(i + 1) * per_thread * M]) for i in range(nthreads)]
    with ThreadPoolExecutor(nthreads) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda p: fn(p, level), pieces))     # ctypes releases the GIL
        t = time.perf_counter() - t0
    total = sum(len(p) for p in pieces)
    return {"value": round(total / t / 1e6, 2), "unit": "MB/s", "cores": nthreads, "kind": kind,
            "sample": f"{nthreads} threads x {per_thread} slabs of {M} B ({total} B) of the same workload, level -{level}",
            "single_thread_MBps": round(rate1 / 1e6, 2), "host_cpus": cores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bytes", type=int, default=1_000_000_000, help="input bytes per GPU")
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--kind", default="text", choices=["text", "rand"])
    ap.add_argument("--slabs", type=int, default=0, help="resident slabs per chunk (0 = all)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--verify", action="store_true", help="decode the stream with Python's bz2 (untimed)")
    ap.add_argument("--no-isolated", action="store_true", help="skip the extra single-stream pass (profiling runs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path exists)"
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import lbzip2_amd
    lib = lbzip2_amd.library()

    n = args.bytes
    M = args.level * 100000
    path = os.environ.get("LBZ_ENWIK9")
    if path and os.path.exists(path) and args.kind == "text":
        data = bytearray(open(path, "rb").read()[:n])
        source = "enwik9 file"
    else:
        data = gen_input(args.kind, n, 2 + rank)
        source = f"synthetic {args.kind}({n}, seed {2 + rank})"
    n = len(data)
    nslabs = (n + M - 1) // M
    slabs = args.slabs or nslabs

    src = torch.frombuffer(data, dtype=torch.uint8).cuda()
    dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    ctx = lib.context(args.level, slabs, 0, local)

    def step():
        return ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    out_len = 0
    for _ in range(args.warmup):
        out_len = step()
    barrier()
    t0 = time.perf_counter()
    kms = {"k_collect": 0.0, "k_bwt_part": 0.0, "k_bwt_batch": 0.0, "k_bwt_fix": 0.0, "k_mtf": 0.0,
           "k_encode": 0.0, "finish": 0.0}
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
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        o = torch.tensor([out_len, n], dtype=torch.int64, device="cuda")
        dist.all_reduce(o)
        total_out, total_in = int(o[0].item()), int(o[1].item())
    else:
        total_out, total_in = out_len, n

    if args.verify:
        import bz2
        assert bz2.decompress(bytes(dst[:out_len].cpu().numpy())) == bytes(data)

    iso = None
    if rank == 0 and not args.no_isolated:
        # the same per-kernel figures with nothing overlapped: one untimed pass, one stream
        os.environ["LBZAMD_STREAMS"] = "1"
        try:
            with lib.context(args.level, slabs, 0, local) as c1:
                c1.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
                c1.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
                s1 = c1.stats()
                iso = {"slots": c1.nslots, "ms": {"k_collect": s1.ms_collect, "k_bwt_part": s1.ms_bwt_part, "k_bwt_batch": s1.ms_bwt_batch,
                                                  "k_mtf": s1.ms_mtf, "k_encode": s1.ms_encode}, "ms_total": s1.ms_total}
        finally:
            del os.environ["LBZAMD_STREAMS"]
    if rank == 0:
        nchunks = (nslabs + slabs - 1) // slabs
        nslots = ctx.nslots
        rounds = sum(-(-min(slabs, nslabs - i * slabs) // nslots) for i in range(nchunks))   # rounds (launches of every per-round kernel) per step
        alg = {"k_collect": st.n_in + st.n_rle, "k_bwt_part": 5.0 * st.n_rle, "k_bwt_batch": 6.0 * st.n_rle,
               "k_mtf": st.n_rle + 2.0 * st.n_mtf, "k_encode": 18.0 * st.n_mtf + st.n_out}     # bytes per step
        nlaunch = {"k_collect": nchunks, "k_bwt_part": rounds, "k_bwt_batch": rounds, "k_mtf": rounds, "k_encode": rounds}
        per_kernel = {k: {"ms_per_step": round(kms[k] / args.steps, 3), "launches_per_step": nlaunch[k],
                          "alg_GB_per_step": round(alg[k] / 1e9, 3),
                          "achieved_GBps": round(alg[k] * args.steps / (kms[k] * 1e-3) / 1e9, 2) if kms[k] > 0 else 0.0}
                      for k in alg}
        if iso:
            iso_rounds = sum(-(-min(slabs, nslabs - i * slabs) // iso["slots"]) for i in range(nchunks))
            iso_nl = {"k_collect": nchunks, "k_bwt_part": iso_rounds, "k_bwt_batch": iso_rounds, "k_mtf": iso_rounds, "k_encode": iso_rounds}
            iso_tab = {k: {"ms_per_step": round(iso["ms"][k], 3), "launches_per_step": iso_nl[k],
                           "achieved_GBps": round(alg[k] / (iso["ms"][k] * 1e-3) / 1e9, 2) if iso["ms"][k] > 0 else 0.0,
                           "frac": round(alg[k] / (iso["ms"][k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if iso["ms"][k] > 0 else 0.0}
                       for k in alg}
        dom = max(alg, key=lambda k: kms[k])
        launches = nlaunch[dom] * args.steps
        # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, separate runs; profiles/pmc_traffic.json holds KB per slab of the same workload).
        # FETCH_SIZE is doubled (gfx950 tallies wide reads at half, MI355X_MICROARCH.md, HBM).
        traffic = None
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pt.get("workload") == f"{args.kind} -{args.level}" and dom in pt["kernels"]:
                e = pt["kernels"][dom]
                slabs_per_launch = nslabs * args.steps / launches
                traffic = round((2.0 * e["fetch_kb_per_slab"] + e["write_kb_per_slab"]) * 1024.0 * slabs_per_launch)
        except (OSError, ValueError, KeyError):
            traffic = None
        achieved = per_kernel[dom]["achieved_GBps"]
        pipe_alg = st.n_in + 13.0 * st.n_rle + 20.0 * st.n_mtf + st.n_out  # SURVEY 8(d), per step
        res = {
            "metric": "compress MB/s (whole node) + ratio, enwik9 -9, at 1/2/4/8 MI355X", "value": round(total_in * args.steps / elapsed / 1e6, 1),
            "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic" if "synthetic" in source else "enwik9",
            "config": {"workload": f"{source}, level -{args.level}, {nslabs} slabs of {M} B per GPU, "
                                   f"{slabs} resident per chunk, one bzip2 block per workgroup, rounds of {nslots} slabs on two streams",
                       "bytes_per_gpu": n, "level": args.level, "parallelism": f"{world} independent shard(s)"},
            "ratio": round(total_in / total_out, 4), "out_bytes": total_out,
            "bit_exact": "vs reference lbzip2 (tests/test_gpu_parity.py); periodic blocks: origin pointer only",
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "alg_bytes_per_launch": round(alg[dom] * args.steps / launches), "launches": launches,
                         "avg_launch_ms": round(kms[dom] / launches, 3), "per_kernel": per_kernel,
                         "pipeline_alg_bytes_per_step": round(pipe_alg),
                         "pipeline_achieved_GBps": round(pipe_alg * args.steps / (tot_ms * 1e-3) / 1e9, 2),
                         "pipeline_frac": round(pipe_alg * args.steps / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "isolated": ({"note": "one untimed single-stream pass, no overlap", "slots": iso["slots"],
                                       "ms_total": round(iso["ms_total"], 2), "per_kernel": iso_tab} if iso else None)},
            "kernel_ms_per_step": {k: round(v / args.steps, 2) for k, v in kms.items()},
            "sorter": {"elements_per_block_byte": round(st.sort_elems / max(1, st.n_rle), 3), "blocks": st.nblocks,
                       "periodic_blocks": st.nperiodic},
        }
        if not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(data, args.level)
        print(json.dumps(res), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
