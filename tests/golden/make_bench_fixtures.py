#!/usr/bin/env python3
"""Generate reference fixtures for the BASELINE.json configurations and the benchmark inputs.

Run in the build container only (needs oracle/_ref/libref.so = the compiled reference):
    python tests/golden/make_bench_fixtures.py [quick]
Writes
    tests/golden/bench_fixtures.json   per workload (generator kind, bytes, seed, level): md5 of the
        input, length + md5 of the reference's .bz2 stream (ref_md5), md5 of the same stream with
        the origin pointer of exactly periodic blocks set to the smallest equal row (canon_md5 --
        equal to ref_md5 when periodic_blocks == 0), number of blocks, combined CRC, and md5s of
        the stream body cut at every 100th slab (to localise a mismatch).
    tests/golden/periodic_blocks.json  every exactly periodic block (T = u^k) of the reference's own
        compress corpora at -9 and -1: the reference's origin pointer, k, and the smallest equal
        row this repository emits -- the enumerated list of the one documented divergence.
Inputs come from lbzip2_amd/host/gen_inputs.c (seeded, integer-only), the streams from the
reference's encode.c/divbwt.c driven by oracle/cpu_mt.h on all host cores.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as L  # noqa: E402
from golden_util import suite_inputs  # noqa: E402

PIECE_SLABS = 100


def md5(b):
    return hashlib.md5(b).hexdigest()


def record(kind, n, seed, level, config):
    data = L.gen_kind(kind, n, seed)
    nt = os.cpu_count() or 1
    ref, blocks, sec = L.ref_compress_mt(data, level, nt)
    nper = sum(1 for b in blocks if b[3] > 1)
    canon = L.ref_compress_mt(data, level, nt, canon=True)[0] if nper else ref
    # body pieces: blocks are byte aligned; a piece = the blocks of PIECE_SLABS consecutive slabs
    pieces, o, cur, start = [], 4, 0, 4
    for olen, _crc, _idx, _k, slab in blocks:
        if slab // PIECE_SLABS != cur:
            pieces.append(md5(canon[start:o]))
            start, cur = o, slab // PIECE_SLABS
        o += olen
    pieces.append(md5(canon[start:o]))
    assert o + 10 == len(canon)
    rec = {"config": config, "kind": kind, "n": n, "seed": seed, "level": level, "in_md5": md5(data),
           "out_len": len(ref), "ref_md5": md5(ref), "canon_md5": md5(canon), "blocks": len(blocks),
           "periodic_blocks": nper, "combined_crc": int.from_bytes(ref[-4:], "big"),
           "piece_slabs": PIECE_SLABS, "piece_md5": pieces,
           "ref_MBps_all_cores_build_container": round(n / sec / 1e6, 1)}
    print(kind, n, seed, level, "->", len(ref), "ratio %.3f" % (n / len(ref)), "periodic", nper, "%.1f MB/s" % (n / sec / 1e6), flush=True)
    return rec


def record_c4_full(n=10_000_000_000, seed=4, level=9, piece_slabs=1000):
    """C4 as BASELINE.json words it: 10 GB of random bytes at -9.  Neither the input nor the reference's stream is held at once:
    slab-aligned pieces of the generator's sequence (lbzgen_rand_from) go through the compiled reference one by one; their
    block bytes are the stream's body in order (every block is byte aligned) and the block CRCs fold into the trailer."""
    import ctypes as C
    g = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(HERE)), "lbzip2_amd", "host", "libgen_inputs.so"))
    g.lbzgen_rand_from.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    g.lbzgen_rand_from.restype = None
    M = level * 100000
    state = C.c_uint32(seed)
    hin, hout = hashlib.md5(), hashlib.md5()
    hout.update(b"BZh" + bytes([48 + level]))
    out_len, nblocks, cc, pieces, sec, done = 4, 0, 0, [], 0.0, 0
    nt = os.cpu_count() or 1
    while done < n:
        ln = min(n - done, piece_slabs * M)
        buf = bytearray(ln)
        cb = (C.c_uint8 * ln).from_buffer(buf)
        g.lbzgen_rand_from(cb, ln, C.byref(state))
        del cb
        hin.update(buf)
        ref, blocks, s = L.ref_compress_mt(buf, level, nt)
        body = ref[4:len(ref) - 10]
        assert sum(b[0] for b in blocks) == len(body) and all(b[3] <= 1 for b in blocks)
        hout.update(body)
        pieces.append(md5(body))
        for b in blocks:
            cc = (((cc << 1) | (cc >> 31)) ^ b[1] ^ 0xFFFFFFFF) & 0xFFFFFFFF
        out_len += len(body)
        nblocks += len(blocks)
        sec += s
        done += ln
        print("  C4 full:", done, "of", n, "bytes,", nblocks, "blocks", flush=True)
    trailer = bytes.fromhex("177245385090") + cc.to_bytes(4, "big")
    hout.update(trailer)
    out_len += 10
    m = hout.hexdigest()
    return {"config": "C4 random, 10 GB as specified (streamed: rand continues over the pieces)", "kind": "rand", "n": n, "seed": seed,
            "level": level, "in_md5": hin.hexdigest(), "out_len": out_len, "ref_md5": m, "canon_md5": m, "blocks": nblocks,
            "periodic_blocks": 0, "combined_crc": cc, "piece_slabs": piece_slabs, "piece_md5": pieces,
            "ref_MBps_all_cores_build_container": round(n / sec / 1e6, 1)}


def periodic_corpus_blocks():
    out = []
    for name, raw in sorted(suite_inputs().items()):
        if not raw:
            continue
        for lvl in (9, 1):
            _, blocks, _ = L.ref_compress_mt(raw, lvl, 1)
            for i, (olen, crc, idx, k, _slab) in enumerate(blocks):
                if k > 1:
                    out.append({"input": name, "level": lvl, "block": i, "copies": k, "ref_bwt_idx": idx,
                                "canon_bwt_idx": idx - idx % k})
    return out


def main():
    assert L.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    path = os.path.join(HERE, "bench_fixtures.json")
    if len(sys.argv) > 1 and sys.argv[1] == "c4full":        # only (re)make the 10 GB record: ~3 minutes of the reference on 8 cores
        rec = record_c4_full()
        recs = [r for r in json.load(open(path)) if not (r["kind"] == "rand" and r["n"] == rec["n"])] + [rec]
        json.dump(recs, open(path, "w"), indent=1)
        print("C4 full:", rec["out_len"], rec["ref_md5"], rec["blocks"], "blocks")
        return
    G = 1_000_000_000
    work = []
    # C2 (headline): enwik9-sized, level -9; one seed per rank of the weak-scaling bench (2 + rank)
    for seed in range(2, 10):
        work.append(("wiki", G, seed, 9, "C2 enwik9 stand-in (enwik-like)"))
    work.append(("text", G, 2, 9, "C2 enwik9 stand-in of round 1 (28-symbol word soup)"))
    # C1: enwik8-sized at -9 (the reference's own CPU-runnable case)
    work.append(("wiki", 100_000_000, 1, 9, "C1 enwik8 stand-in"))
    # C3: Silesia-sized mixed entropy at -1 and -9
    work.append(("mixed", 211_938_580, 3, 1, "C3 Silesia stand-in"))
    work.append(("mixed", 211_938_580, 3, 9, "C3 Silesia stand-in"))
    # C4: random bytes at -9, one GPU's share of 10 GB on 8 GPUs is 1.25 GB; 112 slabs pinned here
    work.append(("rand", 100_000_000, 4, 9, "C4 random (112 slabs)"))
    work.append(("rand", 1_250_000_000, 4, 9, "C4 random, one GPU's eighth of 10 GB"))
    # C5: kernel-tarball-like at -9: the whole 1.4 GB and one GPU's eighth
    work.append(("tar", 175_000_000, 5, 9, "C5 tarball stand-in, one GPU's eighth"))
    work.append(("tar", 1_400_000_000, 5, 9, "C5 tarball stand-in"))
    # strong-scaling slices of the headline input are prefixes of it at slab boundaries: the piece
    # md5s above cover them.  Small cases for the CPU-side tests (emulator, gloo):
    work.append(("wiki", 2_000_000, 2, 9, "small"))
    work.append(("wiki", 350_000, 2, 1, "small"))
    work.append(("tar", 1_000_000, 5, 9, "small"))
    work.append(("mixed", 3_000_000, 3, 1, "small"))
    if quick:
        work = [w for w in work if w[1] <= 3_000_000]
    recs = [record(*w) for w in work]
    if not quick:
        recs.append(record_c4_full())
    if quick and os.path.exists(path):
        old = [r for r in json.load(open(path)) if r["n"] > 3_000_000]
        recs = old + recs
    json.dump(recs, open(path, "w"), indent=1)
    pb = periodic_corpus_blocks()
    json.dump(pb, open(os.path.join(HERE, "periodic_blocks.json"), "w"), indent=0)
    print("fixtures:", len(recs), "periodic corpus blocks:", len(pb))


if __name__ == "__main__":
    main()
