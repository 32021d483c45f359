"""Differential fuzz on the GPU against the COMPILED REFERENCE: random slices (offset, length, level, seed) of the
workload generators (wiki / tar / mixed / text / rand and splices of them) -- blocks with deep ties, oversized
groups, mixed alphabets -- vs oracle/_ref through the pthreads driver (canon = smallest equal row for exactly
periodic blocks).  `python tests/fuzz_corpora_gpu.py SEED COUNT`; test_gpu_parity.py runs a slice of it."""
import random
import sys
import time


def make(rng, L):
    parts = []
    for _ in range(rng.choice([1, 1, 2, 3])):
        kind = rng.choice(["wiki", "wiki", "tar", "tar", "mixed", "text", "rand"])
        n = rng.choice([rng.randint(1, 5000), rng.randint(50_000, 400_000), rng.randint(800_000, 3_000_000)])
        skip = rng.randint(0, 200_000) if kind != "mixed" else 0
        buf = L.gen_kind(kind, n + skip, rng.randint(1, 1000))
        parts.append(bytes(buf[skip:]))
    if rng.random() < 0.2:                                   # a verbatim duplicate of everything so far: very deep ties
        parts.append(parts[0][:rng.randint(1, len(parts[0]))])
    return b"".join(parts)


def run(lib, L, seed, count):
    rng = random.Random(seed)
    bad = []
    for i in range(count):
        data = make(rng, L)
        level = rng.choice([1, 3, 9, 9])
        want = (L.ref_compress_mt(data, level, 16, canon=True)[0] if L.have_ref() else L.orc_compress_mt(data, level, 16)[0])
        if lib.compress(data, level) != want:
            bad.append((i, len(data), level))
    return bad


if __name__ == "__main__":
    sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
    import torch, lbzip2_amd  # noqa: F401
    import oracle_lib as L
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    t0 = time.time()
    bad = run(lbzip2_amd.library(), L, seed, count)
    print("corpora fuzz seed", seed, "cases", count, "mismatches", bad, "in %.1f s" % (time.time() - t0), flush=True)
