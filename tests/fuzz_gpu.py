"""Differential fuzz on the GPU: structured random inputs (small alphabets, runs at the RLE1 limits,
near-periodic text, word soups) vs the CPU oracle.  `python tests/fuzz_gpu.py SEED COUNT [big]`;
test_gpu_parity.py runs a slice of it."""
import random, sys, time

BIG = False
SMALL = False


def make(rng):
    kind = rng.choice(["alpha", "markov", "repeat", "runs", "mix", "periodic", "words"])
    n = rng.choice([rng.randint(1, 300), rng.randint(4000, 9000), rng.randint(20000, 120000), rng.randint(150000, 400000)])
    if BIG: n = rng.randint(900000, 2200000)
    if SMALL: n = rng.choice([rng.randint(1, 300), rng.randint(4000, 9000)])
    if kind == "alpha":
        k = rng.choice([1, 2, 3, 4, 7, 16, 60, 64, 65, 128, 129, 200, 256])
        syms = rng.sample(range(256), k)
        return bytes(rng.choice(syms) for _ in range(n))
    if kind == "markov":
        k = rng.choice([2, 3, 5, 20]); syms = rng.sample(range(256), k); p = rng.random() * 0.2
        out = bytearray(); c = syms[0]
        for _ in range(n):
            if rng.random() < p: c = rng.choice(syms)
            out.append(c)
        return bytes(out)
    if kind == "repeat":
        unit = bytes(rng.randrange(256) for _ in range(rng.randint(1, 2000)))
        out = bytearray()
        while len(out) < n:
            out += unit
            if rng.random() < 0.05: out += bytes([rng.randrange(256)])
        return bytes(out[:n])
    if kind == "runs":
        out = bytearray()
        while len(out) < n:
            out += bytes([rng.randrange(4)]) * rng.choice([1, 2, 3, 4, 5, 254, 255, 256, 259, 260, 1000])
        return bytes(out[:n])
    if kind == "periodic":
        unit = bytes(rng.randrange(3) for _ in range(rng.randint(1, 50)))
        return (unit * (n // len(unit) + 1))[:n]
    if kind == "words":
        words = [bytes(rng.randrange(97, 123) for _ in range(rng.randint(1, 9))) for _ in range(rng.randint(3, 300))]
        out = bytearray()
        while len(out) < n: out += rng.choice(words) + b" "
        return bytes(out[:n])
    a, b = make(rng), make(rng)
    return (a + b)[:max(1, n)]

def run(lib, oracle_compress, seed, count, big=False, save=None, small=False):
    """Returns the list of (index, length, level) of mismatching cases."""
    global BIG, SMALL
    BIG, SMALL = big, small
    rng = random.Random(seed)
    bad = []
    for i in range(count):
        data = make(rng)
        level = rng.choice([1, 1, 2, 9])
        if lib.compress(data, level) != oracle_compress(data, level):
            bad.append((i, len(data), level))
            if save:
                open(f"{save}/fuzz_fail_{seed}_{i}.bin", "wb").write(data)
    return bad


def run_decode(lib, seed, count, small=False, both_widths=False):
    """The inverse path on the same inputs: streams written by Python's bz2 (bzip2's own encoder: bit-aligned blocks, its
    own code tables) and by this library, one or two streams per file.  Returns the mismatching case indices."""
    import bz2, os
    global BIG, SMALL
    BIG, SMALL = False, small
    rng = random.Random(seed)
    bad = []
    for i in range(count):
        data = make(rng)
        level = rng.choice([1, 1, 2, 9])
        z = bz2.compress(data, level) if rng.random() < 0.6 else lib.compress(data, level)
        want = data
        if rng.random() < 0.3:
            more = make(rng)[:50000]
            z += bz2.compress(more, rng.choice([1, 9]))
            want = data + more
        for wide in (("0", "1", "2") if both_widths else (None,)):
            if wide is not None:
                os.environ["LBZAMD_DWIDE"] = wide
            try:
                ok = lib.decompress(z) == want
            except Exception as ex:                                   # noqa: BLE001 -- a refused valid stream is a failure too
                ok = False
            if not ok:
                bad.append((i, len(data), level, wide))
    if both_widths:
        os.environ.pop("LBZAMD_DWIDE", None)
    return bad


if __name__ == "__main__":
    sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
    import torch, lbzip2_amd  # noqa: F401
    import oracle_lib as L
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    t0 = time.time()
    bad = run(lbzip2_amd.library(), L.orc_compress, seed, count, len(sys.argv) > 3 and sys.argv[3] == "big", "/root/repo/gpurun_out")
    print("fuzz seed", seed, "cases", count, "mismatches", bad, "in %.1f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    print("decode fuzz seed", seed, "cases", count, "mismatches", run_decode(lbzip2_amd.library(), seed, count, both_widths=True), "in %.1f s" % (time.time() - t0), flush=True)
