"""Shared helpers for the golden fixtures (tests/golden/*)."""
import bz2
import hashlib
import json
import os
import tarfile

import oracle_lib as L

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def md5(b):
    return hashlib.md5(b).hexdigest()


def load(name):
    return json.load(open(os.path.join(GOLD, name)))


def gen(kind, n, seed):
    """Seeded inputs; same definitions as tests/golden/make_golden.py::gen."""
    if kind == "rand":
        return L.gen_rand(n, seed)
    if kind == "text":
        return L.gen_text(n, seed)
    if kind == "zero":
        return bytes(n)
    if kind == "ab":
        return (b"ab" * (n // 2 + 1))[:n]
    if kind == "runs":
        r = L.gen_rand(n // 4 + 1, seed)
        out = bytearray()
        prev = -1
        for b in r:
            if b == prev:
                b = (b + 1) & 255
            out += bytes([b]) * 4
            prev = b
        return bytes(out[:n])
    if kind == "lines":
        # source-code-like: lines drawn from a small pool (long exact repeats, indentation runs)
        pool_txt = L.gen_text(40000, seed + 1000).split(b"\n")
        pool = [b" " * (4 * (i % 5)) + l[:20 + (i * 7) % 90] for i, l in enumerate(pool_txt) if l][:150]
        r = L.gen_rand(4 * (n // 20 + 64), seed)
        out = bytearray()
        i = 0
        while len(out) < n:
            k = int.from_bytes(r[4 * i:4 * i + 4], "little"); i += 1
            if k % 7 == 0 and len(out) > 4000:           # repeat a recent passage verbatim
                start = len(out) - 1 - (k >> 8) % 3500
                out += out[start:start + 200 + (k >> 20) % 800]
            else:
                out += pool[(k >> 4) % len(pool)] + b"\n"
        return bytes(out[:n])
    if kind in ("wiki", "mixed", "tar"):          # lbzip2_amd/host/gen_inputs.c: the BASELINE.json stand-ins
        return L.gen_kind(kind, n, seed)
    raise ValueError(kind)


def bench_fixtures(max_n=None, min_n=0):
    """Records of tests/golden/bench_fixtures.json (reference streams of the BASELINE configs)."""
    return [r for r in load("bench_fixtures.json") if (max_n is None or r["n"] <= max_n) and r["n"] >= min_n]


_suite_cache = None


def suite_inputs():
    """dict name -> raw bytes of the reference's compress-test corpora."""
    global _suite_cache
    if _suite_cache is None:
        d = {}
        with tarfile.open(os.path.join(GOLD, "suite_inputs.tar")) as tf:
            for m in tf.getmembers():
                d[m.name] = bz2.decompress(tf.extractfile(m).read())
        _suite_cache = d
    return _suite_cache
