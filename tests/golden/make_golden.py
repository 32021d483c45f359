#!/usr/bin/env python3
"""Generate the committed golden vectors from the COMPILED REFERENCE.

Run in the build container only (needs /root/reference and oracle/_ref/libref.so):
    python tests/golden/make_golden.py
Writes:
    tests/golden/suite_inputs.tar   the reference's own compress-test corpora
                                    (tests/suite/{manual-compress,fuzz-collect,fuzz-divbwt}/*.bz2,
                                    test DATA, CMakeLists.txt:36-64), unchanged
    tests/golden/suite_expected.json   per input, level -9/-1: length+md5 of the reference's
                                    .bz2 output, number of blocks, number of exactly-periodic
                                    blocks, and md5 of the canonical-origin-pointer variant
    tests/golden/streams.json       tiny literal known-answer streams + seeded-generator md5s
    tests/golden/stages.json        per-stage values for a handful of blocks
"""
import bz2
import glob
import hashlib
import io
import json
import os
import sys
import tarfile
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as L  # noqa: E402

SUITE = "/root/reference/tests/suite"
SUITES = ["manual-compress", "fuzz-collect", "fuzz-divbwt"]


def md5(b):
    return hashlib.md5(b).hexdigest()


def suite_case(path):
    raw = bz2.decompress(open(path, "rb").read())
    rec = {"raw_len": len(raw), "raw_md5": md5(raw)}
    for lvl in (9, 1):
        r = L.ref_compress(raw, lvl)
        o = L.orc_compress(raw, lvl)
        blocks = L.orc_blocks(raw, lvl)
        rec[str(lvl)] = {"len": len(r), "ref_md5": md5(r), "canon_md5": md5(o),
                         "blocks": len(blocks),
                         "periodic_blocks": sum(b["periodic"] for b in blocks)}
        assert len(o) == len(r)
        assert (o == r) == (rec[str(lvl)]["periodic_blocks"] == 0) or o == r
    return "/".join(path.split("/")[-2:]), rec


def stage_record(name, data, lvl, blk=0):
    rb = L.ref_blocks(data, lvl)[blk]
    return {"name": name, "level": lvl, "block": blk, "consumed": rb["consumed"],
            "nblock": rb["nblock"], "crc": rb["crc"], "inuse_md5": md5(rb["inuse"]),
            "block_md5": md5(rb["block"]), "bwt_md5": md5(rb["bwt"]), "bwt_idx": rb["bwt_idx"],
            "nmtf": rb["nmtf"], "mtfv_md5": md5(rb["mtfv"]), "alpha": rb["alpha"],
            "num_trees": rb["num_trees"], "num_selectors": rb["num_selectors"],
            "tree_pad": rb["tree_pad"], "selector_md5": md5(rb["selector"]),
            "lengths": [l.hex() for l in rb["lengths"]],
            "out_len": rb["out_len"], "out_md5": md5(rb["out"])}


def main():
    assert L.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    files = []
    for s in SUITES:
        files += sorted(glob.glob(os.path.join(SUITE, s, "*.bz2")))
    with tarfile.open(os.path.join(HERE, "suite_inputs.tar"), "w") as tf:
        for f in files:
            ti = tarfile.TarInfo("/".join(f.split("/")[-2:]))
            data = open(f, "rb").read()
            ti.size = len(data)
            tf.addfile(ti, io.BytesIO(data))
    with ProcessPoolExecutor(os.cpu_count()) as ex:
        expected = dict(ex.map(suite_case, files, chunksize=8))
    json.dump(expected, open(os.path.join(HERE, "suite_expected.json"), "w"), indent=0, sort_keys=True)

    streams = {"literals": {}, "seeded": []}
    for s in [b"", b"a", b"aaaa", b"banana", b"abababab", b"\x00" * 300, bytes(range(256)),
              b"mississippi", b"a" * 259 + b"b", b"ab" * 20 + b"c"]:
        streams["literals"][s.hex()] = {str(l): L.ref_compress(s, l).hex() for l in (9, 1)}
    for kind, n, seed, lvl in [("rand", 2000000, 1, 9), ("text", 2000000, 1, 9),
                               ("text", 2000000, 1, 1), ("zero", 1000000, 0, 9),
                               ("ab", 900000, 0, 9), ("text", 5000000, 7, 9),
                               ("rand", 1000000, 9, 1), ("text", 1000000, 3, 5)]:
        data = gen(kind, n, seed)
        r = L.ref_compress(data, lvl)
        o = L.orc_compress(data, lvl)
        streams["seeded"].append({"kind": kind, "n": n, "seed": seed, "level": lvl,
                                  "in_md5": md5(data), "out_len": len(r), "ref_md5": md5(r),
                                  "canon_md5": md5(o)})
    json.dump(streams, open(os.path.join(HERE, "streams.json"), "w"), indent=1)

    stages = []
    for kind, n, seed, lvl in [("text", 2000000, 1, 9), ("text", 2000000, 1, 1),
                               ("rand", 2000000, 1, 9), ("rand", 300000, 5, 1),
                               ("zero", 1000000, 0, 9), ("text", 30000, 11, 9),
                               ("text", 700, 12, 9), ("text", 100, 13, 9)]:
        stages.append(dict(stage_record(f"{kind}({n},{seed})", gen(kind, n, seed), lvl),
                           kind=kind, n=n, seed=seed))
    # a spill block (RLE1 expansion forces a second block in the slab)
    data = gen("runs", 900000, 21)
    for b in range(len(L.ref_blocks(data, 9))):
        stages.append(dict(stage_record("runs(900000,21)", data, 9, b), kind="runs", n=900000, seed=21))
    json.dump(stages, open(os.path.join(HERE, "stages.json"), "w"), indent=1)
    print("suite cases:", len(expected), "stage records:", len(stages))


def gen(kind, n, seed):
    if kind == "rand":
        return L.gen_rand(n, seed)
    if kind == "text":
        return L.gen_text(n, seed)
    if kind == "zero":
        return bytes(n)
    if kind == "ab":
        return (b"ab" * (n // 2 + 1))[:n]
    if kind == "runs":                      # runs of exactly 4: worst-case RLE1 expansion
        r = L.gen_rand(n // 4 + 1, seed)
        out = bytearray()
        prev = -1
        for b in r:
            if b == prev:
                b = (b + 1) & 255
            out += bytes([b]) * 4
            prev = b
        return bytes(out[:n])
    raise ValueError(kind)


if __name__ == "__main__":
    main()
