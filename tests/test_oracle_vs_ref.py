"""CPU, build container only: pin every oracle stage against the compiled reference
(oracle/_ref/libref.so = the reference's encode.c/divbwt.c/crctab.c).  Skipped where the
reference build is absent."""
import pytest

import oracle_lib as L
from golden_util import gen, suite_inputs

pytestmark = pytest.mark.skipif(not L.have_ref(), reason="oracle/_ref/libref.so not built")

KEYS = ["consumed", "nblock", "crc", "inuse", "block", "bwt", "bwt_idx", "nmtf", "mtfv", "alpha",
        "num_trees", "num_selectors", "tree_pad", "selector", "lengths", "out_len", "out"]


def _cmp(data, lvl):
    ob, rb = L.orc_blocks(data, lvl), L.ref_blocks(data, lvl)
    assert len(ob) == len(rb)
    for o, r in zip(ob, rb):
        diff = [k for k in KEYS if o[k] != r[k]]
        if o["periodic"]:
            assert set(diff) <= {"bwt_idx", "out"}
            a, b = bytearray(o["out"]), bytearray(r["out"])
            a[10:14] = b[10:14] = b"\0" * 4          # the 24-bit origin pointer lives in bytes 10..13
            assert a == b
        else:
            assert not diff, diff


@pytest.mark.parametrize("kind,n,seed,lvl", [("text", 1000000, 31, 9), ("rand", 250000, 32, 1),
                                              ("runs", 900000, 33, 9), ("text", 4000, 34, 9),
                                              ("zero", 950000, 0, 9), ("ab", 123456, 0, 9)])
def test_stages_seeded(kind, n, seed, lvl):
    _cmp(gen(kind, n, seed), lvl)


def test_stages_suite_sample():
    names = sorted(suite_inputs())
    for name in names[::23]:
        _cmp(suite_inputs()[name], 9)
        _cmp(suite_inputs()[name], 1)


def test_rle_boundary_sweep():
    """Runs of length 1..8, 258..264, 518, 519 placed 0..8 bytes before the block limit
    (the sweep SURVEY.md App. B2 describes), at -1."""
    M = 100000
    for run in list(range(1, 9)) + list(range(258, 265)) + [518, 519]:
        for before in range(0, 9):
            head = bytes((i * 7 + 1) & 0xFF or 1 for i in range(M - before))
            # avoid accidental runs in the filler
            head = bytes(b if (i == 0 or b != head[i - 1]) else (b + 1) & 0xFF for i, b in enumerate(head))
            data = head + bytes([0xAA]) * run + b"xyz"
            for o, r in zip(L.orc_blocks(data[:M], 1), L.ref_blocks(data[:M], 1)):
                for k in ("consumed", "nblock", "crc", "block"):
                    assert o[k] == r[k], (run, before, k)
