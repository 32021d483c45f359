"""CPU: the oracle restatement against the committed golden vectors (generated from the
compiled reference by tests/golden/make_golden.py)."""
import bz2
import os
from concurrent.futures import ProcessPoolExecutor

import pytest

import oracle_lib as L
from golden_util import gen, load, md5, suite_inputs


def test_literal_streams():
    for hx, by_level in load("streams.json")["literals"].items():
        data = bytes.fromhex(hx)
        for lvl, want in by_level.items():
            got = L.orc_compress(data, int(lvl))
            if data == b"abababab":            # exactly periodic: origin pointer differs (documented)
                assert len(got) == len(bytes.fromhex(want))
                assert bz2.decompress(got) == data
                continue
            assert got.hex() == want, (hx, lvl)


def test_generators_selfcheck():
    assert L.gen_rand(8, 1).hex() == "00049d128e2c2519"
    assert L.gen_text(80, 1) == (b"avqlqq shdqj cg yubylkic smys xwr qp fo iinjmrio brbm\n"
                                 b"bxhyqjx unbawqk qblr hk wu")


@pytest.mark.parametrize("rec", load("streams.json")["seeded"],
                         ids=lambda r: f"{r['kind']}-{r['n']}-{r['seed']}-L{r['level']}")
def test_seeded_streams(rec):
    data = gen(rec["kind"], rec["n"], rec["seed"])
    assert md5(data) == rec["in_md5"]
    out = L.orc_compress(data, rec["level"])
    assert len(out) == rec["out_len"]
    assert md5(out) == rec["canon_md5"]
    if rec["kind"] != "ab":
        assert rec["canon_md5"] == rec["ref_md5"]


@pytest.mark.parametrize("rec", load("stages.json"),
                         ids=lambda r: f"{r['name']}-L{r['level']}-b{r['block']}")
def test_stage_goldens(rec):
    data = gen(rec["kind"], rec["n"], rec["seed"])
    b = L.orc_blocks(data, rec["level"])[rec["block"]]
    for k in ("consumed", "nblock", "crc", "bwt_idx", "nmtf", "alpha", "num_trees",
              "num_selectors", "tree_pad", "out_len"):
        assert b[k] == rec[k], k
    assert md5(b["inuse"]) == rec["inuse_md5"]
    assert md5(b["block"]) == rec["block_md5"]
    assert md5(b["bwt"]) == rec["bwt_md5"]
    assert md5(b["mtfv"]) == rec["mtfv_md5"]
    assert md5(b["selector"]) == rec["selector_md5"]
    assert [l.hex() for l in b["lengths"]] == rec["lengths"]
    assert md5(b["out"]) == rec["out_md5"]


def _suite_chunk(names):
    inputs = suite_inputs()
    exp = load("suite_expected.json")
    bad = []
    for name in names:
        raw = inputs[name]
        for lvl in ("9", "1"):
            out = L.orc_compress(raw, int(lvl))
            e = exp[name][lvl]
            if len(out) != e["len"] or md5(out) != e["canon_md5"]:
                bad.append((name, lvl))
            if e["periodic_blocks"] == 0 and e["canon_md5"] != e["ref_md5"]:
                bad.append((name, lvl, "fixture"))
    return bad


def test_reference_suite_corpora():
    """All 1093 inputs of the reference's compress suites, -9 and -1: byte-identical to the
    reference except the origin pointer of exactly-periodic blocks."""
    names = sorted(suite_inputs())
    assert len(names) == 1093
    chunks = [names[i::16] for i in range(16)]
    with ProcessPoolExecutor(min(8, os.cpu_count() or 1)) as ex:
        bad = [b for r in ex.map(_suite_chunk, chunks) for b in r]
    assert not bad, bad[:10]


def test_periodic_blocks_enumerated():
    """The one documented divergence, enumerated: tests/golden/periodic_blocks.json lists every exactly
    periodic block (T = u^k) of the reference's own compress corpora with the reference's origin
    pointer and the smallest equal row (the reference's row rounded down to a multiple of k: the k equal
    rows of a rotation class are consecutive and classes start at multiples of k).  The oracle must
    emit exactly that row; with the compiled reference present, its row must be the listed one."""
    from golden_util import load, suite_inputs
    pb = load("periodic_blocks.json")
    assert len(pb) >= 100
    inputs = suite_inputs()
    seen = 0
    for e in pb:
        raw = inputs[e["input"]]
        assert e["copies"] >= 2 and e["canon_bwt_idx"] == e["ref_bwt_idx"] - e["ref_bwt_idx"] % e["copies"]
        if len(raw) > 120000:
            continue
        blk = L.orc_blocks(raw, e["level"])[e["block"]]
        assert blk["periodic"] and blk["bwt_idx"] == e["canon_bwt_idx"], e
        if L.have_ref():
            assert L.ref_blocks(raw, e["level"])[e["block"]]["bwt_idx"] == e["ref_bwt_idx"], e
        seen += 1
    assert seen >= 12
