"""CPU: the parts of bench.py that do not need a GPU -- the input generator (the SURVEY.md 8d
stand-in text) and the cpu_baseline leg (the only place outside tests/ and smoke() that may call
into oracle/)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
import oracle_lib as L  # noqa: E402


def test_generator_matches_oracle_generator():
    n = 300000
    assert bytes(bench.gen_input("text", n, 2)) == L.gen_text(n, 2)
    assert bytes(bench.gen_input("rand", n, 7)) == L.gen_rand(n, 7)
    # known answers of SURVEY.md App. B
    assert L.gen_rand(8, 1).hex() == "00049d128e2c2519"


def test_cpu_baseline_leg():
    data = bench.gen_input("text", 4 * 900000, 2)
    r = bench.cpu_baseline(data, 9)
    assert r["kind"] in ("reference", "port") and r["unit"] == "MB/s"
    assert r["value"] > 0 and 1 <= r["cores"] <= (os.cpu_count() or 1)
    assert "slabs of 900000 B" in r["sample"]
