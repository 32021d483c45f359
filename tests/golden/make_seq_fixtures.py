"""tests/golden/seq_fixtures.json: the reference's -u / --sequential streams (compress.c:129-198) of seeded inputs,
from the COMPILED reference codec driven as do_collect_seq drives it (oracle/ref_probe.c: ref_compress_seq; checked here
against the reference CLI `lbzip2 -u` on the smaller cases).  Build container only."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as L
from golden_util import gen

CASES = [("wiki", 100_000_000, 1, 9), ("mixed", 60_000_000, 3, 1), ("tar", 50_000_000, 5, 9), ("runs", 20_000_000, 4, 5),
         ("rand", 10_000_000, 4, 1), ("text", 30_000_000, 2, 9), ("wiki", 1_000_000_000, 2, 9),
         ("wiki", 350_000, 7, 1), ("runs", 450_000, 7, 1), ("rand", 250_000, 7, 1)]
STOCK = os.path.join(ROOT, "oracle", "_ref", "lbzip2_stock")
recs = []
for kind, n, seed, level in CASES:
    data = bytes(gen(kind, n, seed))
    z = L.ref_compress_seq(data, level)
    if n <= 100_000_000:
        cli = subprocess.run([STOCK, "-u", "-%d" % level, "-c"], input=data, capture_output=True).stdout
        assert cli == z, (kind, n)
    dflt = None
    recs.append({"kind": kind, "n": n, "seed": seed, "level": level, "in_md5": hashlib.md5(data).hexdigest(),
                 "out_len": len(z), "ref_md5": hashlib.md5(z).hexdigest(), "nblocks": z.count(bytes.fromhex("314159265359"))})
    print(recs[-1], flush=True)
json.dump({"source": "compiled reference, sequential mode (lbzip2 -u)", "records": recs},
          open(os.path.join(ROOT, "tests", "golden", "seq_fixtures.json"), "w"), indent=1)
