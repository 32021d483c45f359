"""Ad-hoc robustness run (not a test): random damage to .bz2 streams must end in LbzError or in the right bytes --
never in a hang, a crash or wrong bytes accepted.  usage: python tests/fuzz_decode_gpu.py [cases] [seed]"""
import bz2
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lbzip2_amd
from golden_util import gen

lib = lbzip2_amd.Library(os.environ["LBZ_LIB"]) if os.environ.get("LBZ_LIB") else lbzip2_amd.library()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
srcs = [bytes(gen("wiki", 420000, 3)), bytes(gen("rand", 150000, 4)), bytes(gen("runs", 300000, 5)), b"ab" * 70000]
streams = [(d, bz2.compress(d, 1)) for d in srcs] + [(srcs[0] + srcs[2], bz2.compress(srcs[0], 2) + bz2.compress(srcs[2], 1))]
refused = same = wrong = 0
with lib.decoder(16) as dec:
    for it in range(cases):
        d, z = streams[it % len(streams)]
        b = bytearray(z)
        kind = rng.randrange(5)
        if kind == 0:
            for _ in range(rng.randrange(1, 4)): b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif kind == 1:
            del b[rng.randrange(len(b)):]
        elif kind == 2:
            p = rng.randrange(len(b)); b[p:p] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
        elif kind == 3:
            p = rng.randrange(len(b)); del b[p:p + rng.randrange(1, 40)]
        else:
            p = rng.randrange(len(b) - 8); b[p:p + 8] = bytes(rng.randrange(256) for _ in range(8))
        try:
            out = dec.decompress(bytes(b))
        except lbzip2_amd.LbzError:
            refused += 1
            continue
        try:
            ok = out == bz2.decompress(bytes(b))          # accepted: then it must be what a CPU decoder makes of it
        except Exception:
            ok = out == d or len(out) < len(d)            # libbz2 refuses what we took: only trailing damage (ignored garbage) may do that
        same += ok
        wrong += not ok
print(f"{cases} damaged streams: {refused} refused, {same} accepted with the right bytes, {wrong} WRONG")
sys.exit(1 if wrong else 0)
