"""Ad-hoc: damaged .bz2 streams through `lbzamd -dc` (emulator build) and the compiled reference program: same exit status,
same diagnostic (program name aside), same bytes on stdout."""
import bz2, os, random, subprocess, sys, hashlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from golden_util import gen
ROOT = "/root/repo"
STOCK = ROOT + "/oracle/_ref/lbzip2_stock"
EMU = ROOT + "/tests/emu/_build/lbzamd_emu"
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
srcs = [bytes(gen("wiki", 130000, 3)), bytes(gen("rand", 40000, 4)), bytes(gen("runs", 90000, 5)), b"ab" * 30000, bytes(gen("text", 250000, 6))]
streams = [bz2.compress(d, 1) for d in srcs] + [bz2.compress(srcs[0], 1) + bz2.compress(srcs[2], 1), bz2.compress(srcs[4], 1) + b"\0\0trailing"]
env = {k: v for k, v in os.environ.items() if k not in ("LBZIP2", "BZIP2", "BZIP")}
env.update({"LBZ_EMU_THREADS": "2", "LBZAMD_POOL_SLABS": "4", "LBZAMD_DWIDE": "0"})
def run(prog, data):
    p = subprocess.run([prog, "-dc"], input=data, env=env, capture_output=True, timeout=600)
    name = os.path.basename(prog).encode()
    err = b"\n".join(l[len(name) + 2:] if l.startswith(name + b": ") else l for l in p.stderr.split(b"\n"))
    return p.returncode, err, hashlib.md5(p.stdout).hexdigest(), len(p.stdout)
bad = 0
for it in range(cases):
    z = streams[it % len(streams)]
    b = bytearray(z)
    kind = rng.randrange(6)
    if kind == 0:
        for _ in range(rng.randrange(1, 4)): b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
    elif kind == 1:
        del b[rng.randrange(len(b)):]
    elif kind == 2:
        p = rng.randrange(len(b)); b[p:p] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
    elif kind == 3:
        p = rng.randrange(len(b)); del b[p:p + rng.randrange(1, 40)]
    elif kind == 4:
        p = rng.randrange(len(b) - 8); b[p:p + 8] = bytes(rng.randrange(256) for _ in range(8))
    else:
        p = rng.randrange(min(len(b), 60)); b[p] ^= 1 << rng.randrange(8)      # the headers
    refs = set()
    for k in range(4):
        p = subprocess.run([STOCK, "-dc"] + (["-n", "1"] if k >= 2 else []), input=bytes(b), env=env, capture_output=True, timeout=600)
        refs.add((p.returncode, p.stderr.split(b"stdin: ")[-1].strip()))
    c0 = run(EMU, bytes(b)); c = (c0[0], c0[1].split(b"stdin: ")[-1].strip()); a = sorted(refs)
    if c not in refs:
        bad += 1
        open("/tmp/campaign_damaged_fail_%d.bz2" % it, "wb").write(bytes(b))
        print("case", it, "kind", kind, "stream", it % len(streams), "\n  ref", a, "\n  got", c, flush=True)
    if it % 20 == 19: print("..", it + 1, "cases,", bad, "differ", flush=True)
print("done:", cases, "cases,", bad, "differ")
