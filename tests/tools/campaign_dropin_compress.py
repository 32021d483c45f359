"""Ad-hoc: the reference's own program linked against the library (oracle/_ref/lbzip2_dropin_emu: encode.h's five symbols on the
emulated device) and the stock program, random inputs, levels, worker counts, -u: the same bytes.  usage: campaign_dropin_compress.py cases seed"""
import os, random, subprocess, sys, hashlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from golden_util import gen
REF = "/root/repo/oracle/_ref"
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
env = dict(os.environ, LBZAMD_POOL_SLABS="8", LBZ_EMU_THREADS="2", LBZ_EMU_CHECK_SITES="2")
bad = 0
for it in range(cases):
    kind = rng.choice(["wiki", "text", "rand", "runs", "lines", "mixed"])
    n = rng.choice([0, 1, 7, 99999, 100000, 100001, 250000, 420000])
    data = bytes(gen(kind, n, rng.randrange(1000))) if n else b""
    argv = ["-%d" % rng.choice([1, 1, 1, 2, 3]), "-n", str(rng.randrange(1, 5))] + (["-u"] if rng.random() < 0.25 else [])
    a = subprocess.run([REF + "/lbzip2_stock"] + argv, input=data, capture_output=True, timeout=600)
    b = subprocess.run([REF + "/lbzip2_dropin_emu"] + argv, input=data, capture_output=True, timeout=900, env=env)
    if (a.returncode, a.stdout) != (b.returncode, b.stdout):
        bad += 1
        print("case", it, kind, n, argv, a.returncode, b.returncode, hashlib.md5(a.stdout).hexdigest(), hashlib.md5(b.stdout).hexdigest(), b.stderr[-200:], flush=True)
    if it % 10 == 9: print("..", it + 1, "cases,", bad, "differ", flush=True)
print("done:", cases, "cases,", bad, "differ")
