"""Ad-hoc: random command lines, both directions, existing outputs, suffixes, -f/-k/-t/-c, names through lbzamd (emulator) and the compiled reference."""
import os, random, sys, pathlib, tempfile, shutil, bz2
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ.setdefault("LBZ_EMU_CHECK_SITES", "2")
import test_cli as T
from golden_util import gen
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cli = os.path.join(T.EMU_DIR, "_build", "lbzamd_emu")
kinds = ["wiki", "text", "rand", "runs", "lines"]
bad = 0
for it in range(cases):
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="fzm"))
    try:
        files = {}
        names = []
        for k in range(rng.choice([1, 1, 2, 3])):
            n = rng.choice([0, 1, 4000, 120000])
            data = bytes(gen(rng.choice(kinds), n, rng.randrange(1000))) if n else b""
            stem = rng.choice(["a", "b.txt", "c", "d"]) + str(k)
            what = rng.random()
            if what < 0.55:                                     # a compressed operand, under one of the names lbzip2 knows
                suf = rng.choice([".bz2", ".bz2", ".tbz2", ".tbz", ".tz2", ".bz", "", ".out"])
                z = bz2.compress(data, rng.choice([1, 9])) if rng.random() < 0.8 else bz2.compress(data[:len(data) // 2], 1) + bz2.compress(data[len(data) // 2:], 1)
                if rng.random() < 0.1: z += b"\0trailing"
                if rng.random() < 0.07: z = data[:50] or b"not bzip2"
                files[stem + suf] = T.F(z, rng.choice([0o644, 0o600]), T.T0 + rng.randrange(100000))
                names.append(stem + suf)
                if rng.random() < 0.25:                         # the output exists already
                    out = stem + {".bz2": "", ".bz": "", ".tbz2": ".tar", ".tbz": ".tar", ".tz2": ".tar"}.get(suf, suf + ".out")
                    if out and out not in files: files[out] = T.F(b"old", 0o644, T.T0)
            else:
                files[stem] = T.F(data, rng.choice([0o644, 0o640]), T.T0 + rng.randrange(100000))
                names.append(stem)
                if rng.random() < 0.25: files[stem + ".bz2"] = T.F(b"old", 0o644, T.T0)
        mode = rng.choice(["-d", "-d", "-t", "-dk", "-dc", "-df", "-tv", "-1", "-1f", "-dkf", "-dq", "-dv", "-1k"])
        argv = [mode]
        if rng.random() < 0.15: argv += ["-n", str(rng.randrange(1, 4))]
        argv0 = rng.choice([None, None, None, "bunzip2", "bzcat", "lbunzip2"])
        rng.shuffle(names)
        # damaged streams excluded: which of two diagnostics the reference prints for them is a race (tests/golden/damaged_cases.json)
        T._both(tmp, cli, files, argv + names, argv0=argv0)
    except AssertionError as e:
        bad += 1
        print("case", it, "DIFF", str(e)[:700], flush=True)
    except Exception as e:
        bad += 1
        print("case", it, "EXC", repr(e)[:300], flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if it % 10 == 9: print("..", it + 1, "cases,", bad, "differ", flush=True)
print("done:", cases, "cases,", bad, "differ")
