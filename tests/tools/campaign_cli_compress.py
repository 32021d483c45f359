"""Ad-hoc: random command lines (compression side, files and pipes) through lbzamd (emulator) and the compiled reference, side by side."""
import os, random, sys, pathlib, tempfile, shutil, traceback
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ.setdefault("LBZ_EMU_CHECK_SITES", "2")
import test_cli as T
from golden_util import gen
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cli = os.path.join(T.EMU_DIR, "_build", "lbzamd_emu")
kinds = ["wiki", "text", "rand", "runs", "lines", "mixed"]
bad = 0
for it in range(cases):
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="fzc"))
    try:
        nfiles = rng.choice([0, 1, 1, 2, 3])
        files = {}
        for k in range(nfiles):
            n = rng.choice([0, 1, 5, 4000, 99999, 100000, 100001, 180000, 250000])
            data = bytes(gen(rng.choice(kinds), n, rng.randrange(1000))) if n else b""
            name = rng.choice(["a", "b.txt", "c.tar", "d.bz2", "e.tbz", "f f", "-g"]) + str(k)
            files[name] = T.F(data, rng.choice([0o644, 0o600, 0o640]), T.T0 + rng.randrange(100000))
        argv = [rng.choice(["-1", "-1", "-2", "-1k", "-1v", "-1q", "--fast", "-1kf"])]
        if rng.random() < 0.2: argv.append("-u")
        if rng.random() < 0.2: argv += ["-n", str(rng.randrange(1, 5))]
        if rng.random() < 0.15: argv.append("-c")
        if rng.random() < 0.1: argv.append("-z")
        if rng.random() < 0.1: argv += ["-S", ".zz"] if False else []
        stdin = b""
        if nfiles == 0:
            n = rng.choice([0, 1, 3000, 150000])
            stdin = bytes(gen(rng.choice(kinds), n, rng.randrange(1000))) if n else b""
            if rng.random() < 0.5: argv.append("-c")
        names = list(files)
        rng.shuffle(names)
        if names and rng.random() < 0.1: names.append("missing-file")
        dashdash = ["--"] if any(n.startswith("-") for n in names) else []
        env = {}
        if rng.random() < 0.15: env["LBZIP2"] = rng.choice(["-k", "-v", "-2 -k"])
        T._both(tmp, cli, files, argv + dashdash + names, env=env, stdin=stdin)
    except AssertionError as e:
        bad += 1
        print("case", it, "DIFF", str(e)[:600], flush=True)
    except Exception as e:
        bad += 1
        print("case", it, "EXC", repr(e)[:300], flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if it % 10 == 9: print("..", it + 1, "cases,", bad, "differ", flush=True)
print("done:", cases, "cases,", bad, "differ")
