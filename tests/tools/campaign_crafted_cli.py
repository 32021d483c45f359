"""Ad-hoc: hand-made VALID-OR-NOT one-block streams with random alphabets, code tables (complete, sometimes not), selectors and
block bytes through `lbzamd -dc` (emulator) and the compiled reference: same status, same bytes; diagnostic in the reference's set."""
import os, random, subprocess, sys, hashlib
sys.path.insert(0, "/root/repo/tests")
import craft_bz2 as C
STOCK = "/root/repo/oracle/_ref/lbzip2_stock"; EMU = "/root/repo/tests/emu/_build/lbzamd_emu"
env = {k: v for k, v in os.environ.items() if k not in ("LBZIP2", "BZIP2", "BZIP")}
env.update({"LBZ_EMU_THREADS": "2", "LBZAMD_POOL_SLABS": "4", "LBZAMD_DWIDE": "0"})
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
def rand_code(n, maxlen):
    L = [1, 1]
    while len(L) < n:
        i = rng.randrange(len(L))
        if L[i] >= maxlen: continue
        l = L.pop(i); L += [l + 1, l + 1]
    rng.shuffle(L); return L
def skew_code(n):                                   # a comb of c leaves at depths 1..c, the other n - c in a subtree below depth c
    c = rng.randrange(1, min(n - 1, 13))
    rest = n - c
    while (1 << (20 - c)) < rest: c -= 1; rest += 1
    sub = rand_code(rest, 20 - c) if rest > 1 else None
    return list(range(1, c + 1)) + ([c + x for x in sub] if sub else [c])
bad = 0; stats = {}
for it in range(cases):
    k = rng.choice([1, 2, 3, 6, 17, 40, 120, 256])
    used = sorted(rng.sample(range(256), k))
    n = rng.choice([1, 2, 5, 60, 200, 450])
    t = bytearray()
    while len(t) < n:
        b = rng.choice(used)
        t += bytes([b]) * rng.choice([1, 1, 1, 2, 3, 4, 5, 9])
        if len(t) >= 4 and t[-1] == t[-2] == t[-3] == t[-4] and rng.random() < 0.8: t.append(rng.choice(used + [0, 1, 3, 255]) if False else rng.randrange(0, 6))
    t = bytes(b if b in used else used[0] for b in t)
    alpha = k + 2
    ntab = rng.randrange(2, 7)
    tabs = []
    for _ in range(ntab):
        r = rng.random()
        L = skew_code(alpha) if r < 0.25 and alpha > 3 else rand_code(alpha, rng.choice([20, 20, 12, 9]) if alpha > 2 ** 8 else 20)
        if len(L) != alpha or max(L) > 20: L = rand_code(alpha, 20)
        tabs.append(L)
    if rng.random() < 0.15:                          # a table that does not fill the code space, selected or not
        L = list(tabs[rng.randrange(ntab)]); i = rng.randrange(alpha); L[i] = min(20, L[i] + 1) if rng.random() < 0.6 else max(1, L[i] - 1)
        tabs[rng.randrange(ntab)] = L
    syms = C._encode_symbols(C._bwt(list(t))[0], used)
    ng = (len(syms) + 1 + 49) // 50
    sel = [rng.randrange(ntab) for _ in range(ng)]
    kw = {}
    if rng.random() < 0.1: kw["orig"] = rng.choice([len(t), len(t) + 1, 0, max(0, len(t) - 1)])
    try:
        z, data = C.block_stream(t, level=rng.choice([1, 5, 9]), used=used, tables=tabs, selectors=sel, **kw)
    except (KeyError, AssertionError) as e:         # (a symbol without a code in an oversubscribed table)
        stats['skipped'] = stats.get('skipped', 0) + 1
        continue
    if rng.random() < 0.3: z = z + C.block_stream(b"ABCDEF" * 3 + b"FED")[0]          # a second stream behind it
    refs = set()
    for extra in ([], ["-n", "1"], []):
        p = subprocess.run([STOCK, "-dc"] + extra, input=z, env=env, capture_output=True, timeout=300)
        refs.add((p.returncode, p.stderr.split(b"stdin: ")[-1].strip(), hashlib.md5(p.stdout).hexdigest() if p.returncode == 0 else None))
    p = subprocess.run([EMU, "-dc"], input=z, env=env, capture_output=True, timeout=300)
    got = (p.returncode, p.stderr.split(b"stdin: ")[-1].strip(), hashlib.md5(p.stdout).hexdigest() if p.returncode == 0 else None)
    stats[(got[0], got[1][-40:])] = stats.get((got[0], got[1][-40:]), 0) + 1
    if got not in refs:
        bad += 1
        open("/tmp/campaign_crafted_fail_%d.bz2" % it, "wb").write(z)
        print("case", it, "alphabet", k, "bytes", len(t), "tables", ntab, "maxlen", max(max(x) for x in tabs), kw, "\n  ref", sorted(refs), "\n  got", got, flush=True)
    if it % 25 == 24: print("..", it + 1, "cases,", bad, "differ", flush=True)
print("done:", cases, "cases,", bad, "differ"); print(stats)
