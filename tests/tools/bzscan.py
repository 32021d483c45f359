"""Ad-hoc: a .bz2 file symbol by symbol (headers, tables, symbols, decoded size of every block) -- to see where a damaged stream
stops being one.  usage: bzscan.py FILE"""
import sys
data = open(sys.argv[1], "rb").read()
bits = int.from_bytes(data, "big"); nbits = len(data) * 8
pos = 0
def rd(n):
    global pos
    if pos + n > nbits: raise EOFError
    v = (bits >> (nbits - pos - n)) & ((1 << n) - 1); pos += n; return v
assert rd(24) == 0x425a68; lvl = rd(8) - 48
print("level", lvl)
blk = 0
while True:
    start = pos
    magic = rd(48)
    if magic == 0x177245385090: print("EOS at bit", start, "crc %08x" % rd(32)); break
    if magic != 0x314159265359: print("bad magic at bit", start, hex(magic)); break
    crc = rd(32); rnd = rd(1); orig = rd(24)
    big = rd(16); used = []
    for i in range(16):
        if big >> (15 - i) & 1:
            sm = rd(16)
            used += [16 * i + j for j in range(16) if sm >> (15 - j) & 1]
    alpha = len(used) + 2
    ng = rd(3); ns = rd(15)
    sel = []; mtf = list(range(ng))
    ok = True
    for i in range(ns):
        j = 0
        while rd(1): j += 1
        if j >= ng: print("blk", blk, "bad selector"); ok = False; break
        t = mtf.pop(j); mtf.insert(0, t); sel.append(t)
    if not ok: break
    lens = []
    for t in range(ng):
        cur = rd(5); L = []
        for s in range(alpha):
            while True:
                if cur < 1 or cur > 20: print("blk", blk, "bad delta"); ok = False; break
                if not rd(1): break
                cur += 1 - 2 * rd(1)
            if not ok: break
            L.append(cur)
        if not ok: break
        lens.append(L)
    if not ok: break
    # canonical decode tables
    tabs = []
    for L in lens:
        code = 0; d = {}
        for l in range(1, 21):
            for s in range(alpha):
                if L[s] == l: d[(l, code)] = s; code += 1
            code <<= 1
        tabs.append(d)
    total = 0; run = 0; shift = 0; nsym = 0; first_over = None; maxdig = 0; dig = 0
    eob = False; err = None
    try:
        for g in range(ns):
            d = tabs[sel[g]]
            for k in range(50):
                c = 0; l = 0
                while True:
                    c = (c << 1) | rd(1); l += 1
                    if (l, c) in d: s = d[(l, c)]; break
                    if l >= 20: err = "prefix"; break
                if err: break
                nsym += 1
                if s == alpha - 1: eob = True; break
                if s < 2:
                    run += (s + 1) << shift; shift += 1; dig += 1; maxdig = max(maxdig, dig)
                else:
                    total += run + 1; run = 0; shift = 0; dig = 0
                if first_over is None and total + run > 900000: first_over = nsym
            if eob or err: break
    except EOFError:
        err = "eof"
    total += run
    print("blk", blk, "start bit", start, "orig", orig, "alpha", alpha, "ngroups", ng, "nsel", ns, "nsym", nsym, "eob", eob, "err", err, "total", total, "first>900000 at sym", first_over, "max digits", maxdig, "end bit", pos)
    blk += 1
    if not eob: break
