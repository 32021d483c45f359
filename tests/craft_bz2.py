"""Hand-made bzip2 streams whose block PAYLOAD contains a 48-bit block magic (0x314159265359) or end-of-stream
magic (0x177245385090) -- what a magic scan alone cannot tell from a real block boundary.  TEST INFRASTRUCTURE.

The block uses 6 distinct bytes, so its alphabet is RUNA, RUNB, five move-to-front positions and the end-of-block
symbol: 8 symbols, all given 3-bit codes (canonical: symbol k = code k, end of block = 111).  Any bit string cut
into 3-bit groups none of which is 111 is therefore a valid symbol sequence, and the symbols are chosen to spell
the magic.  The stream is a valid .bz2 file: Python's bz2 (libbzip2) decodes it; the tests compare with that.
"""
BLOCK_MAGIC = 0x314159265359
END_MAGIC = 0x177245385090


def _crc32_bz(data):
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b << 24
        for _ in range(8):
            crc = ((crc << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if crc & 0x80000000 else (crc << 1) & 0xFFFFFFFF
    return crc ^ 0xFFFFFFFF


def _spell(magic):
    """3-bit symbols (never 7) whose bits contain the 48-bit magic; returns the symbols."""
    bits = format(magic, "048b")
    for pre in ("", "0", "1", "00", "01", "10", "11"):          # bits of a symbol in front of the magic
        s = pre + bits
        s += "0" * (-len(s) % 3)
        groups = [int(s[i:i + 3], 2) for i in range(0, len(s), 3)]
        if 7 not in groups:
            return groups
    raise AssertionError("magic cannot be spelled")


def _decode_symbols(syms, used):
    """MTF + zero-run decoding of the symbol list (without the end-of-block symbol) -> BWT bytes."""
    order = list(used)
    out = []
    run, weight = 0, 1
    for s in syms + [None]:
        if s in (0, 1):
            run += weight << s
            weight <<= 1
            continue
        if run:
            out += [order[0]] * run
            run, weight = 0, 1
        if s is None:
            break
        b = order.pop(s - 1)
        order.insert(0, b)
        out.append(b)
    return out


def _inverse_bwt(L, orig):
    n = len(L)
    order = sorted(range(n), key=lambda i: (L[i], i))            # stable: the k-th occurrence maps to the k-th
    out, p = [], order[orig]
    for _ in range(n):
        out.append(L[p])
        p = order[p]
    return out


def _inverse_rle1(b):
    out, i = [], 0
    while i < len(b):
        c, run = b[i], 1
        out.append(c)
        i += 1
        while i < len(b) and run < 4 and b[i] == c:
            out.append(c); run += 1; i += 1
        if run == 4 and i < len(b):
            out += [c] * b[i]
            i += 1
    return bytes(out)


def crafted_stream(magic=BLOCK_MAGIC, level=9, filler=60):
    """A complete one-block .bz2 stream whose payload contains `magic`; returns (stream, decoded bytes)."""
    used = [65, 66, 67, 68, 69, 70]
    # the walk over the move-to-front list must stay meaningful: a prelude that touches every byte, the magic, a tail
    body = [2, 3, 4, 5, 6, 2, 3] * 3 + [0, 1, 4] * (filler // 3)
    syms = body + _spell(magic) + [2, 6, 3, 0, 5]
    L = _decode_symbols(syms, used)
    n = len(L)
    orig = n // 3
    data = _inverse_rle1(_inverse_bwt(L, orig))
    crc = _crc32_bz(data)

    bits = []

    def put(nb, v):
        bits.append(format(v, "0%db" % nb))

    put(48, BLOCK_MAGIC); put(32, crc); put(1, 0); put(24, orig)
    put(16, 1 << (15 - 4))                                         # bytes 64..79 in use
    put(16, sum(1 << (15 - (u - 64)) for u in used))
    all_syms = syms + [7]
    nsel = (len(all_syms) + 49) // 50
    put(3, 2); put(15, nsel)
    for _ in range(nsel):
        put(1, 0)                                                  # every group uses table 0
    for _ in range(2):                                             # two identical tables: every length 3
        put(5, 3)
        for _ in range(8):
            put(1, 0)
    for s in all_syms:
        put(3, s)
    put(48, END_MAGIC); put(32, crc)                               # one block: combined CRC = rotl(0,1) ^ crc
    s = "".join(bits)
    s += "0" * (-len(s) % 8)
    stream = b"BZh" + bytes([48 + level]) + int(s, 2).to_bytes(len(s) // 8, "big")
    return stream, data


def _bwt(t):
    """(last column, row of rotation 0) of the sorted rotations of t -- naive, for blocks of a few hundred bytes."""
    n = len(t)
    rows = sorted(range(n), key=lambda i: (t[i:] + t[:i], i))
    return [t[(i - 1) % n] for i in rows], rows.index(0)


def _encode_symbols(L, used):
    """move-to-front + zero runs (RUNA / RUNB, bijective base 2) of the BWT string -> symbols without the end-of-block one"""
    order = list(used)
    syms, run = [], 0

    def flush():
        nonlocal run
        while run:
            run -= 1
            syms.append(run & 1)
            run >>= 1

    for b in L:
        j = order.index(b)
        if j == 0:
            run += 1
            continue
        flush()
        syms.append(j + 1)
        order.insert(0, order.pop(j))
    flush()
    return syms


def _canonical(lens):
    """bzip2's code assignment: lengths ascending, symbols ascending inside a length -> {symbol: (length, code)}"""
    code, out = 0, {}
    for l in range(1, 21):
        for s_, ls in enumerate(lens):
            if ls == l:
                out[s_] = (l, code)
                code += 1
        code <<= 1
    return out


def block_stream(rle1, level=9, orig=None, stored_crc=None, cut_bits=None, tail=b"", tables=None, selectors=None, nsel=None, used=None):
    """A one-block .bz2 stream whose block -- the bytes BEHIND the initial run-length coding -- is `rle1` (bytes out of
    A..F, all six of them in use: eight symbols with 3-bit codes, as crafted_stream).  The block need not be what an
    encoder would write: it may end where a run's count should stand, `orig` may point behind it, `stored_crc` may be wrong.
    cut_bits: the stream is cut after that many bits (then padded to a byte); tail: bytes appended.
    Returns (stream, bytes a decoder that accepts the block writes, or None if the block ends inside a run's count)."""
    default_alphabet = used is None
    used = [65, 66, 67, 68, 69, 70] if used is None else sorted(used)   # used: the bytes of the block's map (with `tables` of len(used) + 2 lengths each)
    t = list(rle1)
    assert set(t) <= set(used) and (not default_alphabet or len(set(t)) == 6 or not t)
    eob = len(used) + 1
    L, o = _bwt(t) if t else ([], 0)
    syms = _encode_symbols(L, used)
    assert _decode_symbols(syms, used) == L
    state, prev = 0, None                                          # the run-length state behind the last byte (4: a count is due)
    for b in t:
        if state == 4:
            state, prev = 0, None
            continue
        state = state + 1 if b == prev else 1
        prev = b
    ends_in_count = state == 4
    data = None if ends_in_count else _inverse_rle1(t)
    crc = _crc32_bz(data) if data is not None else 0
    if stored_crc is not None:
        crc = stored_crc
    bits = []

    def put(nb, v):
        bits.append(format(v, "0%db" % nb))

    put(48, BLOCK_MAGIC); put(32, crc); put(1, 0); put(24, o if orig is None else orig)
    put(16, sum(1 << (15 - i) for i in range(16) if any(u >> 4 == i for u in used)))
    for i in range(16):
        if any(u >> 4 == i for u in used):
            put(16, sum(1 << (15 - (u & 15)) for u in used if u >> 4 == i))
    all_syms = syms + [eob]
    ngroups = (len(all_syms) + 49) // 50
    tables = tables or [[3] * 8, [3] * 8]                          # tables: code lengths of the len(used) + 2 symbols, 2..6 tables
    selectors = selectors or [0] * ngroups                         # table of each group of 50 symbols
    put(3, len(tables)); put(15, ngroups if nsel is None else nsel)
    order = list(range(len(tables)))
    for t in (selectors if nsel is None else selectors[:nsel]):    # move-to-front, unary
        j = order.index(t)
        put(j + 1, (1 << (j + 1)) - 2)
        order.insert(0, order.pop(j))
    for lens in tables:
        put(5, lens[0])
        cur = lens[0]
        for l in lens:
            while cur != l:
                put(2, 2 if l > cur else 3)
                cur += 1 if l > cur else -1
            put(1, 0)
    for i, s_ in enumerate(all_syms):
        l, code = _canonical(tables[selectors[i // 50]])[s_]
        put(l, code)
    put(48, END_MAGIC); put(32, crc)
    s = "".join(bits)
    if cut_bits is not None:
        s = s[:max(0, len(s) + cut_bits)] if cut_bits < 0 else s[:cut_bits]          # negative: that many bits off the end
    s += "0" * (-len(s) % 8)
    stream = b"BZh" + bytes([48 + level]) + (int(s, 2).to_bytes(len(s) // 8, "big") if s else b"") + tail
    return stream, data


def random_block_streams(seed, count):
    """VALID one-block streams no encoder would write: alphabets of 1..256 bytes, 2..6 code tables with lengths up to 20 (random
    trees and combs), a random table for every group of 50 symbols, blocks of 1..450 bytes with runs of every length.
    Yields (stream, the bytes it decodes to)."""
    import random
    rng = random.Random(seed)

    def rand_code(n, maxlen):
        L = [1, 1]
        while len(L) < n:
            i = rng.randrange(len(L))
            if L[i] >= maxlen:
                continue
            l = L.pop(i)
            L += [l + 1, l + 1]
        rng.shuffle(L)
        return L

    def comb_code(n):                                # c leaves at depths 1..c, the other n - c in a subtree below depth c
        c = rng.randrange(1, min(n - 1, 13))
        rest = n - c
        while (1 << (20 - c)) < rest:
            c -= 1
            rest += 1
        sub = rand_code(rest, 20 - c) if rest > 1 else None
        L = list(range(1, c + 1)) + ([c + x for x in sub] if sub else [c])
        rng.shuffle(L)
        return L

    made = 0
    while made < count:
        k = rng.choice([1, 2, 3, 6, 17, 40, 120, 256])
        used = sorted(rng.sample(range(256), k))
        n = rng.choice([1, 2, 5, 60, 200, 450])
        t = bytearray()
        while len(t) < n:
            t += bytes([rng.choice(used)]) * rng.choice([1, 1, 1, 2, 3, 4, 5, 9])
            if len(t) >= 4 and t[-1] == t[-2] == t[-3] == t[-4]:
                t.append(used[rng.randrange(min(len(used), 6))])       # a count that is one of the block's bytes
        t = bytes(t)
        alpha = k + 2
        tabs = [comb_code(alpha) if alpha > 3 and rng.random() < 0.3 else rand_code(alpha, 20) for _ in range(rng.randrange(2, 7))]
        nsyms = len(_encode_symbols(_bwt(list(t))[0], used)) + 1
        sel = [rng.randrange(len(tabs)) for _ in range((nsyms + 49) // 50)]
        z, data = block_stream(t, level=rng.choice([1, 5, 9]), used=used, tables=tabs, selectors=sel)
        if data is None:                             # the block happens to end where a count is due
            continue
        if rng.random() < 0.3:
            z2, d2 = block_stream(b"ABCDEF" * 3 + b"FED")
            z, data = z + z2, data + d2
        made += 1
        yield z, data
