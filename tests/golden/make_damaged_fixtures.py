"""Generates tests/golden/damaged_cases.json: hand-made and randomly damaged .bz2 streams with what the COMPILED reference
(oracle/_ref/lbzip2_stock -dc) says about each -- exit status and diagnostic -- in the schema of expand_cases.json.

The reference's decompressor is a pipeline of threads and WHICH error it reports for a damaged stream can depend on which
thread gets there first (measured: a third of randomly damaged streams have two possible diagnostics).  Only streams for
which eight runs (default threads, -n 1, -n 3) agree are kept.  The hand-made ones (tests/craft_bz2.py: block_stream) each
hold ONE defect, so that the order in which the reference looks at things is pinned case by case:
  a block that ends where a run's count should stand (decode.c:1009), an origin pointer behind the block (:753), an empty
  block (:751), a block larger than the stream's level allows with a good and with a bad CRC (expand.c:725 looks at the size
  first), a bad block CRC alone, and a stream cut or overwritten at every 16-bit word of the end-of-stream marker
  (parse.c:152-262 takes headers a word at a time: a word that is not there is ERR_EOF, one that does not fit ERR_HEADER).
Run in the build container only (needs oracle/_ref)."""
import bz2
import hashlib
import json
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import craft_bz2 as C                                                 # noqa: E402
from golden_util import gen                                           # noqa: E402

STOCK = os.path.join(ROOT, "oracle", "_ref", "lbzip2_stock")


def reference_says(z):
    seen = set()
    for extra in ([], [], [], ["-n", "1"], ["-n", "1"], ["-n", "3"], ["-n", "3"], []):
        r = subprocess.run([STOCK, "-d", "-c"] + extra, input=z, capture_output=True, timeout=300)
        ok = r.returncode == 0
        seen.add((r.returncode, r.stderr.decode(errors="replace").strip(), len(r.stdout) if ok else 0, hashlib.md5(r.stdout).hexdigest() if ok else None))
    return seen


def patch_bits(z, bit, nbits, value):
    v = int.from_bytes(z, "big")
    total = len(z) * 8
    shift = total - bit - nbits
    v = (v & ~(((1 << nbits) - 1) << shift)) | (value << shift)
    return v.to_bytes(len(z), "big")


plain = b"ABCDEF" * 7 + b"CAFE" + b"FADE" * 3
made = []
made.append(("block-ends-where-a-count-is-due", C.block_stream(b"ABCDEF" * 7 + b"BAAAA")[0]))
made.append(("block-ends-where-a-count-is-due-2", C.block_stream(b"ABCDEF" * 3 + b"EEEE")[0]))
made.append(("origin-pointer-is-the-block-size", C.block_stream(plain, orig=len(plain))[0]))
made.append(("origin-pointer-far-behind-the-block", C.block_stream(plain, orig=0xFFFFFF)[0]))
made.append(("empty-block", C.block_stream(b"")[0]))
made.append(("bad-block-crc", C.block_stream(plain, stored_crc=0x12345678)[0]))
big = (bytes(range(65, 91)) * 6000)[:150000]
z2 = bz2.compress(big, 2)
assert z2[:4] == b"BZh2" and z2[4:10] == bytes.fromhex("314159265359")
made.append(("block-of-150000-in-a-level-1-stream", b"BZh1" + z2[4:]))
zz = b"BZh1" + z2[4:]
zz = patch_bits(zz, 32 + 48, 32, 0x0BADC0DE)                           # the block's CRC ...
# ... and the stream's (one block: the same value), wherever the end-of-stream marker stands
v = int.from_bytes(zz, "big"); total = len(zz) * 8
for bit in range(total - 48, 0, -1):
    if (v >> (total - bit - 48)) & ((1 << 48) - 1) == C.END_MAGIC:
        zz = patch_bits(zz, bit + 48, 32, 0x0BADC0DE)
        break
made.append(("block-of-150000-in-a-level-1-stream-and-a-bad-crc", zz))
for keep in (80, 72, 64, 48, 32, 16, 8):                                # bits of the 80-bit trailer that are kept
    made.append(("trailer-cut-%d-of-80-bits" % keep, C.block_stream(plain, cut_bits=-(80 - keep) if keep < 80 else None)[0]))
for keep, tail in ((0, b"\x17\x73\x45\x38\x50\x90\0\0\0\0"), (0, b"\x31\x41\x59\x26\x53\x58\0\0\0\0\0\0"), (16, b"\x45\x39\x50\x90\0\0\0\0"),
                   (32, b"\x50\x91\0\0\0\0"), (0, b"\xAB"), (0, b"\xAB\xCD"), (0, b"\x17\x72"), (0, b"\x17\x72\x45"), (0, b"\x31\x41\x59\x26\x53\x59\x00\x00")):
    made.append(("trailer-%d-bits-then-%s" % (keep, tail.hex()), C.block_stream(plain, cut_bits=-(80 - keep), tail=tail)[0]))

# files of 0..15 bytes: a header and what fits behind it (14 bytes are the shortest whole stream); the verdict is the
# parser's, a 16-bit word at a time over the zero-filled last 32-bit word
full = b"BZh9" + bytes.fromhex("177245385090") + b"\0\0\0\0"
seen_short = set()
for n in range(0, 16):
    for k, v in enumerate((full[:n], (full[:max(0, n - 1)] + b"\x31") if n else b"", (b"BZh9" + bytes.fromhex("314159265359") + b"\1\2\3\4\5\6")[:n],
                           (b"BZh9" + bytes.fromhex("177245385090") + b"\0\0\1\0")[:n], (full[:n] + b"\0" * (15 - n)) if n < 15 else full)):
        if v not in seen_short:
            seen_short.add(v)
            made.append(("short-%02d-bytes-%d" % (len(v), k), v))

# code tables and selectors (decode.c:226-235, :554-563, :640): a table whose lengths do not fill the code space exactly is an
# error only where a group SELECTS it -- python's bz2 takes an incomplete table that the reference refuses
inc, over, skew = [3] * 7 + [4], [3] * 7 + [2], [1, 2, 3, 4, 5, 6, 7, 7]
made.append(("tables-of-uneven-lengths", C.block_stream(plain, tables=[skew, [3] * 8])[0]))
made.append(("incomplete-table-that-no-group-selects", C.block_stream(plain, tables=[[3] * 8, inc])[0]))
made.append(("incomplete-table-selected", C.block_stream(plain, tables=[inc, [3] * 8])[0]))
made.append(("oversubscribed-table-that-no-group-selects", C.block_stream(plain, tables=[[3] * 8, over])[0]))
made.append(("oversubscribed-table-selected", C.block_stream(plain, tables=[over, [3] * 8])[0]))
made.append(("second-group-selects-an-incomplete-table", C.block_stream(b"ABCDEF" * 30 + b"FEDCBA" * 11, tables=[[3] * 8, inc], selectors=[0, 1, 0, 0, 0, 0][:(len(C._encode_symbols(C._bwt(list(b"ABCDEF" * 30 + b"FEDCBA" * 11))[0], [65, 66, 67, 68, 69, 70])) + 50) // 50])[0]))
made.append(("no-selectors", C.block_stream(plain, nsel=0)[0]))
made.append(("one-table", C.block_stream(plain, tables=[[3] * 8])[0]))
made.append(("seven-tables", C.block_stream(plain, tables=[[3] * 8] * 7)[0]))
made.append(("fewer-selectors-than-groups", C.block_stream(b"ABCDEF" * 30 + b"FEDCBA" * 11, nsel=1)[0]))

rng = random.Random(20260928)
srcs = [bytes(gen("wiki", 16000, 3)), bytes(gen("rand", 5000, 4)), bytes(gen("runs", 12000, 5)), b"ab" * 3000, bytes(gen("text", 23000, 6))]
streams = [bz2.compress(d, 1) for d in srcs] + [bz2.compress(srcs[0], 1) + bz2.compress(srcs[2], 1), bz2.compress(srcs[4], 1) + b"\0\0trailing"]
tried = 0
while tried < 70:
    z = streams[tried % len(streams)]
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
        p = rng.randrange(min(len(b), 60)); b[p] ^= 1 << rng.randrange(8)
    tried += 1
    made.append(("random-%03d-kind%d" % (tried, kind), bytes(b)))

cases, skipped, kept_random = [], [], 0
for name, z in made:
    if name.startswith("random-") and kept_random >= 16:
        continue
    seen = reference_says(z)
    if len(seen) == 1 and name.startswith("random-"):
        kept_random += 1
    also = []
    if len(seen) != 1:
        # a hand-made stream with ONE defect inside the block's tables or codes: the reference refuses it in every run, with the
        # block's own error or with the "bad block header magic" its parser finds where the block broke off -- both are kept
        msgs = sorted(s[1] for s in seen)
        if name.startswith("random-") or any(s[0] == 0 for s in seen) or len(seen) != 2 or not any("bad block header magic" in m for m in msgs):
            skipped.append((name, msgs))
            continue
        also = [m for m in msgs if "bad block header magic" in m]
        seen = {s for s in seen if s[1] not in also}
    rc, msg, out_len, out_md5 = next(iter(seen))
    cases.append({"case": hashlib.sha1(z).hexdigest(), "name": name, "bz2_hex": z.hex(), "ok": rc == 0, "ref_exit": rc, "ref_message": msg,
                  "also": also, "out_len": out_len, "out_md5": out_md5})
    print("%-58s %5d B  %d  %s" % (name, len(z), rc, msg[-60:]))
for s in skipped:
    print("skipped (the reference's runs disagree):", s)
json.dump({"source": "tests/golden/make_damaged_fixtures.py: hand-made and randomly damaged streams, outcomes from the compiled reference (lbzip2 -d -c, eight runs that agree)",
           "cases": cases}, open(os.path.join(ROOT, "tests", "golden", "damaged_cases.json"), "w"), indent=1)
print(len(cases), "cases,", len(skipped), "skipped;", os.path.getsize(os.path.join(ROOT, "tests", "golden", "damaged_cases.json")), "bytes")
