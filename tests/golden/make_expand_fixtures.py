"""Generates tests/golden/expand_cases.json from the reference's own decompressor test data
(/root/reference/tests/suite/manual-expand/*.bz2, described in tests/README of the reference) and the COMPILED
reference (oracle/_ref/lbzip2_stock -d): per case the stream itself (they are 0..4140 bytes), whether the reference
accepts it, and length + md5 of what it writes.  Also lbzip2_amd/csrc/lbz_rand.h: the format's randomisation table
read out of libbz2 (BZ2_rNums).  Run in the build container only (needs /root/reference and oracle/_ref)."""
import ctypes
import ctypes.util
import glob
import hashlib
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STOCK = os.path.join(ROOT, "oracle", "_ref", "lbzip2_stock")
NAMES = {}                                     # sha1-named suite file -> the name tests/README describes it under
for f in glob.glob("/root/reference/tests/*.bz2"):
    NAMES[hashlib.sha1(open(f, "rb").read()).hexdigest()] = os.path.basename(f)[:-4]

cases = []
for f in sorted(glob.glob("/root/reference/tests/suite/manual-expand/*.bz2")):
    z = open(f, "rb").read()
    r = subprocess.run([STOCK, "-d", "-c"], input=z, capture_output=True, timeout=300)
    sha = os.path.basename(f)[:-4]
    cases.append({"case": sha, "name": NAMES.get(hashlib.sha1(z).hexdigest(), "?"), "bz2_hex": z.hex(),
                  "ok": r.returncode == 0, "ref_exit": r.returncode, "ref_message": r.stderr.decode(errors="replace").strip(),
                  "out_len": len(r.stdout) if r.returncode == 0 else 0,
                  "out_md5": hashlib.md5(r.stdout).hexdigest() if r.returncode == 0 else None})
    print(cases[-1]["name"], cases[-1]["ok"], cases[-1]["out_len"], cases[-1]["ref_message"][:60])
json.dump({"source": "reference tests/suite/manual-expand, outputs from the compiled reference (lbzip2 -d -c)", "cases": cases},
          open(os.path.join(ROOT, "tests", "golden", "expand_cases.json"), "w"), indent=1)

lib = ctypes.CDLL(ctypes.util.find_library("bz2"))
tab = list((ctypes.c_int * 512).in_dll(lib, "BZ2_rNums"))
with open(os.path.join(ROOT, "lbzip2_amd", "csrc", "lbz_rand.h"), "w") as h:
    h.write("/* The bzip2 format's randomisation table (512 step lengths; bzip2's randtable.c), read out of libbz2's exported\n"
            " * BZ2_rNums by tests/golden/make_expand_fixtures.py and checked against it by tests/test_decode.py.  Byte k of a\n"
            " * randomised block's inverse-BWT output is flipped (xor 1) iff k + 2 is a partial sum of the table taken cyclically. */\n"
            "#ifndef LBZ_RAND_H\n#define LBZ_RAND_H\n#if defined(__HIPCC__) && !defined(LBZ_EMULATED)\n__device__\n#endif\nstatic const unsigned short LBZ_RNUMS[512] = {\n")
    for i in range(0, 512, 16):
        h.write("  " + ", ".join(str(v) for v in tab[i:i + 16]) + ",\n")
    h.write("};\n#endif\n")
