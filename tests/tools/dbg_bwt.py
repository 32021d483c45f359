"""Ad-hoc: the BWT stage of one wiki slab on the GPU against the oracle, per library variant: where the rows differ."""
import os, sys, subprocess
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/tools")
if len(sys.argv) > 1:
    lib_path, kind, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    os.environ["LBZAMD_STREAMS"] = "1"
    import numpy as np, lbzip2_amd, inputs, oracle_lib as L
    if lib_path != "default": lbzip2_amd.LIB_PATH = lib_path
    lib = lbzip2_amd.library()
    data = bytes(inputs.get(kind, n, 2))
    ob = L.orc_blocks(data, 9)
    for rep in range(2):
        with lib.context(9, max(1, (n + 899999) // 900000), 8) as ctx:
            gb = ctx.blocks(data, 1)
        for g, o in zip(gb, ob):
            a = np.frombuffer(g["bwt"], dtype=np.uint8); b = np.frombuffer(o["bwt"], dtype=np.uint8)
            d = np.nonzero(a != b)[0]
            print(f"rep {rep} blk {g['blk']} n {g['nblock']} mismatches {len(d)} idx {g['bwt_idx']} vs {o['bwt_idx']}", end=" ")
            if len(d):
                runs = np.split(d, np.nonzero(np.diff(d) > 2000)[0] + 1)
                print("ranges:", [(int(r[0]), int(r[-1]), len(r)) for r in runs[:12]], "zeros in gpu bwt", int((a == 0).sum()), flush=True)
            else:
                print(flush=True)
    sys.exit(0)
import resource
resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
V = "/root/repo/lbzip2_amd/csrc/variants/"
for lib, segs in [(V + x, "0") for x in sorted(os.listdir(V)) if x.endswith(".so")]:
    r = subprocess.run([sys.executable, __file__, lib, "wiki", "900000"], capture_output=True, text=True, timeout=200, env=dict(os.environ, LBZAMD_SEGS=segs))
    err = [l for l in r.stderr.splitlines() if "fault" in l.lower()]
    print("==", os.path.basename(lib), "segs", segs, "rc", r.returncode, err[:1])
    print(r.stdout[-3000:], flush=True)
