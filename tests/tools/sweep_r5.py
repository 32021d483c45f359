"""Ad-hoc (round 5): one process, the inputs made once, a list of environment settings of the library's tuning knobs each
timed on every input.  usage: sweep_r5.py slabs kinds "K=V,K=V;K=V;..."   (LBZAMD_STREAMS / LBZAMD_SLOTS are read at context creation)"""
import os, sys, time, hashlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/tools")
import torch, lbzip2_amd, inputs
if os.environ.get("LBZ_LIB"): lbzip2_amd.LIB_PATH = os.environ["LBZ_LIB"]
lib = lbzip2_amd.library()
slabs = int(sys.argv[1]); kinds = sys.argv[2].split(","); settings = sys.argv[3].split(";")
n = slabs * 900000
ref = {}
for kind in kinds:
    data = inputs.get(kind, n, 2)
    src = torch.frombuffer(data, dtype=torch.uint8).cuda()
    dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    for st in settings:
        env = dict(kv.split("=") for kv in st.split(",") if kv)
        for k, v in env.items(): os.environ[k] = v
        ctx = lib.context(9, slabs, int(os.environ.get("LBZ_SLOTS", "371")))
        best = None
        for it in range(3):
            t = time.time()
            m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
            dt = time.time() - t
            if it: best = dt if best is None or dt < best else best
        s = ctx.stats()
        h = hashlib.md5(dst[:m].cpu().numpy().tobytes()).hexdigest()
        ref.setdefault(kind, h)
        print(f"{kind:8s} {st:60s} {n/best/1e6:8.1f} MB/s  part={s.ms_bwt_part:.1f} batch={s.ms_bwt_batch:.1f} ties={s.ms_bwt_fix:.1f} mtf={s.ms_mtf:.1f} enc={s.ms_encode:.1f}  {'same stream' if h == ref[kind] else 'STREAM DIFFERS'} {h[:8]}", flush=True)
        ctx.close()
        for k in env: del os.environ[k]
    del src, dst
    torch.cuda.empty_cache()
