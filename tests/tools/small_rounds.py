"""Ad-hoc: device-resident time of inputs of a few slabs (one round each): wiki of 16 / 64 / 112 / 139 / 200 slabs, with the
library's tuning knobs from the environment.  usage: small_rounds.py "K=V,K=V;..." [kind]"""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/tools")
import torch, lbzip2_amd, inputs
if os.environ.get("LBZ_LIB"): lbzip2_amd.LIB_PATH = os.environ["LBZ_LIB"]
lib = lbzip2_amd.library()
settings = sys.argv[1].split(";") if len(sys.argv) > 1 else [""]
kind = sys.argv[2] if len(sys.argv) > 2 else "wiki"
big = inputs.get(kind, 200 * 900000, 1)
for slabs in (16, 64, 112, 139, 200):
    n = slabs * 900000 if slabs != 112 else 100_000_000
    src = torch.frombuffer(big[:n], dtype=torch.uint8).cuda()
    dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    for st in settings:
        env = dict(kv.split("=") for kv in st.split(",") if kv)
        for k, v in env.items(): os.environ[k] = v
        with lib.context(9, (n + 899999) // 900000, 0) as ctx:
            best = None
            for it in range(6):
                torch.cuda.synchronize(); t = time.perf_counter()
                m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
                torch.cuda.synchronize(); dt = time.perf_counter() - t
                if it: best = dt if best is None or dt < best else best
            s = ctx.stats()
        print(f"{kind} {slabs:4d} slabs {st:28s} {best*1e3:7.2f} ms = {n/best/1e6:7.1f} MB/s  collect={s.ms_collect:.2f} part={s.ms_bwt_part:.2f} batch={s.ms_bwt_batch:.2f} ties={s.ms_bwt_fix:.2f} mtf={s.ms_mtf:.2f} enc={s.ms_encode:.2f}", flush=True)
        for k in env: del os.environ[k]
