"""Ad-hoc microbenchmark (round 4): random gathers / scatters on MI355X by working-set size and element width.
What decides whether the deep-tie rounds should run over waves of few blocks (rank arrays resident in the
256 MiB Infinity Cache / the XCD's 4 MiB L2) or over all blocks of a round at once."""
import torch, time, json, sys
dev = "cuda"
res = []
def bench(ws_mb, dtype, nacc=100_000_000, mode="gather"):
    esz = 4 if dtype == torch.int32 else 8
    n = int(ws_mb * (1 << 20)) // esz
    tab = torch.arange(n, dtype=dtype, device=dev)
    idx = torch.randint(0, n, (nacc,), device=dev)
    out = torch.empty(nacc, dtype=dtype, device=dev)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        if mode == "gather":
            torch.index_select(tab, 0, idx, out=out)
        else:
            tab.index_copy_(0, idx, out)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    r = {"mode": mode, "bytes": esz, "ws_mb": ws_mb, "G_acc_s": round(nacc / best / 1e9, 2)}
    res.append(r); print(r, flush=True)
for mode in ("gather", "scatter"):
    for dt in (torch.int32, torch.int64):
        for ws in (2, 8, 24, 64, 128, 200, 400, 1000, 2700):
            bench(ws, dt, mode=mode)
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/micro_random2.json", "w"), indent=1)
