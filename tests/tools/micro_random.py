"""Ad-hoc microbenchmark: random 4-byte gathers/scatters on MI355X by working-set size (MALL = 256 MB)."""
import torch, time
dev = "cuda"
def bench(ws_mb, nacc=200_000_000, mode="gather", local=False):
    n = ws_mb * (1 << 20) // 4
    tab = torch.arange(n, dtype=torch.int32, device=dev)
    if local:
        # 256 regions; accesses of a contiguous chunk of the index array stay in one region (like one block per CU)
        reg = n // 256
        base = (torch.arange(nacc, device=dev) // (nacc // 256)).clamp_(max=255) * reg
        idx = base + torch.randint(0, reg, (nacc,), device=dev)
    else:
        idx = torch.randint(0, n, (nacc,), device=dev)
    out = torch.empty(nacc, dtype=torch.int32, device=dev)
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        if mode == "gather":
            torch.index_select(tab, 0, idx, out=out)
        else:
            tab.index_copy_(0, idx, out)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{mode:8s} ws={ws_mb:5d} MB local={local}: {nacc/dt/1e9:6.2f} G acc/s  ({dt*1e3:.1f} ms)", flush=True)
for ws in (32, 128, 230, 460, 920, 2048):
    bench(ws)
for ws in (230, 920):
    bench(ws, local=True)
for ws in (230, 920):
    bench(ws, mode="scatter")
