#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests
cat > /tmp/t5.py <<'PY'
import sys, time, bz2, hashlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, lbzip2_amd
import oracle_lib as L
lib = lbzip2_amd.library()
for kind, n in (("wiki", 100_000_000), ("wiki", 1_000_000_000), ("rand", 100_000_000), ("tar", 175_000_000)):
    data = L.gen_kind(kind, n, 2 if kind != "tar" else 5)
    src = torch.frombuffer(data, dtype=torch.uint8).cuda()
    dst = torch.empty(lib.bound(n), dtype=torch.uint8, device="cuda")
    M = 900000
    with lib.context(9, (n + M - 1) // M) as ctx:
        m = ctx.compress_device(src.data_ptr(), n, dst.data_ptr(), dst.numel())
    out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    with lib.decoder(1200) as d:
        for rep in range(2):
            t = time.time()
            k = d.decompress_device(dst.data_ptr(), m, out.data_ptr(), out.numel())
            torch.cuda.synchronize(); dt = time.time() - t
        s = d.stats()
    ok = k == n and bool(torch.equal(out[:n], src))
    print(kind, n, "->", m, "decoded ok:", ok, "%.1f MB/s (output bytes / wall)" % (n / dt / 1e6),
          "ms: scan %.1f huff %.1f sort %.1f walk %.1f emit %.1f" % (s.ms_scan, s.ms_huff, s.ms_sort, s.ms_walk, s.ms_emit), "blocks", s.nblocks, flush=True)
PY
timeout 300 python /tmp/t5.py 2>&1 | grep -v amdgpu | tail -6
