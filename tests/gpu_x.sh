#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests
cat > /tmp/t3.py <<'PY'
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import lbzip2_amd, oracle_lib as L
lbzip2_amd.LIB_PATH = "/root/repo/lbzip2_amd/csrc/variants/ldst.so"
lib = lbzip2_amd.library()
for kind in ("wiki", "text"):
    data = bytes(L.gen_kind(kind, 512 * 900000, 2))
    with lib.context(9, 512) as ctx:
        ctx.run_stages(data, 1)
        for b in (0, 2, 4):
            bi = ctx.block_info(b); t = list(bi.ticks); f = list(bi.fticks)
            print(kind, "blk", b, "BATCH total %.2f: load %.2f scan %.2f waves %.2f; whole-workgroup sorts %.2f; oversized groups: %d groups, %d rows (%.1f%%), %.2f ms of which HBM sort %.2f" % (t[0]/1e5, t[3]/1e5, t[4]/1e5, t[5]/1e5, bi.rounds/1e5, t[1], t[6], 100.0*t[6]/bi.n, t[7]/1e5, t[2]/1e5),
                  "| FIX total %.2f: listbuild %.2f load %.2f runs %.2f sort %.2f write %.2f, %d tiles" % (f[7]/1e5, f[6]/1e5, f[2]/1e5, f[3]/1e5, f[4]/1e5, f[5]/1e5, f[0]), flush=True)
PY
timeout 120 python /tmp/t3.py 2>&1 | grep blk
