#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo:/root/repo/tests
cat > /tmp/t4.py <<'PY'
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import lbzip2_amd, oracle_lib as L
lbzip2_amd.LIB_PATH = "/root/repo/lbzip2_amd/csrc/variants/mtft.so"
lib = lbzip2_amd.library()
for kind in ("wiki", "text", "rand"):
    data = bytes(L.gen_kind(kind, 512 * 900000, 2))
    with lib.context(9, 512) as ctx:
        ctx.run_stages(data, 2)
        for b in (0, 2):
            bi = ctx.block_info(b); t = list(bi.ticks)
            print(kind, "blk", b, "nmtf/n %.2f" % (bi.nmtf / bi.n), "MTF ms: prelude %.2f ranks %.2f zrle %.2f" % (t[3]/1e5, t[4]/1e5, t[5]/1e5), flush=True)
PY
timeout 120 python /tmp/t4.py 2>&1 | grep blk
