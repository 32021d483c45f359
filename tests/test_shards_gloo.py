"""CPU, world_size 2 and 3 over gloo: the N>1 path of bench.py --scaling strong / lbzip2_amd.shard --
slab ranges of ONE input dealt over the ranks, body-only bytes + 12-byte CRC partials gathered on
rank 0 (grouped send/recv) into ONE stream that must be byte-identical to the single-rank stream,
to the oracle's and to the reference fixture.  The per-rank compressor here is the emulated kernel
build (there is no GPU in this container); on the GPU box the same code runs over RCCL."""
import os
import subprocess
import sys

import pytest

from lbzip2_amd.shard import fold_parts, shard_plan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, bz2, hashlib
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
from lbzip2_amd._binding import Library
from lbzip2_amd.shard import compress_sharded
import oracle_lib as L
from golden_util import gen, bench_fixtures
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = Library(os.path.join({root!r}, "tests", "emu", "_build", "liblbzamd_emu_1024.so"))
# (1) a slab with a spill block (RLE1 expansion) in the middle of the range boundaries, level -1
data = bytes(gen("text", 330000, 17) + gen("runs", 120000, 18) + gen("wiki", 95000, 3))
whole = compress_sharded(lib, data, 1, dist, "cpu")
if rank == 0:
    assert whole == lib.compress(data, 1), "gathered stream differs from the single-rank stream"
    assert whole == L.orc_compress(data, 1)
    assert bz2.decompress(whole) == data
# (2) the reference fixture wiki(350000, seed 2) at -1: md5 of the reference's own stream
rec = [r for r in bench_fixtures(max_n=400000) if r["kind"] == "wiki" and r["level"] == 1][0]
d2 = bytes(gen(rec["kind"], rec["n"], rec["seed"]))
w2 = compress_sharded(lib, d2, rec["level"], dist, "cpu")
if rank == 0:
    assert len(w2) == rec["out_len"] and hashlib.md5(w2).hexdigest() == rec["ref_md5"]
# (3) fewer slabs than ranks, and the empty input
for d3 in (bytes(gen("text", 70000, 5)), b""):
    w3 = compress_sharded(lib, d3, 1, dist, "cpu")
    if rank == 0:
        assert w3 == L.orc_compress(d3, 1)
if rank == 0:
    print("SHARDS_OK", world, len(whole))
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_plan_covers_input():
    for n in (0, 1, 899999, 900000, 900001, 10**9, 10**10 + 7):
        for world in (1, 2, 3, 8):
            plan = shard_plan(n, world, 9)
            assert sum(l for _, l in plan) == n
            pos = 0
            for off, l in plan:
                assert off == pos and off % 900000 == 0 or l == 0
                pos += l


def test_fold_parts_is_the_block_fold():
    """rotl(cc, m mod 32) ^ fold_from_zero == folding the m blocks one by one (encode.h:38)."""
    import random
    from lbzip2_amd import combine_crc
    rng = random.Random(3)
    crcs = [rng.getrandbits(32) for _ in range(200)]
    whole = 0
    for c in crcs:
        whole = combine_crc(whole, c)
    for cuts in ([0, 200], [0, 1, 200], [0, 31, 32, 33, 64, 200], [0, 0, 7, 7, 200]):
        parts = []
        for a, b in zip(cuts, cuts[1:]):
            f = 0
            for c in crcs[a:b]:
                f = combine_crc(f, c)
            parts.append((b - a, f))
        assert fold_parts(0, parts) == whole


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_gloo_single_stream(tmp_path, world):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "WG=1024"])
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", LBZ_EMU_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(29517 + world), str(script)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "SHARDS_OK" in out.stdout
