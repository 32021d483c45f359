"""CPU, world_size 2 over gloo: the N>1 path of bench.py / lbzip2_amd.shard -- slab-aligned
shards, one complete stream per rank, sizes by all_gather, concatenation = valid multi-stream
.bz2 of the whole input.  The per-rank compressor here is the emulated kernel build (there is no
GPU in this container); on the GPU box the same code runs over RCCL."""
import bz2
import os
import subprocess
import sys

import pytest

from lbzip2_amd.shard import shard_plan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, bz2
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
from lbzip2_amd._binding import Library
from lbzip2_amd.shard import shard_plan, gather_sizes, gather_streams
import oracle_lib as L
from golden_util import gen
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
data = gen("text", 330000, 17) + gen("runs", 120000, 18)
off, n = shard_plan(len(data), world, 1)[rank]
lib = Library(os.path.join({root!r}, "tests", "emu", "_build", "liblbzamd_emu_1024.so"))
mine = lib.compress(data[off:off + n], 1)
assert mine == L.orc_compress(data[off:off + n], 1)
sizes = gather_sizes(len(mine), dist)
assert sizes[rank] == len(mine) and len(sizes) == world
whole = gather_streams(mine, dist, 0)
if rank == 0:
    assert len(whole) == sum(sizes)
    assert bz2.decompress(whole) == data
    print("SHARDS_OK", sizes)
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


def test_two_ranks_gloo(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "WG=1024"])
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", LBZ_EMU_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "SHARDS_OK" in out.stdout
