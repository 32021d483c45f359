"""Multi-GPU sharding of the block compressor: independent slabs, no data-path collective.

Every bzip2 block depends on one slab of bs100k*100000 input bytes only (reference
src/process.c:631, src/compress.c:73-118), so N ranks take N contiguous, slab-aligned byte
ranges and each writes a COMPLETE .bz2 stream of its range.  The concatenation of the rank
streams in rank order is a valid multi-stream .bz2 file of the whole input (what
`cat a.bz2 b.bz2` produces; lbzip2/bzip2 decode it).  torch.distributed is only needed to
learn the sizes (all_gather of one int64 per rank) or to collect the streams on one rank.
"""
from typing import List, Tuple


def shard_plan(nbytes: int, world: int, level: int = 9) -> List[Tuple[int, int]]:
    """(offset, length) per rank: contiguous ranges of whole slabs, sizes differing by <= 1 slab."""
    M = level * 100000
    nslabs = (nbytes + M - 1) // M
    out = []
    for r in range(world):
        s0 = nslabs * r // world
        s1 = nslabs * (r + 1) // world
        off = min(s0 * M, nbytes)
        end = min(s1 * M, nbytes)
        out.append((off, end - off))
    return out


def gather_sizes(local_size: int, dist=None) -> List[int]:
    """Sizes of all ranks' streams (all_gather of one int64; works with gloo and nccl/RCCL)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_size]
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([local_size], dtype=torch.int64, device=dev)
    allv = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allv, mine)
    return [int(t.item()) for t in allv]


def gather_streams(stream: bytes, dist=None, dst: int = 0):
    """Variable-size gather of the rank streams to rank dst (padded all_gather); returns the
    concatenated multi-stream .bz2 on dst, None elsewhere."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return stream
    import torch
    sizes = gather_sizes(len(stream), dist)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    pad = max(sizes)
    mine = torch.zeros(pad, dtype=torch.uint8, device=dev)
    mine[:len(stream)] = torch.frombuffer(bytearray(stream), dtype=torch.uint8).to(dev)
    allv = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allv, mine)
    if dist.get_rank() != dst:
        return None
    return b"".join(bytes(t[:n].cpu().numpy()) for t, n in zip(allv, sizes))
