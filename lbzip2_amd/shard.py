"""Multi-GPU sharding of the block compressor into ONE .bz2 stream.

Every bzip2 block depends on one slab of bs100k*100000 input bytes only (reference
src/process.c:631, src/compress.c:73-118) and every block is a whole number of bytes
(src/encode.c:514-525), so the reference's stream is

    "BZh" level | blocks in slab order | 0x177245385090 | combined CRC

(src/compress.c:238-250 reorder, :291-321 header/trailer).  With N ranks, rank r compresses the
contiguous slab range shard_plan()[r] into body-only bytes (lbzamd_compress_device_body: no header,
no trailer) and reports {bytes, nblocks, crc_fold}: the stream CRC fold cc' = rotl(cc,1) ^ ~crc
(src/encode.h:38) is linear over GF(2), so a range of m blocks acts as
cc -> rotl(cc, m mod 32) ^ fold_from_zero and 12 bytes per rank describe it.  The muxer (rank 0):

    all_gather of the 3-word partials          (torch.distributed; RCCL on GPUs, gloo on CPU)
    bodies -> rank 0 at prefix-sum offsets     (grouped send/recv: batch_isend_irecv = ncclSend/ncclRecv
                                                on device buffers over xGMI; one message per rank)
    header, trailer with the folded CRC        (rank 0)

The result is byte-identical to the single-GPU stream and to reference lbzip2's output for the
same input.  Volumes per 900 kB slab: ~0.25 MB back to rank 0 -- far below one xGMI link.
"""
from typing import List, Tuple

HEADER = b"BZh"
TRAILER_MAGIC = bytes([0x17, 0x72, 0x45, 0x38, 0x50, 0x90])


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


def fold_parts(cc: int, parts) -> int:
    """cc = rotl32(cc, nblocks mod 32) ^ crc_fold, range by range (encode.h:38 applied to whole ranges)."""
    for nblocks, fold in parts:
        r = nblocks & 31
        cc = ((((cc << r) | (cc >> (32 - r))) & 0xFFFFFFFF) if r else cc) ^ fold
    return cc


class StreamMux:
    """Gathers the ranks' body bytes into one stream on rank 0.  Buffers are torch tensors on
    `device` ("cuda": RCCL send/recv between device buffers; "cpu": gloo).  dist=None: one rank."""

    def __init__(self, dist, level: int, out_cap: int, device: str = "cuda"):
        import torch
        self.torch = torch
        self.dist = dist
        self.level = level
        self.device = device
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.out = torch.empty(out_cap + 16 if self.rank == 0 else 1, dtype=torch.uint8, device=device)
        self.meta = torch.zeros(3, dtype=torch.int64, device=device)
        self.allmeta = [torch.zeros(3, dtype=torch.int64, device=device) for _ in range(self.world)]
        self.head = torch.tensor(list(HEADER + bytes([0x30 + level])), dtype=torch.uint8, device=device)

    def gather(self, body, nbytes: int, nblocks: int, crc_fold: int) -> int:
        """body: uint8 tensor holding this rank's nbytes of blocks.  Returns the stream length on
        rank 0 (the stream is self.out[:length]), 0 elsewhere."""
        torch, dist = self.torch, self.dist
        if self.world == 1:
            parts = [(nbytes, nblocks, crc_fold)]
        else:
            self.meta[0], self.meta[1], self.meta[2] = nbytes, nblocks, crc_fold
            dist.all_gather(self.allmeta, self.meta)
            parts = [tuple(int(x) for x in t.tolist()) for t in self.allmeta]
        offs, o = [], 4
        for b, _, _ in parts:
            offs.append(o)
            o += b
        total = o + 10
        if self.rank == 0:
            assert total <= self.out.numel(), "stream does not fit the mux buffer"
            self.out[:4] = self.head
            self.out[4:4 + nbytes] = body[:nbytes]
            ops = [dist.P2POp(dist.irecv, self.out[offs[r]:offs[r] + parts[r][0]], r)
                   for r in range(1, self.world) if parts[r][0]]
        else:
            ops = [dist.P2POp(dist.isend, body[:nbytes], 0)] if nbytes else []
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if self.rank != 0:
            return 0
        cc = fold_parts(0, [(nb, f) for _, nb, f in parts])
        tail = TRAILER_MAGIC + cc.to_bytes(4, "big")
        self.out[o:o + 10] = torch.tensor(list(tail), dtype=torch.uint8, device=self.device)
        return total


def compress_sharded(lib, data: bytes, level: int = 9, dist=None, device: str = "cpu"):
    """Host-buffer convenience used by the CPU tests: every rank holds `data`, compresses its
    slab range through the C ABI (lbzamd_compress_host_body) and rank 0 returns the single stream."""
    import torch
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    off, ln = shard_plan(len(data), world, level)[rank]
    M = level * 100000
    with lib.context(level, max(1, (ln + M - 1) // M)) as ctx:
        body, nblocks, fold = ctx.compress_body(data[off:off + ln])
    mux = StreamMux(dist, level, lib.bound(len(data)), device)
    t = torch.frombuffer(bytearray(body), dtype=torch.uint8).to(device) if body else torch.empty(0, dtype=torch.uint8, device=device)
    n = mux.gather(t, len(body), nblocks, fold)
    return bytes(mux.out[:n].cpu().numpy()) if rank == 0 else None
