"""lbzip2_amd -- MI355X-native bzip2 block-compression core.

The per-block hot path of lbzip2 (src/encode.c + src/divbwt.c: RLE1+CRC, Burrows-Wheeler
block sort, MTF+zero-run coding, prefix-code selection, bit packing) as hand-written HIP
kernels for gfx950, behind the reference's own work-unit interface (src/encode.h:22-38).

    import lbzip2_amd
    bz = lbzip2_amd.compress(data, level=9)      # bit-exact .bz2 of reference lbzip2

The package is only a thin ctypes mirror of the C ABI in include/lbzip2_amd.h; it loads
lbzip2_amd/csrc/liblbzamd.so (built by hipcc for gfx950) and has no CPU implementation.
"""
import os

from ._binding import (CLUSTER_FACTOR, HEADER_SIZE, TRAILER_SIZE, STAGE_BWT, STAGE_MTFV, STAGE_OUT,
                       STAGE_RLE, BlockInfo, Context, Decoder, DStats, Encoder, EXPORTS, LbzError, Library, Part, Stats,
                       combine_crc, fold_parts)

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "liblbzamd.so")
_lib = None


def library() -> Library:
    """The product library (HIP, gfx950). Raises LbzError if it has not been built."""
    global _lib
    if _lib is None:
        try:
            # torch ships its own libamdhip64; load it first so the process holds ONE HIP runtime
            # (loading ROCm's copy first leaves torch with "No HIP GPUs are available").
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = Library(LIB_PATH)
    return _lib


def compress(data: bytes, level: int = 9) -> bytes:
    return library().compress(data, level)


def decompress(data: bytes) -> bytes:
    return library().decompress(data)


def encoder_alloc_size(mbs: int) -> int:
    return library().lib.encoder_alloc_size(mbs)


def context(level=9, max_slabs=64, nslots=0, device=-1) -> Context:
    return library().context(level, max_slabs, nslots, device)


def encoder(max_block_size: int) -> Encoder:
    return library().encoder(max_block_size)
