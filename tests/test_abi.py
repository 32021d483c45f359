"""CPU: the C-ABI library loads and exports every symbol include/lbzip2_amd.h declares;
host-side logic that needs no device (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

import lbzip2_amd
from lbzip2_amd import _binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(lbzip2_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return lbzip2_amd.Library(lbzip2_amd.LIB_PATH)


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "lbzip2_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(lbzamd_\w+|encoder_alloc_size|encoder_init|collect|encode|transmit)\s*\(", hdr))
    names -= {"lbzamd_ctx", "lbzamd_stats", "lbzamd_block_info"}
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(lib.lib, n), n
    assert set(_binding.EXPORTS) <= names | {"lbzamd_last_error"}


def test_alloc_size_and_bound(lib):
    for k in range(1, 10):
        assert lib.lib.encoder_alloc_size(k * 100000) >= k * 100000 + 64
    assert lib.bound(0) >= 14
    assert lib.bound(10**9) > 10**9


def test_combine_crc_matches_macro():
    # encode.h:38 on uint32
    for cc, c in [(0, 0), (0xFFFFFFFF, 1), (0x80000001, 0x12345678), (0xDEADBEEF, 0xFFFFFFFF)]:
        ref = ((cc << 1) ^ (cc >> 31) ^ c ^ 0xFFFFFFFF) & 0xFFFFFFFF
        assert _binding.combine_crc(cc, c) == ref


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(lbzip2_amd.LbzError):
        lbzip2_amd.Library(str(tmp_path / "nope.so"))


def test_no_device_fails_loudly(lib):
    """Without a GPU the batch interface must refuse, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.lib.lbzamd_create(C.byref(h), -1, 9, 1, 1)
    assert rc != 0 and lib.error()


def test_kernels_contain_no_device_function_calls():
    """Every gfx950 code object of the product library is free of s_swappc: the sorter's out-of-line functions once made a
    build that sorted wrongly on the device only (k_bwt.hip, note at lds_radix_sort; csrc/check_no_calls.sh).  The same script
    refuses a 1024-thread kernel built for two workgroups a CU (<= 64 vector registers) with more than 80 scalar registers:
    k_mtf at 86 computed wrong ranks on the device (round 6)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([os.path.join(root, "lbzip2_amd", "csrc", "check_no_calls.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no calls" in r.stdout
