import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# the emulator runs a workgroup's lanes as fibers: the decoder's 1024-thread kernel is for tests/test_decode.py::test_wide_workgroups
if not os.path.exists("/dev/kfd"):
    os.environ.setdefault("LBZAMD_DWIDE", "0")


# the emulator (tests/emu) checks that the lanes of a wave enter every wave collective from the same source line -- lanes that
# do not have diverged, and what they exchange is not what a GPU wave would see -- and aborts when they do not
os.environ.setdefault("LBZ_EMU_CHECK_SITES", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")
