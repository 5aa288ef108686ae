import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("PF_LIB_VARIANT"):      # development: run the suite on a variant build of the library (tools/build_variant.sh)
        from pointfoot_amd import _abi
        _abi.lib_path = lambda: os.path.abspath(os.environ["PF_LIB_VARIANT"])


def _emu_requested():
    return os.environ.get("PF_TEST_EMU") == "1"


def pytest_report_header(config):
    if _emu_requested():
        return ["PF_TEST_EMU=1: the env tests run on the HOST EMULATION of the step program (tests/emu), not on the HIP library -- "
                "a CPU rehearsal; it is refused on a machine with a GPU"]


def pytest_runtest_setup(item):
    """PF_TEST_EMU=1 swaps the HIP mirror for the host emulation (tests/common.py).  That is a CPU rehearsal of the `-m gpu`
    env tests; on a GPU box it would turn the GPU column green from CPU code, so there a gpu-marked test FAILS instead."""
    if _emu_requested() and item.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            pytest.fail("PF_TEST_EMU=1 is set on a machine with a GPU: gpu-marked tests must run on the HIP library (unset it)", pytrace=False)
