import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_LUTS = "/root/reference/LUTs/"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_first():
    """PyTorch carries its own copy of the HIP runtime: in one process it has to initialise BEFORE the engine's (/opt/rocm) copy
    touches the device, otherwise torch finds "No HIP GPUs" later (bench.py has the same order). Harmless without a GPU."""
    try:
        import torch
        if torch.cuda.device_count() > 0 and torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (builds oracle/liboracle.so on first use)."""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def synth():
    from raytracedicom_amd import luts
    return luts.synth_luts()


@pytest.fixture(scope="session")
def engine():
    """The HIP engine through its C ABI; fails loudly when the library or a GPU is missing."""
    from raytracedicom_amd import engine as eng
    return eng
