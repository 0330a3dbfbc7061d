import os
import sys

import pytest

# The test process creates dozens of spectral solvers of different shapes one after another — exactly what trips the rocFFT plan-cache defect
# (csrc/ins_fftcheck.hip).  The library no longer resets rocFFT's process-wide state on its own; the test host opts in (a product host decides
# for itself).
os.environ.setdefault("INS_FFT_ALLOW_RESET", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    from oracle import ins_oracle

    return ins_oracle


@pytest.fixture(autouse=True)
def _collect_between_tests():
    """Destroy solver handles promptly: rocFFT (ROCm 7.2) corrupts a real plan created while another live plan
    has the same nx*ny but a different shape (tools/dbg_hipfft2.cpp); libinship detects that at plan creation
    (csrc/ins_fftcheck.hip) and raises, so tests must not leave solvers of earlier tests alive."""
    yield
    import gc

    gc.collect()
