import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """The kernels compiled for the CPU emulator of the HIP execution model (test infrastructure)."""
    from emu.build_emu import build_emu
    from cosyvoice_amd._lib import Lib
    from guard import guard
    lib = Lib(build_emu(), allow_emulated=True)
    lib.tensor_hook = lambda t: guard(lib, t)          # every operand ends flush against an inaccessible page
    return lib


@pytest.fixture(scope="session")
def hip_lib():
    """The real gfx950 library; fails loudly if it was not built."""
    import torch
    from cosyvoice_amd._lib import get_lib
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return get_lib()


BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def lib(request):
    """Every kernel parity test runs twice: under the emulator (CPU, `-m "not gpu"`) and on the MI355X (`-m gpu`)."""
    return request.getfixturevalue(request.param + "_lib")


@pytest.fixture(autouse=True)
def _collect_dead_handles():
    """Destroy dead library handles BETWEEN tests (not from a finaliser firing in the middle of the next test's threads)."""
    yield
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass
