import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) is dominated by kernel tests under the HIP-execution-model emulator, single-threaded per test: run the test
    files on 6 pytest-xdist workers (13.5 -> 5.5 min on the 8-core build container with the round-2 test set).  GPU runs (`-m gpu`) stay in ONE process: one model at a time on
    the device, and the process that loads libcosyvoice_amd.so is the pytest process itself.  `CV_TEST_SERIAL=1` or an explicit `-n` overrides."""
    try:
        import xdist  # noqa: F401
    except ImportError:
        return None
    if (config.getoption("markexpr", "") or "").strip() == "not gpu" and getattr(config.option, "numprocesses", None) is None \
            and not os.environ.get("CV_TEST_SERIAL") and not os.environ.get("PYTEST_XDIST_WORKER"):
        config.option.numprocesses = max(2, min(8, os.cpu_count() or 6))      # round 6: one worker per core (the suite is ~6500 core-seconds of single-threaded emulator work)
        config.option.dist = "loadfile"
        # xdist would re-sort the files by their NUMBER of tests (--loadscope-reorder, on by default), which puts the heavy single-test files (a whole model under the
        # emulator each) last; pytest_collection_modifyitems below orders the files by measured weight instead
        if hasattr(config.option, "loadscopereorder"):
            config.option.loadscopereorder = False
    return None


# CPU suite under pytest-xdist (--dist loadfile hands whole files to idle workers in COLLECTION order): heaviest files first, so that the long ones do not
# start last and leave five workers idle behind them (measured file totals under the emulator, seconds: round-4 `--durations=30` run).
_HEAVY_FIRST = ["test_fullsize_pinned_model.py", "test_flow_big.py", "test_fullsize_pinned.py", "test_flow.py", "test_dropin_reference.py", "test_model_cv3.py", "test_llm_ras.py", "test_model_batch_padded.py", "test_model_cv3_filter.py", "test_causal_hift.py",
                "test_zz_llm_batch.py", "test_dropin_reference_cv1.py", "test_bench_cv1_dryrun.py", "test_zzz_cosyvoice1_hip_model.py", "test_model_load.py", "test_model.py", "test_model_batch.py",
                "test_model_cv3_batch.py", "test_dit.py", "test_zzz_cosyvoice1_hip.py", "test_zzz_cosyvoice1_hip_r6.py", "test_zzz_cosyvoice1_hip_hift.py", "test_zz_fullsize.py", "test_hift.py", "test_llm.py"]


def pytest_collection_modifyitems(config, items):
    # `experiments`: a test (or one of its parameter sets) of a measured no-go variant that only exists in a library built with CV_BUILD_EXPERIMENTS.
    # The emulator build always has them (the CPU suite keeps every variant under test); the product build does not: on the hardware these items are deselected
    # unless the run says the library under test was built with them (CV_BUILD_EXPERIMENTS=1 in the environment of both the build and the test run).
    if os.environ.get("CV_BUILD_EXPERIMENTS") != "1":
        drop = [it for it in items if it.get_closest_marker("experiments") and getattr(getattr(it, "callspec", None), "params", {}).get("lib") == "hip"]
        if drop:
            config.hook.pytest_deselected(items=drop)
            items[:] = [it for it in items if it not in drop]
    if (config.getoption("markexpr", "") or "").strip() != "not gpu":
        return                                                   # GPU runs keep the alphabetical order (one process; the full-size tests last)
    rank = {name: i for i, name in enumerate(_HEAVY_FIRST)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), len(rank)))          # stable: the order inside a file does not change


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments: exercises a measured no-go variant that exists only in a CV_BUILD_EXPERIMENTS build (always under the emulator)")
    if os.environ.get("PYTEST_XDIST_WORKER"):
        # CPU suite under pytest-xdist (6 workers): the emulator is single-threaded and the oracle's torch ops are small, so one or two intra-op
        # threads per worker are enough - 6 workers x all cores each only fight over the machine
        import torch
        torch.set_num_threads(1)


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
