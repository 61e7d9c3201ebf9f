import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _have_gpu() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the gpu tests instead of failing in adsb_create."""
    expr = config.getoption("-m") or ""
    if _have_gpu() or ("gpu" in expr and "not gpu" not in expr):
        return  # asked for explicitly (the GPU box): run, and fail loudly if the device is missing
    skip = pytest.mark.skip(reason="no HIP device: gpu tests run on the MI355X box (-m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    return json.loads((GOLDEN / "reference_frames.json").read_text())


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle binding (test infrastructure)."""
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def fixture_iq(golden):
    """{file name: (131072, 2) int16 [re, im]} -- file order is [im][re] (src/utils.rs:29-31)."""
    out = {}
    for fx in golden["fixtures"]:
        raw = np.fromfile(GOLDEN / fx["file"], dtype="<i2").reshape(-1, 2)
        out[fx["file"]] = np.ascontiguousarray(raw[:, ::-1])
    return out


@pytest.fixture(scope="session")
def hip_lib():
    """libadsb_hip.so, built if stale (hipcc cross-compiles without a GPU)."""
    from dump1090_rs_amd import _lib, build
    build.build_library()
    build.build_abi_host()   # the compiled-C host of test_compiled_c_host_runs_the_reference_test_routine
    return _lib.lib()
