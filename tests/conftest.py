import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """prints what the GPU parity tests measured (tests/helpers.py: report) and keeps it as JSON"""
    import json

    from tests import helpers
    if not helpers.REPORT:
        return
    terminalreporter.section("parity report (measured errors)")
    for row in helpers.REPORT:
        terminalreporter.write_line(json.dumps(row))
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.json"), "w") as fh:
            json.dump(helpers.REPORT, fh, indent=1)
    except OSError:
        pass


@pytest.fixture
def oracle_backend():
    """Routes the host-side mirror to the CPU oracle for the duration of one test (test-only:
    the product never selects a CPU provider)."""
    from gaussian_splatting_amd import backend
    from oracle import gs_oracle

    prev = backend._backend
    gs_oracle.set_modes(0, 0)
    gs_oracle.set_sh_band1_mode(0)
    backend.use(gs_oracle)
    yield gs_oracle
    gs_oracle.set_modes(0, 0)
    gs_oracle.set_sh_band1_mode(0)
    backend.use(prev)


@pytest.fixture(params=["shim", "native"])
def hip_backend(request):
    """the two providers of the reference's `splat_cuda` functions on the GPU: the ctypes shim
    (gaussian_splatting_amd.splat_cuda) and the compiled pybind11 module (csrc/bindings_hip.cpp)"""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gaussian_splatting_amd import backend, splat_cuda, splat_cuda_native

    prev = backend._backend
    mod = splat_cuda if request.param == "shim" else splat_cuda_native.load()
    backend.use(mod)
    yield mod
    backend.use(prev)
