import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASE9 = os.path.join(GOLDEN, "case9.m")
CASE118 = os.path.join(GOLDEN, "case118.m")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def case9_grid(oracle):
    return oracle.grid_from_matpower(CASE9)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """-m gpu runs: say which build of the library and which HIP runtime the tests ran on (a Python process maps
    PyTorch's bundled runtime first -- exaadmm.jl_amd/__init__.py -- a C / Julia host the system one)."""
    if "gpu" not in (config.getoption("-m") or "") or "not gpu" in (config.getoption("-m") or ""):
        return
    try:
        import exaadmm_loader
        pkg = exaadmm_loader.load_package()
        lib = pkg.library()
        terminalreporter.write_line("exaadmm library : %s" % lib.exaadmm_version().decode())
        terminalreporter.write_line("HIP runtime     : %s" % pkg.bound_hip_runtime())
    except Exception as e:   # never turn a report line into a failure
        terminalreporter.write_line("exaadmm library : (not loaded: %s)" % e)
