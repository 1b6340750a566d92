import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("rust-pathtracer_amd")


@pytest.fixture(scope="session")
def oracle(pkg):
    """The CPU oracle (oracle/libptref.so), built on demand.  Test infrastructure only."""
    import oracle_loader
    return oracle_loader.load(pkg)


@pytest.fixture(scope="session")
def engine(pkg):
    """The product: the HIP engine behind the C ABI.  No fallback."""
    return pkg.load()


def pytest_terminal_summary(terminalreporter):
    """Which film cases needed the 8-ulp allowance of tests/parity_suite.py (pixels too bright for 1e-4 to be more than their own rounding):
    printed with every run and left in gpurun_out/ulp_bar.json, so that the tolerance in use is a reported fact."""
    try:
        import json
        import parity_suite
        log = parity_suite.ULP_BAR_LOG
        terminalreporter.write_line("film cases that used the 8-ulp allowance: %d%s" % (len(log), "" if log else " (every film within L-inf 1e-4 flat)"))
        for case, (n, brightest, linf) in sorted(log.items()):
            terminalreporter.write_line("  %s: %d pixel(s), brightest value %.4g, L-inf %.3g" % (case, n, brightest, linf))
        out = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out) and log:
            json.dump({k: {"pixels": v[0], "brightest": v[1], "linf": v[2]} for k, v in log.items()}, open(os.path.join(out, "ulp_bar.json"), "w"), indent=1)
    except Exception as e:   # a report, never a failure
        terminalreporter.write_line("ulp-bar report unavailable: %r" % (e,))
