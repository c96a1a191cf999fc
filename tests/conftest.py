import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# OpenMP (the oracle, the host harness) must not start one thread per visible core: a GPU box shows 256 cores and grants ~16
from oracle.pyoracle import usable_cpus  # noqa: E402

os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()))
# the driver's worker threads poll for ~100 us between jobs; next to the oracle's OpenMP team (the CPU tests drive the oracle through the
# product's driver) they only steal cores, so CPU test processes run the driver on one thread unless a test asks otherwise (the GPU tests
# of the full-size runs and tests/test_dist_gloo.py use the pool)
os.environ.setdefault("FGOICP_HOST_SPIN", "0")
try:
    import torch as _torch
    if not _torch.cuda.is_available():
        os.environ.setdefault("FGOICP_HOST_THREADS", "1")
except Exception:
    os.environ.setdefault("FGOICP_HOST_THREADS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "dev_knobs: sets FGOICP_* A/B knobs, which only the development build reads (libfgoicp_amd_dev.so; "
                            "tests/test_gpu_dev_build.py runs these in one child process with FGOICP_LIB pointing at it)")


# What the SHIPPED library still reads from the environment (csrc/host/knobs.hpp); setting anything else is an A/B knob.
_DEPLOYMENT_VARS = {"FGOICP_HOST_THREADS", "FGOICP_HOST_SPIN", "FGOICP_MULTI_DEVICES", "FGOICP_LIB", "FGOICP_ORACLE_THREADS", "FGOICP_EXTRA_CXXFLAGS"}


def pytest_collection_modifyitems(config, items):
    """GPU tests that set an FGOICP_* knob (monkeypatch.setenv / an env dict) exercise variants the shipped build does not carry or
    thresholds it does not read: they get the `dev_knobs` marker, and are skipped unless the loaded library is the development build."""
    import inspect
    import re
    dev = None
    for item in items:
        if item.get_closest_marker("gpu") is None:
            continue
        fn = getattr(item, "function", None)
        try:
            src = inspect.getsource(fn) if fn is not None else ""
        except (OSError, TypeError):
            src = ""
        # every FGOICP_* name the test mentions (a knob may be set through a variable: for name, val in env.items(): monkeypatch.setenv(name, val)),
        # minus the deployment variables and the ABI's constants
        knobs = {k for k in re.findall(r"FGOICP_[A-Z][A-Z_0-9]+", src) if not re.match(r"FGOICP_(FLAG|ERR|OK|SCHEDULE|TRANSPORT|LOG|ABI)(_|$)", k)} - _DEPLOYMENT_VARS
        if "monkeypatch" not in src and "env=" not in src and "environ" not in src:
            knobs = set()
        if knobs and item.get_closest_marker("dev_knobs") is None:
            item.add_marker(pytest.mark.dev_knobs)
        if item.get_closest_marker("dev_knobs") is not None:
            if dev is None:
                try:
                    import fgoicp_amd
                    dev = fgoicp_amd.dev_knobs()
                except Exception:
                    dev = False
            if not dev:
                item.add_marker(pytest.mark.skip(reason="needs the development build (FGOICP_LIB=.../libfgoicp_amd_dev.so): run by tests/test_gpu_dev_build.py"))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (checker only)."""
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def fg():
    import fgoicp_amd
    return fgoicp_amd


@pytest.fixture(scope="session")
def tiny_case(fg):
    """Pre-processed (centred, scaled) tiny cloud pair + LUT bounds, shared by operator tests."""
    tgt, src, R_gt, t_gt = fg.synth.workload("tiny", angle_deg=25.0)
    t_c, s_c, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
    return dict(pct=t_c, pcs=s_c, bounds=bounds, res=0.05, R_gt=R_gt, t_gt=t_gt, raw=(tgt, src))


@pytest.fixture(scope="session")
def gpu_required():
    if not _has_gpu():
        pytest.fail("this test is marked gpu but no GPU is visible (run it with -m gpu on the GPU box)")
    return True
