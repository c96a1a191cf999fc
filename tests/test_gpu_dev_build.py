"""GPU: the tests that set FGOICP_* A/B knobs (marked `dev_knobs` by tests/conftest.py) need the development build of the library
(-DFGOICP_DEV_KNOBS: csrc/host/knobs.hpp) — the shipped build reads none of those variables and does not instantiate the rejected
kernel variants.  They run here, in ONE child process that loads libfgoicp_amd_dev.so (FGOICP_LIB), so that `pytest -m gpu` covers
both builds: every other GPU test runs against the shipped library in this process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shipped_build_reads_no_knobs(fg, gpu_required):
    assert not fg.dev_knobs() or os.environ.get("FGOICP_LIB"), "the default library must be the shipped build"


def test_knob_tests_pass_on_the_development_build(fg, gpu_required):
    if fg.dev_knobs():
        pytest.skip("already running on the development build")
    dev = fg.build.DEV_LIB
    assert os.path.exists(dev), "python __graft_entry__.py build makes both libraries"
    env = dict(os.environ, FGOICP_LIB=dev)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests"), "-m", "gpu and dev_knobs", "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    tail = (p.stdout[-3000:] + p.stderr[-1500:])
    assert p.returncode == 0, tail
    assert " passed" in p.stdout and "skipped" not in p.stdout.splitlines()[-1], tail
