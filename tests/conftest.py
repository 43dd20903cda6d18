import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# The MFMA modes every golden-vector suite runs in: exact fp32 (the reference's arithmetic) and the mode bench.py reports.
# A test that takes the `mfma` fixture runs once per mode inside the ordinary `-m gpu` run; the mirror, the Engine and
# the dynamics module pick the mode up through ops.default_mfma(), direct C-ABI calls through golden_util.dyn_kw / the value.
BENCH_MFMA = "f16x2"
MFMA_MODES = ["f32", BENCH_MFMA]


@pytest.fixture(params=MFMA_MODES)
def mfma(request, monkeypatch):
    monkeypatch.setenv("MOBODY_MFMA", request.param)
    return request.param
