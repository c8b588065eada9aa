import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "attention-lvcsr_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU work (full-size oracle checks); run with --runslow")


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False, help="also run the tests marked slow")
    parser.addoption("--knob", action="append", default=[], metavar="NAME=INT",
                     help="run the GPU tests under a kernel variant (include/lvsr_hip.h LVSR_KNOB_*), e.g. --knob dec_cluster=8")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--runslow"):
        return
    skip = pytest.mark.skip(reason="slow: pass --runslow")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


def golden_path(name):
    return os.path.join(GOLDEN, name + ".npz")


def load_golden(name):
    import json
    import numpy
    path = golden_path(name)
    if not os.path.exists(path):
        pytest.skip("golden fixture %s missing" % name)
    z = numpy.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta


@pytest.fixture(scope="session")
def gpu_device(request):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    # A/B runs of the GPU suite under a kernel variant: `pytest --knob dec_cluster=8 ...`; nothing given = the defaults
    knobs = request.config.getoption("--knob")
    if knobs:
        from lvsr_amd import native
        native.get().set_knobs(knobs)
    return torch.device("cuda:0")
