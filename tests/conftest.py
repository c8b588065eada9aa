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
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    # A/B runs of the GPU suite under a kernel variant: LVSR_KNOB_<NAME>=<int> (tools/r3*.sh); nothing set = the defaults
    if any(k.startswith("LVSR_KNOB_") for k in os.environ):
        from lvsr_amd import native
        native.get().knobs_from_env()
    return torch.device("cuda:0")
