import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """libdsp_hip.so is git-ignored: (re)build it in-tree when it is missing or older than its sources, so that both
    test tiers exercise the current kernels (hipcc cross-compiles gfx950 without a GPU)."""
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def rts309():
    d = np.load(os.path.join(ROOT, "dispatches_amd", "data", "rts_gmlc_309.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def rts303():
    d = np.load(os.path.join(ROOT, "dispatches_amd", "data", "rts_gmlc_303.npz"))
    return {k: d[k] for k in d.files}
