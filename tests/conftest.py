import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "image-processing_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "server_numpy_ops.npz"))
    with open(os.path.join(ROOT, "tests", "golden", "server_contract.json")) as f:
        meta = json.load(f)
    return g, meta


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ipx._lib as L
    L.load()
    L.call("ipx_check_device")
    return torch.device("cuda:0")
