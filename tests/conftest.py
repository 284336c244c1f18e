import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "state_dict_manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def savi_sd(manifest):
    from textocvp_amd import synth
    return synth.synth_state_dict(manifest["SAVi"], prefix="savi.")


@pytest.fixture(scope="session")
def pred_sd(manifest):
    from textocvp_amd import synth
    return synth.synth_state_dict(manifest["PredictorWrapper"], prefix="pred.")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def max_abs(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max())
