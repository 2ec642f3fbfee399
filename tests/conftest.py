import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(GOLDEN_DIR, "golden_meta.json")) as f:
        return json.load(f)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))


def case_inputs(meta_case):
    """Rebuild (state_dict, inputs, hparams) of a golden case from its seeds."""
    from onepose_amd import synthetic
    kind, seed = meta_case["weights"]
    sd = synthetic.make_state_dict(seed) if kind == "random" else synthetic.make_passthrough_state_dict(seed)
    data = synthetic.make_inputs(**meta_case["inputs"])
    return sd, data, meta_case["hparams"]
