import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _reset_training_globals():
    """The mode of the weight gradients' X operands is a per-pass global set by forward_train (policy role `wgx`): operator-level tests
    that follow a training pass in the same process must not inherit it."""
    yield
    mod = sys.modules.get("craft_amd.autograd")
    if mod is not None:
        mod.use_modes((None, None, None))
