import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


os.environ.pop("VLB_LIB", None)      # the tests always exercise the in-tree product library, never the tools build


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel_err(a, ref):
    """max |a-ref| / max |ref| on fp32 CPU copies."""
    import torch
    a = a.detach().float().cpu()
    ref = ref.detach().float().cpu()
    return float((a - ref).abs().max() / (ref.abs().max() + 1e-12))
