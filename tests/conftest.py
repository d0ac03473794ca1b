import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_ready():
    """A HIP device is visible and libfdjac is built -- what every `gpu`-marked test needs."""
    try:
        import torch
        if not torch.cuda.is_available():
            return False, "no HIP device visible"
    except Exception as e:  # pragma: no cover
        return False, "torch unavailable: %s" % e
    so = os.path.join(ROOT, "finitediff.jl_amd", "lib", "libfdjac.so")
    if not os.path.exists(so):
        return False, "libfdjac.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    return True, ""


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not failed) on a box without an MI355X, so a plain `pytest` is green on CPU.
    On the GPU box nothing is skipped: a missing library there is a failure the tests must show."""
    ok, why = _gpu_ready()
    if ok:
        return
    skip = pytest.mark.skip(reason="needs an MI355X: " + why)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
