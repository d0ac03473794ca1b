import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the library's variant switches (FDJAC_WINDOW, FDJAC_LAZY_STORE, ...) are honoured only in a process that opts in: the tests do, to run
# both sides of every "same bits" claim (csrc/fdjac_internal.h, test_switch)
os.environ.setdefault("FDJAC_TEST_SWITCHES", "1")


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


# Tests that compare DECOMPRESSION-kernel variants with each other (row windows, tile sizes, periodic codes, computed
# descriptors, gather kernels, the differences hand-over ...) exercise the hand-over path: since round 3 a verified exact band
# with a FD_LAZY_CAP_STORE launcher never reaches those kernels (f!'s launch stores the Jacobian itself), so they switch the
# store capability off.  Oracle-parity tests run with the library's defaults.  A test's own monkeypatch.setenv overrides this.
_HANDOVER_TESTS = ("kernel_bit_identical", "tile_sizes", "tile_order", "periodic_entry_codes", "band_descriptors", "band_direct",
                   "band_index_arithmetic", "row_strips", "rolling_row_windows", "lazy_differences", "lds_dma", "heuristic")


@pytest.fixture(autouse=True)
def _handover_path_for_variant_tests(request, monkeypatch):
    name = request.node.name
    if "lazy_store" not in name and any(k in name for k in _HANDOVER_TESTS):
        monkeypatch.setenv("FDJAC_LAZY_STORE", "0")
    yield
