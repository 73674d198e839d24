import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "no_parity_coverage: the kernels this test launches do not count as parity-tested "
                                       "(property / repeatability / plumbing tests)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


import pytest


def compares(fn):
    """Marks a parity test whose assertions compare device results with the oracle / a reference DIRECTLY (torch.equal,
    allclose, exact integer outputs) instead of through rel_err: when the test body has run to its end - every assertion
    passed - the kernels it launched count as compared (confirm_compared)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        out = fn(*args, **kwargs)
        confirm_compared()
        return out

    return wrapper


@pytest.fixture(autouse=True)
def _seed_global_rngs():
    """Every test starts from the same global RNG state: a few tests draw shape-only inputs from the global generators,
    and their values must not depend on which tests ran before (one finite-difference check flipped a nearest-code
    choice once in a fresh checkout)."""
    import numpy as np
    import torch

    torch.manual_seed(1234)
    np.random.seed(1234)
    yield


# Which kernel variants did the parity tests launch?  tests/test_gpu_zz_coverage.py (collected last) compares this set with
# the kernels one optimizer step of every benchmarked workload launches: a kernel that only the benchmark reaches (a
# dispatch rule keyed on the batch size, say) fails the suite instead of producing an unverified number.
PARITY_MODULES = ("test_gpu_parity", "test_gpu_pixelcnn", "test_gpu_vdvae", "test_gpu_vqvae", "test_gpu_celeba",
                  "test_gpu_masking", "test_gpu_data", "test_gpu_eval_paths", "test_gpu_importer", "test_gpu_vade", "test_gpu_lookahead")
PARITY_KERNELS = {}          # kernel key -> first test that COMPARED a result computed after launching it
UNCOMPARED_KERNELS = {}      # kernel key -> a test that launched it without a comparison behind the launch
_current = {"node": None, "exact": False}
# modules whose assertions are exact comparisons of integer / bit-pattern outputs (torch.equal, array_equal): everything a
# test of these launches counts as compared
EXACT_MODULES = ("test_gpu_masking", "test_gpu_data")


def confirm_compared() -> None:
    """Called by the comparison helpers of the parity modules (rel_err) - a kernel variant only counts as parity-tested
    once a comparison against the oracle / a reference value has run AFTER its launch in the same test (round 3 recorded
    every launch of a parity test, also those whose result no assertion looked at)."""
    if _current["node"] is None:
        return
    from posterior_matching_amd import ops

    for k in ops.coverage_take():
        PARITY_KERNELS.setdefault(k, _current["node"])


@pytest.fixture(autouse=True)
def _record_parity_kernels(request):
    mod = request.module.__name__.rsplit(".", 1)[-1]
    if (mod not in PARITY_MODULES or "gpu" not in request.keywords or "no_parity_coverage" in request.keywords
            or not _has_gpu()):
        yield
        return
    from posterior_matching_amd import ops

    ops.coverage_begin()
    _current["node"], _current["exact"] = request.node.nodeid, mod in EXACT_MODULES
    try:
        yield
    finally:
        _current["node"] = None
        for k in ops.coverage_end():                # launched after the test's last comparison (or never compared)
            if mod in EXACT_MODULES:
                PARITY_KERNELS.setdefault(k, request.node.nodeid)
            else:
                UNCOMPARED_KERNELS.setdefault(k, request.node.nodeid)
