import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def keys():
    import fixtures
    return fixtures.load_keys()


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (the HIP path has no CPU fallback)")
    from multi_party_ecdsa_amd import engine
    return engine.Context(0)


@pytest.fixture(scope="session")
def gpu_ctx_serial():
    """A context pinned to the code paths LARGE batches take — one stream (no forks), 18 limbs per lane, one item per lane in
    the EC round kernels, two-base ladders in the verifiers — so that the small parity cases cover those paths too (the
    library reads no environment: the switches are options of the context, mpe_ctx_set_option)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (the HIP path has no CPU fallback)")
    from multi_party_ecdsa_amd import engine
    return engine.Context(0, options={"no_par": 1, "no_adaptive_lanes": 1})
