import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """SPLAT_TEST_SHUFFLE=<seed>: the tests in a seeded random order.  The GPU tests share one context (below): another order is
    another history of that context — of its buffers' addresses and contents, its learnt sizes, its streams' timing — and a result
    that depends on any of it is a bug (tools/gpu_test_matrix.sh runs a few seeds; the default order stays the file order)."""
    seed = os.environ.get("SPLAT_TEST_SHUFFLE")
    if seed:
        import random
        random.Random(int(seed)).shuffle(items)


@pytest.fixture(scope="session")
def device():
    """One splat ctx for the whole GPU session (a ctx is one device + one stream)."""
    # torch bundles its own HIP runtime: in a process that uses both, torch's must be the one that gets
    # loaded (libsplat_hip then binds to it); the other order leaves torch with "No HIP GPUs are available"
    import torch  # noqa: F401
    import splat_renderer_amd as sr
    dev = sr.Device(0)
    yield dev
    # The default frame ranks with returning LDS atomics, checks every finished tile list, and on a failed check renders
    # the frame again with ballots — after which every list comparison passes.  A recovery on the context ~370 tests share
    # must therefore fail the session, not hide in it (the injection tests use contexts of their own).
    status = dev.rankStatus()
    dev.destroy()
    expected = "checked" if not os.environ.get("SPLAT_RANK") else status["policy"]  # (SPLAT_RANK forces a policy: then only the counter)
    assert status["orderFaults"] == 0 and status["policy"] == expected, (
        f"the shared test context recovered from a misranked tile list during this session: {status}")
