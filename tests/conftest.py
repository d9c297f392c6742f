"""Test configuration: registers the `gpu` marker, puts the repo root on sys.path and checks that
the native pieces are built (they are built beforehand by __graft_entry__.build(), in-tree; tests
never compile on the GPU box)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(TESTS, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    need = [os.path.join(ROOT, "ray-tracing_amd", "librt_hip.so"), os.path.join(ROOT, "ray-tracing_amd", "librt_host.so"),
            os.path.join(ROOT, "oracle", "_build", "liboracle.so")]
    if all(os.path.exists(p) for p in need):
        return
    if os.path.exists("/opt/rocm/bin/hipcc"):
        import __graft_entry__ as g

        g.build()
    missing = [p for p in need if not os.path.exists(p)]
    if missing:
        raise RuntimeError(f"native libraries missing (run __graft_entry__.build()): {missing}")


_ensure_built()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gpu():
    """Initialise device 0 through the C-ABI; GPU tests fail (not skip) when the HIP path is unusable."""
    import torch

    import ray_tracing_amd as rt

    assert torch.cuda.is_available(), "GPU test selected but no GPU is visible"
    desc = rt.init(0)
    assert "gfx950" in desc, desc
    return desc
