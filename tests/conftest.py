import os
import sys

import pytest

try:                # torch first: its bundled HIP runtime is then the one the whole process shares (a test that imports torch only after
    import torch    # libmrt_hip.so has loaded /opt/rocm's copy finds "No HIP GPUs"); the full suite always had this order through collection
except ImportError:
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mrt():
    import metal_raytracing_amd
    return metal_raytracing_amd


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu_ctx(mrt):
    ctx = mrt.Context(0)
    yield ctx
    ctx.close()
