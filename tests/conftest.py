import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (TMC2_PACKAGE_DIR: a copy of the package next to another build of the library -- tools/asan_host.sh)
for p in (os.environ.get("TMC2_PACKAGE_DIR") or os.path.join(ROOT, "mpeg-pcc-tmc2_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


try:                       # torch brings its own HIP runtime: when both it and libtmc2hip.so live in one process (the zero-copy
    import torch  # noqa: F401  canvas views of the multi-GPU gather), torch's must be loaded first or it finds no device
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding as ob
    return ob.Oracle()


@pytest.fixture(scope="session")
def reference():
    """The compiled, unmodified reference (oracle/_ref); skipped where it has not been built."""
    import oracle_binding as ob
    if not os.path.exists(ob.REF_PATH):
        pytest.skip("oracle/_ref/libtmc2ref.so not built (needs /root/reference)")
    return ob.Reference()


@pytest.fixture(scope="session")
def gpu_ctx():
    import tmc2_amd as T
    return T.Context(0)
