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


class _CtxOptions:
    """monkeypatch-shaped access to the per-context options of the session's context (tmc2_ctx_set_option: the library reads the
    environment once, when a context is created -- a test that wants another form of a stage says so to ITS context)."""

    def __init__(self, ctx):
        self.ctx, self.saved = ctx, {}

    @staticmethod
    def _key(name):
        return name[5:] if name.startswith("TMC2_") else name

    def setenv(self, name, value):
        key = self._key(name)
        self.saved.setdefault(key, self.ctx.get_option(key))
        self.ctx.set_option(key, value)

    def delenv(self, name):
        self.setenv(name, None)

    def undo(self):
        for key, old in self.saved.items():
            self.ctx.set_option(key, old)


@pytest.fixture
def ctx_options(gpu_ctx):
    o = _CtxOptions(gpu_ctx)
    yield o
    o.undo()
