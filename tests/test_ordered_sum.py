"""CPU tier: the block form of an ordered fp64 sum (csrc/ordered_sum.h, S23's `sse += dist` of PCCMetrics.cpp:187-198) against
the plain loop, staged as the kernels stage it (tests/helpers/ordered_sum_check.cpp, compiled with g++: no device code)."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def term_families(rng, n):
    """Term arrays that stress what the form rests on: ties (terms that end exactly half an ulp of the running sum), long runs
    of zeros, binade crossings at every scale, terms far below and far above the sum."""
    fams = {
        "colour": (np.where(rng.random(n) < 0.3, 0, rng.standard_normal(n) * 0.02).astype(np.float32) ** 2).astype(np.float32).astype(np.float64),
        "uniform": rng.random(n),
        "d2": (rng.standard_normal(n) * rng.integers(0, 4, n)) ** 2 / rng.integers(1, 5, n),
        "ties": rng.integers(0, 64, n) * (2.0 ** rng.integers(-30, 4, n)),
        "wide": np.exp(rng.uniform(-40, 20, n)),
        "zeros": np.zeros(n),
        "tenth": np.full(n, 0.1),
        "ones": np.full(n, 1.0),
        "integers": rng.integers(0, 1000, n).astype(np.float64),
        "tiny_then_huge": np.concatenate([np.full(n // 2, 1e-30), np.full(n - n // 2, 1e10)]),
        "subnormal": np.full(n, 5e-324) * rng.integers(0, 1000, n),
    }
    if n:
        neg = rng.random(n)
        neg[n // 2] = -0.25                      # not a case of the metric; the form must notice and add the block as the loop does
        fams["negative"] = neg
    return fams


@pytest.fixture(scope="module")
def osum(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    so = str(tmp_path_factory.mktemp("osum") / "libosum_check.so")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "mpeg-pcc-tmc2_amd", "csrc"),
                    os.path.join(ROOT, "tests", "helpers", "ordered_sum_check.cpp"), "-o", so], check=True)
    L = ctypes.CDLL(so)
    L.osum_sequential.restype = ctypes.c_double
    L.osum_sequential.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
    L.osum_block_form.restype = ctypes.c_double
    L.osum_block_form.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
    return L


def bits(x):
    return np.float64(x).view(np.uint64)


@pytest.mark.parametrize("n,block", [(0, 4), (1, 4), (5, 4), (1000, 4), (1000, 64), (4096, 1024), (100003, 64), (100003, 1024),
                                     (700001, 1024)])
def test_block_form_is_the_loop(osum, n, block):
    rng = np.random.default_rng(n + block)
    for name, t in term_families(rng, n).items():
        t = np.ascontiguousarray(t, np.float64)
        fb = ctypes.c_uint64(0)
        a = osum.osum_sequential(t.ctypes.data, t.size, 1)
        b = osum.osum_block_form(t.ctypes.data, t.size, 1, block, ctypes.byref(fb))
        assert bits(a) == bits(b), (name, n, block, a, b)
        if n:                                      # numpy's accumulate is the same loop: what the GPU tier compares with
            assert bits(np.cumsum(t)[-1]) == bits(a), name
        nb = (n + block - 1) // block
        if nb >= 64 and name not in ("negative", "subnormal", "tiny_then_huge"):
            assert fb.value <= 40, (name, fb.value, nb)   # the form pays: only the blocks around powers of two fall back


def test_strided_terms(osum):
    """Terms as the metric keeps them: [n][5], one chain per column."""
    rng = np.random.default_rng(11)
    t = rng.random((50000, 5)) ** 3
    for c in range(5):
        col = np.ascontiguousarray(t[:, c])
        a = osum.osum_sequential(col.ctypes.data, col.size, 1)
        b = osum.osum_block_form(t.ctypes.data + 8 * c, t.shape[0], 5, 1024, None)
        assert bits(a) == bits(b)
