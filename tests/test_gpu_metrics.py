"""GPU tier (-m gpu): D1 / D2 / colour distortion (S23) through the C-ABI against the oracle -- every value bit-equal."""
import numpy as np
import pytest

from tmc2_amd.synth import synth_cloud

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _recon(oracle, xyz, rgb):
    a = oracle.phase_a([(xyz, rgb)], 10)
    b = oracle.phase_b([(xyz, rgb)], a)[0]
    return b["recon_xyz"], b["recon_rgb"]


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_gpu_metrics_match_oracle(gpu_ctx, oracle, name):
    xyz, rgb = synth_cloud(name)
    rec, col = _recon(oracle, xyz, rgb)
    nrm = oracle.normals(xyz)
    for normals in (None, nrm):
        got, gc = gpu_ctx.metrics_compute(xyz, rgb, rec, col, normals)
        exp, ec = oracle.metrics(xyz, rgb, rec, col, normals)
        assert np.array_equal(gc, ec)
        assert np.array_equal(bits(got), bits(exp)), (got, exp)


def test_gpu_metrics_lossy_reconstruction(gpu_ctx, oracle):
    """Perturbed geometry (duplicates after rounding, holes -> reconstructed points nobody votes for) and colours."""
    xyz, rgb = synth_cloud("small", 1)
    rec, col = _recon(oracle, xyz, rgb)
    rng = np.random.default_rng(7)
    rec2 = (rec[::2] + rng.integers(-2, 3, rec[::2].shape)).astype(np.int16)
    col2 = np.clip(col[::2].astype(np.int32) + rng.integers(-9, 10, col[::2].shape), 0, 255).astype(np.uint8)
    extra = (xyz[:50] + np.array([40, 0, 0], np.int16)).astype(np.int16)      # isolated blob: orphans for scaleNormals
    rec2 = np.concatenate([rec2, extra])
    col2 = np.concatenate([col2, rgb[:50]])
    nrm = oracle.normals(xyz)
    got, gc = gpu_ctx.metrics_compute(xyz, rgb, rec2, col2, nrm)
    exp, ec = oracle.metrics(xyz, rgb, rec2, col2, nrm)
    assert np.array_equal(gc, ec) and np.array_equal(bits(got), bits(exp)), (got, exp)


def test_gpu_metrics_identity(gpu_ctx):
    """A cloud against itself: zero distortion, infinite PSNR (division by zero as in the reference)."""
    xyz, rgb = synth_cloud("tiny")
    got, counts = gpu_ctx.metrics_compute(xyz, rgb, xyz, rgb, None)
    assert counts[0] == counts[1] == len(xyz)
    assert got[2, 0] == 0.0 and np.isinf(got[2, 1]) and got[2, 4] == 0.0
