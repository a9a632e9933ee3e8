"""The condition bench.py times -- the 32-frame longdress_vox10 GOF with 16 frames in flight on one GPU -- against per-frame
MD5 fixtures of the unmodified reference (tests/golden/full_size.npz, case longdress_vox10_ai_r3_gof32: every canvas, patch
list, reconstructed cloud, colours and pointToPixel of all 32 frames), REPEATED: a scheduling-dependent slip of the lock-free
union-finds (S3 contraction, S7 components: possibly stale views, see patches.hip / orient_contract.hip) shows up here or
nowhere.  The soak runs with the debug invariants of the union passes on (TMC2_UF_CHECK=1: every link falls in priority and
stays inside the plane / raw set, both ends of every mutual edge share a root at agent scope), and once more with the
agent-scope-only fallback forced, which must give the same bytes."""
import multiprocessing as mp
import os

import pytest

import tmc2_amd as T
from test_gpu_full_size import CASES as FULL_CASES  # noqa: F401  (kept in step with make_golden.py)
from test_gpu_full_size import check_against_fixture, digest, fixture

NAME = "longdress_vox10_ai_r3_gof32"
CASE = dict(workload="longdress_vox10", frames=32, iterations=50, vox_dim=4, bits3d=11, precision=4, min_w=1280, min_h=1280)


def _gen(i):
    from tmc2_amd.synth import synth_cloud
    return synth_cloud(CASE["workload"], i)


@pytest.fixture(scope="module")
def gof32():
    g = fixture(NAME)
    if not g:
        pytest.fail("fixture of %s missing from tests/golden/full_size.npz" % NAME)
    with mp.get_context("spawn").Pool(min(16, os.cpu_count() or 4)) as pool:      # (spawn: a HIP context may exist already)
        frames = pool.map(_gen, range(CASE["frames"]))
    assert "".join(digest(x) + digest(col) for x, col in frames) == str(g["input_md5"]), "synthetic input differs from the fixture's"
    return g, frames


def run_and_check(enc, frs, g, what):
    for fr in frs:
        fr.reset()
    W, H = enc.phase_a(frs)
    enc.phase_b(frs)
    per = enc.per_frame(frs, lambda fr, i: (fr.get_patches()[0][fr.get_patch_order()], fr.get_geometry_images(),
                                            fr.get_reconstruction(), fr.get_attribute_images()))
    try:
        check_against_fixture(g, W, H, per)
    except AssertionError as e:
        raise AssertionError("%s: %s" % (what, e))


@pytest.mark.gpu
def test_gpu_gof32_sixteen_in_flight_soak(gof32, monkeypatch):
    g, frames = gof32
    monkeypatch.setenv("TMC2_UF_CHECK", "1")
    enc = T.GofEncoder(0, workers=16, iterations=CASE["iterations"], bits3d=CASE["bits3d"], occ_precision=CASE["precision"],
                       min_w=CASE["min_w"], min_h=CASE["min_h"], vox_dim=CASE["vox_dim"])
    try:
        frs = enc.upload(frames)
        for rep in range(5):
            run_and_check(enc, frs, g, "repetition %d, 16 frames in flight, invariants on" % rep)
        monkeypatch.setenv("TMC2_UF_CHECK", "0")
        for rep in range(3):
            run_and_check(enc, frs, g, "repetition %d, 16 frames in flight" % rep)
        # the conservative forms must give the same bytes: no stale pre-check; every hop of every find at agent scope
        monkeypatch.setenv("TMC2_UF_PRECHECK", "0")
        run_and_check(enc, frs, g, "pre-check off")
        monkeypatch.setenv("TMC2_UF_SCOPE", "agent")
        run_and_check(enc, frs, g, "agent-scope finds")
    finally:
        enc.close()
