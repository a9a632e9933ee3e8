"""Every BASELINE configuration as it runs in production -- a GOF with MANY FRAMES IN FLIGHT on one GPU -- against per-frame
MD5 fixtures of the unmodified reference (tests/golden/full_size.npz: every canvas, patch list, reconstructed cloud, colours and
pointToPixel of every frame), REPEATED: a scheduling-dependent slip of the lock-free union-finds (S3 contraction, S7 components:
possibly stale views, see patches.hip / orient_contract.hip), of the closure's spill ring (S5) or of the speculative round trips
shows up here or nowhere.

  config 2   longdress_vox10 all-intra, 32 frames, 16 in flight (the condition bench.py times)
  config 3   loot / redandblack / soldier all-intra: voxels of 2 (CSR rows, 3 911-cell ball, the 1 GiB dense table per
             context), 10 iterations, 8 frames, 8 in flight
  config 4   basketball_player_vox11 random-access: 2.9 M points per frame, 12 bits, occupancyPrecision 2, 2560-wide canvas,
             spatial-consistency chain + global patch allocation over the GOF, 8 frames, 8 in flight

Each soak runs with the debug invariants of the union passes on (TMC2_UF_CHECK=1: every link falls in priority and stays inside
the plane / raw set, both ends of every mutual edge share a root at agent scope), then without, then with the conservative
forms (no stale pre-check; every hop of every find at agent scope), which must give the same bytes."""
import multiprocessing as mp
import os

import pytest

import tmc2_amd as T
from tmc2_amd.configs import FULL_SIZE_CASES, constrained_pack
from test_gpu_full_size import check_against_fixture, digest, fixture

# case -> (frames in flight, repetitions with invariants on, repetitions with them off)
SOAKS = {
    "longdress_vox10_ai_r3_gof32": (16, 5, 3),
    "loot_vox10_ai_r3_gof8": (8, 5, 2),
    "redandblack_vox10_ai_r3_gof8": (8, 5, 2),
    "soldier_vox10_ai_r3_gof8": (8, 5, 2),
    "basketball_player_vox11_ra_r5_gof8": (8, 5, 2),
}


def _gen(arg):
    from tmc2_amd.synth import synth_cloud
    return synth_cloud(arg[0], arg[1])


def gof_input(name):
    c, g = FULL_SIZE_CASES[name], fixture(name)
    if not g:
        pytest.fail("fixture of %s missing from tests/golden/full_size.npz" % name)
    with mp.get_context("spawn").Pool(min(16, c["frames"], os.cpu_count() or 4)) as pool:      # (spawn: a HIP context may exist already)
        frames = pool.map(_gen, [(c["workload"], i) for i in range(c["frames"])])
    assert "".join(digest(x) + digest(col) for x, col in frames) == str(g["input_md5"]), "synthetic input differs from the fixture's"
    return c, g, frames


def run_and_check(enc, frs, c, g, what):
    for fr in frs:
        fr.reset()
    W, H = enc.phase_a(frs, constrained_pack=constrained_pack(c))
    enc.phase_b(frs)
    per = enc.per_frame(frs, lambda fr, i: (fr.get_patches()[0][fr.get_patch_order()], fr.get_geometry_images(),
                                            fr.get_reconstruction(), fr.get_attribute_images()))
    try:
        check_against_fixture(g, W, H, per)
    except AssertionError as e:
        raise AssertionError("%s: %s" % (what, e))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SOAKS))
def test_gpu_gof_frames_in_flight_soak(name, monkeypatch):
    c, g, frames = gof_input(name)
    workers, checked, plain = SOAKS[name]
    monkeypatch.setenv("TMC2_UF_CHECK", "1")
    enc = T.GofEncoder(0, workers=workers, iterations=c["iterations"], bits3d=c["bits3d"], occ_precision=c["precision"],
                       min_w=c["min_w"], min_h=c["min_h"], vox_dim=c["vox_dim"])
    try:
        frs = enc.upload(frames)
        for rep in range(checked):
            run_and_check(enc, frs, c, g, "%s: repetition %d, %d frames in flight, invariants on" % (name, rep, workers))
        enc.set_option("UF_CHECK", "0")
        for rep in range(plain):
            run_and_check(enc, frs, c, g, "%s: repetition %d, %d frames in flight" % (name, rep, workers))
        # the conservative forms must give the same bytes: no stale pre-check; every hop of every find at agent scope
        enc.set_option("UF_PRECHECK", "0")
        run_and_check(enc, frs, c, g, name + ": pre-check off")
        enc.set_option("UF_SCOPE", "agent")
        run_and_check(enc, frs, c, g, name + ": agent-scope finds")
    finally:
        enc.close()
