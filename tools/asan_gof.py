#!/usr/bin/env python
"""The whole path with many frames in flight, WITHOUT torch (its HIP start-up does not survive a preloaded sanitizer runtime):
what tools/asan_host_gcc.sh runs on the GPU box against the gcc-ASan build of the host side of libtmc2hip.so.

    bash tools/asan_host_gcc.sh run python tools/asan_gof.py [--config longdress] [--frames 16] [--workers 16] [--steps 3]

Per step: reset, S0-S22 with the config's packing condition, canvases into page-locked host memory (tmc2_host_alloc); then the
post-reconstruction tail, a decoder-side GOF (decoder frames, reconstruct, tail, metric against the source), the frame-resident
metric, PLY round trips -- every host-side code path of the library (C-ABI glue, packers, global patch allocation, orientation
walk, PLY parser / writer, metric text) under the sanitizer, with the worker threads' contexts created and destroyed in order.
Checks the canvases against the reference's digests where the fixture has the case.  Exit code 0 = ran clean."""
import argparse
import hashlib
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("TMC2_PACKAGE_DIR") or os.path.join(ROOT, "mpeg-pcc-tmc2_amd"))
import numpy as np  # noqa: E402
import tmc2_amd as T  # noqa: E402
from tmc2_amd.configs import BENCH_CONFIGS, FULL_SIZE_CASES, constrained_pack  # noqa: E402
from tmc2_amd.synth import synth_cloud  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="longdress")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    name = BENCH_CONFIGS.get(a.config, a.config)
    c = FULL_SIZE_CASES[name]
    nf = min(a.frames, c["frames"])
    clouds = [synth_cloud(c["workload"], i) for i in range(nf)]
    enc = T.GofEncoder(0, min(a.workers, nf), c["iterations"], c["bits3d"], c["precision"], c["min_w"], c["min_h"], vox_dim=c["vox_dim"])
    frames = enc.upload(clouds)
    P, res = c["precision"], float((1 << (c["bits3d"] - 1)) - 1)
    t0 = time.time()
    for step in range(a.steps):
        for fr in frames:
            fr.reset()
        W, H = enc.phase_a(frames, constrained_pack=constrained_pack(c))
        enc.phase_b(frames)
        outs = [(dict(occupancy=T.host_array((H, W), np.uint8), occ_video=T.host_array((H // P, W // P), np.uint8),
                      block_to_patch=T.host_array((H // 16, W // 16), np.uint32), geo0=T.host_array((H, W), np.uint16),
                      geo1=T.host_array((H, W), np.uint16)), T.host_array((2, 3, H, W), np.uint8)) for _ in frames]
        enc.per_frame(frames, lambda fr, i: (fr.get_geometry_images(outs[i][0]), fr.get_attribute_images(outs[i][1])))
    print("steps: %d x %d frames, %.1f frames/s under the sanitizer, canvas %dx%d" % (a.steps, nf, a.steps * nf / (time.time() - t0), W, H), flush=True)
    # parity of the last step (frames of an all-intra GOF do not depend on how many of them run)
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_size.npz"))
    md5 = lambda x: hashlib.md5(np.ascontiguousarray(x).tobytes()).hexdigest()
    if nf == c["frames"] or (c["pack"] == 0 and (W, H) == tuple(int(x) for x in g[name + "/canvas"])):
        bad = [(i, k) for i in range(nf) for k in ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1")
               if md5(outs[i][0][k]) != str(g["%s/f%d_%s_md5" % (name, i, k)])]
        bad += [(i, "attribute") for i in range(nf) if md5(outs[i][1]) != str(g["%s/f%d_attribute_md5" % (name, i)])]
        print("canvases against the reference's digests:", "equal" if not bad else "DIFFER %s" % bad[:6], flush=True)
        if bad:
            return 1
    # tail, decoder side, metrics
    enc.phase_c(frames)

    def cut(fr, i):
        patches = fr.get_patches()[0][fr.get_patch_order()]
        sent = np.zeros(len(patches), patches.dtype)
        for k in ("u0", "v0", "sizeU0", "sizeV0", "patchOrientation", "u1", "v1", "d1", "normalAxis", "tangentAxis", "bitangentAxis", "projectionMode"):
            sent[k] = patches[k]
        sent["sizeU"], sent["sizeV"] = sent["sizeU0"] * 16, sent["sizeV0"] * 16
        img = fr.get_geometry_images()
        return sent, img["occ_video"], np.stack([img["geo0"], img["geo1"]]), fr.encoder_attribute_to_yuv420(4), fr.get_normals()
    cuts = enc.per_frame(frames, cut)
    dec = enc.per_frame(frames, lambda fr, i: fr.ctx.decoder_frame(cuts[i][0], W, H, P, cuts[i][1], cuts[i][2]))

    def chain(fr, i):
        fr.set_decoded_geometry(cuts[i][1], cuts[i][2])
        fr.codec_generate_point_cloud()
        fr.codec_set_decoded_attribute_yuv420(cuts[i][3], 0)
        fr.codec_post_reconstruct(None)
        return fr.metrics_compute_source(clouds[i][0], clouds[i][1], cuts[i][4], 1, res)
    got = enc.per_frame(dec, chain)
    mine = enc.per_frame(frames, lambda fr, i: fr.metrics_compute(1, True, res))
    assert all(np.array_equal(x[0].view(np.uint64), y[0].view(np.uint64)) for x, y in zip(got, mine)), "decoder-side metric differs from the encoder-side one"
    print("decoder side + metric: D1 %.3f dB, D2 %.3f dB" % (got[0][0][2, 1], got[0][0][2, 3]), flush=True)
    print(T.metrics_display(got[0][0], len(clouds[0][0]), int(got[0][1][1]), got[0][1], int(res)).splitlines()[0], flush=True)
    with tempfile.TemporaryDirectory() as d:
        post = dec[0].get_post_reconstruction()
        for ascii_ in (True, False):
            path = os.path.join(d, "c%d.ply" % ascii_)
            T.ply_write(path, post["xyz"], post["rgb"], None, ascii=ascii_)
            x, col, _ = T.ply_read(path, threads=8)
            assert np.array_equal(x, post["xyz"]) and np.array_equal(col, post["rgb"])
        T.point_set_checksum(post["xyz"], post["rgb"])
    for fr in dec + frames:
        fr.close()
    enc.close(join=True)
    print("clean exit", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
