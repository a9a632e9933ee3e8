"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference (oracle/_ref).

Run in the build container (needs /root/reference, via `make -C oracle ref`):
    python tests/golden/make_golden.py
Inputs are the seeded synthetic clouds of tmc2_amd.synth (the generator is part of the repo, so the
fixtures store only outputs + a checksum of the input).  Everything stored is DATA produced by
running the reference -- no reference source text."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "mpeg-pcc-tmc2_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402
from tmc2_amd.synth import synth_cloud, synth_decoded_attribute, two_body_gof  # noqa: E402


def digest(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref = ob.Reference()
    for name in ("tiny", "small"):
        xyz, rgb = synth_cloud(name)
        out = {"input_md5": np.array(digest(xyz) + digest(rgb))}
        knn = ref.knn_self(xyz, 16)
        out["knn16"] = knn
        q = (xyz[::5] + np.array([3, -2, 5], np.int16)).astype(np.int16)
        out["knn8_offcloud"] = ref.knn(xyz, q, 8)
        out["knn1_offcloud"] = ref.knn(xyz, q, 1)
        cnt, idx = ref.radius(xyz, q[:512], 30.0, 64)
        out["radius30_count"], out["radius30_idx"] = cnt, idx
        out["normals_raw"] = ref.normals(xyz, 16, oriented=False)
        out["normals_oriented"] = ref.normals(xyz, 16, oriented=True)
        w = ref.weight_normal(xyz, 11, 0.6)
        out["weight_normal"] = w
        out["partition_initial"] = ref.initial_segmentation(out["normals_oriented"], w).astype(np.uint8)
        if name == "small":  # keep the committed fixtures small: large arrays as digests only
            for k in ("knn16", "knn8_offcloud", "normals_raw", "normals_oriented"):
                out[k + "_md5"] = np.array(digest(out.pop(k)))
        np.savez_compressed(os.path.join(HERE, "segmenter_%s.npz" % name), **out)
        print(name, len(xyz), {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


def gof():
    """Whole path S0-S23 on a 2-frame GOF through the reference's own PCCEncoder / PCCMetrics members."""
    ref = ob.Reference()
    frames = [synth_cloud("tiny", f) for f in range(2)]
    a = ref.phase_a(frames, 10, 11, 4)
    b = ref.phase_b(frames, a, 4)
    out = {"input_md5": np.array("".join(digest(x) + digest(c) for x, c in frames)),
           "canvas": np.array([a[0]["width"], a[0]["height"]])}
    for i, (pa, pb) in enumerate(zip(a, b)):
        p = pa["patches"]
        out["f%d_patches" % i] = np.stack([p[n] for n in p.dtype.names if n not in ("depthOffset", "occOffset")], 1).astype(np.int32)
        out["f%d_block_to_patch" % i] = pa["block_to_patch"].astype(np.uint16)
        out["f%d_occ_video" % i] = np.packbits(pa["occ_video"])
        for k in ("occupancy", "geo0", "geo1"):
            out["f%d_%s_md5" % (i, k)] = np.array(digest(pa[k]))
        for k in ("recon_xyz", "recon_rgb", "point_to_pixel", "attribute"):
            out["f%d_%s_md5" % (i, k)] = np.array(digest(pb[k]))
        nrm = ref.normals(frames[i][0], 16, oriented=True)
        q, counts = ref.metrics(frames[i][0], frames[i][1], pb["recon_xyz"], pb["recon_rgb"], nrm)
        out["f%d_metrics" % i] = q
        out["f%d_metric_counts" % i] = counts
    np.savez_compressed(os.path.join(HERE, "gof_tiny2.npz"), **out)
    print("gof_tiny2", {k: (v.shape if getattr(v, "shape", ()) else str(v)) for k, v in out.items()})


def gof_low_delay():
    """S0-S16 of a 4-frame GOF under the low-delay packing (constrainedPack = 1: frames after the first are packed by
    spatialConsistencyPackFlexible against their predecessor), through the reference's own placeSegments."""
    ref = ob.Reference()
    frames = [synth_cloud("tiny", f) for f in range(4)]
    a = ref.phase_a(frames, 10, 11, 4, constrained_pack=True)
    out = {"input_md5": np.array("".join(digest(x) + digest(c) for x, c in frames)),
           "canvas": np.array([a[0]["width"], a[0]["height"]])}
    for i, pa in enumerate(a):
        p = pa["patches"]
        out["f%d_patches" % i] = np.stack([p[n] for n in p.dtype.names if n not in ("depthOffset", "occOffset")], 1).astype(np.int32)
        out["f%d_matches" % i] = pa["matches"].astype(np.int32)
        out["f%d_block_to_patch" % i] = pa["block_to_patch"].astype(np.uint16)
        for k in ("occupancy", "geo0", "geo1"):
            out["f%d_%s_md5" % (i, k)] = np.array(digest(pa[k]))
    np.savez_compressed(os.path.join(HERE, "gof_tiny4_low_delay.npz"), **out)
    print("gof_tiny4_low_delay", {k: (v.shape if getattr(v, "shape", ()) else str(v)) for k, v in out.items()})


RANDOM_ACCESS_CANVAS = (128, 192)  # small enough that the allocation restarts its sub-context three different ways


def gof_random_access():
    """S0-S16 of a 6-frame GOF under the random-access packing (globalPatchAllocation = 1: performDataAdaptiveGPAMethod
    re-packs the frames of a sub-context around the unions of their tracked patches), through the reference's own
    generateSegments / placeSegments / performDataAdaptiveGPAMethod.  On this canvas the sequence takes the accepting
    branch, the bad-packing restart and the unions-too-tall restart."""
    ref = ob.Reference()
    frames = two_body_gof("tiny", 6)
    a = ref.phase_a(frames, 10, 11, 4, min_w=RANDOM_ACCESS_CANVAS[0], min_h=RANDOM_ACCESS_CANVAS[1], constrained_pack=2)
    out = {"input_md5": np.array("".join(digest(x) + digest(c) for x, c in frames)),
           "canvas": np.array([a[0]["width"], a[0]["height"]])}
    for i, pa in enumerate(a):
        p = pa["patches"]
        out["f%d_patches" % i] = np.stack([p[n] for n in p.dtype.names if n not in ("depthOffset", "occOffset")], 1).astype(np.int32)
        out["f%d_matches" % i] = pa["matches"].astype(np.int32)
        out["f%d_block_to_patch" % i] = pa["block_to_patch"].astype(np.uint16)
        for k in ("occupancy", "geo0", "geo1"):
            out["f%d_%s_md5" % (i, k)] = np.array(digest(pa[k]))
    np.savez_compressed(os.path.join(HERE, "gof_twobody6_random_access.npz"), **out)
    print("gof_twobody6_random_access", {k: (v.shape if getattr(v, "shape", ()) else str(v)) for k, v in out.items()})


def gof_post_reconstruction():
    """The post-reconstruction tail (encode() :571-719) on the 2-frame GOF of gof_tiny2.npz, through the reference's own
    colorPointCloud / smoothPointCloudPostprocess / transferColors16bitBP / convertYUV16ToRGB8.  The "decoded" attribute
    frames are synth_decoded_attribute() of the generated attribute canvases (the generator is part of the repo)."""
    ref = ob.Reference()
    frames = [synth_cloud("tiny", f) for f in range(2)]
    a = ref.phase_a(frames, 10, 11, 4)
    b = ref.phase_b(frames, a, 4)
    dec = [synth_decoded_attribute(x["attribute"]) for x in b]
    c = ref.phase_c(b, dec)
    out = {"input_md5": np.array("".join(digest(x) + digest(col) for x, col in frames)),
           "decoded_md5": np.array("".join(digest(d) for d in dec))}
    for i, pc in enumerate(c):
        out["f%d_counts" % i] = np.array([len(pc["xyz"]), int((pc["boundary_before"] == 1).sum()), int((pc["boundary"] == 3).sum())])
        out["f%d_boundary_before" % i] = np.packbits(pc["boundary_before"].astype(np.uint8))
        out["f%d_moved" % i] = np.flatnonzero(pc["boundary"] == 3).astype(np.uint32)
        out["f%d_moved_xyz" % i] = pc["xyz"][pc["boundary"] == 3]
        out["f%d_moved_colors16" % i] = pc["colors16"][pc["boundary"] == 3]
        for k in ("xyz", "colors16", "rgb", "boundary"):
            out["f%d_%s_md5" % (i, k)] = np.array(digest(pc[k]))
    np.savez_compressed(os.path.join(HERE, "gof_tiny2_post.npz"), **out)
    print("gof_tiny2_post", {k: (v.shape if getattr(v, "shape", ()) else str(v)) for k, v in out.items()})


def gof_color_chain():
    """The attribute video's way around an identity codec on the 2-frame GOF of gof_tiny2.npz, through the reference's own
    PCCInternalColorConverter: RGB444 -> YUV420 (8 bits, downsampling filter 4) -> YUV444 (16 bits, upsampling filter 0),
    then the post-reconstruction tail on those frames."""
    ref = ob.Reference()
    frames = [synth_cloud("tiny", f) for f in range(2)]
    a = ref.phase_a(frames, 10, 11, 4)
    b = ref.phase_b(frames, a, 4)
    out = {"input_md5": np.array("".join(digest(x) + digest(col) for x, col in frames))}
    dec = []
    for i, pb in enumerate(b):
        planes = []
        for m in range(2):
            y, u, v = ref.convert_rgb444_to_yuv420(pb["attribute"][m], 4)
            yuv444 = ref.convert_yuv420_to_yuv444(y, u, v, 0)
            out["f%d_m%d_yuv420_md5" % (i, m)] = np.array(digest(np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)])))
            out["f%d_m%d_yuv444_md5" % (i, m)] = np.array(digest(yuv444))
            if i == 0 and m == 0:   # a readable corner next to the digests
                out["f0_m0_u_rows"] = u[u.any(1)][:4].copy()
            planes.append(yuv444)
        dec.append(np.stack(planes))
    c = ref.phase_c(b, dec)
    for i, pc in enumerate(c):
        out["f%d_moved" % i] = np.array(int((pc["boundary"] == 3).sum()))
        for k in ("xyz", "colors16", "rgb", "boundary"):
            out["f%d_%s_md5" % (i, k)] = np.array(digest(pc[k]))
    np.savez_compressed(os.path.join(HERE, "gof_tiny2_color.npz"), **out)
    print("gof_tiny2_color", {k: (v.shape if getattr(v, "shape", ()) else str(v)) for k, v in out.items()})


def io_golden():
    """PCCPointSet3::read on the PLY variants of tests/ply_cases.py and PCCPointSet3::computeChecksum, through the reference."""
    import tempfile
    from ply_cases import cases
    ref = ob.Reference()
    xyz, rgb = synth_cloud("tiny", 0)
    xyz, rgb = xyz[:1500].copy(), rgb[:1500].copy()
    out = {"input_md5": np.array(digest(xyz) + digest(rgb))}
    with tempfile.TemporaryDirectory() as tmp:
        for name, data in cases(xyz, rgb).items():
            path = os.path.join(tmp, name + ".ply")
            with open(path, "wb") as f:
                f.write(data)
            out[name + "_file_md5"] = np.array(hashlib.md5(data).hexdigest())
            for rn in (0, 1):
                got = ref.ply_read(path, bool(rn))
                if name == "binary_short" and got[2] is not None:      # the property the file ends in is uninitialised there
                    got = (got[0], got[1], got[2][:int(np.flatnonzero(got[0].any(1))[-1])])
                out["%s_n%d" % (name, rn)] = np.array(["" if a is None else digest(a) for a in got])
    big = np.concatenate([xyz, xyz[:300]]), np.concatenate([rgb, rgb[300:600]])
    for reorder in (0, 1):
        out["checksum_r%d" % reorder] = np.frombuffer(ref.checksum(big[0], big[1], bool(reorder)), np.uint8)
        out["checksum_nocolor_r%d" % reorder] = np.frombuffer(ref.checksum(big[0], None, bool(reorder)), np.uint8)
    np.savez_compressed(os.path.join(HERE, "io_golden.npz"), **out)
    print("io_golden", {k: (v.shape if getattr(v, "shape", ()) else str(v)) for k, v in out.items()})


# BASELINE.json configurations at their real sizes and CTC settings: ONE table for the generator, the GPU tests and bench.py
from tmc2_amd.configs import FULL_SIZE_CASES  # noqa: E402
FULL_SIZE_PHASE_A = ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1")
FULL_SIZE_PHASE_B = ("recon_xyz", "recon_rgb", "point_to_pixel", "attribute")


def full_size_digests(a, b):
    """What the fixture keeps of a GOF: canvas size, and per frame the patch count, the reconstructed point count and the MD5
    of every canvas / cloud (the same bytes PCCVideoEncoder.cpp:389-396 and PCCEncoder.cpp:620-626 log per picture / cloud)."""
    out = {"canvas": np.array([a[0]["width"], a[0]["height"]])}
    for i, (pa, pb) in enumerate(zip(a, b)):
        out["f%d_counts" % i] = np.array([len(pa["patches"]), len(pb["recon_xyz"])])
        p = pa["patches"]
        out["f%d_patches_md5" % i] = np.array(digest(np.stack([p[n] for n in p.dtype.names if n not in ("depthOffset", "occOffset")], 1).astype(np.int32)))
        for k in FULL_SIZE_PHASE_A:
            out["f%d_%s_md5" % (i, k)] = np.array(digest(np.ascontiguousarray(pa[k]).astype(pa[k].dtype)))
        for k in FULL_SIZE_PHASE_B:
            out["f%d_%s_md5" % (i, k)] = np.array(digest(pb[k]))
    return out


FULL_SIZE_PHASE_C = ("xyz", "colors16", "rgb", "boundary")


def full_size_decoder_side(ref, frames, a, b, resolution, metric_frames):
    """Config 5 (the decoder's frame finish = the encoder's post-reconstruction tail) at BASELINE size, from the unmodified
    reference: the attribute canvases through "RGB444ToYUV420_8_4" (the I420 frames the video encoder reads) and back through
    "YUV420ToYUV444_8_0" (PCCInternalColorConverter.cpp:355-482: what an identity video codec hands the reconstruction),
    colorPointCloud, smoothPointCloudPostprocess (PCCCodec.cpp:54, 1067-1106), transferColors16bitBP (PCCPointSet.cpp:1126),
    convertYUV16ToRGB8 -- MD5s per frame -- and PCCMetrics::compute (PCCMetrics.cpp:324-375) of the source frame against the
    finished cloud, raw doubles, for the frames in metric_frames."""
    out, decoded = {}, []
    for i, pb in enumerate(b):
        planes = [ref.convert_rgb444_to_yuv420(pb["attribute"][m]) for m in range(2)]
        i420 = np.stack([np.concatenate([p.ravel() for p in planes[m]]) for m in range(2)])
        d444 = np.stack([ref.convert_yuv420_to_yuv444(*planes[m]) for m in range(2)])
        out["f%d_i420_md5" % i], out["f%d_dec444_md5" % i] = np.array(digest(i420)), np.array(digest(d444))
        decoded.append(d444)
    post = ref.phase_c(b, decoded)
    for i, pc in enumerate(post):
        for k in FULL_SIZE_PHASE_C:
            out["f%d_post_%s_md5" % (i, k)] = np.array(digest(pc[k]))
        out["f%d_post_moved" % i] = np.array(int((pc["boundary"] == 3).sum()))
        if i in metric_frames:
            q, counts = ref.metrics(frames[i][0], frames[i][1], pc["xyz"], pc["rgb"], None, resolution)
            out["f%d_post_metrics_no_normals" % i], out["f%d_post_metric_counts" % i] = q, counts
            nrm = ref.normals(frames[i][0], 16, True)                  # (the source's normals: D2 needs them)
            out["f%d_post_metrics" % i], _ = ref.metrics(frames[i][0], frames[i][1], pc["xyz"], pc["rgb"], nrm, resolution)
            out["f%d_src_normals_md5" % i] = np.array(digest(nrm))
    return out


def full_size(only=None, part_dir=None):
    """MD5 fixtures at BASELINE size from the unmodified reference: minutes of CPU time, kilobytes of fixture.  part_dir: write
    each case to its own <part_dir>/<name>.npz (several cases at once, one process each; `full_size_merge` folds them in)."""
    import time
    ref = ob.Reference()
    path = os.path.join(HERE, "full_size.npz")
    out = dict(np.load(path)) if os.path.exists(path) and not part_dir else {}
    for name, c in FULL_SIZE_CASES.items():
        if only and name not in only:
            continue
        t = time.time()
        frames = [synth_cloud(c["workload"], f) for f in range(c["frames"])]
        a = ref.phase_a(frames, c["iterations"], c["bits3d"], c["precision"], c["min_w"], c["min_h"], c["pack"], c["vox_dim"])
        b = ref.phase_b(frames, a, c["precision"])
        d = full_size_digests(a, b)
        d["input_md5"] = np.array("".join(digest(x) + digest(col) for x, col in frames))
        t1 = time.time()
        d.update(full_size_decoder_side(ref, frames, a, b, float((1 << (c["bits3d"] - 1)) - 1), range(c["frames"])))
        for k in [k for k in out if k.startswith(name + "/") and not k.split("/")[1].startswith(("f0_metric", "f0_normals"))]:
            del out[k]
        out.update({name + "/" + k: v for k, v in d.items()})
        if part_dir:
            os.makedirs(part_dir, exist_ok=True)
            np.savez_compressed(os.path.join(part_dir, name + ".npz"), **{k: v for k, v in out.items() if k.startswith(name + "/")})
        else:
            np.savez_compressed(path, **out)
        print(name, [len(f[0]) for f in frames], "canvas", d["canvas"], "counts", [d["f%d_counts" % i].tolist() for i in range(c["frames"])],
              "moved", [int(d["f%d_post_moved" % i]) for i in range(c["frames"])],
              "%.0f s (decoder side %.0f s)" % (time.time() - t, time.time() - t1), flush=True)


def full_size_merge(part_dir):
    """Fold per-case part files into full_size.npz (a case's encoder-side metric doubles, made by full_size_metrics, stay)."""
    path = os.path.join(HERE, "full_size.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for fn in sorted(f for f in os.listdir(part_dir) if f.endswith(".npz")):
        name = fn[:-4]
        new = dict(np.load(os.path.join(part_dir, fn)))
        for k in [k for k in out if k.startswith(name + "/") and not k.split("/")[1].startswith(("f0_metric", "f0_normals"))]:
            del out[k]
        out.update(new)
        print("merged", name, len(new), "entries")
    np.savez_compressed(path, **out)


FULL_SIZE_METRIC_CASES = ("longdress_vox10_ai_r3", "loot_vox10_ai_r3", "redandblack_vox10_ai_r3", "soldier_vox10_ai_r3",
                          "basketball_player_vox11_ra_r5", "longdress_vox10_ra_r3_gof3")


def full_size_metrics(only=None):
    """S23 at BASELINE size: PCCMetrics::compute of the unmodified reference (duplicate removal, normal copy / scaling,
    QualityMetrics::compute both ways) on frame 0 of every single-GOF case -- source cloud + the reference's own normals
    against the reconstruction of its attribute-image step -- as raw doubles (3 x 8) and the two point counts."""
    import time
    ref = ob.Reference()
    path = os.path.join(HERE, "full_size.npz")
    out = dict(np.load(path))
    for name in FULL_SIZE_METRIC_CASES:
        if only and name not in only:
            continue
        c = FULL_SIZE_CASES[name]
        t = time.time()
        frames = [synth_cloud(c["workload"], f) for f in range(c["frames"])]
        a = ref.phase_a(frames, c["iterations"], c["bits3d"], c["precision"], c["min_w"], c["min_h"], c["pack"], c["vox_dim"])
        b = ref.phase_b(frames, a, c["precision"])
        assert digest(b[0]["recon_xyz"]) == str(out[name + "/f0_recon_xyz_md5"])
        nrm = ref.normals(frames[0][0], 16, True)
        q, counts = ref.metrics(frames[0][0], frames[0][1], b[0]["recon_xyz"], b[0]["recon_rgb"], nrm, float((1 << (c["bits3d"] - 1)) - 1))
        q0, counts0 = ref.metrics(frames[0][0], frames[0][1], b[0]["recon_xyz"], b[0]["recon_rgb"], None, float((1 << (c["bits3d"] - 1)) - 1))
        out[name + "/f0_metrics"], out[name + "/f0_metric_counts"] = q, counts
        out[name + "/f0_metrics_no_normals"] = q0
        out[name + "/f0_normals_md5"] = np.array(digest(nrm))
        np.savez_compressed(path, **out)
        print(name, "counts", counts.tolist(), "D1 psnr %.4f D2 psnr %.4f Y psnr %.4f" % (q[2, 1], q[2, 3], q[2, 7]), "%.0f s" % (time.time() - t), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "full_size_metrics":
        full_size_metrics(sys.argv[2:])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "full_size":
        full_size(sys.argv[2:])
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "full_size_part":           # full_size_part <dir> <case> ...
        full_size(sys.argv[3:], part_dir=sys.argv[2])
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "full_size_merge":
        full_size_merge(sys.argv[2])
        sys.exit(0)
    main()
    gof()
    gof_low_delay()
    gof_random_access()
    gof_post_reconstruction()
    gof_color_chain()
    io_golden()
