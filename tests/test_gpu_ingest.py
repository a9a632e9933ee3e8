"""GPU tier (-m gpu): SURVEY.md section 8f row 4 end to end on the device box -- a PLY file is parsed straight into
page-locked buffers (tmc2_ply_read), bound to a frame in HBM, taken through the path, and the reconstructed cloud leaves as
the PLY file / checksum the reference would write (PCCPointSet3::read / write / computeChecksum, PCCPointSet.cpp:222-757)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

import tmc2_amd as T
from tmc2_amd.synth import synth_cloud

pytestmark = pytest.mark.gpu


def _port_io():
    spec = importlib.util.spec_from_file_location("port_io", os.path.join(os.path.dirname(__file__), "..", "oracle", "port_io.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _through_the_path(ctx, xyz, rgb, iterations=10):
    fr = ctx.frame(xyz, rgb)
    p = T.ctc_params(iterations, 11, fr.weight_normal(11, 0.6))
    fr.segmenter_compute(p)
    W, H = T.encoder_canvas_size([fr.encoder_pack_flexible(1280, 2, 1.0)], 1280, 1280, 1280)
    fr.encoder_generate_geometry_images(W, H, 4)
    fr.encoder_generate_attribute_images()
    rx, rc, _ = fr.get_reconstruction()
    return fr.get_geometry_images(), fr.get_attribute_images(), rx, rc


@pytest.mark.parametrize("ascii_", [True, False])
def test_gpu_ply_file_to_frame_to_ply_file(gpu_ctx, tmp_path, ascii_):
    port_io = _port_io()
    xyz, rgb = synth_cloud("small", 1)
    src = tmp_path / "in.ply"
    T.ply_write(str(src), xyz, rgb, None, ascii=ascii_)
    n, has_rgb, _ = T.ply_info(str(src))
    assert n == len(xyz) and has_rgb
    # the parser's destination is page-locked memory: what tmc2_frame_create uploads from without a staging copy
    hx = torch.empty((n, 3), dtype=torch.int16, pin_memory=True).numpy()
    hc = torch.empty((n, 3), dtype=torch.uint8, pin_memory=True).numpy()
    for threads in (1, 8):
        hx[:], hc[:] = 0, 0
        gx, gc, _ = T.ply_read(str(src), threads=threads, out=(hx, hc))
        ex, ec, _ = port_io.ply_read(str(src))
        assert np.array_equal(gx, ex) and np.array_equal(gc, ec) and np.array_equal(gx, xyz) and np.array_equal(gc, rgb)
    # the frame bound to the parsed buffers gives what the frame bound to the arrays gives
    got = _through_the_path(gpu_ctx, hx, hc)
    exp = _through_the_path(gpu_ctx, xyz, rgb)
    for k in ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1"):
        assert np.array_equal(got[0][k], exp[0][k]), k
    assert np.array_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2]) and np.array_equal(got[3], exp[3])
    # ... and the reconstructed cloud leaves as the reference's file and checksum
    rx, rc = got[2], got[3]
    dst = tmp_path / "rec.ply"
    T.ply_write(str(dst), rx, rc, None, ascii=ascii_)
    bx, bc, _ = port_io.ply_read(str(dst))
    assert np.array_equal(bx, rx) and np.array_equal(bc, rc)
    for reorder in (False, True):
        assert T.point_set_checksum(rx, rc, reorder) == port_io.checksum(rx, rc, reorder)


@pytest.mark.gpu
@pytest.mark.parametrize("condition", ["ai", "ld", "ra"])
def test_gpu_native_front_end_sharded_over_device_shards(tmp_path, condition):
    """integration/tmc2_encode_gof.cpp --devices: the GOF's frames sharded over several device shards from ONE native process
    (frame f on shard f mod D; here the same GPU twice and three times, so that the sharded path -- contexts per shard, the
    packing chain over frames that live on different shards, canvases into page-locked host memory -- runs on a one-GPU box)
    must write the same bytes as the single-shard run: occupancy / geometry / attribute videos, reconstructed clouds,
    checksums."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "integration")], check=True, capture_output=True)
    exe = os.path.join(root, "integration", "tmc2_encode_gof")
    frames = 5
    for f in range(frames):
        xyz, rgb = synth_cloud("tiny", f)
        T.ply_write(str(tmp_path / ("fr_%04d.ply" % f)), xyz, rgb)
    outs = {}
    for tag, devices, workers in (("one", "0", "2"), ("two", "0,0", "2"), ("three", "0-0,0,0", "1")):
        r = subprocess.run([exe, "--in", str(tmp_path / "fr_%04d.ply"), "--frames", str(frames), "--out", str(tmp_path / tag),
                            "--condition", condition, "--devices", devices, "--workers", workers, "--min-width", "256",
                            "--min-height", "256", "--repeat", "2"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert '"frames_per_s"' in r.stdout and ("%d device shard(s)" % len(devices.replace("-0", "").split(","))) in r.stdout, r.stdout
        outs[tag] = {p.name[len(tag):]: p.read_bytes() for p in tmp_path.glob(tag + "*")}
        assert len(outs[tag]) == 5 + frames, sorted(outs[tag])
    assert outs["one"] == outs["two"] == outs["three"]
