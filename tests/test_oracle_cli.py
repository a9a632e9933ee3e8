"""CLI-level pin of the oracle: the reference's own PccAppEncoder -- the whole PCCEncoder::encode(), built by
oracle/Makefile from the sources where they lie -- runs a 2-frame GOF with an identity "video codec" behind its HMAPP
wrapper and logs, under CONFORMANCE_TRACE, the MD5 of every picture it hands to / gets back from the codec
(PCCVideoEncoder.cpp:389-396) and of every reconstructed cloud (PCCEncoder.cpp:620-626, 714-718).  With the identity codec
decoded == generated, so these logs pin S11-S22 and the post-reconstruction tail exactly as encode() orders them.  The harness
(oracle/ref_harness.cpp) drives the same members stage by stage and transcribes a few inline lines of encode(); the oracle
restates them.  Both must give the CLI's bytes."""
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

import tmc2_amd as T
from tmc2_amd.synth import synth_cloud

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "oracle", "_ref", "PccAppEncoder")
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not (os.path.exists(APP) and os.path.isdir(os.path.join(REF, "cfg"))),
                                reason="needs the reference tree and oracle/_ref/PccAppEncoder (make -C oracle ref)")

STUB = """#!/bin/sh
# identity "video codec" behind the reference's HMAPP wrapper (PCCHMAppVideoEncoder.cpp:59-90): reconstruction = input
for a in "$@"; do case $a in --InputFile=*) IN=${a#*=};; --ReconFile=*) REC=${a#*=};; --BitstreamFile=*) BIN=${a#*=};; esac; done
cp "$IN" "$REC"; printf '\\000\\000\\000\\001\\100\\001\\014\\001' > "$BIN"
"""


def md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_cli(tmp, frames, iterations):
    stub = os.path.join(tmp, "stub.sh")
    with open(stub, "w") as f:
        f.write(STUB)
    os.chmod(stub, 0o755)
    for i, (xyz, rgb) in enumerate(frames):
        T.ply_write(os.path.join(tmp, "in_%04d.ply" % i), xyz, rgb, ascii=False)
    cfg = os.path.join(REF, "cfg")
    cmd = [APP, "--configurationFolder=" + cfg + "/", "--config=" + cfg + "/common/ctc-common.cfg",
           "--config=" + cfg + "/condition/ctc-all-intra.cfg", "--config=" + cfg + "/sequence/longdress_vox10.cfg",
           "--config=" + cfg + "/rate/ctc-r3.cfg", "--uncompressedDataPath=" + os.path.join(tmp, "in_%04d.ply"),
           "--startFrameNumber=0", "--frameCount=%d" % len(frames), "--groupOfFramesSize=%d" % len(frames),
           "--iterationCountRefineSegmentation=%d" % iterations,
           "--videoEncoderOccupancyCodecId=HMAPP", "--videoEncoderGeometryCodecId=HMAPP", "--videoEncoderAttributeCodecId=HMAPP",
           "--videoEncoderOccupancyPath=" + stub, "--videoEncoderGeometryPath=" + stub, "--videoEncoderAttributePath=" + stub,
           "--keepIntermediateFiles=1", "--nbThread=1", "--computeMetrics=0", "--computeChecksum=0",
           "--reconstructedDataPath=" + os.path.join(tmp, "rec_%04d.ply"), "--compressedStreamPath=" + os.path.join(tmp, "S.bin")]
    subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    pictures, section = {}, None
    with open(os.path.join(tmp, "S_enc_picture_log.txt")) as f:
        for line in f:
            if line.strip() in ("Occupancy", "Geometry", "Attribute"):
                section = line.strip()
                pictures[section] = []
            m = re.findall(r"MD5checksumChan\d = ([0-9a-f]{32})", line)
            if m:
                pictures[section].append(m)
    with open(os.path.join(tmp, "S_enc_pcframe_log.txt")) as f:
        counts = [int(x) for x in re.findall(r"NumProjPoints = (\d+)", f.read())]
    with open(os.path.join(tmp, "S_enc_rec_pcframe_log.txt")) as f:
        rec = re.findall(r"MD5 checksum = ([0-9a-f]{32})", f.read())
    return pictures, counts, rec


@pytest.mark.parametrize("engine", ["reference", "oracle"])
def test_oracle_cli_picture_md5s(tmp_path, reference, oracle, engine):
    frames, iters = [synth_cloud("tiny", f) for f in range(2)], 10
    pictures, counts, rec = run_cli(str(tmp_path), frames, iters)
    eng = reference if engine == "reference" else oracle
    a = eng.phase_a(frames, iters, 11, 4)
    b = eng.phase_b(frames, a, 4)
    # occupancy video: one 8-bit picture per frame (chroma planes of a 4:2:0 picture: zeros)
    assert [p[0] for p in pictures["Occupancy"]] == [md5(x["occ_video"].astype(np.uint8)) for x in a]
    # geometry video: D0 and D1 of every frame, 16-bit samples
    assert [p[0] for p in pictures["Geometry"]] == [md5(x[k].astype(np.uint16)) for x in a for k in ("geo0", "geo1")]
    # attribute video: both maps of every frame after RGB444 -> YUV420 (filter 4), 16-bit samples, all three planes
    exp, dec = [], []
    for x in b:
        planes = []
        for m in range(2):
            y, u, v = eng.convert_rgb444_to_yuv420(x["attribute"][m], 4)
            exp.append([md5(c.astype(np.uint16)) for c in (y, u, v)])
            planes.append(eng.convert_yuv420_to_yuv444(y, u, v, 0))
        dec.append(np.stack(planes))
    assert pictures["Attribute"] == exp
    assert counts == [len(x["recon_xyz"]) for x in b]
    # the finished clouds (geometry smoothing, 16-bit colours from the decoded attribute video, colour transfer onto the moved
    # points, YUV -> RGB): PCCPointSet3::computeChecksum( reorder = true ) of every reconstructed frame
    c = reference.phase_c(b, dec) if engine == "reference" else oracle.phase_c(a, b, dec, 4)
    assert rec == [reference.checksum(x["xyz"], x["rgb"], True).hex() for x in c]
