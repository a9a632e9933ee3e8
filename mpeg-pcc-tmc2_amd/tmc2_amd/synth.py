"""Deterministic synthetic voxelised point clouds (the MPEG 8i/Owlii data sets are not redistributable).

`synth_cloud(name, frame)` returns (xyz int16 [N,3], rgb uint8 [N,3]) with UNIQUE integer positions
(the reference's CTC inputs are voxelised, duplicate-free clouds -- SURVEY.md section 7.3-4), a humanoid
made of capsules/ellipsoids with sinusoidal surface bumps, and a low-frequency colour field plus
per-voxel hash noise.  Sizes follow SURVEY.md section 8(d):
    longdress_vox10-like : bits=10, N ~ 0.8 M      basketball_vox11-like : bits=11, N ~ 2.9 M
"""
import numpy as np

# (centre a, centre b, radius) in a unit-height body frame (y up, body height ~1)
_CAPSULES = [
    ((0.00, 0.93, 0.00), (0.00, 0.93, 0.00), 0.075),   # head
    ((0.00, 0.82, 0.00), (0.00, 0.86, 0.00), 0.035),   # neck
    ((0.00, 0.55, 0.00), (0.00, 0.76, 0.00), 0.120),   # torso
    ((0.00, 0.45, 0.00), (0.00, 0.52, 0.00), 0.125),   # hips
    ((-0.15, 0.76, 0.00), (-0.27, 0.55, 0.03), 0.040),  # upper arm L
    ((-0.27, 0.55, 0.03), (-0.30, 0.36, 0.10), 0.033),  # fore arm L
    ((0.15, 0.76, 0.00), (0.27, 0.55, 0.03), 0.040),   # upper arm R
    ((0.27, 0.55, 0.03), (0.30, 0.36, 0.10), 0.033),   # fore arm R
    ((-0.07, 0.45, 0.00), (-0.09, 0.24, 0.02), 0.060),  # thigh L
    ((-0.09, 0.24, 0.02), (-0.10, 0.03, 0.00), 0.043),  # shin L
    ((0.07, 0.45, 0.00), (0.09, 0.24, 0.02), 0.060),   # thigh R
    ((0.09, 0.24, 0.02), (0.10, 0.03, 0.00), 0.043),   # shin R
]

_PRESETS = {
    # name: (bits, body height in voxels, seed base)
    "tiny": (10, 90, 7),          # ~6 k points   (unit tests)
    "small": (10, 170, 11),       # ~20 k points  (golden vectors)
    "medium": (10, 500, 13),      # ~180 k points
    "longdress_vox10": (10, 1000, 1051),   # ~0.8 M
    "loot_vox10": (10, 990, 1000),
    "redandblack_vox10": (10, 950, 1450),
    "soldier_vox10": (10, 1010, 536),
    "basketball_player_vox11": (11, 1900, 1),  # ~2.9 M
}


def _hash_u32(x):
    x = (x ^ (x >> 16)) * np.uint32(0x7FEB352D)
    x = (x ^ (x >> 15)) * np.uint32(0x846CA68B)
    return x ^ (x >> 16)


def _sdf_capsule(p, a, b, r):
    ab = b - a
    den = float(ab @ ab)
    t = np.clip(((p - a) @ ab) / den, 0.0, 1.0) if den > 0 else np.zeros(len(p))
    c = a + t[:, None] * ab
    return np.linalg.norm(p - c, axis=1) - r


def synth_cloud(name="small", frame=0, height=None, bits=None, seed=None):
    """name + "_noisy": the same body with per-sample jitter along the surface normal on the scale of the voxel pitch (sigma = 0.8
    voxels) -- a thick, rough shell, the way a reconstructed capture looks next to a smooth CAD-like surface: normals of
    neighbouring points disagree, the orientation's strong-edge contraction (S3) meets inconsistent cycles and its threshold ladder
    / point-level fallback get exercised (bench.py --workload longdress_vox10_noisy reports how often)."""
    noisy = name.endswith("_noisy")
    if noisy:
        name = name[:-len("_noisy")]
    pb, ph, ps = _PRESETS[name] if name in _PRESETS else (10, 170, 11)
    bits = pb if bits is None else bits
    H = float(ph if height is None else height)
    seed = (ps if seed is None else seed) + frame
    rng = np.random.default_rng(seed)
    size = 1 << bits
    phase = 0.35 * np.sin(0.2 * frame + np.arange(len(_CAPSULES)))  # per-frame limb sway
    origin = np.array([size * 0.5, (size - H) * 0.5 if H < size else 0.0, size * 0.5])
    caps = []
    for k, (a, b, r) in enumerate(_CAPSULES):
        a = np.array(a, dtype=np.float64)
        b = np.array(b, dtype=np.float64)
        if k >= 4:  # limbs sway in z
            b = b + np.array([0.0, 0.0, 0.06 * phase[k]])
        caps.append((a * H + origin, b * H + origin, 0.72 * r * H))
    pts = []
    for k, (a, b, r) in enumerate(caps):
        ab = b - a
        L = float(np.linalg.norm(ab))
        # sample the capsule surface densely (~3 samples per voxel area), with bumps along the normal
        area = 2 * np.pi * r * L + 4 * np.pi * r * r
        ns = int(area * 3.0) + 64
        ax = ab / L if L > 0 else np.array([0.0, 1.0, 0.0])
        tmp = np.array([1.0, 0.0, 0.0]) if abs(ax[0]) < 0.9 else np.array([0.0, 0.0, 1.0])
        e1 = np.cross(ax, tmp)
        e1 /= np.linalg.norm(e1)
        e2 = np.cross(ax, e1)
        frac_cyl = (2 * np.pi * r * L) / area
        ncyl = int(ns * frac_cyl)
        # cylinder part
        t = rng.random(ncyl)
        th = rng.random(ncyl) * 2 * np.pi
        nrm = np.cos(th)[:, None] * e1 + np.sin(th)[:, None] * e2
        base = a + t[:, None] * ab
        pc = np.concatenate([base, ], 0)
        nc = nrm
        # sphere caps (full spheres at both ends; the interior halves are removed by the SDF test)
        nsph = ns - ncyl
        v = rng.normal(size=(nsph, 3))
        v /= np.linalg.norm(v, axis=1)[:, None]
        ends = np.where(rng.random(nsph)[:, None] < 0.5, a, b)
        pc = np.concatenate([pc, ends], 0)
        nc = np.concatenate([nc, v], 0)
        surf = pc + r * nc
        bump = (0.008 * H) * np.sin(surf[:, 0] * (37.0 / H) + k) * np.sin(surf[:, 1] * (29.0 / H)) * np.sin(
            surf[:, 2] * (31.0 / H) + 0.1 * frame)
        if noisy:
            bump = bump + 0.8 * rng.normal(size=len(bump))
        surf = np.rint(surf + bump[:, None] * nc).astype(np.int64)
        surf = surf[np.all((surf >= 0) & (surf < size), axis=1)]
        kk = np.unique(surf[:, 0] | (surf[:, 1] << 12) | (surf[:, 2] << 24))
        surf = np.stack([kk & 0xFFF, (kk >> 12) & 0xFFF, (kk >> 24) & 0xFFF], 1).astype(np.float64)
        # keep only voxels that are outside every other primitive
        keep = np.ones(len(surf), dtype=bool)
        for j, (a2, b2, r2) in enumerate(caps):
            if j != k:
                lo = np.minimum(a2, b2) - r2 - 1
                hi = np.maximum(a2, b2) + r2 + 1
                cand = np.nonzero(keep & np.all((surf >= lo) & (surf <= hi), axis=1))[0]
                if len(cand):
                    keep[cand] = _sdf_capsule(surf[cand], a2, b2, r2) > -0.5
        pts.append(surf[keep])
    p = np.concatenate(pts, 0).astype(np.int64)
    key = p[:, 0] | (p[:, 1] << 12) | (p[:, 2] << 24)
    key = np.unique(key)
    xyz = np.stack([key & 0xFFF, (key >> 12) & 0xFFF, (key >> 24) & 0xFFF], 1).astype(np.int16)
    # colour: low-frequency field + per-voxel hash noise
    f = xyz.astype(np.float64) / H
    base = np.stack([
        128 + 90 * np.sin(6.0 * f[:, 1] + 1.0) * np.cos(5.0 * f[:, 0]),
        128 + 90 * np.sin(7.0 * f[:, 2] + 2.0 * f[:, 1]),
        128 + 90 * np.cos(4.0 * f[:, 0] - 3.0 * f[:, 2] + 0.5),
    ], 1)
    h = _hash_u32(key.astype(np.uint32) * np.uint32(2654435761) + np.uint32(seed))
    noise = np.stack([(h & 0x1F), ((h >> 5) & 0x1F), ((h >> 10) & 0x1F)], 1).astype(np.float64) - 16.0
    rgb = np.clip(np.rint(base + noise), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(xyz), np.ascontiguousarray(rgb)


def synth_gof(name, frame_count):
    return [synth_cloud(name, f) for f in range(frame_count)]


def two_body_gof(name="tiny", frames=5, seed=0):
    """A GOF of two bodies: one that stays put (its patches keep matching from frame to frame) and one that jumps to an
    unrelated place every frame (its patches never match).  The inter-frame packers (low-delay spatial consistency,
    global patch allocation) see long patch tracks AND a large unmatched remainder, which is what drives the global
    patch allocation into its restart branches on small canvases."""
    rng = np.random.default_rng(1000 + seed)
    out = []
    for f in range(frames):
        a, ca = synth_cloud(name, f)
        b, cb = synth_cloud(name, (7 * f + 3) % 11)
        lo, hi = b.min(0).astype(np.int64), b.max(0).astype(np.int64)
        while True:  # a random resting place whose bounding box is clear of the first body's
            at = rng.integers(0, 1024 - (hi - lo))
            if np.any(at + (hi - lo) < a.min(0)) or np.any(at > a.max(0)):
                break
        xyz = np.concatenate([a, (b + (at - lo)).astype(a.dtype)])
        out.append((np.ascontiguousarray(xyz), np.ascontiguousarray(np.concatenate([ca, cb]))))
    return out


def synth_decoded_attribute(attribute):
    """Stand-in for the attribute video codec + colour conversion: the 8-bit RGB attribute canvases ([2][3][H][W]) as the
    16-bit YUV 4:4:4 frames a decoder hands to the reconstruction (BT.709, full range, chroma centred at 32768), with a
    small deterministic coding error on top."""
    a = np.asarray(attribute, dtype=np.float64) / 255.0
    r, g, b = a[:, 0], a[:, 1], a[:, 2]
    y = 0.2126 * r + 0.7152 * g + 0.0722 * b
    u = (b - y) / 1.8556
    v = (r - y) / 1.5748
    H, W = y.shape[-2:]
    ripple = ((np.arange(H)[:, None] * 7 + np.arange(W)[None, :] * 13) % 11 - 5) * 48.0
    out = np.stack([y * 65535.0 + ripple, u * 65535.0 + 32768.0 - ripple, v * 65535.0 + 32768.0 + ripple], 1)
    return np.clip(np.rint(out), 0, 65535).astype(np.uint16)
