"""PLY files for the ingest tests: every header / body variant the reference's reader (PCCPointSet3::read) distinguishes."""
import numpy as np


def _header(fmt, n, props, extra_element=True, comments=True, eol="\n"):
    lines = ["ply", "format %s 1.0" % fmt]
    if comments:
        lines += ["comment made by tests/ply_cases.py", "comment   another one"]
    lines += ["element vertex %d" % n] + ["property %s %s" % p for p in props]
    if extra_element:
        lines += ["element face 0", "property list uint8 int32 vertex_index"]
    lines += ["end_header"]
    return (eol.join(lines) + eol).encode()


def cases(xyz, rgb, seed=0):
    """-> {name: file bytes}.  xyz int16[n][3] (non-negative), rgb uint8[n][3]."""
    rng = np.random.default_rng(seed)
    n = len(xyz)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    out = {}
    # ASCII, float coordinates as PCCPointSet3::write prints them
    body = "".join("%f %f %f %d %d %d\n" % (*p, *c) for p, c in zip(xyz.tolist(), rgb.tolist()))
    out["ascii_float"] = _header("ascii", n, [("float", "x"), ("float", "y"), ("float", "z"), ("uchar", "red"), ("uchar", "green"),
                                             ("uchar", "blue")]) + body.encode()
    # ASCII with fractions, exponents, signs, tabs, CRLF, blank lines, extra columns and a different property order
    rows = []
    for i, (p, c) in enumerate(zip(xyz.tolist(), rgb.tolist())):
        fx = "%.3f" % (p[0] + 0.999) if i % 3 == 0 else ("%de0" % p[0] if i % 3 == 1 else "+%d" % p[0])
        fy = "%.17g" % (p[1] + 0.25)
        fz = "%d.000" % p[2]
        rows.append("%d\t%s %s  %d %s 7 %d %f %f %f\r" % (c[0], fx, fy, c[1], fz, c[2], *nrm[i]))
        if i % 50 == 7:
            rows.append(" \t\r")
    out["ascii_mixed"] = _header("ascii", n, [("uchar", "red"), ("double", "x"), ("float", "y"), ("uint8", "green"), ("float32", "z"),
                                             ("uchar", "alpha"), ("uchar", "blue"), ("float", "nx"), ("float", "ny"), ("float", "nz")],
                                 eol="\r\n") + ("\n".join(rows) + "\n").encode()
    # ASCII without colours, fewer body lines than announced
    out["ascii_short"] = _header("ascii", n, [("float", "x"), ("float", "y"), ("float", "z")], extra_element=False) + \
        "".join("%d %d %d\n" % tuple(p) for p in xyz[:n // 2].tolist()).encode()
    # binary: float coordinates, float normals, colours (the layout PCCPointSet3::write produces)
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    for k, d in zip(("x", "y", "z"), range(3)):
        rec[k] = xyz[:, d] + np.float32(0.5)
    for k, d in zip(("nx", "ny", "nz"), range(3)):
        rec[k] = nrm[:, d]
    rec["r"], rec["g"], rec["b"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    out["binary_float_normals"] = _header("binary_little_endian", n, [("float", "x"), ("float", "y"), ("float", "z"), ("float", "nx"),
                                                                      ("float", "ny"), ("float", "nz"), ("uchar", "red"),
                                                                      ("uchar", "green"), ("uchar", "blue")]) + rec.tobytes()
    # binary: double x, 2-byte y, float z, an 8-byte and a 2-byte property to skip, colours out of order
    rec = np.zeros(n, dtype=[("b", "u1"), ("x", "<f8"), ("skip8", "<i8"), ("y", "<i2"), ("r", "u1"), ("z", "<f4"), ("skip2", "<u2"), ("g", "u1")])
    rec["x"], rec["y"], rec["z"] = xyz[:, 0] + 0.75, xyz[:, 1], xyz[:, 2] + np.float32(0.125)
    rec["r"], rec["g"], rec["b"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    rec["skip8"], rec["skip2"] = np.arange(n), 77
    out["binary_mixed"] = _header("binary_little_endian", n, [("uchar", "blue"), ("double", "x"), ("int64", "timestamp"), ("int16", "y"),
                                                              ("uchar", "red"), ("float", "z"), ("uint16", "label"), ("uchar", "green")],
                                  comments=False) + rec.tobytes()
    # binary: coordinates declared int32 -- the reference reads any 4-byte coordinate as float
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    out["binary_int32_named"] = _header("binary_little_endian", n, [("int32", "x"), ("int", "y"), ("uint32", "z")]) + rec.tobytes()
    # binary, truncated inside a record, right after its coordinates (27-byte records: 12 + 12 + 3)
    out["binary_short"] = out["binary_float_normals"][:-(27 * (n // 3) + 15)]
    return out
