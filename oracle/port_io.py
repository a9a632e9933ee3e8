"""oracle/port_io.py -- TEST INFRASTRUCTURE (numpy restatement; never imported by the product).

Point-cloud ingest and conformance checksums:
  * PCCPointSet3::read        (PccLibCommon/source/PCCPointSet.cpp:464-757): ASCII and binary_little_endian PLY
  * PCCPointSet3::computeChecksum / computeMd5 / reorder (PCCPointSet.cpp:222-305): MD5 over int16 positions, then uint8 colours

The reference tells vertex properties apart by NAME and BYTE COUNT only: x / y / z are read as uint16 (2 bytes), float (4) or
double (8) whatever type the header declares, ASCII values go through atof / atoi, vertex properties stop counting at the first
other element, a body shorter than the header announces leaves zeros behind.
"""
import hashlib

import numpy as np

_BYTES = {"double": 8, "float64": 8, "uint64": 8, "int64": 8, "float": 4, "float32": 4, "uint32": 4, "int32": 4, "int": 4,
          "uint16": 2, "int16": 2, "uchar": 1, "uint8": 1, "char": 1, "int8": 1}


def _header(raw, read_normals):
    lines, at = [], 0
    while True:
        nl = raw.index(b"\n", at)
        lines.append(raw[at:nl].replace(b"\t", b" ").replace(b"\r", b" ").split())
        at = nl + 1
        if lines[-1][:1] == [b"end_header"]:
            break
    assert lines[0][:1] == [b"ply"]
    ascii_body, points, vertex, props = False, 0, True, []
    for t in lines[1:]:
        if not t or t[0] == b"comment":
            continue
        if t[0] == b"format":
            ascii_body = t[1] == b"ascii"
        elif t[0] == b"element":
            if t[1] == b"vertex":
                points = int(t[2])
            else:
                vertex = False
        elif t[0] == b"property" and vertex:
            props.append((t[2].decode(), _BYTES[t[1].decode()]))
    idx = {}
    for a, (name, nbytes) in enumerate(props):
        if name in ("x", "y", "z") and nbytes in (8, 4, 2):
            idx[name] = a
        elif name in ("red", "green", "blue") and nbytes == 1:
            idx[name] = a
        elif name in ("nx", "ny", "nz") and nbytes == 4 and read_normals:
            idx[name] = a
    return ascii_body, points, props, idx, at


def ply_read(path, read_normals=False):
    """-> (xyz int16[n][3], rgb uint8[n][3] or None, normals float64[n][3] or None)"""
    raw = open(path, "rb").read()
    ascii_body, n, props, idx, body = _header(raw, read_normals)
    colors = all(k in idx for k in ("red", "green", "blue"))
    normals = all(k in idx for k in ("nx", "ny", "nz"))
    xyz = np.zeros((n, 3), np.int16)
    rgb = np.zeros((n, 3), np.uint8) if colors else None
    nrm = np.zeros((n, 3), np.float64) if normals else None      # (ASCII bodies never fill them)
    if ascii_body:
        rows = [ln.replace(b"\t", b" ").replace(b"\r", b" ").split() for ln in raw[body:].split(b"\n")]
        rows = [r for r in rows if r][:n]
        for i, r in enumerate(rows):                             # small test files only: a plain loop
            assert len(r) >= len(props)
            xyz[i] = [np.int16(np.int32(float(r[idx[k]]))) for k in ("x", "y", "z")]
            if colors:
                rgb[i] = [int(r[idx[k]].split(b".")[0]) & 255 for k in ("red", "green", "blue")]
        return xyz, rgb, nrm
    stride = sum(b for _, b in props)
    offs = np.cumsum([0] + [b for _, b in props])
    rec = np.frombuffer(raw, np.uint8, offset=body)
    m = min(n, len(rec) // stride)
    tail = rec[m * stride:] if m < n else rec[:0]               # a record cut off by the end of the file
    rec = rec[:m * stride].reshape(m, stride)

    def column(a, dtype):
        nb = np.dtype(dtype).itemsize
        return np.ascontiguousarray(rec[:, offs[a]:offs[a] + nb]).view(dtype).reshape(-1)
    for d, k in enumerate(("x", "y", "z")):
        nb = props[idx[k]][1]
        col = column(idx[k], {2: "<u2", 4: "<f4", 8: "<f8"}[nb])
        xyz[:m, d] = col.astype(np.int16) if nb == 2 else col.astype(np.int32).astype(np.int16)
    if colors:
        for d, k in enumerate(("red", "green", "blue")):
            rgb[:m, d] = column(idx[k], np.uint8)
    if normals:
        for d, k in enumerate(("nx", "ny", "nz")):
            nrm[:m, d] = column(idx[k], "<f4")
    if len(tail):   # the reference still stores the properties it could read completely, up to the first one that is cut
        whole = 0
        while whole < len(props) and offs[whole] + props[whole][1] <= len(tail):
            whole += 1
        padded = np.zeros(stride, np.uint8)
        padded[:len(tail)] = tail

        def field(a, dtype):
            return padded[offs[a]:offs[a] + np.dtype(dtype).itemsize].view(dtype)[0]
        for d, k in enumerate(("x", "y", "z")):
            if idx[k] < whole:
                nb = props[idx[k]][1]
                v = field(idx[k], {2: "<u2", 4: "<f4", 8: "<f8"}[nb])
                xyz[m, d] = np.int16(v) if nb == 2 else np.int16(np.int32(v))
        if colors:
            for d, k in enumerate(("red", "green", "blue")):
                if idx[k] < whole:
                    rgb[m, d] = field(idx[k], np.uint8)
        if normals:
            for d, k in enumerate(("nx", "ny", "nz")):
                if idx[k] < whole:
                    nrm[m, d] = field(idx[k], "<f4")
    return xyz, rgb, nrm


def checksum(xyz, rgb=None, reorder=False):
    """-> 16 bytes"""
    xyz = np.ascontiguousarray(xyz, dtype=np.int16)
    if reorder:
        order = np.lexsort((xyz[:, 2], xyz[:, 1], xyz[:, 0]))
        xyz = xyz[order]
        if rgb is not None:
            rgb = np.asarray(rgb, dtype=np.uint8)[order]
            first = np.ones(len(xyz), bool)
            first[1:] = (xyz[1:] != xyz[:-1]).any(1)
            start = np.flatnonzero(first)
            counts = np.diff(np.append(start, len(xyz)))
            sums = np.add.reduceat(rgb.astype(np.uint64), start, axis=0) if len(xyz) else np.zeros((0, 3), np.uint64)
            rgb = (sums // counts[:, None].astype(np.uint64)).astype(np.uint8)
            xyz = xyz[start]
    h = hashlib.md5(np.ascontiguousarray(xyz).tobytes())
    if rgb is not None:
        h.update(np.ascontiguousarray(rgb, dtype=np.uint8).tobytes())
    return h.digest()
