"""INRIA-style `.ply` -> `PlanarGaussian3d`.

Mirror of the reference loader `parse_ply_3d` (src/io/ply.rs:23-132), including its quirks:
  * only `float` properties are consumed (`Property::Float` arms, :29-71); anything else is ignored;
  * required vertex properties (:82-100): x y z f_dc_0..2 scale_0 scale_1 opacity rot_0..3
    (scale_2 is NOT required and defaults to 0 before the exp);
  * `opacity` is stored as a logit -> sigmoid (:40-42);
  * `f_rest_i` (planar per channel in the file) -> interleaved coefficient index with
    channel = i / 16 and coefficient = (i % 15) + 1 (:47-70) -- reproduced literally, although
    with 45 f_rest values the `/ 16` makes i = 15, 31 land in the previous channel;
  * scale: clamp each log-scale to mean +- MAX_SIZE_VARIANCE (= 4), then exp (:105-116);
  * rotation [w,x,y,z] normalised by its L2 norm (:118-124);
  * padded with `Gaussian3d::default()` to a multiple of 32, a FULL extra 32 when already aligned
    (:127-129); default = position 0, visibility 1, everything else 0.
The third-party parser `ply-rs` 0.1.3 (header grammar, ascii / binary_little_endian /
binary_big_endian bodies) is restated here for scalar properties; list properties are skipped
for ascii bodies and rejected for binary ones.
"""
from __future__ import annotations

import io
from typing import BinaryIO, Union

import numpy as np

from .gaussian import SH_CHANNELS, SH_COEFF_COUNT, SH_COEFF_COUNT_PER_CHANNEL, PlanarGaussian3d

MAX_SIZE_VARIANCE = 4.0  # src/io/ply.rs:21

_PLY_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
    "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
    "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}
_REQUIRED = ("x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "scale_0", "scale_1", "opacity",
             "rot_0", "rot_1", "rot_2", "rot_3")


def _read_header(f: BinaryIO):
    first = f.readline().strip()
    if first != b"ply":
        raise ValueError("not a PLY file")
    fmt = None
    elements = []  # (name, count, [(prop_name, dtype or None for list)])
    while True:
        line = f.readline()
        if not line:
            raise ValueError("unexpected end of PLY header")
        tok = line.decode("ascii", "replace").split()
        if not tok or tok[0] in ("comment", "obj_info"):
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append((tok[1], int(tok[2]), []))
        elif tok[0] == "property":
            if tok[1] == "list":  # property list <count type> <item type> <name>
                if tok[2] not in _PLY_TYPES or tok[3] not in _PLY_TYPES:
                    raise ValueError(f"unknown PLY list types {tok[2]} {tok[3]}")
                elements[-1][2].append((tok[4], None, _PLY_TYPES[tok[2]], _PLY_TYPES[tok[3]]))
            else:
                if tok[1] not in _PLY_TYPES:
                    raise ValueError(f"unknown PLY property type {tok[1]}")
                elements[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
        elif tok[0] == "end_header":
            break
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise ValueError(f"unsupported PLY format {fmt}")
    return fmt, elements


def _read_element(f: BinaryIO, fmt: str, count: int, props):
    scalars = [(p[0], p[1]) for p in props if p[1] is not None]
    if fmt == "ascii":
        cols = {n: np.zeros(count, np.float64) for n, t in scalars}
        for r in range(count):
            tok = f.readline().split()
            k = 0
            for p in props:
                if p[1] is None:  # list: count followed by that many items
                    k += 1 + int(tok[k])
                else:
                    cols[p[0]][r] = float(tok[k])
                    k += 1
        return {n: (cols[n].astype(t), t) for n, t in scalars}
    order = "<" if fmt == "binary_little_endian" else ">"
    if any(p[1] is None for p in props):
        # an element with list properties (a trailing `face` element of a mesh export, which ply-rs parses
        # and the reference then ignores) has rows of varying length: walk it row by row
        cols = {n: np.zeros(count, np.dtype(order + t)) for n, t in scalars}
        for r in range(count):
            for p in props:
                if p[1] is None:
                    cdt, idt = np.dtype(order + p[2]), np.dtype(order + p[3])
                    raw = f.read(cdt.itemsize)
                    if len(raw) != cdt.itemsize:
                        raise ValueError("truncated PLY payload")
                    k = int(np.frombuffer(raw, cdt)[0])
                    if k < 0 or len(f.read(k * idt.itemsize)) != k * idt.itemsize:
                        raise ValueError("truncated PLY payload")
                else:
                    dt1 = np.dtype(order + p[1])
                    raw = f.read(dt1.itemsize)
                    if len(raw) != dt1.itemsize:
                        raise ValueError("truncated PLY payload")
                    cols[p[0]][r] = np.frombuffer(raw, dt1)[0]
        return {n: (cols[n], t) for n, t in scalars}
    props = scalars
    dt = np.dtype([(n, order + t) for n, t in props])
    raw = f.read(dt.itemsize * count)
    if len(raw) != dt.itemsize * count:
        raise ValueError("truncated PLY payload")
    arr = np.frombuffer(raw, dtype=dt, count=count)
    return {n: (arr[n], t) for n, t in props}


def parse_ply_3d(source: Union[str, bytes, BinaryIO]) -> PlanarGaussian3d:
    """src/io/ply.rs:76-132."""
    if isinstance(source, (bytes, bytearray)):
        f: BinaryIO = io.BytesIO(source)
    elif isinstance(source, str):
        f = open(source, "rb")
    else:
        f = source
    try:
        fmt, elements = _read_header(f)
        cols = None
        count = 0
        for name, cnt, props in elements:
            data = _read_element(f, fmt, cnt, props)
            if name == "vertex":
                have = {p[0] for p in props}
                if any(r not in have for r in _REQUIRED):
                    raise ValueError("missing required properties")  # ply.rs:93-98
                cols, count = data, cnt
    finally:
        if isinstance(source, str):
            f.close()
    if cols is None:
        cols, count = {}, 0

    def col(name):
        # only `Property::Float` values reach the splat (ply.rs:29-71); other types are ignored
        if name in cols and cols[name][1] == "f4":
            return cols[name][0].astype(np.float32)
        return None

    n = count
    pv = np.zeros((n, 4), np.float32)
    pv[:, 3] = 1.0  # PositionVisibility::default (src/gaussian/f32.rs:58-64)
    sh = np.zeros((n, SH_COEFF_COUNT), np.float32)
    rot = np.zeros((n, 4), np.float32)
    so = np.zeros((n, 4), np.float32)
    for i, k in enumerate(("x", "y", "z", "visibility")):
        c = col(k)
        if c is not None:
            pv[:, i] = c
    for c_idx in range(3):
        c = col(f"f_dc_{c_idx}")
        if c is not None:
            sh[:, c_idx] = c
    for i in range(3):
        c = col(f"scale_{i}")
        if c is not None:
            so[:, i] = c
    c = col("opacity")
    if c is not None:
        so[:, 3] = (np.float32(1.0) / (np.float32(1.0) + np.exp(-c))).astype(np.float32)  # ply.rs:40-42
    for i in range(4):
        c = col(f"rot_{i}")
        if c is not None:
            rot[:, i] = c
    # f_rest_*: file order of the properties matters when two of them map to the same slot
    for name in [p for p in cols if p.startswith("f_rest_")]:
        c = col(name)
        if c is None:
            continue
        i = int(name[7:])
        channel = i // SH_COEFF_COUNT_PER_CHANNEL
        coefficient = 1 if SH_COEFF_COUNT_PER_CHANNEL == 1 else (i % (SH_COEFF_COUNT_PER_CHANNEL - 1)) + 1
        interleaved_idx = coefficient * SH_CHANNELS + channel
        if interleaved_idx < SH_COEFF_COUNT:
            sh[:, interleaved_idx] = c

    # ply.rs:103-125
    mean_scale = ((so[:, 0] + so[:, 1]) + so[:, 2]) / np.float32(3.0)
    for i in range(3):
        so[:, i] = np.exp(np.minimum(np.maximum(so[:, i], mean_scale - np.float32(MAX_SIZE_VARIANCE)),
                                     mean_scale + np.float32(MAX_SIZE_VARIANCE))).astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        norm = np.sqrt((rot.astype(np.float32) ** 2).sum(axis=1, dtype=np.float32)).astype(np.float32)
        rot = (rot / norm[:, None]).astype(np.float32)

    # ply.rs:127-129: pad with defaults to a multiple of 32 (a whole extra block when aligned)
    pad = 32 - (n % 32)
    pv = np.concatenate([pv, np.tile(np.array([[0, 0, 0, 1]], np.float32), (pad, 1))])
    sh = np.concatenate([sh, np.zeros((pad, SH_COEFF_COUNT), np.float32)])
    rot = np.concatenate([rot, np.zeros((pad, 4), np.float32)])
    so = np.concatenate([so, np.zeros((pad, 4), np.float32)])
    return PlanarGaussian3d(pv, sh, rot, so)


def write_ply_3d(cloud: PlanarGaussian3d, path: str, binary: bool = True) -> None:
    """Write an INRIA-style PLY that `parse_ply_3d` maps back to `cloud` (up to f32 rounding of
    logit/log): opacity as logit, scale as log, SH DC + the 45 f_rest in the file's planar order
    restricted to the slots the reference's mapping can address. Tooling/test helper; the
    reference itself only reads this format."""
    n = len(cloud)
    names = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"]
    cols = [cloud.position_visibility[:, 0], cloud.position_visibility[:, 1], cloud.position_visibility[:, 2],
            cloud.spherical_harmonic[:, 0], cloud.spherical_harmonic[:, 1], cloud.spherical_harmonic[:, 2]]
    for i in range(45):
        channel = i // SH_COEFF_COUNT_PER_CHANNEL
        coefficient = (i % (SH_COEFF_COUNT_PER_CHANNEL - 1)) + 1
        names.append(f"f_rest_{i}")
        cols.append(cloud.spherical_harmonic[:, coefficient * SH_CHANNELS + channel])
    with np.errstate(divide="ignore", invalid="ignore"):
        op = cloud.scale_opacity[:, 3].astype(np.float64)
        logit = np.log(op / (1.0 - op))
        names.append("opacity"); cols.append(logit.astype(np.float32))
        for i in range(3):
            names.append(f"scale_{i}"); cols.append(np.log(cloud.scale_opacity[:, i].astype(np.float64)).astype(np.float32))
    for i in range(4):
        names.append(f"rot_{i}"); cols.append(cloud.rotation[:, i])
    header = ["ply", "format binary_little_endian 1.0" if binary else "format ascii 1.0",
              f"element vertex {n}"] + [f"property float {k}" for k in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        data = np.stack([np.asarray(c, np.float32) for c in cols], axis=1)
        if binary:
            f.write(data.astype("<f4").tobytes())
        else:
            for row in data:
                f.write((" ".join(repr(float(v)) for v in row) + "\n").encode("ascii"))
