"""VTK XML PolyData feature files (the reference's ``<dir>_00.vtk``) without VTK.

Reading goes through the C++ reader of ``include/lrr_amd/vtp_features.h`` (``read_vtp`` +
``features_from_vtp``, the look-alike of ``cdcl/io/vtkPolyDataToFeaturesICPFilter.cxx:129-237``).
``write_vtp`` is an independent pure-Python writer of the same format — every encoding
``vtkXMLPolyDataWriter`` can be configured to produce (ascii, inline base64, appended base64 / raw,
zlib blocks, 32- or 64-bit headers) — used to generate test fixtures and to export synthetic clouds
for the reference's own ``location_registration``.  Arrays and types follow
``features/vtkITKPointSetToPolyDataFilter.cxx:147-250``: points, ``errorProjectors`` (9), ``strengths``,
``normals`` (3), ``binormals`` (3) as Float32, ``shapes`` as UInt8.
"""
from __future__ import annotations

import base64
import ctypes as C
import struct
import zlib

import numpy as np

from . import estimation as E
from . import synth

_dp = C.POINTER(C.c_double)
_bp = C.POINTER(C.c_uint8)
FORMATS = ("ascii", "binary", "appended-base64", "appended-raw")


def _lib():
    L = E.est_lib()
    if not getattr(L, "_vtp_ready", False):
        L.lrr_vtp_open.restype = C.c_void_p
        L.lrr_vtp_open.argtypes = [C.c_char_p]
        L.lrr_vtp_close.argtypes = [C.c_void_p]
        L.lrr_vtp_count.argtypes = [C.c_void_p]
        L.lrr_vtp_count.restype = C.c_int64
        L.lrr_vtp_arrays.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int)]
        L.lrr_vtp_features.argtypes = [C.c_void_p, _dp, _dp, _dp, _bp, _bp, _dp]
        L._vtp_ready = True
    return L


def _d(a):
    return a.ctypes.data_as(_dp)


def read_vtp_arrays(path):
    """The arrays of the file as stored: dict of points (n,3), errorProjectors (n,9), strengths (n,), shapes (n,),
    normals (n,3), binormals (n,3); absent arrays are None."""
    L = _lib()
    h = L.lrr_vtp_open(str(path).encode())
    if not h:
        raise RuntimeError(L.lrr_est_last_error().decode())
    try:
        n = L.lrr_vtp_count(h)
        out = dict(points=np.zeros((n, 3)), errorProjectors=np.zeros((n, 9)), strengths=np.zeros(n), shapes=np.zeros(n),
                   normals=np.zeros((n, 3)), binormals=np.zeros((n, 3)))
        present = (C.c_int * 6)()
        if L.lrr_vtp_arrays(h, *[_d(out[k]) for k in ("points", "errorProjectors", "strengths", "shapes", "normals", "binormals")],
                            present) != 0:
            raise RuntimeError(L.lrr_est_last_error().decode())
        for k, name in enumerate(("points", "errorProjectors", "strengths", "shapes", "normals", "binormals")):
            if not present[k]:
                out[name] = None
        return out
    finally:
        L.lrr_vtp_close(h)


def read_features(path, return_all_directions=False):
    """The feature vector the reference's filter builds from the file, as a synth.FeatureSet (dirs = directions_[0],
    directions_[1]); with return_all_directions also (ndirs, dirs9)."""
    L = _lib()
    h = L.lrr_vtp_open(str(path).encode())
    if not h:
        raise RuntimeError(L.lrr_est_last_error().decode())
    try:
        n = L.lrr_vtp_count(h)
        loc, proj, st = np.zeros((n, 3)), np.zeros((n, 9)), np.zeros(n)
        shape, ndirs, dirs9 = np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros((n, 9))
        if L.lrr_vtp_features(h, _d(loc), _d(proj), _d(st), shape.ctypes.data_as(_bp), ndirs.ctypes.data_as(_bp), _d(dirs9)) != 0:
            raise RuntimeError(L.lrr_est_last_error().decode())
    finally:
        L.lrr_vtp_close(h)
    fs = synth.FeatureSet(loc, proj, st, shape, np.ascontiguousarray(dirs9[:, :6]))
    return (fs, ndirs, dirs9) if return_all_directions else fs


def disk_arrays(fs: synth.FeatureSet):
    """The on-disk arrays of a feature set (features/vtkITKPointSetToPolyDataFilter.cxx:184-236): a SHEET stores its
    normal and a zero binormal; a TUBE its two normals, which the reader hands back as directions_ = {binormal,
    normal}; a CORNER normal and binormal (its directions_[0] = normal x binormal is rebuilt by the reader)."""
    n = fs.n
    normals, binormals = np.zeros((n, 3)), np.zeros((n, 3))
    sheet, tube, corner = fs.shape == synth.SHEET, fs.shape == synth.TUBE, fs.shape == synth.CORNER
    normals[sheet] = fs.dirs[sheet, 0:3]
    binormals[tube] = fs.dirs[tube, 0:3]
    normals[tube] = fs.dirs[tube, 3:6]
    normals[corner] = fs.dirs[corner, 0:3]
    binormals[corner] = fs.dirs[corner, 3:6]
    return dict(points=fs.loc, errorProjectors=fs.proj, strengths=fs.strength, shapes=fs.shape, normals=normals, binormals=binormals)


def _encode_block(raw: bytes, compress: bool, hfmt: str, b64: bool, block: int = 32768) -> bytes:
    if not compress:
        head = struct.pack(hfmt, len(raw))
        return base64.b64encode(head) + base64.b64encode(raw) if b64 else head + raw  # VTK encodes the length word on its own
    blocks = [raw[i:i + block] for i in range(0, len(raw), block)] or [b""]
    comp = [zlib.compress(b) for b in blocks]
    last = len(blocks[-1]) if len(blocks[-1]) != block else 0
    head = struct.pack("<" + hfmt[1] * (3 + len(comp)), len(comp), block, last, *[len(c) for c in comp])
    body = b"".join(comp)
    return base64.b64encode(head) + base64.b64encode(body) if b64 else head + body


def write_vtp(path, arrays, fmt="appended-raw", compress=False, header_type="UInt32", joint_header=False):
    """arrays: dict as disk_arrays() returns (any of the point-data arrays may be missing).  joint_header: encode
    length word and data as ONE base64 unit (what some non-VTK writers do) instead of VTK's two."""
    assert fmt in FORMATS and header_type in ("UInt32", "UInt64")
    hfmt = "<I" if header_type == "UInt32" else "<Q"
    pts = np.ascontiguousarray(arrays["points"], dtype=np.float32).reshape(-1, 3)
    n = pts.shape[0]
    spec = [("Points", None, pts, "Float32", 3)]
    for name, dt, vt, nc in (("strengths", np.float32, "Float32", 1), ("shapes", np.uint8, "UInt8", 1),
                             ("normals", np.float32, "Float32", 3), ("binormals", np.float32, "Float32", 3),
                             ("errorProjectors", np.float32, "Float32", 9)):
        if arrays.get(name) is not None:
            spec.append(("PointData", name, np.ascontiguousarray(arrays[name], dtype=dt).reshape(n, nc), vt, nc))
    appended, chunks = fmt.startswith("appended"), []

    def element(name, a, vt, nc):
        nm = f' Name="{name}"' if name else ""
        if fmt == "ascii":
            body = "\n".join(" ".join(repr(float(x)) if vt != "UInt8" else str(int(x)) for x in row) for row in a)
            return f'<DataArray type="{vt}"{nm} NumberOfComponents="{nc}" format="ascii">\n{body}\n</DataArray>\n'
        raw = a.tobytes()
        if joint_header and not compress and fmt != "appended-raw":
            enc = base64.b64encode(struct.pack(hfmt, len(raw)) + raw)
        else:
            enc = _encode_block(raw, compress, hfmt, fmt != "appended-raw")
        if fmt == "binary":
            return f'<DataArray type="{vt}"{nm} NumberOfComponents="{nc}" format="binary">\n{enc.decode()}\n</DataArray>\n'
        off = sum(len(c) for c in chunks)
        chunks.append(enc)
        return f'<DataArray type="{vt}"{nm} NumberOfComponents="{nc}" format="appended" offset="{off}"/>\n'

    comp_attr = ' compressor="vtkZLibDataCompressor"' if compress and fmt != "ascii" else ""
    x = ['<?xml version="1.0"?>\n',
         f'<VTKFile type="PolyData" version="0.1" byte_order="LittleEndian" header_type="{header_type}"{comp_attr}>\n',
         "<PolyData>\n",
         f'<Piece NumberOfPoints="{n}" NumberOfVerts="0" NumberOfLines="0" NumberOfStrips="0" NumberOfPolys="0">\n',
         '<PointData Scalars="shapes" Vectors="binormals" Tensors="errorProjectors">\n']
    for sec, name, a, vt, nc in spec[1:]:
        x.append(element(name, a, vt, nc))
    x.append("</PointData>\n<CellData>\n</CellData>\n<Points>\n")
    x.append(element(None, spec[0][2], "Float32", 3))
    x.append("</Points>\n</Piece>\n</PolyData>\n")
    head = "".join(x).encode()
    with open(path, "wb") as f:
        f.write(head)
        if appended:
            f.write(b'<AppendedData encoding="%s">\n_' % (b"raw" if fmt == "appended-raw" else b"base64"))
            f.write(b"".join(chunks))
            f.write(b"\n</AppendedData>\n")
        f.write(b"</VTKFile>\n")
