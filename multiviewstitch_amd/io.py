"""The reference's file formats (include/mvs_io.h) with its float32 text quantisation.

Names follow the reference: ``ReadObj`` / ``WriteObj`` (R/PlyObj/PlyObj.cpp:6-137), ``LoadDepth`` / ``SaveDepth``
(R/Common/Utils.h:166-185), ``LoadParts`` (R/PartRecognition/PartRecognition.cpp:7-48); the ``.npts`` and
``SRT.txt`` helpers cover the inline stream code at R/Processor/Processor.cpp:855-871, 958-963, 1033-1040, 1145-1165.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L


def _p(path) -> bytes:
    return os.fsencode(path)


def ReadObj(path):
    """-> (points[V,3] f64, normals[N,3] f64, facets[F,3] i32); values carry float32 precision, normals are unit."""
    nv, nn, nf = C.c_int64(), C.c_int64(), C.c_int64()
    L.check(L.lib().mvs_obj_read(_p(path), C.byref(nv), C.byref(nn), C.byref(nf), None, None, None))
    pts, nrm, fac = np.empty((nv.value, 3)), np.empty((nn.value, 3)), np.empty((nf.value, 3), np.int32)
    L.check(L.lib().mvs_obj_read(_p(path), C.byref(nv), C.byref(nn), C.byref(nf), L.ptr(pts), L.ptr(nrm), L.ptr(fac)))
    return pts, nrm, fac


def WriteObj(path, points, normals=None, facets=None):
    pts = L.arr(points, np.float64).reshape(-1, 3)
    nrm = None if normals is None or len(normals) != len(pts) else L.arr(normals, np.float64).reshape(-1, 3)
    fac = np.empty((0, 3), np.int32) if facets is None else L.arr(facets, np.int32).reshape(-1, 3)
    L.check(L.lib().mvs_obj_write(_p(path), len(pts), L.ptr(pts), L.ptr(nrm), len(fac), L.ptr(fac)))


def read_npts(path):
    n = C.c_int64()
    L.check(L.lib().mvs_npts_read(_p(path), C.byref(n), None, None))
    pts, nrm = np.empty((n.value, 3)), np.empty((n.value, 3))
    L.check(L.lib().mvs_npts_read(_p(path), C.byref(n), L.ptr(pts), L.ptr(nrm)))
    return pts, nrm


def write_npts(path, points, normals):
    pts, nrm = L.arr(points, np.float64).reshape(-1, 3), L.arr(normals, np.float64).reshape(-1, 3)
    L.check(L.lib().mvs_npts_write(_p(path), len(pts), L.ptr(pts), L.ptr(nrm)))


def read_srt_txt(path, n_seq: int):
    s, R, t = np.empty(n_seq), np.empty((n_seq, 3, 3)), np.empty((n_seq, 3))
    L.check(L.lib().mvs_srt_txt_read(_p(path), n_seq, L.ptr(s), L.ptr(R), L.ptr(t)))
    return s, R, t


def write_srt_txt(path, scales, Rs, ts):
    s = L.arr(scales, np.float64).reshape(-1)
    R, t = L.arr(Rs, np.float64).reshape(len(s), 3, 3), L.arr(ts, np.float64).reshape(len(s), 3)
    L.check(L.lib().mvs_srt_txt_write(_p(path), len(s), L.ptr(s), L.ptr(R), L.ptr(t)))


def LoadDepth(path, w: int, h: int) -> np.ndarray:
    """float32 raster [h, w] (the reference widens to double afterwards; the engine consumes the float32 file)."""
    d = np.empty((h, w), np.float32)
    L.check(L.lib().mvs_depth_raw_read(_p(path), w, h, L.ptr(d)))
    return d


def SaveDepth(path, depth):
    d = L.arr(depth, np.float64).reshape(-1)
    L.check(L.lib().mvs_depth_raw_write(_p(path), len(d), L.ptr(d)))


def LoadParts(path, n_vertices: int) -> np.ndarray:
    lab = np.empty(n_vertices, np.int32)
    L.check(L.lib().mvs_parts_read(_p(path), n_vertices, L.ptr(lab)))
    return lab
