"""``Processor::Deform`` (R/Processor/Processor.cpp:1111-1138) on files, through ``mvs_processor_deform``."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L
from .deformation import _stats, default_params


def Deform(model_obj, template_obj, parts_path, cam_R, dist_thres: float, out_obj, params: L.CParams | None = None) -> dict:
    """./Result/Model.obj + ./Template/meanbody.obj + ./Template/part/parts -> ./Result/deform.obj.
    ``cam_R`` is the rotation of cameras[0][0]; the view ray is its third row (R^T.col(2))."""
    R = L.arr(cam_R, np.float64).reshape(9)
    prm = params if params is not None else default_params()
    st = L.CStats()
    rc = L.check(L.lib().mvs_processor_deform(os.fsencode(model_obj), os.fsencode(template_obj), os.fsencode(parts_path), L.ptr(R),
                                              float(dist_thres), C.byref(prm), os.fsencode(out_obj), C.byref(st)))
    return _stats(st, rc)


def CheckConsistencyCore(curcam, refcams, depth, refdepths, min_dsp: float, max_dsp: float, reproj_err: int) -> np.ndarray:
    """R/Processor/Processor.cpp:72-126 — float32 raster in, filtered float32 raster out (what SaveDepth writes to DATA/CHECK)."""
    d = L.arr(depth, np.float32)
    refs = [L.arr(r, np.float32) for r in refdepths]
    ptrs = (C.c_void_p * max(1, len(refs)))(*[r.ctypes.data for r in refs])
    cams = (L.CCamera * max(1, len(refs)))(*[L.CCamera.of(c) for c in refcams])
    out = np.empty_like(d)
    cc = L.CCamera.of(curcam)
    L.check(L.lib().mvs_check_consistency(L.ptr(d), C.byref(cc), len(refs), ptrs, cams, float(min_dsp), float(max_dsp), int(reproj_err),
                                          L.ptr(out)))
    return out


def CheckConsistency(cameras, depths, min_dsp: float, max_dsp: float, reproj_err: int, out_dev: int | None = None, stream: int | None = None):
    """R/Processor/Processor.cpp:29-70 for one sequence: every frame against its two neighbours.
    ``depths`` is a float32 array [n, h, w] — or a device address when ``out_dev`` (a device address) is given."""
    cams = (L.CCamera * len(cameras))(*[L.CCamera.of(c) for c in cameras])
    if out_dev is not None:
        L.check(L.lib().mvs_check_consistency_seq_dev(len(cameras), L.ptr(int(depths)), cams, float(min_dsp), float(max_dsp), int(reproj_err),
                                                      L.ptr(int(out_dev)), L.ptr(stream)))
        return None
    d = L.arr(depths, np.float32)
    out = np.empty_like(d)
    L.check(L.lib().mvs_check_consistency_seq(len(cameras), L.ptr(d), cams, float(min_dsp), float(max_dsp), int(reproj_err), L.ptr(out)))
    return out


def RenderDepth(points, facets, camera, znear: float = 0.01, zfar: float = 2000.0, out_dev: int | None = None, stream: int | None = None):
    """Model2Depth::RenderDepth for one camera (R/Model2Depth/Model2Depth.cpp:58-156) without GLUT: float32 raster [h, w]
    of inverse depths, 0 where no triangle covers the pixel.  With ``out_dev`` the mesh arguments are device addresses
    (points: V*3 float64, facets: F*3 int32 — pass (address, count) tuples) and nothing is returned."""
    cc = L.CCamera.of(camera)
    if out_dev is not None:
        (pp, V), (fp, F) = points, facets
        L.check(L.lib().mvs_render_depth_dev(L.ptr(int(pp)), int(V), L.ptr(int(fp)), int(F), C.byref(cc), float(znear), float(zfar),
                                             L.ptr(int(out_dev)), L.ptr(stream)))
        return None
    pts = L.arr(points, np.float64).reshape(-1, 3)
    fac = L.arr(facets, np.int32).reshape(-1, 3)
    out = np.empty((camera.h, camera.w), np.float32)
    L.check(L.lib().mvs_render_depth(L.ptr(pts), len(pts), L.ptr(fac), len(fac), C.byref(cc), float(znear), float(zfar), L.ptr(out)))
    return out


def MatchFilter(raw, tex1, valid1, tex2, valid2, img1, img2, ssd_win: int, ssd_err: float, sample_interval: int):
    """The duplicate / SSD / gap cascade in front of RemoveOutliers (R/Processor/Processor.cpp:644-735) for the matches
    between the generated views of one frame pair.  raw [n, 6] = (view1, u1, v1, view2, u2, v2); tex [views, h*w] int32,
    valid [h*w] uint8, img [h, w, 3] uint8.  -> (matches [m, 4] = (u1, v1, u2, v2), sizes after the three stages)."""
    raw = L.arr(raw, np.int32).reshape(-1, 6)
    tex1, tex2 = L.arr(tex1, np.int32), L.arr(tex2, np.int32)
    valid1, valid2 = L.arr(valid1, np.uint8), L.arr(valid2, np.uint8)
    img1, img2 = L.arr(img1, np.uint8), L.arr(img2, np.uint8)
    h, w = img1.shape[:2]
    prm = L.CMatchFilterParams(w, h, tex1.shape[0], int(ssd_win), float(ssd_err), int(sample_interval), 0)
    out = np.empty((max(1, len(raw)), 4), np.int32)
    n_out = C.c_int64()
    cnt = np.zeros(3, np.int64)
    L.check(L.lib().mvs_match_filter(L.ptr(raw), len(raw), L.ptr(tex1), L.ptr(valid1), L.ptr(tex2), L.ptr(valid2), L.ptr(img1), L.ptr(img2),
                                     C.byref(prm), L.ptr(out), C.byref(n_out), L.ptr(cnt)))
    return out[:n_out.value].copy(), cnt
