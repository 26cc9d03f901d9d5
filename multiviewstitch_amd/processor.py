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
    L.check(L.lib().mvs_processor_deform(os.fsencode(model_obj), os.fsencode(template_obj), os.fsencode(parts_path), L.ptr(R),
                                         float(dist_thres), C.byref(prm), os.fsencode(out_obj), C.byref(st)))
    return _stats(st)
