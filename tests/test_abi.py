"""The C-ABI library loads without a GPU, exports every symbol include/mvs.h declares, and fails
loudly (status codes, never a CPU fallback) when no device is present."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    import glob
    src = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mvs_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    from multiviewstitch_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 40
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    assert sorted(_lib.EXPORTS) == decl, "multiviewstitch_amd/_lib.py signature table out of sync with include/*.h"
    assert lib.mvs_abi_version() == 4


def test_struct_layouts_match_the_header():
    from multiviewstitch_amd import _lib
    assert C.sizeof(_lib.CCamera) == 4 * 8 + 9 * 8 + 3 * 8 + 2 * 4
    assert C.sizeof(_lib.CParams) == 80
    assert C.sizeof(_lib.CStats) == 4 * 4 + 8 * 8 + 8 + 2 * 4 + 8 + 4 * 4
    assert _lib.CAND_DTYPE.itemsize == 48
    from oracle import binding as O
    assert C.sizeof(O.Params) == C.sizeof(_lib.CParams) and C.sizeof(O.Camera) == C.sizeof(_lib.CCamera)
    assert O.CAND_DTYPE == _lib.CAND_DTYPE
    p = _lib.CParams()
    _lib.lib().mvs_deform_default_params(C.byref(p))
    assert (p.proj_len_err, p.proj_dist_err, p.min_cos, p.max_result, p.top_k, p.graph_k, p.smooth_sweeps, p.arap_iters,
            p.arap_tol) == (100.0, 100.0, 0.1, 10000, 8, 8, 2, 5, 1e-4)      # Processor.cpp:1136, Deformation.cpp:244,338,359,362,398


def test_host_side_entries_work_without_a_device():
    """triple generator, chain composition and argument validation are host code."""
    from multiviewstitch_amd import srt, _lib
    from oracle import binding as O
    tri, st = srt.make_triples(50, 20, 7)
    otri, ost = O.srt_make_triples(50, 20, 7)
    assert np.array_equal(tri, otri) and st == ost
    assert (np.diff(tri, axis=1) > 0).all() and tri.min() >= 0 and tri.max() < 50      # Shuffle returns sorted distinct indices
    rng = np.random.default_rng(0)
    Ra, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    Rb, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    ta, tb = rng.normal(size=3), rng.normal(size=3)
    got = srt.compose(1.1, Ra, ta, 0.9, Rb, tb)
    ref = O.srt_compose(1.1, Ra, ta, 0.9, Rb, tb)
    assert abs(got[0] - ref[0]) == 0 and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    got = srt.relative(0.9, Rb, tb, 1.1, Ra, ta)
    ref = O.srt_relative(0.9, Rb, tb, 1.1, Ra, ta)
    assert abs(got[0] - ref[0]) == 0 and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])


def test_compute_entries_fail_loudly_without_a_gpu():
    from multiviewstitch_amd import _lib, deformation, srt
    import pytest
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    from multiviewstitch_amd import scene as S
    sc = S.make_scene(0)
    with pytest.raises(_lib.MvsError) as e:
        deformation.Deformation(sc.verts, sc.normals, sc.faces)
    assert e.value.code == -4 and "no CPU fallback" in str(e.value)
    with pytest.raises(_lib.MvsError) as e:
        srt.apply(sc.verts, None, 1.0, np.eye(3), np.zeros(3))
    assert e.value.code == -4
    with pytest.raises(_lib.MvsError) as e:
        srt.depth_to_model(sc.depth[0], sc.cams[0], S.MIN_DSP, S.MAX_DSP, S.SMOOTH)
    assert e.value.code == -4
    # the mesh check is a kernel too (meshbuild.hip, round 3): without a device a bad mesh is refused like any other — there is
    # no host-side restatement of the check to fall back on (tests/test_gpu_deform.py::test_bad_mesh_is_rejected holds the GPU case)
    bad = sc.faces.copy()
    bad[0] = bad[0][::-1]
    with pytest.raises(_lib.MvsError) as e:
        deformation.Deformation(sc.verts, sc.normals, bad)
    assert e.value.code == -4


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing in the product package may import, include or load it."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|#include\s+[\"<][^\">]*(orc_|mvs_oracle)|libmvs_oracle|orc_[a-z_]+\s*\(", re.M)
    pkg = os.path.join(ROOT, "multiviewstitch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not pat.search(src), f"{f} references the oracle"


def test_trace_callback_sees_every_entry():
    """mvs_set_trace (SURVEY §8b "optional callback for tracing"): entered / left with the entry's name and its host time —
    also when the entry fails (here, without a GPU, with MVS_E_NO_DEVICE)."""
    import ctypes as C
    from multiviewstitch_amd import _lib, srt
    seen = []
    CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_int, C.c_double)
    cb = CB(lambda ctx, name, phase, ms: seen.append((name.decode(), phase, ms)))
    L = _lib.lib()
    assert L.mvs_set_trace(C.cast(cb, C.c_void_p), None) == 0
    try:
        try:
            srt.apply(np.zeros((4, 3)), None, 1.0, np.eye(3), np.zeros(3))
        except _lib.MvsError:
            pass
    finally:
        assert L.mvs_set_trace(None, None) == 0
    assert [(n, p) for n, p, _ in seen] == [("mvs_srt_apply", 0), ("mvs_srt_apply", 1)] and seen[1][2] >= 0.0
    n = len(seen)
    try:
        srt.apply(np.zeros((4, 3)), None, 1.0, np.eye(3), np.zeros(3))
    except _lib.MvsError:
        pass
    assert len(seen) == n                                   # switched off
