"""Experiment (EXPERIMENTS build): does assigning a row's matrix entries to the ELL slots so that the 32-lane groups of the
Chebyshev step's LDS gathers meet few bank conflicts pay?  A fresh handle per variant (a long fit drifts into the late regime:
the same passes must be compared), the entry tables permuted per row on the host and written back (mvs_test_mesh_table 112-114)."""
import ctypes as C
import itertools
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import _lib as L, deformation, srt as srt_mod, scene as S
import bench

dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
fn = L.lib().mvs_test_mesh_table
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(1)


def measure(mode):
    d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
    d.UniformSampling(16)
    d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)

    def table(what, dtype):
        n = C.c_int64()
        L.check(fn(d._h, what, None, C.byref(n)))
        out = np.empty(n.value // np.dtype(dtype).itemsize, dtype)
        L.check(fn(d._h, what, L.ptr(out), C.byref(n)))
        return out

    if mode != "built":
        NP, LS, W, *_ = table(0, np.int64)
        pnloc = table(7, np.int32)
        lcol = table(12, np.int16).reshape(NP, W, LS).copy()
        gent = table(13, np.int32).reshape(NP, W, LS).copy()
        gcol = table(14, np.int32).reshape(NP, W, LS).copy()
        perms = np.array(list(itertools.permutations(range(W))))          # perm[s] = entry that goes to slot s
        ar = np.arange(W)
        for p in range(NP):
            nloc = int(pnloc[p])
            for g in range(2 * -(-nloc // 64)):
                load = np.zeros((W, 32), np.int64)
                seen = np.zeros((W, LS + 1), bool)
                for r in range(32 * g, 32 * g + 32):
                    a = lcol[p, :, r].astype(np.int64)
                    a = np.where(a < 0, r, np.minimum(a, LS))               # padding reads the row's own slot; halo columns one zero slot
                    cost = np.where(seen[:, a].T, 0, (load[:, a % 32].T + 1) ** 2)        # [entry][slot]
                    pm = perms[np.argmin(cost[perms, ar].sum(1))]
                    if mode == "identity": pm = ar
                    if mode == "random": pm = rng.permutation(W)
                    lcol[p, :, r] = lcol[p, pm, r]; gent[p, :, r] = gent[p, pm, r]; gcol[p, :, r] = gcol[p, pm, r]
                    a = a[pm]
                    new = ~seen[ar, a]
                    load[ar[new], a[new] % 32] += 1
                    seen[ar, a] = True
        nb = C.c_int64()
        for what, t in ((112, lcol), (113, gent), (114, gcol)):
            L.check(fn(d._h, what, L.ptr(np.ascontiguousarray(t)), C.byref(nb)))
    out = []
    d.iterate(5)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = d.iterate(20)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"{mode:9s}", " ".join("%.4f" % x for x in out), "ms per outer iteration (passes 5-24, 25-44, 45-64)",
          {k: st[k] for k in ("cg_launches", "cg_active", "unconverged_solves")}, flush=True)
    d.close()


for mode in sys.argv[1:] or ["built", "opt", "built", "opt"]:
    measure(mode)
