"""Host mirror of ``class Deformation`` (R/Deformation/Deformation.h:224-252).

Same method names and argument meaning as the reference; every method calls
the HIP engine through the C-ABI (include/mvs.h).  Differences a caller sees:

* errors raise ``MvsError`` instead of ``exit(-1)`` (Deformation.cpp:41-45,393-397);
* no files are written (``./Result/sample.obj``, Deformation.cpp:105);
* ``Deform`` can run more than the reference's single pass (``counter = 1``,
  Deformation.cpp:252) and the target may be a view shard of a larger set.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def default_params(**kw) -> L.CParams:
    p = L.CParams()
    L.lib().mvs_deform_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _stats(st: L.CStats, status: int = 0) -> dict:
    """``status``: the call's return value (0, or MVS_W_UNCONVERGED when a global solve of the batch ended above
    cg_tol); ``converged`` says the same as a bool."""
    return dict(outer_done=st.outer_done, arap_iters_run=st.arap_iters_run, cg_iters=st.cg_iters,
                n_valid=st.n_valid, energy=np.array(st.energy[:]), cg_rel_residual=st.cg_rel_residual,
                cg_launches=st.cg_launches, cg_active=st.cg_active,
                worst_rel_residual_in_batch=st.worst_rel_residual_in_batch, solves_in_batch=st.solves_in_batch,
                unconverged_solves=st.unconverged_solves, escalated=bool(st.escalated),
                status=int(status), converged=(status == 0 and st.unconverged_solves == 0))


class Deformation:
    """``Deformation(points, normals, facets)`` — Deformation.cpp:29-46."""

    def __init__(self, points, normals, facets, device: int | None = None):
        lib = L.lib()
        if device is not None:
            L.check(lib.mvs_set_device(int(device)))
        pts = L.arr(points, np.float64).reshape(-1, 3)
        nrm = L.arr(normals, np.float64).reshape(-1, 3)
        fcs = L.arr(facets, np.int32).reshape(-1, 3)
        if nrm.shape != pts.shape:
            raise ValueError("normals must match points")
        self._h = C.c_void_p()
        L.check(lib.mvs_deform_create(len(pts), L.ptr(pts), L.ptr(nrm), len(fcs), L.ptr(fcs), C.byref(self._h)))
        self.V, self.F = len(pts), len(fcs)
        self._faces = fcs
        self.params = default_params()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            L.lib().mvs_deform_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------- reference API --
    def UniformSampling(self, knn: int = 16) -> int:
        """Deformation.cpp:63-106; returns sampIdx.size()."""
        K = C.c_int64()
        L.check(L.lib().mvs_deform_sample_nodes(self._h, knn, C.byref(K)))
        return K.value

    def Deform(self, tpts, tnormals, projLenErr: float = 100.0, projDistErr: float = 100.0, n_outer: int = 1) -> dict:
        """``Deform(tpts, tnormals, projLenErr, projDistErr)`` — Deformation.cpp:232-402
        (call site R/Processor/Processor.cpp:1136 passes 100.0, 100.0)."""
        self.set_target(tpts, tnormals)
        if self.K == 0:
            self.UniformSampling()                       # Deformation.cpp:248-250
        self.params.proj_len_err = projLenErr
        self.params.proj_dist_err = projDistErr
        return self.iterate(n_outer)

    def Polyhedron(self):
        """Vertices and facets of the (deformed) mesh — what exportOBJ walks, Deformation.h:193-217."""
        return self.vertices(), self._faces.copy()

    # ----------------------------------------------------------------- engine --
    @property
    def K(self) -> int:
        k = C.c_int64()
        L.check(L.lib().mvs_deform_sizes(self._h, None, None, C.byref(k), None))
        return k.value

    @property
    def P(self) -> int:
        p = C.c_int64()
        L.check(L.lib().mvs_deform_sizes(self._h, None, None, None, C.byref(p)))
        return p.value

    def set_nodes(self, idx):
        idx = L.arr(idx, np.int32)
        L.check(L.lib().mvs_deform_set_nodes(self._h, L.ptr(idx), len(idx)))

    def nodes(self) -> np.ndarray:
        out = np.empty(self.K, np.int32)
        L.check(L.lib().mvs_deform_get_nodes(self._h, L.ptr(out)))
        return out

    def set_target(self, tpts, tnormals, index_base: int = 0):
        tp = L.arr(tpts, np.float64).reshape(-1, 3)
        tn = L.arr(tnormals, np.float64).reshape(-1, 3)
        if tp.shape != tn.shape:
            raise ValueError("tnormals must match tpts")
        L.check(L.lib().mvs_deform_set_target(self._h, len(tp), L.ptr(tp), L.ptr(tn), index_base))

    def set_target_dev(self, pts_dev: int, nrm_dev: int, P: int, index_base: int = 0):
        """Target already in HBM (raw device addresses, e.g. torch ``tensor.data_ptr()``)."""
        L.check(L.lib().mvs_deform_set_target_dev(self._h, P, L.ptr(int(pts_dev)), L.ptr(int(nrm_dev)), index_base))

    def iterate(self, n_outer: int = 1) -> dict:
        st = L.CStats()
        rc = L.check(L.lib().mvs_deform_iterate(self._h, C.byref(self.params), n_outer, C.byref(st)))
        return _stats(st, rc)

    def enqueue(self, n_outer: int = 1):
        """``iterate`` without the host synchronisation (needs one earlier synchronous call); pair with ``collect``."""
        L.check(L.lib().mvs_deform_iterate(self._h, C.byref(self.params), n_outer, None))

    def collect(self) -> dict:
        st = L.CStats()
        rc = L.check(L.lib().mvs_deform_collect(self._h, C.byref(self.params), C.byref(st)))
        return _stats(st, rc)

    # sharded phases (mvs.h: mvs_deform_assoc_*), device addresses in / out
    def assoc_dmin(self, d2min_dev: int):
        L.check(L.lib().mvs_deform_assoc_dmin(self._h, C.byref(self.params), L.ptr(int(d2min_dev))))

    def assoc_select(self, d2min_dev: int, records_dev: int, counts_dev: int):
        L.check(L.lib().mvs_deform_assoc_select(self._h, C.byref(self.params), L.ptr(int(d2min_dev)),
                                                L.ptr(int(records_dev)), L.ptr(int(counts_dev))))

    def assoc_merge(self, records_all_dev: int, counts_all_dev: int, nranks: int):
        L.check(L.lib().mvs_deform_assoc_merge(self._h, C.byref(self.params), L.ptr(int(records_all_dev)),
                                               L.ptr(int(counts_all_dev)), nranks))

    def assoc_merge_packed(self, packed_all_dev: int, nranks: int):
        """rank r's block of ``packed_all``: [K*8 records (48 B)][K*2 int32 counts] — one all-gather instead of two."""
        L.check(L.lib().mvs_deform_assoc_merge_packed(self._h, C.byref(self.params), L.ptr(int(packed_all_dev)), nranks))

    def assoc_merge_block(self, records_blk_dev: int, counts_blk_dev: int, nranks: int, k0: int, k1: int, block_nodes: int, block_dev: int):
        """owner-merges exchange: merge the node block [k0, k1) from every rank's records of that block into
        block_dev = [block_nodes*3 doubles | block_nodes bytes]"""
        L.check(L.lib().mvs_deform_assoc_merge_block(self._h, C.byref(self.params), L.ptr(int(records_blk_dev)), L.ptr(int(counts_blk_dev)), nranks,
                                                     k0, k1, block_nodes, L.ptr(int(block_dev))))

    def set_node_targets_dev(self, blocks_dev: int, nblocks: int, block_nodes: int, block_stride_bytes: int):
        L.check(L.lib().mvs_deform_set_node_targets_dev(self._h, L.ptr(int(blocks_dev)), nblocks, block_nodes, block_stride_bytes))

    def solve(self, sync: bool = True):
        """sync=False: enqueue only (no host synchronisation, no stats) — for back-to-back sharded steps."""
        if not sync:
            L.check(L.lib().mvs_deform_solve(self._h, C.byref(self.params), None))
            return None
        st = L.CStats()
        rc = L.check(L.lib().mvs_deform_solve(self._h, C.byref(self.params), C.byref(st)))
        return _stats(st, rc)

    def iterate_sharded(self, comm: "Comm", n_outer: int = 1) -> dict:
        """``mvs_deform_iterate_sharded``: the view-sharded passes with the collectives issued by the library itself (RCCL)."""
        st = L.CStats()
        rc = L.check(L.lib().mvs_deform_iterate_sharded(self._h, comm._c, C.byref(self.params), n_outer, C.byref(st)))
        return _stats(st, rc)

    def set_vertices(self, points, normals=None):
        """new positions (and normals) for the same topology — e.g. the template again for the next scan; tables, nodes and
        launch plans are kept (mvs_deform_set_vertices)"""
        p = L.arr(points, np.float64).reshape(-1, 3)
        n = L.arr(normals, np.float64).reshape(-1, 3) if normals is not None else None
        if len(p) != self.V or (n is not None and len(n) != self.V):
            raise ValueError("set_vertices: the vertex count of the handle's mesh is fixed")
        L.check(L.lib().mvs_deform_set_vertices(self._h, L.ptr(p), L.ptr(n)))

    def sync(self):
        L.check(L.lib().mvs_deform_sync(self._h))

    def stream(self) -> int:
        return int(L.lib().mvs_deform_stream(self._h) or 0)

    def set_stream(self, hip_stream: int | None):
        """Enqueue on a caller-owned stream (``torch.cuda.Stream(dev).cuda_stream``); None restores the own one.
        torch's DEFAULT stream has handle 0, which the C-ABI reads as NULL = "own stream": refuse it rather than let a
        caller believe the engine follows the default stream."""
        if hip_stream is not None and int(hip_stream) == 0:
            raise ValueError("stream handle 0 (the default stream) cannot be set: create a torch.cuda.Stream and pass its cuda_stream")
        L.check(L.lib().mvs_deform_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def arap(self, ctrl_targets) -> dict:
        """CGAL-equivalent ARAP with explicit targets for the handle's nodes (Deformation.cpp:383-400)."""
        ct = L.arr(ctrl_targets, np.float64).reshape(-1, 3)
        if len(ct) != self.K:
            raise ValueError("one target per node")
        st = L.CStats()
        rc = L.check(L.lib().mvs_deform_arap(self._h, C.byref(self.params), L.ptr(ct), C.byref(st)))
        return _stats(st, rc)

    # ---------------------------------------------------------------- getters --
    def vertices(self) -> np.ndarray:
        out = np.empty((self.V, 3))
        L.check(L.lib().mvs_deform_get_vertices(self._h, L.ptr(out)))
        return out

    def normals(self) -> np.ndarray:
        out = np.empty((self.V, 3))
        L.check(L.lib().mvs_deform_get_normals(self._h, L.ptr(out)))
        return out

    def rotations(self) -> np.ndarray:
        out = np.empty((self.V, 3, 3))
        L.check(L.lib().mvs_deform_get_rotations(self._h, L.ptr(out)))
        return out

    def compute_normals(self) -> np.ndarray:
        """exportOBJ's vertex normals of the current geometry (Deformation.h:86-150)."""
        out = np.empty((self.V, 3))
        L.check(L.lib().mvs_deform_compute_normals(self._h, L.ptr(out)))
        return out

    def node_targets(self, smoothed: bool = False) -> dict:
        K = self.K
        ctrl, valid = np.empty((K, 3)), np.empty(K, np.uint8)
        d2, cnt, top = np.empty(K, np.float32), np.empty((K, 2), np.int32), np.empty((K, 8), np.int64)
        L.check(L.lib().mvs_deform_get_node_targets(self._h, int(smoothed), L.ptr(ctrl), L.ptr(valid), L.ptr(d2),
                                                    L.ptr(cnt), L.ptr(top)))
        return dict(controls=ctrl, valid=valid, d2min=d2, counts=cnt, top_idx=top)

    def node_graph(self) -> np.ndarray:
        out = np.empty((self.K, self.params.graph_k + 1), np.int32)
        L.check(L.lib().mvs_deform_get_node_graph(self._h, L.ptr(out)))
        return out

    # ----------------------------------------------------------------- timing --
    def solver_info(self) -> dict:
        """which global solver `self.params` selects: {'kind': 'cg' | 'patch', 'patches', 'local_rows', 'width'}"""
        kind, width = C.c_int32(), C.c_int32()
        patches, rows = C.c_int64(), C.c_int64()
        L.check(L.lib().mvs_deform_solver_info(self._h, C.byref(self.params), C.byref(kind), C.byref(patches), C.byref(rows), C.byref(width)))
        return {"kind": "patch" if kind.value else "cg", "patches": patches.value, "local_rows": rows.value, "width": width.value}

    def enable_timing(self, on: int = 1):
        """0 off, 1 every phase, 2 only the "cg" / "tail" groups, 3 the planned sweeps ("cg") of every eighth pass with the idle flags
        of exactly those launches read back: "cg_idle" counts them, "cg:a<active>:i<idle>" holds the brackets of one composition
        (mvs_deform_enable_timing)."""
        L.check(L.lib().mvs_deform_enable_timing(self._h, int(on)))

    def kernel_time(self, name: str):
        ms, n = C.c_double(), C.c_int64()
        L.check(L.lib().mvs_deform_kernel_time(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


class Comm:
    """``mvs_comm_t``: the library's own RCCL communicator (one rank per GPU).  ``Comm.unique_id()`` on rank 0, share the
    bytes with the other ranks, then ``Comm(rank, nranks, id)`` everywhere (a collective)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        L.check(L.lib().mvs_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def __init__(self, rank: int, nranks: int, uid: bytes):
        self._c = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        L.check(L.lib().mvs_comm_init(rank, nranks, C.cast(buf, C.c_void_p), C.cast(C.byref(self._c), C.c_void_p)))
        self.rank, self.nranks = rank, nranks

    def set_exchange(self, mode: int):
        """0 auto (= all-gather at every rank count), 1 all-gather, 2 owner-merges on request (mvs_comm_set_exchange)"""
        L.check(L.lib().mvs_comm_set_exchange(self._c, int(mode)))

    def close(self):
        if getattr(self, "_c", None):
            L.lib().mvs_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def knn_points(pts, k: int) -> np.ndarray:
    """KNearestNeighbor on arbitrary points (Deformation.cpp:108-134): (n,k) indices, self included."""
    pts = L.arr(pts, np.float64).reshape(-1, 3)
    out = np.empty((len(pts), k), np.int32)
    L.check(L.lib().mvs_knn_points(L.ptr(pts), len(pts), k, L.ptr(out)))
    return out
