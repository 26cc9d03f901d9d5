"""Per-part deformation graphs (BASELINE.json config 5: "PartRecognition-segmented per-part deformation graphs").

The reference labels template vertices with its 16 body parts (``Template/part/parts``,
R/PartRecognition/PartRecognition.cpp:7-48), transfers the labels to the scan by 1-NN (``PartRecog`` :50-77) and uses
them for the per-limb rigid fits of ``LocalAlignment`` (R/Alignment/Alignment.cpp:316-421); its non-rigid stage then
runs ONE ``Deformation`` over the whole template (R/Processor/Processor.cpp:1135-1137).  Config 5 carries the
segmentation one step further: one ``Deformation`` per part — its own sub-mesh, node sampling, node graph and ARAP
system — fitted to the scan points that ``PartRecog`` gave the same label.  Nothing new runs on the device: every
part is an ordinary handle of the C-ABI (include/mvs.h), so the per-part results are the reference's
``Deformation::Deform`` on the part's sub-mesh (checked against the oracle part by part in tests/test_partwise.py).

What this module adds is host logic only:

* ``split_parts``   — the sub-mesh of a label: the faces whose three vertices carry it, minus the faces that would
                      leave a vertex with two separate fans (``Deformation``'s half-edge builder rejects such
                      vertices, Deformation.cpp:38-45), vertices renumbered in ascending global order;
* ``PartwiseDeformation`` — the handles, each on its own HIP stream: after the first (calibrating, synchronous)
                      pass, ``iterate`` only ENQUEUES every part (``mvs_deform_iterate`` with stats = NULL) and then
                      collects them, so the launch-latency-bound chains of the 16 parts overlap on the device.
"""
from __future__ import annotations

import numpy as np

from .deformation import Deformation


def sector_labels(points, n_parts: int = 16) -> np.ndarray:
    """16 angular sectors about the z axis standing in for ``enum PART`` on the synthetic template."""
    p = np.asarray(points, np.float64).reshape(-1, 3)
    az = np.arctan2(p[:, 1], p[:, 0])
    return np.minimum(((az + np.pi) / (2 * np.pi) * n_parts).astype(np.int32), n_parts - 1)


def _single_fan_faces(faces: np.ndarray) -> np.ndarray:
    """Mask of the faces to keep so that every vertex of the sub-mesh has ONE fan of faces (connected through edges
    at that vertex).  Corners (face, vertex) are joined across every shared edge; a vertex whose corners fall into
    several groups keeps the largest (ties: the group holding its lowest face id).  Repeats until stable, because
    dropping a face can split another vertex's fan."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    keep = np.ones(len(faces), bool)
    while True:
        fid = np.flatnonzero(keep)
        f = faces[fid]
        if len(f) == 0:
            return keep
        nf = len(f)
        # directed half-edges (a -> b) of face row r at corner k; the twin of (a, b) is (b, a) in a manifold mesh
        a = f.reshape(-1)
        b = np.roll(f, -1, axis=1).reshape(-1)
        corner_a = np.arange(3 * nf)                                   # corner of vertex a in its face
        corner_b = (np.arange(3 * nf) // 3) * 3 + (np.arange(3 * nf) % 3 + 1) % 3
        key = a.astype(np.int64) * (faces.max() + 1) + b
        twin = b.astype(np.int64) * (faces.max() + 1) + a
        order = np.argsort(key, kind="stable")
        pos = np.searchsorted(key[order], twin)
        pos = np.minimum(pos, len(order) - 1)
        has = key[order][pos] == twin
        he = np.flatnonzero(has)
        th = order[pos[he]]                                            # twin half-edge (b -> a) in the other face
        # same vertex on both sides of the edge: a here = b there, b here = a there
        rows = np.concatenate([corner_a[he], corner_b[he]])
        cols = np.concatenate([corner_b[th], corner_a[th]])
        g = coo_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(3 * nf, 3 * nf))
        _, comp = connected_components(g, directed=False)
        vert = f.reshape(-1)
        # per vertex: size of each corner group, keep the largest
        pair = np.stack([vert, comp], 1)
        uniq, inv, cnt = np.unique(pair, axis=0, return_inverse=True, return_counts=True)
        inv = inv.reshape(-1)
        first_face = np.full(len(uniq), np.iinfo(np.int64).max)
        np.minimum.at(first_face, inv, np.arange(3 * nf) // 3)
        # best group of each vertex: max count, then lowest first face
        o = np.lexsort((first_face, -cnt, uniq[:, 0]))
        uv = uniq[o, 0]
        lead = np.concatenate([[True], uv[1:] != uv[:-1]])
        best_group = np.zeros(int(faces.max()) + 1, np.int64) - 1
        best_group[uv[lead]] = o[lead]
        bad_corner = best_group[vert] != inv
        if not bad_corner.any():
            return keep
        keep[fid[np.unique(np.flatnonzero(bad_corner) // 3)]] = False


def split_parts(faces, labels, n_parts: int | None = None):
    """-> list (one entry per label 0..n_parts-1) of dicts: ``vid`` ascending global vertex ids of the part's sub-mesh
    (int64), ``faces`` its triangles in local numbering (int32, original order and orientation).  A face belongs to a
    part when all three of its vertices carry the label; faces straddling two parts belong to none."""
    faces = np.ascontiguousarray(faces, np.int32).reshape(-1, 3)
    labels = np.asarray(labels, np.int32).reshape(-1)
    n_parts = int(labels.max()) + 1 if n_parts is None else n_parts
    lf = labels[faces]
    same = (lf[:, 0] == lf[:, 1]) & (lf[:, 1] == lf[:, 2])
    out = []
    for part in range(n_parts):
        pf = faces[same & (lf[:, 0] == part)]
        if len(pf):
            pf = pf[_single_fan_faces(pf)]
        vid = np.unique(pf)
        local = np.searchsorted(vid, pf).astype(np.int32)
        out.append(dict(vid=vid.astype(np.int64), faces=local))
    return out


def assign_parts(sizes, world: int) -> list[int]:
    """Owner rank of every part: largest part first onto the least loaded rank (ties: lower rank, lower part index) —
    deterministic, so every rank computes the same map without talking."""
    load = [0] * world
    owner = [0] * len(sizes)
    for k in sorted(range(len(sizes)), key=lambda q: (-int(sizes[q]), q)):
        r = min(range(world), key=lambda q: (load[q], q))
        owner[k] = r
        load[r] += int(sizes[k])
    return owner


class PartwiseDeformation:
    """One ``Deformation`` per part label; same call sequence as ``Deformation`` (UniformSampling -> set_target ->
    iterate -> vertices).

    ``world > 1`` (one process per GPU, torch.distributed initialised): the parts are INDEPENDENT fits, so they shard
    naturally — every rank owns the handles of its parts only (``assign_parts``: balanced by vertex count), takes the scan
    points of those parts, iterates them without any communication, and the one collective of the whole fit is the
    all-gather of the part vertices in ``vertices()`` (padded to the largest rank's share; ``group`` = process group)."""

    def __init__(self, points, normals, facets, labels, n_parts: int | None = None, device: int | None = None,
                 rank: int = 0, world: int = 1, group=None, handle_factory=None):
        self.points = np.array(points, np.float64).reshape(-1, 3)
        self.normals = np.array(normals, np.float64).reshape(-1, 3)
        self.labels = np.asarray(labels, np.int32).reshape(-1)
        if len(self.labels) != len(self.points):
            raise ValueError("one label per vertex")
        self.parts = split_parts(facets, self.labels, n_parts)
        self.rank, self.world, self.group = int(rank), int(world), group
        self.owner = assign_parts([len(p["vid"]) for p in self.parts], self.world)
        make = handle_factory if handle_factory is not None else (lambda pts, nrm, fcs: Deformation(pts, nrm, fcs, device))
        self.handles: list[Deformation | None] = []
        for k, part in enumerate(self.parts):
            if len(part["faces"]) == 0 or self.owner[k] != self.rank:
                self.handles.append(None)
                continue
            vid = part["vid"]
            self.handles.append(make(self.points[vid], self.normals[vid], part["faces"]))
        self._calibrated = False
        self._pool = None
        self.host_threads = 8
        self._group = None
        self.use_group = True            # step the parts as one sequence of launches when the library allows it
        self.group_split = 2             # ... as this many groups side by side (each on its own stream and host thread)
        self.group_declined = ""
        self.group_passes = 0

    def close(self):
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None
        if getattr(self, "_group", None) is not None:
            from . import _lib as L
            for g, _, _ in self._group:
                L.lib().mvs_deform_group_destroy(g)
            self._group = None
        for h in self.handles:
            if h is not None:
                h.close()
        self.handles = []

    @property
    def live(self):
        return [(k, h) for k, h in enumerate(self.handles) if h is not None]

    def set_params(self, **kw):
        for _, h in self.live:
            for key, val in kw.items():
                setattr(h.params, key, val)

    def UniformSampling(self, knn: int = 16) -> int:
        return sum(h.UniformSampling(knn) for _, h in self.live)

    @property
    def K(self) -> int:
        """nodes of the parts this rank owns (all parts when world == 1)"""
        return sum(h.K for _, h in self.live)

    def set_target(self, tpts, tnormals, tlabels):
        """Scan points and normals with the labels ``PartRecog`` gave them; part k is fitted to the points labelled k
        (a part that received no point keeps its rest shape: every node is invalid, as in Deformation.cpp:355-356)."""
        tp = np.asarray(tpts, np.float64).reshape(-1, 3)
        tn = np.asarray(tnormals, np.float64).reshape(-1, 3)
        tl = np.asarray(tlabels, np.int32).reshape(-1)
        order = np.argsort(tl, kind="stable")                      # one pass; the points of a part keep their order
        bounds = np.searchsorted(tl[order], np.arange(len(self.handles) + 1))
        for k, h in self.live:
            sel = order[bounds[k]:bounds[k + 1]]
            h.set_target(tp[sel], tn[sel])
        self._calibrated = False

    def iterate(self, n_outer: int = 1) -> list:
        """-> statistics per live part (same dict as ``Deformation.iterate``).  The first call after ``set_target``
        runs the parts one after the other (each handle calibrates its solver plan synchronously); later calls enqueue
        all parts first and collect afterwards."""
        if not self._calibrated:
            stats = [h.iterate(n_outer) for _, h in self.live]
            self._calibrated = True
            return stats
        # The parts as ONE sequence of launches (include/mvs.h, mvs_deform_group_*: every kernel of an outer iteration once for
        # all parts).  The library declines (MVS_E_STATE) until every part has stepped twice on its own — its first two
        # associations search unbounded — or when a part runs the CG solver; the parts then go on as separate launch chains.
        if self.use_group and len(self.live) > 1 and all(isinstance(h, Deformation) for _, h in self.live):
            got = self._group_iterate(n_outer)
            if got is not None:
                return got
        # one host thread per part: the C-ABI call releases the GIL, and a single thread's launch rate (~3.5 us per
        # kernel) would otherwise cap 16 overlapping parts at the speed of ~4
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=min(self.host_threads, max(1, len(self.live))))
        list(self._pool.map(lambda kh: kh[1].enqueue(n_outer), self.live))
        return [h.collect() for _, h in self.live]

    def _group_iterate(self, n_outer: int):
        """-> statistics per live part, or None when the library declines (see iterate).  The parts step as ``group_split`` groups,
        each a sequence of launches of its own on its own stream and host thread: two groups of eight parts fill the chip better
        than one of sixteen (a launch of ~500 workgroups leaves its last round half empty; the other group's launch runs beside
        it) — 1.02 against 1.58 ms per outer iteration early in a config-5 fit, four groups 1.21 (scripts/config5_split.py)."""
        import ctypes as C
        from . import _lib as L
        from .deformation import _stats
        live = [h for _, h in self.live]
        if self._group is None:
            ng = max(1, min(int(self.group_split), len(live) // 2 or 1))
            groups = []
            for gi in range(ng):
                idx = list(range(gi, len(live), ng))
                arr = (C.c_void_p * len(idx))(*[live[i]._h for i in idx])
                g = C.c_void_p()
                L.check(L.lib().mvs_deform_group_create(C.cast(arr, C.c_void_p), len(idx), C.cast(C.byref(g), C.c_void_p)))
                groups.append((g, idx, (L.CStats * len(idx))()))
            self._group = groups

        def run(item):
            g, idx, st = item
            rc = L.lib().mvs_deform_group_iterate(g, C.byref(live[idx[0]].params), n_outer, C.cast(st, C.c_void_p))
            return rc, (L.lib().mvs_last_error().decode(errors="replace") if rc < 0 else "")

        # (every group must be able to step: probe with zero outer iterations first, so that a decline leaves ALL parts untouched)
        for g, idx, st in self._group:
            rc = L.lib().mvs_deform_group_iterate(g, C.byref(live[idx[0]].params), 0, C.cast(st, C.c_void_p))
            if rc == -8:                                         # MVS_E_STATE: not yet
                self.group_declined = L.lib().mvs_last_error().decode(errors="replace")
                return None
            L.check(rc)
        if len(self._group) == 1:
            res = [run(self._group[0])]
        else:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=min(self.host_threads, max(1, len(self.live))))
            res = list(self._pool.map(run, self._group))
        for rc, msg in res:
            if rc < 0:
                raise L.MvsError(rc, msg)
        self.group_passes += n_outer
        out = [None] * len(live)
        for (g, idx, st) in self._group:
            for i, s_ in zip(idx, st):
                out[i] = _stats(s_, 1 if s_.unconverged_solves else 0)       # (a part's own verdict is in its statistics)
        return out

    def vertices(self, comm_device=None) -> np.ndarray:
        """[V,3]: each part's vertices at their place; vertices of no sub-mesh (isolated by the split) stay at rest.
        world > 1: every rank contributes the vertices of the parts it owns to ONE all-gather (comm_device: the torch
        device the collective runs on — the rank's GPU under nccl = RCCL, None = CPU tensors for gloo)."""
        out = self.points.copy()
        for k, h in self.live:
            out[self.parts[k]["vid"]] = h.vertices()
        if self.world > 1:
            import torch
            import torch.distributed as dist
            mine = [k for k in range(len(self.parts)) if self.owner[k] == self.rank and len(self.parts[k]["faces"])]
            share = [sum(len(self.parts[k]["vid"]) for k in range(len(self.parts)) if self.owner[k] == r and len(self.parts[k]["faces"]))
                     for r in range(self.world)]
            width = max(max(share), 1) * 3
            send = torch.zeros(width, dtype=torch.float64, device=comm_device)
            if mine:
                flat = np.concatenate([out[self.parts[k]["vid"]].reshape(-1) for k in mine])
                send[:len(flat)] = torch.from_numpy(flat).to(send.device)
            recv = torch.empty(self.world * width, dtype=torch.float64, device=comm_device)
            dist.all_gather_into_tensor(recv, send, group=self.group)
            recv = recv.cpu().numpy().reshape(self.world, width)
            for r in range(self.world):
                o = 0
                for k in range(len(self.parts)):
                    if self.owner[k] != r or len(self.parts[k]["faces"]) == 0:
                        continue
                    n = len(self.parts[k]["vid"]) * 3
                    out[self.parts[k]["vid"]] = recv[r, o:o + n].reshape(-1, 3)
                    o += n
        return out
