"""World-size-2 run (gloo, CPU) of the view-sharded outer iteration protocol
(multiviewstitch_amd/dist.py): all-reduce(MIN) of d2min, all-gather of the per-rank best-8
records, identical merge on every rank.  The shard behind the protocol is a CHECKER shard
built on the oracle (the product shard needs a GPU); the exchange code under test is the one
bench.py runs over RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class OracleShard:
    """Shard protocol (dist.py) on top of oracle/ — CPU tensors, gloo."""

    def __init__(self, O, g, lo, hi):
        self.O, self.p = O, O.Params.default()
        self.nodes = g["nodes"]
        self.pts, self.nrm, self.faces = g["verts"].copy(), g["normals"], g["faces"]
        self.tgt = O.Target(g["tp"][lo:hi], g["tn"][lo:hi], lo)
        self.result = None
        self.packed = True

    def buffers(self, K, world):
        if self.packed:                                     # the layout bench.py exchanges: one all-gather per step
            R, Cn = K * 8 * 48, K * 2 * 4
            pack = torch.empty(R + Cn, dtype=torch.uint8)
            return dict(d2min=torch.empty(K, dtype=torch.float32), pack=pack, rec=pack[:R], cnt=pack[R:].view(torch.int32),
                        pack_all=torch.empty(world * (R + Cn), dtype=torch.uint8))
        return dict(d2min=torch.empty(K, dtype=torch.float32), rec=torch.empty(K * 8 * 48, dtype=torch.uint8),
                    cnt=torch.empty(K * 2, dtype=torch.int32), rec_all=torch.empty(world * K * 8 * 48, dtype=torch.uint8),
                    cnt_all=torch.empty(world * K * 2, dtype=torch.int32))

    def dmin(self, b):
        b["d2min"].copy_(torch.from_numpy(self.tgt.dmin(self.pts[self.nodes])))

    def select(self, b):
        rec, cnt = self.tgt.select(self.pts[self.nodes], self.nrm[self.nodes], self.p, b["d2min"].numpy())
        b["rec"].copy_(torch.from_numpy(rec.view(np.uint8).reshape(-1)))
        b["cnt"].copy_(torch.from_numpy(cnt.reshape(-1)))

    def merge(self, b, world):
        K = len(self.nodes)
        if "pack_all" in b:
            blk = b["pack_all"].numpy().reshape(world, -1)
            rec = np.ascontiguousarray(blk[:, :K * 8 * 48]).view(self.O.CAND_DTYPE).reshape(world, K, 8)
            cnt = np.ascontiguousarray(blk[:, K * 8 * 48:]).view(np.int32).reshape(world, K, 2)
        else:
            rec = b["rec_all"].numpy().view(self.O.CAND_DTYPE).reshape(world, K, 8)
            cnt = b["cnt_all"].numpy().reshape(world, K, 2)
        self.merged = self.O.assoc_merge(self.pts[self.nodes], self.nrm[self.nodes], self.p, rec, cnt)

    # owner-merges exchange (dist.py, N >= 4 on the GPU): same protocol, checker side
    def buffers_owner(self, K, world, rank):
        from multiviewstitch_amd.dist import EngineShard
        return EngineShard._buffers_owner(K, world, rank, torch.device("cpu"))

    def merge_block(self, b, world):
        o = b["owner"]
        k0, k1 = o["blocks"][o["rank"]]
        n, bn = k1 - k0, o["bn"]
        if n == 0:
            return
        rec = b["rec_in"].numpy()[:world * n * 8 * 48].view(self.O.CAND_DTYPE).reshape(world, n, 8)
        cnt = b["cnt_in"].numpy()[:world * n * 2].reshape(world, n, 2)
        nodes = self.nodes[k0:k1]
        m = self.O.assoc_merge(self.pts[nodes], self.nrm[nodes], self.p, rec, cnt)
        blk = b["blk"].numpy()
        blk[:bn * 24].view(np.float64).reshape(bn, 3)[:n] = m["controls"]
        blk[bn * 24:bn * 24 + n] = m["valid"].astype(np.uint8)

    def install(self, b, world):
        o = b["owner"]
        K, bn = len(self.nodes), o["bn"]
        allb = b["blk_all"].numpy().reshape(world, o["stride"])
        ctrl = np.concatenate([allb[r, :bn * 24].view(np.float64).reshape(bn, 3) for r in range(world)])[:K]
        valid = np.concatenate([allb[r, bn * 24:bn * 25] for r in range(world)])[:K]
        self.merged = dict(controls=ctrl.copy(), valid=valid.astype(bool))

    def solve(self, sync=True):
        O = self.O
        npts = self.pts[self.nodes]
        ctrl = O.smooth(npts, self.merged["controls"], O.knn_points(npts, 9), 2)
        r = O.arap(self.pts, self.faces, self.nodes, ctrl, 5, 1e-4)
        self.pts = r["pts"]
        return r


def _worker(rank, world, port, q, packed=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as O
    from multiviewstitch_amd import dist as mdist
    g = np.load(os.path.join(GOLD, "deform_cfg0.npz"))
    P = len(g["tp"])
    # uneven split, one rank may even be empty of useful points
    cuts = [0, P // 3, P]
    offs, counts = mdist.exclusive_offsets(cuts[rank + 1] - cuts[rank], world, torch.device("cpu"))
    assert offs[rank] == cuts[rank] and counts.sum() == P
    shard = OracleShard(O, g, cuts[rank], cuts[rank + 1])
    shard.packed = packed
    bufs = shard.buffers_owner(len(g["nodes"]), world, rank) if packed == "owner" else shard.buffers(len(g["nodes"]), world)
    for _ in range(2):
        mdist.sharded_step(shard, bufs, world)
    q.put((rank, shard.pts, shard.merged["valid"]))
    dist.barrier()
    dist.destroy_process_group()


import pytest


# one all-gather of [records | counts]; the two arrays separately; owner-merges (all-to-all by node block, all-gather of the targets)
@pytest.mark.parametrize("packed", [True, False, "owner"])
def test_sharded_step_world2_matches_single_rank(packed):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, packed)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, pts, valid = q.get(timeout=120)
        res[r] = (pts, valid)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = np.load(os.path.join(GOLD, "deform_cfg0.npz"))
    # every rank ends with the same mesh, and it is the unsharded answer (fixture: 2 outer iterations)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.sqrt(np.mean(np.sum((res[0][0] - g["it2_pts"]) ** 2, axis=1))) < 1e-9


def test_view_shards_cover_every_view_once():
    from multiviewstitch_amd import dist as mdist
    for n, w in ((8, 1), (8, 2), (8, 4), (8, 8), (16, 8), (4, 8), (5, 3)):
        sh = mdist.view_shards(n, w)
        assert len(sh) == w and sorted(v for s in sh for v in s) == list(range(n))
        assert max(len(s) for s in sh) - min(len(s) for s in sh) <= 1


def test_bench_launcher_starts_its_own_ranks():
    """`python bench.py --gpus N` with no external launcher: the parent starts N ranks of itself (before touching any GPU),
    relays rank 0's JSON line and fails when a rank fails.  Rehearsed here on the CPU with --dry-run (gloo)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["backend"] == "gloo" and out["sum_of_rank_ids_plus_1"] == 3.0
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"],
                       env=dict(env, MVS_BENCH_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


# ---------------------------------------------------------------- parts over ranks (BASELINE config 5) ----
class _OraclePart:
    """The handle protocol PartwiseDeformation drives, on top of oracle/ (the product handle needs a GPU)."""

    def __init__(self, pts, nrm, faces):
        from oracle import binding as O
        self.O, self.o = O, O.Deform(pts, nrm, faces)
        self.params = O.Params.default()

    def UniformSampling(self, knn=16):
        self.K = self.o.sample_nodes(knn)
        return self.K

    def set_target(self, tp, tn):
        self.o.set_target(tp, tn)

    def iterate(self, n=1):
        return self.o.iterate(self.params, n)

    def enqueue(self, n=1):
        self._st = self.o.iterate(self.params, n)

    def collect(self):
        return self._st

    def vertices(self):
        return self.o.vertices()

    def close(self):
        pass


def _parts_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multiviewstitch_amd import partwise as PW
    from tests.util import scene_and_target
    sc, tp, tn, _ = scene_and_target(0)
    labels, tl = PW.sector_labels(sc.verts, 5), PW.sector_labels(tp, 5)
    pd = PW.PartwiseDeformation(sc.verts, sc.normals, sc.faces, labels, 5, rank=rank, world=world, handle_factory=_OraclePart)
    pd.UniformSampling(16)
    pd.set_target(tp, tn, tl)
    pd.iterate(1)
    pd.iterate(1)
    q.put((rank, pd.vertices(), [k for k, _ in pd.live], pd.owner))
    dist.barrier()
    dist.destroy_process_group()


def test_parts_shard_over_ranks_with_one_all_gather():
    """config 5's per-part graphs are independent fits: two ranks each own some parts, iterate them without talking, and
    the all-gather in vertices() gives every rank the mesh a single process computes."""
    from multiviewstitch_amd import partwise as PW
    from tests.util import scene_and_target
    assert PW.assign_parts([5, 9, 9, 1], 2) == [0, 0, 1, 1]          # largest first onto the least loaded rank, ties -> lower rank
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_parts_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc, tp, tn, _ = scene_and_target(0)
    labels, tl = PW.sector_labels(sc.verts, 5), PW.sector_labels(tp, 5)
    ref = PW.PartwiseDeformation(sc.verts, sc.normals, sc.faces, labels, 5, handle_factory=_OraclePart)
    ref.UniformSampling(16)
    ref.set_target(tp, tn, tl)
    ref.iterate(1)
    ref.iterate(1)
    want = ref.vertices()
    assert np.abs(want - sc.verts).max() > 1e-4                        # the fit moved something
    assert sorted(got[0][2] + got[1][2]) == [k for k, _ in ref.live] and not set(got[0][2]) & set(got[1][2])
    assert got[0][3] == got[1][3]
    for _, v, _, _ in got:
        assert np.array_equal(v, want)


# ---------------------------------------------------------------- Alignment reductions by view ----
def _align_worker(rank, world, port, q, engine):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as O
    from multiviewstitch_amd import dist as mdist
    from tests.util import body_scene
    sc = body_scene()
    gr, p, _, _ = O.remove_ground(sc["tgt"], sc["t_nrm"], sc["t_faces"], 0.81)
    cuts = [0, len(p) // 3, len(p)] if world == 2 else [0, len(p) // 3, len(p) // 3, len(p)]      # uneven; with 3 ranks one is empty
    mine = p[cuts[rank]:cuts[rank + 1]]
    red = mdist.host_reducer()
    if engine:                                       # the product entry (GPU box): mvs_init_alignment_sharded
        from multiviewstitch_amd import alignment
        R, t, s = alignment.Alignment().InitAlignmentSharded(sc["src"], mine, gr, sc["view_ray"], red)
    else:                                            # the checker, driven through the same reducer
        R, t, s = O.init_alignment_sharded(sc["src"], mine, gr, sc["view_ray"], red)
    q.put((rank, R, t, s))
    dist.barrier()
    dist.destroy_process_group()


def _run_align(world, engine):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_align_worker, args=(r, world, port, q, engine)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, R, t, sc = q.get(timeout=180)
        res[r] = (R, t, sc)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from oracle import binding as O
    from tests.util import body_scene
    b = body_scene()
    gr, p, _, _ = O.remove_ground(b["tgt"], b["t_nrm"], b["t_faces"], 0.81)
    want = O.init_alignment(b["src"], p, gr, b["view_ray"])
    for r in range(1, world):                                                   # every rank holds the same similarity, bit for bit
        assert all(np.array_equal(np.asarray(a), np.asarray(c)) for a, c in zip(res[0], res[r]))
    # ... and it is the unsharded one up to the order of the sums (12 reduced moments / extents)
    assert np.abs(res[0][0] - want[0]).max() < 1e-9 and np.abs(res[0][1] - want[1]).max() < 1e-9 and abs(res[0][2] - want[2]) < 1e-11


@pytest.mark.parametrize("world", [2, 3])
def test_init_alignment_reductions_shard_by_view(world):
    _run_align(world, engine=False)


# RemoveGround (Alignment.cpp:79-233) and LocalAlignmentCore (:423-546) with the scan sharded by view -------------------
def _two_bodies():
    """the body scene's scan cut into its connected pieces — the body on one "view", the ground patch under it on the other:
    facets never join points of two ranks, and RemoveGround's last step keeps the larger piece, which lives on ONE rank"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from tests.util import body_scene
    b = body_scene()
    P, N, F = b["tgt"], b["t_nrm"], b["t_faces"]
    e = np.concatenate([F[:, [0, 1]], F[:, [1, 2]]])
    _, comp = connected_components(coo_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(len(P), len(P))), directed=False)
    big = np.bincount(comp).argmax()
    views = []
    for sel in (comp == big, comp != big):
        idx = np.flatnonzero(sel)
        remap = -np.ones(len(P), np.int64)
        remap[idx] = np.arange(len(idx))
        f = F[sel[F[:, 0]]]
        views.append((P[idx], N[idx], remap[f].astype(np.int32)))
    return b, views


def _ground_worker(rank, world, port, q, engine):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as O
    from multiviewstitch_amd import dist as mdist
    b, views = _two_bodies()
    empty = (np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3), np.int32))
    p, n, f = views[rank] if rank < 2 else empty                       # with 3 ranks the last one holds nothing
    red = mdist.host_reducer()
    if engine:
        from multiviewstitch_amd import alignment
        A = alignment.Alignment()
        gr, p2, n2, f2 = A.RemoveGroundSharded(p, n, f, red, rank)
    else:
        gr, p2, n2, f2 = O.remove_ground_sharded(p, n, f, red, rank)
    # LocalAlignmentCore on the shares: the body's labelled points split over the ranks (nearest template vertex's label)
    tl_full = O.part_recog(b["src"], b["s_labels"], b["tgt"])
    cuts = np.linspace(0, len(b["tgt"]), world + 1).astype(int) if world == 2 else np.array([0, len(b["tgt"]) // 2, len(b["tgt"]) // 2, len(b["tgt"])])
    sl = slice(cuts[rank], cuts[rank + 1])
    group, label = (1 << 2) | (1 << 3) | (1 << 4), 4                   # left arm, LeftHand
    if engine:
        R, t, s = A.LocalAlignmentCoreSharded(b["src"], b["s_labels"], b["tgt"][sl], tl_full[sl], group, label, red, rank)
    else:
        R, t, s = O.local_alignment_core_sharded(b["src"], b["s_labels"], b["tgt"][sl], tl_full[sl], group, label, red, rank)
    # the far end is reached on two ranks at once (the arm's points on rank 0 AND, relabelled, on rank 1): the reference's loop
    # keeps the first point of the stitched scan (strict >, Alignment.cpp:521), i.e. rank 0's label — not the smaller label
    arm = np.isin(tl_full, (2, 3, 4))
    tp_, tl_ = (b["tgt"][arm], tl_full[arm] if rank == 0 else np.full(int(arm.sum()), 3, np.int32)) if rank < 2 else (np.zeros((0, 3)), np.zeros(0, np.int32))
    ties = []
    for lab in (4, 2):
        if engine:
            ties.append(A.LocalAlignmentCoreSharded(b["src"], b["s_labels"], tp_, tl_, group, lab, red, rank))
        else:
            ties.append(O.local_alignment_core_sharded(b["src"], b["s_labels"], tp_, tl_, group, lab, red, rank))
    # a rank that fails in a LOCAL stage (here: called with a negative rank) must not leave the others in the next reduce
    failed = None
    if engine:
        try:
            A.LocalAlignmentCoreSharded(b["src"], b["s_labels"], b["tgt"][sl], tl_full[sl], group, label, red, -1 if rank == 1 else rank)
            failed = ""
        except Exception as e:                                          # noqa: BLE001
            failed = str(e)
        A.LocalAlignmentCoreSharded(b["src"], b["s_labels"], b["tgt"][sl], tl_full[sl], group, label, red, rank)   # and the next call is in step again
    q.put((rank, gr, p2, f2, R, t, s, ties, failed))
    dist.barrier()
    dist.destroy_process_group()


def _run_ground(world, engine):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ground_worker, args=(r, world, port, q, engine)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=240)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from oracle import binding as O
    b, views = _two_bodies()
    # the unsharded answer on the stitched scan (rank order = point order)
    P = np.concatenate([v[0] for v in views]); N = np.concatenate([v[1] for v in views])
    F = np.concatenate([views[0][2], views[1][2] + len(views[0][0])])
    gr, p_ref, _, f_ref = O.remove_ground(P, N, F, 0.81)
    for r in range(world):
        assert np.abs(res[r][0] - gr).max() < 1e-9                                          # one ground ray everywhere
    kept = [r for r in range(world) if len(res[r][1])]
    assert len(kept) == 1                                                                   # the largest component lives on one rank
    p_got, f_got = res[kept[0]][1], res[kept[0]][2]
    assert len(p_got) == len(p_ref) and np.array_equal(f_got, f_ref if kept[0] == 0 else f_ref)
    assert np.abs(p_got - p_ref).max() == 0.0
    tl_full = O.part_recog(b["src"], b["s_labels"], b["tgt"])
    want = O.local_alignment_core(b["src"], b["s_labels"], b["tgt"], tl_full, (1 << 2) | (1 << 3) | (1 << 4), 4)
    for r in range(1, world):
        assert all(np.array_equal(np.asarray(a), np.asarray(c)) for a, c in zip(res[0][3:6], res[r][3:6]))
    assert np.abs(res[0][3] - want[0]).max() < 1e-9 and np.abs(res[0][4] - want[1]).max() < 1e-8 and abs(res[0][5] - want[2]) < 1e-10
    arm = np.isin(tl_full, (2, 3, 4))
    P2, L2 = np.concatenate([b["tgt"][arm], b["tgt"][arm]]), np.concatenate([tl_full[arm], np.full(int(arm.sum()), 3, np.int32)])
    differ = 0
    for k, lab in enumerate((4, 2)):
        want = O.local_alignment_core(b["src"], b["s_labels"], P2, L2, (1 << 2) | (1 << 3) | (1 << 4), lab)
        by_label = O.local_alignment_core(b["src"], b["s_labels"], P2[::-1].copy(), L2[::-1].copy(), (1 << 2) | (1 << 3) | (1 << 4), lab)   # label 3 first
        differ += abs(want[2] - by_label[2]) > 1e-6
        for r in range(world):
            g = res[r][6][k]
            assert np.abs(g[0] - want[0]).max() < 1e-9 and np.abs(g[1] - want[1]).max() < 1e-8 and abs(g[2] - want[2]) < 1e-10
    assert differ >= 1                                                   # (the order of the tied points does decide one of the two)
    if engine:
        assert "bad arguments" in res[1][7] and "another rank failed" in res[0][7]


@pytest.mark.parametrize("world", [2, 3])
def test_remove_ground_and_local_alignment_reductions_shard_by_view(world):
    _run_ground(world, engine=False)


@pytest.mark.gpu
def test_gpu_remove_ground_and_local_alignment_sharded_two_ranks_on_one_gpu():
    """mvs_remove_ground_sharded / mvs_local_alignment_core_sharded in two processes sharing the box's GPU, reduced over gloo."""
    _run_ground(2, engine=True)


@pytest.mark.gpu
def test_gpu_init_alignment_sharded_two_ranks_on_one_gpu():
    """mvs_init_alignment_sharded in two processes sharing the box's GPU, reduced over gloo by dist.host_reducer."""
    _run_align(2, engine=True)
