"""View-sharded outer iteration: one process per GPU, target points sharded by
view, template mesh / node graph replicated.

Per outer iteration the ranks exchange, over ``torch.distributed`` (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests):

  1. all-reduce(MIN)  of d2min[K]           float32   (32 KB at K = 8 K)
  2. all-gather       of ONE packed buffer per rank: the local best-8 records[K,8] (48 B each) | counts[K,2] int32
  3. every rank merges the N record sets with the same total order -> identical
     node targets everywhere; smoothing + ARAP run replicated, so the meshes
     never diverge and nothing else is communicated.
  2'/3'. (owner-merges, for N >= 4: ``buffers_owner``) all-to-all of the records / counts by node block, each rank merges the block
     it owns, all-gather of the merged targets (25 B per node) — K * 392 B into a rank instead of N * K * 392 B, 1/N of the merge.

The exchange is written against a small "shard" protocol so the same code
drives the HIP engine (``EngineShard``) and, in the CPU tests, a checker shard
built on the oracle (tests/test_dist_gloo.py).  A shard provides
``buffers(K)``, ``dmin(buf)``, ``select(d2min, rec, cnt)``,
``merge(rec_all, cnt_all, nranks)``, ``solve()``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from ._lib import CAND_DTYPE

REC_BYTES = CAND_DTYPE.itemsize          # 48


class EngineShard:
    """Adapter of ``Deformation`` (HIP engine) to the shard protocol; buffers are torch CUDA tensors
    whose addresses go straight into the C-ABI (mvs_deform_assoc_*)."""

    def __init__(self, deform, device):
        self.d = deform
        self.device = device
        # Engine kernels and the collectives must be ordered without host syncs: the engine is put on a torch stream of
        # its own and ``sharded_step`` makes that stream torch's current one, against which ProcessGroupNCCL orders its
        # collectives (events).  torch's DEFAULT stream cannot serve: its handle is 0, which mvs_deform_set_stream
        # reads as "restore the handle's own stream" — a non-blocking stream the collectives would not wait for.
        self.stream = torch.cuda.Stream(device)
        self.d.set_stream(self.stream.cuda_stream)

    def buffers(self, K, world):
        dev = self.device
        with torch.cuda.stream(self.stream):
            return self._buffers(K, world, dev)

    @staticmethod
    def _buffers(K, world, dev):
        # records and counts of a rank live in ONE buffer [K*8 records | K*2 counts] so that one all-gather moves both
        R, Cn = K * 8 * REC_BYTES, K * 2 * 4
        pack = torch.empty(R + Cn, dtype=torch.uint8, device=dev)
        return dict(d2min=torch.empty(K, dtype=torch.float32, device=dev), pack=pack, rec=pack[:R], cnt=pack[R:].view(torch.int32),
                    pack_all=torch.empty(world * (R + Cn), dtype=torch.uint8, device=dev))

    @staticmethod
    def _buffers_two_arrays(K, world, dev):
        return dict(d2min=torch.empty(K, dtype=torch.float32, device=dev),
                    rec=torch.empty(K * 8 * REC_BYTES, dtype=torch.uint8, device=dev),
                    cnt=torch.empty(K * 2, dtype=torch.int32, device=dev),
                    rec_all=torch.empty(world * K * 8 * REC_BYTES, dtype=torch.uint8, device=dev),
                    cnt_all=torch.empty(world * K * 2, dtype=torch.int32, device=dev))

    @staticmethod
    def _buffers_owner(K, world, rank, dev):
        bn, blocks = node_blocks(K, world)
        mine = blocks[rank][1] - blocks[rank][0]
        stride = (bn * 25 + 31) // 32 * 32
        return dict(d2min=torch.empty(K, dtype=torch.float32, device=dev),
                    rec=torch.empty(K * 8 * REC_BYTES, dtype=torch.uint8, device=dev), cnt=torch.empty(K * 2, dtype=torch.int32, device=dev),
                    rec_in=torch.empty(max(1, world * mine * 8 * REC_BYTES), dtype=torch.uint8, device=dev),
                    cnt_in=torch.empty(max(1, world * mine * 2), dtype=torch.int32, device=dev),
                    blk=torch.zeros(stride, dtype=torch.uint8, device=dev), blk_all=torch.empty(world * stride, dtype=torch.uint8, device=dev),
                    owner=dict(rank=rank, bn=bn, blocks=blocks, stride=stride))

    def buffers_owner(self, K, world, rank):
        with torch.cuda.stream(self.stream):
            return self._buffers_owner(K, world, rank, self.device)

    def merge_block(self, b, world):
        o = b["owner"]
        k0, k1 = o["blocks"][o["rank"]]
        if k1 > k0:
            self.d.assoc_merge_block(b["rec_in"].data_ptr(), b["cnt_in"].data_ptr(), world, k0, k1, o["bn"], b["blk"].data_ptr())

    def install(self, b, world):
        o = b["owner"]
        self.d.set_node_targets_dev(b["blk_all"].data_ptr(), world, o["bn"], o["stride"])

    def dmin(self, b):
        self.d.assoc_dmin(b["d2min"].data_ptr())

    def select(self, b):
        self.d.assoc_select(b["d2min"].data_ptr(), b["rec"].data_ptr(), b["cnt"].data_ptr())

    def merge(self, b, world):
        if "pack_all" in b:
            self.d.assoc_merge_packed(b["pack_all"].data_ptr(), world)
        else:
            self.d.assoc_merge(b["rec_all"].data_ptr(), b["cnt_all"].data_ptr(), world)

    def solve(self, sync=True):
        return self.d.solve(sync)


def sharded_step(shard, bufs, world: int, group=None, sync: bool = True, timers: dict | None = None):
    """One outer iteration of Deformation::Deform's body over view-sharded targets.
    sync=False leaves the step enqueued (no host synchronisation, returns None).
    timers (optional, bench.py): {"all_reduce": [], "all_gather": [], "sync": bool} — every collective is bracketed by a
    pair of events on the current stream (or, with "sync", by drained-stream wall-clock stamps: gloo collectives are host
    round trips) and the pair appended to the list."""
    stream = getattr(shard, "stream", None)
    if stream is not None:
        with torch.cuda.stream(stream):
            return _sharded_step(shard, bufs, world, group, sync, timers)
    return _sharded_step(shard, bufs, world, group, sync, timers)


class _Bracket:
    def __init__(self, timers, name):
        self.t, self.name = timers, name

    def __enter__(self):
        if self.t is None:
            return
        if self.t.get("sync"):
            torch.cuda.synchronize()
            import time
            self.a = time.perf_counter()
        else:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.t is None:
            return
        if self.t.get("sync"):
            torch.cuda.synchronize()
            import time
            b = time.perf_counter()
        else:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
        self.t[self.name].append((self.a, b))


def _sharded_exchange_owner(shard, bufs, world, group, timers):
    """owner-merges exchange (N >= 4): every rank receives the records of ITS node block only (K * 392 B in, not N * K * 392 B),
    merges that block, and the blocks' targets (25 B per node) are all-gathered and installed"""
    o = bufs["owner"]
    sizes = [b[1] - b[0] for b in o["blocks"]]
    mine = sizes[o["rank"]]
    with _Bracket(timers, "all_gather"):
        if world > 1:
            dist.all_to_all_single(bufs["rec_in"][:world * mine * 8 * REC_BYTES], bufs["rec"], [mine * 8 * REC_BYTES] * world,
                                   [n * 8 * REC_BYTES for n in sizes], group=group)
            dist.all_to_all_single(bufs["cnt_in"][:world * mine * 2], bufs["cnt"], [mine * 2] * world, [n * 2 for n in sizes], group=group)
        else:
            bufs["rec_in"].copy_(bufs["rec"])
            bufs["cnt_in"].copy_(bufs["cnt"])
        shard.merge_block(bufs, world)
        if world > 1:
            dist.all_gather_into_tensor(bufs["blk_all"], bufs["blk"], group=group)
        else:
            bufs["blk_all"].copy_(bufs["blk"])
    shard.install(bufs, world)


def _sharded_step(shard, bufs, world, group, sync, timers=None):
    shard.dmin(bufs)
    if world > 1:
        with _Bracket(timers, "all_reduce"):
            dist.all_reduce(bufs["d2min"], op=dist.ReduceOp.MIN, group=group)
    shard.select(bufs)
    if "owner" in bufs:
        _sharded_exchange_owner(shard, bufs, world, group, timers)
    elif world > 1 and "pack" in bufs:
        with _Bracket(timers, "all_gather"):
            dist.all_gather_into_tensor(bufs["pack_all"], bufs["pack"], group=group)      # records and counts in one collective
        shard.merge(bufs, world)
    elif world > 1:
        with _Bracket(timers, "all_gather"):
            dist.all_gather_into_tensor(bufs["rec_all"], bufs["rec"], group=group)
            dist.all_gather_into_tensor(bufs["cnt_all"], bufs["cnt"], group=group)
        shard.merge(bufs, world)
    elif "pack" in bufs:
        shard.merge(dict(bufs, pack_all=bufs["pack"]), 1)
    else:
        shard.merge(dict(bufs, rec_all=bufs["rec"], cnt_all=bufs["cnt"]), 1)
    return shard.solve(sync) if not sync else shard.solve()


_REDUCE_FN = None


def host_reducer(group=None, device=None):
    """The all-reduce the view-sharded Alignment entries call back (mvs.h: mvs_reduce_fn — small HOST vectors of doubles,
    op 0 = sum, 1 = min), over ``torch.distributed``: on the CPU for gloo, through a device tensor for nccl (RCCL).
    Returns a ctypes function pointer; keep it referenced while the call runs."""
    import ctypes as C
    global _REDUCE_FN
    if _REDUCE_FN is None:
        _REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)

    def reduce(_ctx, v, n, op):
        try:
            if not dist.is_initialized() or dist.get_world_size(group) == 1:
                return 0
            a = np.ctypeslib.as_array(v, shape=(n,))
            t = torch.from_numpy(a.copy())
            if device is not None:
                t = t.to(device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MIN, group=group)
            a[:] = t.cpu().numpy()
            return 0
        except Exception:                       # noqa: BLE001  (an exception must not unwind through the C frame: reported as an error code)
            import traceback
            traceback.print_exc()
            return 1
    return _REDUCE_FN(reduce)


def node_blocks(K: int, world: int):
    """owner-merges exchange: (block_nodes, [(k0, k1) per rank]) — block_nodes = ceil(K / world), the layout mvs.h fixes"""
    bn = max(1, -(-K // world))
    return bn, [(min(K, r * bn), min(K, (r + 1) * bn)) for r in range(world)]


def view_shards(n_views: int, world: int):
    """views of rank r: contiguous blocks, so global point indices follow view order."""
    per = [n_views // world + (1 if r < n_views % world else 0) for r in range(world)]
    out, s = [], 0
    for r in range(world):
        out.append(list(range(s, s + per[r])))
        s += per[r]
    return out


def exclusive_offsets(local_count: int, world: int, device, group=None):
    """global index base of this rank's points = sum of the counts of the lower ranks."""
    t = torch.zeros(world, dtype=torch.int64, device=device)
    mine = torch.tensor([local_count], dtype=torch.int64, device=device)
    if world > 1:
        dist.all_gather_into_tensor(t, mine, group=group)
    else:
        t[0] = local_count
    counts = t.cpu().numpy()
    return np.concatenate([[0], np.cumsum(counts)])[:-1], counts
