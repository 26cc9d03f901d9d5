#!/usr/bin/env python3
"""Design data for the point-streaming association (round 4, CPU only, oracle as the source of node trajectories):
per node and outer iteration — the temporal search bound (sqrt(d2min of the previous pass) + the node's move), how many target
points lie inside the candidate radius sqrt(2) * bound, how many grid tiles the candidate sphere touches, against the exact ball
population.  Usage: scripts/stream_stats.py [config] [outer iterations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scipy.spatial import cKDTree  # noqa: E402

from oracle import binding as O  # noqa: E402
from tests.util import scene_and_target  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n_outer = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    t0 = time.time()
    sc, tp, tn, _ = scene_and_target(cfg)
    print(f"config {cfg}: P={len(tp)} V={len(sc.verts)} ({time.time() - t0:.1f}s)")
    o = O.Deform(sc.verts, sc.normals, sc.faces)
    K = o.sample_nodes(16)
    nodes = o.nodes()
    o.set_target(tp, tn)
    tgt = O.Target(tp, tn)
    tree = cKDTree(tp.astype(np.float32).astype(np.float64))
    p = O.Params.default()
    # the engine's grid: ~16 points per occupied cell
    ext = float((tp.max(0) - tp.min(0)).max())
    hh = ext / 128
    cells = np.floor((tp - tp.min(0)) / hh).astype(np.int64)
    occ = len(np.unique(cells[:, 0] + 1000 * cells[:, 1] + 1000000 * cells[:, 2]))
    h = hh * np.sqrt(16.0 / (len(tp) / occ))
    print(f"K={K} grid h={h:.5f} (ext {ext:.3f}), coarse cell {8 * h:.4f}")
    prev_d2, prev_pos = None, None
    for it in range(n_outer):
        v, nrm = o.vertices(), o.normals()
        npos, nnrm = v[nodes], nrm[nodes]
        a = tgt.associate(npos, nnrm, p)
        d = np.sqrt(a["d2min"].astype(np.float64))
        nb = a["counts"][:, 0]
        line = f"it {it}: valid {int(a['valid'].sum())}  dmin/h median {np.median(d) / h:.3f} p90 {np.percentile(d, 90) / h:.2f} max {d.max() / h:.1f} | ball n: median {int(np.median(nb))} p90 {int(np.percentile(nb, 90))} p99 {int(np.percentile(nb, 99))} max {nb.max()}"
        if prev_d2 is not None:
            move = np.linalg.norm(npos - prev_pos, axis=1)
            bound = np.sqrt(prev_d2.astype(np.float64)) + move
            cand_r = np.sqrt(2.0) * bound * 1.001
            ncand = np.array([len(x) for x in tree.query_ball_point(npos, cand_r)])
            tiles = np.prod(np.floor((npos + cand_r[:, None] - tp.min(0)) / (8 * h)) - np.floor((npos - cand_r[:, None] - tp.min(0)) / (8 * h)) + 1, axis=1)
            line += (f"\n      move/h median {np.median(move) / h:.3f} p90 {np.percentile(move, 90) / h:.3f} | bound/dmin median {np.median(bound / np.maximum(d, 1e-12)):.2f}"
                     f" | candidates (d <= sqrt2*bound): median {int(np.median(ncand))} p90 {int(np.percentile(ncand, 90))} p99 {int(np.percentile(ncand, 99))}"
                     f" | nodes with <=32 / <=64 / <=128 candidates: {np.mean(ncand <= 32):.3f} {np.mean(ncand <= 64):.3f} {np.mean(ncand <= 128):.3f}"
                     f" | coarse tiles touched: mean {tiles.mean():.2f} p99 {np.percentile(tiles, 99):.0f}; cand_r/h p50 {np.median(cand_r) / h:.2f} p90 {np.percentile(cand_r, 90) / h:.2f}")
            for cap in (32, 64, 128):
                sel = ncand <= cap
                line += f"\n      cap {cap}: {sel.sum()} nodes stream, their tiles {tiles[sel].sum():.0f}, rest {K - sel.sum()} (of which ball>25 rows approx: {(2 * np.sqrt(2.0) * d[~sel] / h + 1 > 5).sum()})"
        print(line, flush=True)
        prev_d2, prev_pos = a["d2min"].copy(), npos.copy()
        o.iterate(p, 1)


if __name__ == "__main__":
    main()
