"""Generate tests/golden/*.npz from the independent numpy/scipy restatement (tests/ref_numpy.py).

    python tests/make_golden.py

The reference (zjuzly/MultiViewStitch) holds no fixtures and cannot be built here, so these
vectors are the build's own pins: inputs come from the seeded scene generator,
expected outputs from ref_numpy.  Everything is float64/int arrays; no pickles.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from multiviewstitch_amd import scene as S  # noqa: E402
from tests import ref_numpy as N            # noqa: E402

OUT = os.path.join(HERE, "golden")


def cam_arrays(cam):
    return np.array([cam.fx, cam.fy, cam.cx, cam.cy, *np.asarray(cam.R).ravel(), *np.asarray(cam.t).ravel(), cam.w, cam.h])


def main():
    os.makedirs(OUT, exist_ok=True)
    sc = S.make_scene(0)

    # ---- depth -> model (view 0)
    pts, nrm, tex, faces = N.depth_to_model(sc.depth[0], sc.cams[0], S.MIN_DSP, S.MAX_DSP, S.SMOOTH)
    np.savez_compressed(os.path.join(OUT, "depth_to_model.npz"), depth=sc.depth[0], cam=cam_arrays(sc.cams[0]),
                        min_dsp=S.MIN_DSP, max_dsp=S.MAX_DSP, smooth=S.SMOOTH, points=pts, normals=nrm, tex=tex, faces=faces)

    # ---- target set of both views in the world frame
    tp, tn = [], []
    for k, cam in enumerate(sc.cams):
        p, n, _, _ = N.depth_to_model(sc.depth[k], cam, S.MIN_DSP, S.MAX_DSP, S.SMOOTH)
        s, R, t = sc.srt[k]
        tp.append(s * p @ R.T + t)
        tn.append(n @ R.T)
    tp, tn = np.concatenate(tp), np.concatenate(tn)

    # ---- deformation: sampling, graph, association, smoothing, ARAP, 2 outer iterations
    nodes = N.uniform_sampling(sc.verts, 16)
    it1 = N.deform_iterate(sc.verts, sc.normals, sc.faces, nodes, tp, tn, 1)
    it2 = N.deform_iterate(sc.verts, sc.normals, sc.faces, nodes, tp, tn, 2)
    ang = 0.3
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    rigid = N.arap(sc.verts, sc.faces, nodes, sc.verts[nodes] @ Rz.T + np.array([0.1, -0.2, 0.05]))
    np.savez_compressed(
        os.path.join(OUT, "deform_cfg0.npz"), verts=sc.verts, normals=sc.normals, faces=sc.faces, tp=tp, tn=tn, nodes=nodes,
        knn9=N.knn(sc.verts[nodes], 9), cgal_normals=N.vertex_normals_cgal(sc.verts, sc.faces),
        d2min=it1["assoc"]["d2min"], counts=it1["assoc"]["counts"], top_idx=it1["assoc"]["top_idx"],
        valid=it1["assoc"]["valid"], controls=it1["assoc"]["controls"], controls_smooth=it1["ctrl"],
        it1_pts=it1["pts"], it1_rot=it1["rot"], it1_energies=it1["energies"], it1_iters=it1["iters"],
        it2_pts=it2["pts"], rigid_targets=sc.verts[nodes] @ Rz.T + np.array([0.1, -0.2, 0.05]),
        rigid_pts=rigid["pts"], rigid_rot=rigid["rot"], rigid_energies=rigid["energies"], rigid_iters=rigid["iters"])

    # ---- SRT: closed form and RANSAC on seeded matches with 20 % outliers
    rng = np.random.default_rng(2000)
    c1, c2 = sc.cams[0], sc.cams[1]
    s_gt, ang = 1.07, 0.4
    R_gt = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t_gt = np.array([0.3, -0.1, 0.2])
    m = S.make_matches(rng, c1, c2, s_gt, R_gt, t_gt, n=64)
    tri, st = N.msvc_triples(len(m), 50, 12345)
    s0, R0, t0 = N.srt_fit(m)
    s1, R1, t1 = N.srt_fit(m, c1, c2, tri)
    e1, pm1 = N.srt_residual(m, c1, c2, s1, R1, t1)
    cs, cR, ct = N.srt_compose(s1, R1, t1, s0, R0, t0)
    rs, rR, rt = N.srt_relative(s0, R0, t0, s1, R1, t1)
    np.savez_compressed(os.path.join(OUT, "srt.npz"), matches=m, cam1=cam_arrays(c1), cam2=cam_arrays(c2), triples=tri,
                        seed=12345, state_after=st, closed_s=s0, closed_R=R0, closed_t=t0, ransac_s=s1, ransac_R=R1, ransac_t=t1,
                        ransac_err=e1, ransac_per_match=pm1, compose_s=cs, compose_R=cR, compose_t=ct,
                        rel_s=rs, rel_R=rR, rel_t=rt)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
