"""Seeded synthetic scenes for the SRT + deformation path.

Host-side input generation only (numpy, optionally torch for the ray caster);
nothing here is on the timed path.  A scene holds

* a closed 2-manifold template (geodesic sphere of frequency ``n``:
  V = 10 n^2 + 2, radius ``1 + 0.15 * sum a_m Y_m(dir)``),
* a target surface = the template's radial function times a smooth warp
  (8 Gaussian bumps, amplitude <= 0.05), so the non-rigid answer is known,
* ``n_views`` pinhole cameras on a ring (elevation 10 deg) with
  ``cx = w/2 - 0.5`` so the reference's Camera derives w,h exactly
  (R/Camera/Camera.cpp:135-136), each expressed in its own "sequence frame"
  related to the world by a ground-truth similarity (s_k, R_k, t_k),
* float32 inverse-depth rasters (R/Common/Utils.h:166-185) rendered by
  ray/surface intersection inside [MinDsp, MaxDsp] = [0.0025, 0.3]
  (R/config.txt:23-24).
"""
from __future__ import annotations

import dataclasses
import math
from typing import List

import numpy as np

MIN_DSP, MAX_DSP = 0.0025, 0.3
SMOOTH = 1.0   # Depth2Model m_fSmoothThreshold for the synthetic rasters (R/config.txt:38 uses 0.12 on real data)

# (n_views, w, h, mesh frequency n, focal factor f = fx / w)  — BASELINE.json configs
CONFIGS = {
    0: dict(n_views=2, w=80, h=60, n=6, f=1.40),          # tiny, unit tests
    1: dict(n_views=2, w=320, h=240, n=19, f=1.40),       # ~50K pts, ~512 nodes
    2: dict(n_views=4, w=640, h=480, n=37, f=1.56),       # ~500K pts, ~2K nodes
    3: dict(n_views=8, w=1280, h=960, n=74, f=1.09),      # ~2M pts, ~8K nodes
    4: dict(n_views=16, w=1280, h=960, n=105, f=1.09),    # ~4M pts, ~16K nodes
    5: dict(n_views=8, w=1280, h=960, n=147, f=1.09),     # ~2M pts, ~32K nodes in 16 per-part graphs (partwise.py)
}


@dataclasses.dataclass
class Camera:
    """Mirror of mvs_camera (include/mvs.h); R row-major, Xc = R Xw + t."""
    fx: float
    fy: float
    cx: float
    cy: float
    R: np.ndarray
    t: np.ndarray
    w: int
    h: int


@dataclasses.dataclass
class Scene:
    config: int
    seed: int
    verts: np.ndarray          # (V,3) float64 template
    normals: np.ndarray        # (V,3) template vertex normals (PlyObj rule)
    faces: np.ndarray          # (F,3) int32
    cams: List[Camera]         # local-frame cameras, one per view
    srt: List[tuple]           # ground-truth (s, R, t): world = s R local + t
    depth: List[np.ndarray]    # (h,w) float32 inverse depth per view
    bumps_a: np.ndarray
    warp_c: np.ndarray
    warp_A: np.ndarray

    def target_radius(self, d):
        return _r_target(d, self.bumps_a, self.warp_c, self.warp_A)


# ------------------------------------------------------------------ template --
def geodesic_sphere(n: int):
    """Class-I geodesic subdivision of the icosahedron, frequency n."""
    phi = (1.0 + math.sqrt(5.0)) / 2.0
    ico = np.array([[-1, phi, 0], [1, phi, 0], [-1, -phi, 0], [1, -phi, 0],
                    [0, -1, phi], [0, 1, phi], [0, -1, -phi], [0, 1, -phi],
                    [phi, 0, -1], [phi, 0, 1], [-phi, 0, -1], [-phi, 0, 1]], dtype=np.float64)
    ico /= np.linalg.norm(ico, axis=1, keepdims=True)
    tri = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
           (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
           (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    key2id = {}
    pts = []

    def vid(a, b, c, i, j, k):
        # canonical key: corner, edge point (sorted endpoints + parameter) or face interior
        w = [(a, i), (b, j), (c, k)]
        nz = [(v, x) for v, x in w if x != 0]
        if len(nz) == 1:
            key = (nz[0][0],)
        elif len(nz) == 2:
            (v0, x0), (v1, x1) = sorted(nz)
            key = (v0, v1, x0)
        else:
            key = (a, b, c, i, j)
        r = key2id.get(key)
        if r is None:
            r = len(pts)
            key2id[key] = r
            p = (ico[a] * i + ico[b] * j + ico[c] * k) / n
            pts.append(p / np.linalg.norm(p))
        return r

    faces = []
    for (a, b, c) in tri:
        ids = {}
        for i in range(n + 1):
            for j in range(n + 1 - i):
                ids[(i, j)] = vid(a, b, c, n - i - j, i, j)
        for i in range(n):
            for j in range(n - i):
                faces.append((ids[(i, j)], ids[(i + 1, j)], ids[(i, j + 1)]))
                if i + j < n - 1:
                    faces.append((ids[(i + 1, j)], ids[(i + 1, j + 1)], ids[(i, j + 1)]))
    return np.asarray(pts, dtype=np.float64), np.asarray(faces, dtype=np.int32)


def _bumps(d, a):
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    Y = (x, y, z, 2.0 * x * y, 2.0 * y * z, x * x - y * y)
    s = 0.0
    for m in range(6):
        s = s + a[m] * Y[m]
    return 1.0 + 0.15 * s / 3.0


def _r_target(d, a, c, A, sigma=0.5):
    r = _bumps(d, a)
    wsum = 0.0
    for k in range(c.shape[0]):
        diff = d - c[k]
        wsum = wsum + A[k] * np.exp(-(diff * diff).sum(-1) / (2.0 * sigma * sigma))
    return r * (1.0 + wsum)


def vertex_normals_plyobj(verts, faces):
    """Mesh::CalculateVertexNormals (R/PlyObj/PlyObj.cpp:139-185), numpy, input generation only."""
    p0, p1, p2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    n = np.cross(p1 - p0, p2 - p1)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    acc = np.zeros_like(verts)
    cnt = np.zeros(len(verts))
    for k in range(3):
        np.add.at(acc, faces[:, k], n)
        np.add.at(cnt, faces[:, k], 1.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        m = acc / cnt[:, None]
        return m / np.linalg.norm(m, axis=1, keepdims=True)


# ------------------------------------------------------------------- cameras --
def _look_at(eye):
    fwd = -eye / np.linalg.norm(eye)                 # camera +z looks at the origin
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd])                 # rows: camera axes in world coords
    t = -R @ eye
    return R, t


def _rot_axis(axis, ang):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)


def render_inverse_depth(cam_R, cam_t, fx, fy, cx, cy, w, h, radius_fn, r_max, steps=64, bisect=44):
    """Inverse depth (1/z_cam) of the star-shaped surface |p| = radius_fn(p/|p|); 0 = background."""
    o = -cam_R.T @ cam_t
    u = np.arange(w, dtype=np.float64)
    v = np.arange(h, dtype=np.float64)
    uu, vv = np.meshgrid(u, v)
    dc = np.stack([(uu - cx) / fx, (vv - cy) / fy, np.ones_like(uu)], -1).reshape(-1, 3)
    dw = dc @ cam_R                                    # R^T applied to rows
    # bounding sphere hit
    a = (dw * dw).sum(-1)
    b = 2.0 * (dw @ o)
    c = float(o @ o) - r_max * r_max
    disc = b * b - 4 * a * c
    hit = np.nonzero(disc > 0)[0]
    out = np.zeros(w * h, dtype=np.float64)
    if hit.size == 0:
        return out.reshape(h, w).astype(np.float32)
    sq = np.sqrt(disc[hit])
    z0 = (-b[hit] - sq) / (2 * a[hit])
    z1 = (-b[hit] + sq) / (2 * a[hit])
    d = dw[hit]

    def f(z):
        p = o[None, :] + z[:, None] * d
        r = np.linalg.norm(p, axis=1)
        return r - radius_fn(p / r[:, None])

    lo = z0.copy()
    flo = f(lo)
    found = np.zeros(hit.size, dtype=bool)
    hi = z0.copy()
    for s in range(1, steps + 1):
        z = z0 + (z1 - z0) * (s / steps)
        fz = f(z)
        newly = (~found) & (flo > 0) & (fz <= 0)
        hi = np.where(newly, z, hi)
        found |= newly
        adv = ~found
        lo = np.where(adv, z, lo)
        flo = np.where(adv, fz, flo)
    idx = np.nonzero(found)[0]
    lo, hi = lo[idx], hi[idx]
    dsel = d[idx]
    for _ in range(bisect):
        mid = 0.5 * (lo + hi)
        p = o[None, :] + mid[:, None] * dsel
        r = np.linalg.norm(p, axis=1)
        fm = r - radius_fn(p / r[:, None])
        pos = fm > 0
        lo = np.where(pos, mid, lo)
        hi = np.where(pos, hi, mid)
    z = 0.5 * (lo + hi)
    out[hit[idx]] = 1.0 / z
    return out.reshape(h, w).astype(np.float32)


def render_inverse_depth_torch(cam_R, cam_t, fx, fy, cx, cy, w, h, a, c, A, r_max, device, steps=64, bisect=44):
    """Same ray caster on torch tensors (float64), for large rasters on the GPU box.  Input
    generation only; last-bit differences from the numpy path are irrelevant because oracle
    and engine always consume the same rasters."""
    import torch
    dd = dict(dtype=torch.float64, device=device)
    R = torch.tensor(cam_R, **dd)
    o = -(R.T @ torch.tensor(cam_t, **dd))
    a_t, c_t, A_t = torch.tensor(a, **dd), torch.tensor(c, **dd), torch.tensor(A, **dd)

    def radius(d):
        x, y, z = d[:, 0], d[:, 1], d[:, 2]
        s = a_t[0] * x + a_t[1] * y + a_t[2] * z + a_t[3] * 2.0 * x * y + a_t[4] * 2.0 * y * z + a_t[5] * (x * x - y * y)
        r = 1.0 + 0.15 * s / 3.0
        diff = d[:, None, :] - c_t[None, :, :]
        ws = (A_t[None, :] * torch.exp(-(diff * diff).sum(-1) / (2.0 * 0.5 * 0.5))).sum(-1)
        return r * (1.0 + ws)

    vv, uu = torch.meshgrid(torch.arange(h, **dd), torch.arange(w, **dd), indexing="ij")
    dc = torch.stack([(uu - cx) / fx, (vv - cy) / fy, torch.ones_like(uu)], -1).reshape(-1, 3)
    dw = dc @ R
    qa = (dw * dw).sum(-1)
    qb = 2.0 * (dw @ o)
    qc = float(o @ o) - r_max * r_max
    disc = qb * qb - 4 * qa * qc
    hit = torch.nonzero(disc > 0)[:, 0]
    out = torch.zeros(w * h, **dd)
    if hit.numel():
        sq = torch.sqrt(disc[hit])
        z0 = (-qb[hit] - sq) / (2 * qa[hit])
        z1 = (-qb[hit] + sq) / (2 * qa[hit])
        d = dw[hit]

        def f(z, dsel=d):
            p = o[None, :] + z[:, None] * dsel
            r = torch.linalg.norm(p, dim=1)
            return r - radius(p / r[:, None])

        lo, hi = z0.clone(), z0.clone()
        flo = f(lo)
        found = torch.zeros(hit.numel(), dtype=torch.bool, device=device)
        for s in range(1, steps + 1):
            z = z0 + (z1 - z0) * (s / steps)
            fz = f(z)
            newly = (~found) & (flo > 0) & (fz <= 0)
            hi = torch.where(newly, z, hi)
            found |= newly
            adv = ~found
            lo = torch.where(adv, z, lo)
            flo = torch.where(adv, fz, flo)
        idx = torch.nonzero(found)[:, 0]
        lo, hi, dsel = lo[idx], hi[idx], d[idx]
        for _ in range(bisect):
            mid = 0.5 * (lo + hi)
            pos = f(mid, dsel) > 0
            lo = torch.where(pos, mid, lo)
            hi = torch.where(pos, hi, mid)
        out[hit[idx]] = 1.0 / (0.5 * (lo + hi))
    return out.reshape(h, w).to(torch.float32).cpu().numpy()


def make_scene(config: int = 1, seed: int | None = None, device=None, views=None, **override) -> Scene:
    """device: None -> numpy ray caster; a torch device -> torch ray caster.
    views: optional iterable of view ids to render (others get depth None) — a rank renders its shard."""
    cfg = dict(CONFIGS[config])
    cfg.update(override)
    seed = 1000 + config if seed is None else seed
    rng = np.random.default_rng(seed)
    n_views, w, h, n, f = cfg["n_views"], cfg["w"], cfg["h"], cfg["n"], cfg["f"]
    a = rng.uniform(-1, 1, 6)
    c = rng.normal(size=(8, 3))
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    A = rng.uniform(-0.05, 0.05, 8)
    dirs, faces = geodesic_sphere(n)
    verts = dirs * _bumps(dirs, a)[:, None]
    normals = vertex_normals_plyobj(verts, faces)
    r_max = 1.02 * float(_r_target(dirs, a, c, A).max())
    cams, srt, depth = [], [], []
    dist = cfg.get("dist", 5.0)
    for k in range(n_views):
        yaw = 2 * math.pi * k / n_views
        el = math.radians(10.0)
        eye = dist * np.array([math.cos(el) * math.cos(yaw), math.cos(el) * math.sin(yaw), math.sin(el)])
        Rc, tc = _look_at(eye)
        fx = fy = f * w
        cx, cy = w / 2 - 0.5, h / 2 - 0.5
        if views is not None and k not in views:
            d = None
        elif device is not None:
            d = render_inverse_depth_torch(Rc, tc, fx, fy, cx, cy, w, h, a, c, A, r_max, device)
        else:
            d = render_inverse_depth(Rc, tc, fx, fy, cx, cy, w, h, lambda q: _r_target(q, a, c, A), r_max)
        # ground-truth similarity of this view's sequence frame: world = s R local + t
        s = float(rng.uniform(0.9, 1.1))
        tilt = _rot_axis(rng.normal(size=3), math.radians(float(rng.uniform(0, 5))))
        R = _rot_axis(np.array([0, 0, 1.0]), float(rng.uniform(-math.pi, math.pi))) @ tilt
        t = rng.uniform(-0.2, 0.2, 3)
        # local-frame camera: Xc_local = (Rc R) p_l + (Rc t + tc)/s ; depth_local = depth / s
        cams.append(Camera(fx, fy, cx, cy, (Rc @ R).copy(), ((Rc @ t + tc) / s).copy(), w, h))
        srt.append((s, R.copy(), t.copy()))
        depth.append(None if d is None else (d.astype(np.float64) * s).astype(np.float32))
    return Scene(config, seed, verts, normals, faces, cams, srt, depth, a, c, A)


def make_matches(rng, cam1: Camera, cam2: Camera, s, R, t, n=64, outlier_frac=0.2, noise_px=0.5):
    """3-D matches p (frame 1) <-> q = s R p + t (frame 2) seen by both cameras."""
    # points in front of camera 1 at depth ~5, spread over the image
    u = rng.uniform(0.2 * cam1.w, 0.8 * cam1.w, n)
    v = rng.uniform(0.2 * cam1.h, 0.8 * cam1.h, n)
    z = rng.uniform(4.0, 6.0, n)
    pc = np.stack([(u - cam1.cx) * z / cam1.fx, (v - cam1.cy) * z / cam1.fy, z], 1)
    p = (pc - cam1.t) @ cam1.R                      # R^T (pc - t)
    q = s * (p @ R.T) + t
    q = q + rng.normal(scale=noise_px * 5.0 / cam1.fx, size=q.shape)
    n_out = int(round(outlier_frac * n))
    if n_out:
        idx = rng.choice(n, n_out, replace=False)
        q[idx] += rng.normal(scale=0.5, size=(n_out, 3))
    return np.ascontiguousarray(np.concatenate([p, q], 1))


def make_sequence(n_frames: int = 5, w: int = 320, h: int = 240, dyaw_deg: float = 3.0, seed: int = 77, f: float = 1.2,
                  dist: float = 5.0, device=None):
    """One key-frame sequence in ONE frame of reference (the input of Processor::CheckConsistency): the synthetic
    surface seen from `n_frames` cameras `dyaw_deg` apart on the ring.  -> (cams, depths float32 [n, h, w])."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, 6)
    c = rng.normal(size=(8, 3))
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    A = rng.uniform(-0.05, 0.05, 8)
    dirs, _ = geodesic_sphere(12)
    r_max = 1.02 * float(_r_target(dirs, a, c, A).max())
    fx = fy = f * w
    cx, cy = w / 2 - 0.5, h / 2 - 0.5
    cams, depths = [], []
    for k in range(n_frames):
        yaw = math.radians(dyaw_deg) * k
        el = math.radians(10.0)
        eye = dist * np.array([math.cos(el) * math.cos(yaw), math.cos(el) * math.sin(yaw), math.sin(el)])
        Rc, tc = _look_at(eye)
        if device is not None:
            d = render_inverse_depth_torch(Rc, tc, fx, fy, cx, cy, w, h, a, c, A, r_max, device)
        else:
            d = render_inverse_depth(Rc, tc, fx, fy, cx, cy, w, h, lambda q: _r_target(q, a, c, A), r_max)
        cams.append(Camera(fx, fy, cx, cy, Rc.copy(), tc.copy(), w, h))
        depths.append(d)
    return cams, np.stack(depths)
