"""Offline experiment (CPU, numpy / scipy): the late regime of the config-3 fit (past outer iteration ~170 a patch solve needs
16-35 sweeps: two or three healthy sweeps, then 13 % per sweep) and what mixing successive sweeps does to it.
The system of ARAP iteration 0 is built from the ORACLE's state at the given outer iteration (scripts/two_level_offline.py's
construction), cut into the engine's patches and swept with the engine's strong local solve (Chebyshev, a = 0.01, 26 steps), with
  plain       : x <- G(x)
  mix global  : Anderson(1), one coefficient per coordinate from sums over all free rows
  mix patch   : one coefficient per patch and coordinate from sums over the patch's owned rows, applied to its owned rows
Printed: relative residual (M^-1 norm) of every sweep's input."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
import two_level_offline as T                         # noqa: E402
from oracle import binding as O                       # noqa: E402
from tests.util import scene_and_target               # noqa: E402


def sweep_fn(Aff, dg, owned, loc, a, m):
    ALL = [Aff[L][:, L].tocsr() for L in loc]
    c0, c1, c2 = T.cheb_coefs(a, m)

    def G(x, b):
        r = b - Aff @ x
        xn = x.copy()
        for k, L in enumerate(loc):
            M, d = ALL[k], dg[L][:, None]
            rr = r[L].copy()
            e = np.zeros_like(rr)
            dd = c0 * rr / d
            for s in range(m):
                e += dd
                rr = rr - M @ dd
                dd = c1[s] * dd + c2[s] * rr / d
            xn[owned[k]] += e[:len(owned[k])]
        return xn, r
    return G


def run(G, b, x0, dg, owned, mode, sweeps, start=3, cap=50.0):
    bn = np.sqrt(((b * b) / dg[:, None]).sum(0))
    y = x0.copy()
    hist, gy_prev, f_prev = [], None, None
    for k in range(sweeps):
        gy, r = G(y, b)
        hist.append(float((np.sqrt(((r * r) / dg[:, None]).sum(0)) / bn).max()))
        f = gy - y
        ynew = gy
        if mode != "plain" and f_prev is not None and k >= start:
            df = f - f_prev
            if mode == "global":
                den = (df * df).sum(0)
                gam = np.where(den > 0, (f * df).sum(0) / np.where(den > 0, den, 1), 0.0)
                gam = np.clip(gam, -cap, cap)
                ynew = gy - gam[None, :] * (gy - gy_prev)
            else:
                ynew = gy.copy()
                for o in owned:
                    den = (df[o] * df[o]).sum(0)
                    gam = np.where(den > 0, (f[o] * df[o]).sum(0) / np.where(den > 0, den, 1), 0.0)
                    gam = np.clip(gam, -cap, cap)
                    ynew[o] = gy[o] - gam[None, :] * (gy[o] - gy_prev[o])
        gy_prev, f_prev, y = gy, f, ynew
    return hist


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    upto = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    t0 = time.time()
    sc, tp, tn, _ = scene_and_target(cfg)
    print(f"scene {time.time() - t0:.0f} s", flush=True)
    o = O.Deform(sc.verts, sc.normals, sc.faces)
    o.sample_nodes(16)
    o.set_target(tp, tn)
    p = O.Params.default()
    t0 = time.time()
    for q in range(0, upto, 10):
        o.iterate(p, min(10, upto - q))
        print(f"outer iteration {q + 10} ({time.time() - t0:.0f} s)", flush=True)
    pts, free, Aff, b, x0, dg = T.build(o, sc.faces, p)
    owned, loc = T.patches_of(pts[free], Aff)
    np.savez_compressed(f"/tmp/late_system_{cfg}_{upto}.npz", data=Aff.data, indices=Aff.indices, indptr=Aff.indptr, b=b, x0=x0, dg=dg, pts=pts[free])
    print(f"{Aff.shape[0]} free rows, {len(owned)} patches")
    G = sweep_fn(Aff, dg, owned, loc, 0.01, 26)
    for mode in ("plain", "global", "patch"):
        h = run(G, b, x0, dg, owned, mode, 24)
        print(f"{mode:7s} " + " ".join(f"{v:.1e}" for v in h), flush=True)


if __name__ == "__main__":
    main()
