"""The reference's file hand-off formats at scan scale (f1; csrc/host_io.cpp, host code): a 2.03 M-vertex / 4.05 M-facet scan with
normals as .obj (PlyObj.cpp's WriteObj / ReadObj layout, 324 MB) and as .npts (113 MB), one 1280 x 960 raster — wall clock of the
second call of each (the first pays the Python mirror's `import torch`).  Files under /dev/shm.  Prints markdown rows."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiviewstitch_amd import io as mio
from tests.util import body_scene

sc = body_scene(5, 30, 450)
P, N, F = sc["tgt"], sc["t_nrm"], sc["t_faces"]
d = tempfile.mkdtemp(dir="/dev/shm")
rows = []


def best(fn, reps=3):
    t = []
    for _ in range(reps):
        a = time.perf_counter()
        fn()
        t.append(time.perf_counter() - a)
    return min(t)


obj, npts, raw = os.path.join(d, "scan.obj"), os.path.join(d, "scan.npts"), os.path.join(d, "d.raw")
mio.WriteObj(obj, P, N, F)
mb = os.path.getsize(obj) / 1e6
rows.append(("mvs_obj_write", f"{len(P)} vertices + normals, {len(F)} facets ({mb:.0f} MB)", best(lambda: mio.WriteObj(obj, P, N, F)), mb))
rows.append(("mvs_obj_read (count call + fill call)", "the same file", best(lambda: mio.ReadObj(obj)), mb))
mio.write_npts(npts, P, N)
mb2 = os.path.getsize(npts) / 1e6
rows.append(("mvs_npts_write", f"{len(P)} points + normals ({mb2:.0f} MB)", best(lambda: mio.write_npts(npts, P, N)), mb2))
rows.append(("mvs_npts_read (count call + fill call)", "the same file", best(lambda: mio.read_npts(npts)), mb2))
dep = np.random.default_rng(0).random((960, 1280)).astype(np.float32)
rows.append(("mvs_depth_raw_write", "1280 x 960 raster", best(lambda: mio.SaveDepth(raw, dep)), 4.9))
rows.append(("mvs_depth_raw_read", "1280 x 960 raster", best(lambda: mio.LoadDepth(raw, 1280, 960)), 4.9))
shutil.rmtree(d)
print("| entry | input | s per call | MB/s |")
print("|---|---|---|---|")
for name, what, s, mb_ in rows:
    print(f"| `{name}` | {what} | {s:.3f} | {mb_ / s:.0f} |")
