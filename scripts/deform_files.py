"""Processor::Deform on FILES at scan scale (f1; R/Processor/Processor.cpp:1119-1137 = mvs_processor_deform): Model.obj with the
2.03 M-vertex / 4.05 M-facet scan (324 MB), meanbody.obj with the 9 K-vertex template, the parts file -> deform.obj.  Wall clock of
the whole call (second of two: the first pays `import torch` and the cold GPU), files under /dev/shm."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiviewstitch_amd import io as mio, processor
from tests.util import body_scene
from tests.test_io import write_parts

sc = body_scene(5, 30, 450)
d = tempfile.mkdtemp(dir="/dev/shm")
model, templ, parts, out = (os.path.join(d, n) for n in ("Model.obj", "meanbody.obj", "parts", "deform.obj"))
mio.WriteObj(model, sc["tgt"], sc["t_nrm"], sc["t_faces"])
mio.WriteObj(templ, sc["src"], sc["s_nrm"], sc["s_faces"])
write_parts(parts, sc["s_labels"])
cam_R = np.linalg.qr(np.random.default_rng(3).normal(size=(3, 3)))[0]
cam_R[2] = sc["view_ray"] / np.linalg.norm(sc["view_ray"])
for rep in range(3):
    a = time.perf_counter()
    st = processor.Deform(model, templ, parts, cam_R, 0.81, out)
    print(f"mvs_processor_deform, files -> file, call {rep}: {time.perf_counter() - a:.3f} s "
          f"(Model.obj {os.path.getsize(model) / 1e6:.0f} MB, {len(sc['tgt'])} vertices; n_valid {st['n_valid']})", flush=True)
shutil.rmtree(d)
