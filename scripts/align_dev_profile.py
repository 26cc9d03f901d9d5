"""mvs_align_dev on the 2 M-vertex / 4 M-facet scan (tests/util.py body_scene(5, 30, 450)), three calls — run under
`rocprofv3 --kernel-trace --stats` to see which kernels the 8 ms are (scripts/align_dev_profile.sh)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiviewstitch_amd import alignment
from tests.util import body_scene

sc = body_scene(5, 30, 450)
A = alignment.Alignment()
dv = torch.device("cuda", 0)
Vt, Ft = len(sc["tgt"]), len(sc["t_faces"])
h = [torch.from_numpy(sc["tgt"]), torch.from_numpy(sc["t_nrm"]), torch.from_numpy(np.ascontiguousarray(sc["t_faces"], np.int32))]
dt, dtn, dtf = (x.to(dv) for x in h)
dl = torch.empty(Vt, dtype=torch.int32, device=dv)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    dt.copy_(h[0]); dtn.copy_(h[1]); dtf.copy_(h[2])
    torch.cuda.synchronize()
    a = time.perf_counter()
    r = A.AlignDev(sc["src"], sc["s_nrm"], sc["s_labels"], dt.data_ptr(), dtn.data_ptr(), Vt, dtf.data_ptr(), Ft, dl.data_ptr(), sc["view_ray"], 0.81)
    print(f"mvs_align_dev call {rep}: {1e3 * (time.perf_counter() - a):.3f} ms  (n_t {r['n_t']}, n_f {r['n_f']})", flush=True)
