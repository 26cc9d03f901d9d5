"""Secondary rows of the path (a2, a9, a10-a15, f2) at full size, to be run under
`rocprofv3 --kernel-trace --stats` (scripts/profile_round.sh): each entry is called REPS times on inputs resident in
HBM where the ABI allows it.  Prints the algorithmic bytes per call so the kernel durations of the trace turn into GB/s
(scripts/rows_summary.py)."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import alignment, processor, scene as S, srt as srt_mod
from tests.util import body_scene

REPS = 10
dev = torch.device("cuda", 0)
rows = {}

# f2: consistency filter, 8 frames of 1280x960
cams, d = S.make_sequence(8, 1280, 960, 2.0, device=dev)
din = torch.from_numpy(d).to(dev)
dout = torch.empty_like(din)
for _ in range(REPS):
    processor.CheckConsistency(cams, din.data_ptr(), S.MIN_DSP, S.MAX_DSP, 4, out_dev=dout.data_ptr())
npx = d.size
valid = int(((d >= S.MIN_DSP) & (d <= S.MAX_DSP)).sum())
rows["k_check_seq"] = {"bytes": 8 * npx + 8 * valid, "note": "4 B read + 4 B written per pixel, 2 x 4 B gathers per valid pixel", "pixels": npx}

# a2: depth -> points/normals/triangles of one 1280x960 raster; a9: similarity map of its points
raster = torch.from_numpy(np.ascontiguousarray(d[0])).to(dev)
npnt, nfac = srt_mod.depth_to_model_dev(raster.data_ptr(), cams[0], S.MIN_DSP, S.MAX_DSP, S.SMOOTH)
p = torch.empty((npnt, 3), dtype=torch.float64, device=dev)
n = torch.empty_like(p)
for _ in range(REPS):
    srt_mod.depth_to_model_dev(raster.data_ptr(), cams[0], S.MIN_DSP, S.MAX_DSP, S.SMOOTH, p.data_ptr(), n.data_ptr())
rows["k_depth_emit"] = {"bytes": 4 * d[0].size + 52 * npnt, "note": "4 B per pixel in, 48 + 4 B per valid pixel out", "points": npnt}
R = np.linalg.qr(np.random.default_rng(0).normal(size=(3, 3)))[0]
q, m = torch.empty_like(p), torch.empty_like(n)
for _ in range(REPS):
    srt_mod.apply_dev(p.data_ptr(), n.data_ptr(), npnt, 1.1, R, np.array([0.1, 0.2, 0.3]), q.data_ptr(), m.data_ptr())
torch.cuda.synchronize()
rows["k_srt_apply"] = {"bytes": 96 * npnt, "note": "48 B in + 48 B out per point", "points": npnt}

# a10-a15: alignment of a 9 K-vertex labelled template to a 2.03 M-vertex / 4.05 M-facet scan mesh (the size Processor.cpp:1119-1131
# hands it; with a 41 K-vertex scan here, round 3's k_label_nn row was the mean of three small and three large calls)
sc = body_scene(5, 30, 450)
for _ in range(3):
    alignment.Alignment().Align(sc["src"], sc["s_nrm"], sc["s_labels"], sc["tgt"], sc["t_nrm"], sc["t_faces"], sc["view_ray"], 0.81)
# a14 at BASELINE config 5's sizes: the scan of config 5 (2.06 M points) against its 216 K-vertex template with 16 sector labels
import bench as _bench
from multiviewstitch_amd import partwise as PW
sc5 = S.make_scene(5, device=dev)
tp5, _ = _bench.build_target(torch, srt_mod, S, sc5, range(8), dev)
big = tp5.cpu().numpy()
lab5 = PW.sector_labels(sc5.verts, 16)
for _ in range(3):
    alignment.part_recog(sc5.verts, lab5, big)
rows["k_label_nn"] = {"bytes": 28 * len(big), "note": f"24 B query + 4 B label per scan point; mean of three calls against the 9 K-vertex template (~1.9 M queries) and three against the {len(sc5.verts)}-vertex template ({len(big)} queries); the template grids stay in L2", "points": len(big)}
# f3: render the 9 K-vertex template and a 314 K-vertex depth mesh back into a 1280x960 raster
_, _, _, faces0 = srt_mod.depth_to_model(d[0], cams[0], S.MIN_DSP, S.MAX_DSP, S.SMOOTH)
mesh_p = p
mesh_f = torch.from_numpy(np.ascontiguousarray(faces0)).to(dev)
out = torch.empty((960, 1280), dtype=torch.float32, device=dev)
for _ in range(REPS):
    processor.RenderDepth((mesh_p.data_ptr(), npnt), (mesh_f.data_ptr(), len(faces0)), cams[1], out_dev=out.data_ptr())
rows["k_rd_raster"] = {"bytes": 12 * len(faces0) + 3 * 16 * len(faces0) + 4 * 1280 * 960, "note": "12 B indices + 3 x 16 B window vertices per triangle, one 4 B depth atomic per covered pixel", "triangles": len(faces0)}
print(json.dumps(rows))
