"""Diagnostic: k_assoc_merge time for 1/2/4/8 gathered rank blocks (the same block repeated: timing only)."""
import sys
sys.path.insert(0, ".")
import torch
from multiviewstitch_amd import deformation, dist as mdist, srt as srt_mod, scene as S
import bench
dev = torch.device("cuda", 0)
sc = S.make_scene(3, device=dev)
d = deformation.Deformation(sc.verts, sc.normals, sc.faces)
K = d.UniformSampling(16)
tp, tn = bench.build_target(torch, srt_mod, S, sc, range(8), dev)
d.set_target_dev(tp.data_ptr(), tn.data_ptr(), tp.shape[0], 0)
sh = mdist.EngineShard(d, dev)
b = sh.buffers(K, 1)
with torch.cuda.stream(sh.stream):
    sh.dmin(b); sh.select(b)
    torch.cuda.synchronize()
    for world in (1, 2, 4, 8):
        allb = b["pack"].repeat(world).contiguous()
        torch.cuda.synchronize()
        d.assoc_merge_packed(allb.data_ptr(), world)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            d.assoc_merge_packed(allb.data_ptr(), world)
        e.record()
        torch.cuda.synchronize()
        print(f"merge of {world} rank blocks: {a.elapsed_time(e) * 100:.1f} us")
