"""Two-rank gloo collectives of the sizes bench.py --gpus 2 --backend gloo moves per outer iteration (K floats all-reduced,
K * 392 B per rank all-gathered), on CPU tensors and — when a GPU is visible — on device tensors shared by both ranks, with
nothing else loaded: tells a slow box (two processes time-slicing one card) from a slow library.
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/gloo_probe.py [mvs]"""
import sys
import time

import torch
import torch.distributed as dist

dist.init_process_group("gloo")
r = dist.get_rank()
devs = [torch.device("cpu")] + ([torch.device("cuda", 0)] if torch.cuda.is_available() else [])
if len(sys.argv) > 1 and sys.argv[1] == "mvs":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from multiviewstitch_amd import _lib
    _lib.check(_lib.lib().mvs_set_device(0))
for dev in devs:
    for n, dt in ((8142, torch.float32), (8142 * 392, torch.uint8)):
        t = torch.zeros(n, dtype=dt, device=dev)
        out = torch.empty(n * dist.get_world_size(), dtype=dt, device=dev)
        for _ in range(3):
            dist.all_reduce(t); dist.all_gather_into_tensor(out, t)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        a = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if dev.type == "cuda":
                torch.cuda.synchronize()
        b = time.perf_counter()
        for _ in range(10):
            dist.all_gather_into_tensor(out, t)
            if dev.type == "cuda":
                torch.cuda.synchronize()
        c = time.perf_counter()
        if r == 0:
            print(f"gloo probe [{dev.type}{' + libmvs' if len(sys.argv) > 1 else ''}]: {n} x {dt}: all_reduce {1e3 * (b - a) / 10:.3f} ms, all_gather {1e3 * (c - b) / 10:.3f} ms", flush=True)
dist.destroy_process_group()
