"""Is this GPU box its usual self?  Dependent-launch floor (1-element torch adds on one stream), device-to-device copy rate, clocks.
A box whose launch floor is 2-3x the usual ~4-5 us makes every launch-bound number of this repo (the whole step is a chain of ~35
dependent launches) 2-3x slower without any change of code."""
import subprocess
import time

import torch

dev = torch.device("cuda", 0)
x = torch.zeros(1, device=dev)
for _ in range(200):
    x.add_(1.0)
torch.cuda.synchronize()
t = time.perf_counter()
N = 2000
for _ in range(N):
    x.add_(1.0)
torch.cuda.synchronize()
floor = 1e6 * (time.perf_counter() - t) / N
a = torch.empty(1 << 26, dtype=torch.float64, device=dev).normal_()
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    b.copy_(a)
e1.record()
torch.cuda.synchronize()
gbps = 10 * 2 * a.numel() * 8 / (1e-3 * e0.elapsed_time(e1)) / 1e9
print(f"box probe: {floor:.2f} us per dependent 1-element launch (host-paced), copy {gbps:.0f} GB/s, device {torch.cuda.get_device_name(0)}")
try:
    print(subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout[-600:])
except Exception as e:      # noqa: BLE001
    print("rocm-smi:", e)
