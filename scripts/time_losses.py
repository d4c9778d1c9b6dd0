#!/usr/bin/env python3
"""GPU box: HIP-event timing of the loss-side kernels (K4 photometric, K5 smoothness, K6 range map) at the benchmark's full resolution,
through the same entry points the model uses.  usage: python scripts/time_losses.py [B H W]"""
import sys
import torch

sys.path.insert(0, ".")
from ocflow_amd import ops  # noqa: E402
from ocflow_amd.runtime import KernelTimer  # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 448, 1024)
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
img1 = torch.rand((B, 3, H, W), device=dev, generator=g) * 2 - 1
img2 = torch.rand((B, 3, H, W), device=dev, generator=g) * 2 - 1
flow = (torch.randn((B, 2, H, W), device=dev, generator=g) * 3).requires_grad_()
occ = (torch.rand((B, 1, H, W), device=dev, generator=g) < 0.3).float()
timer = KernelTimer()
ops.set_kernel_timer(timer)
for it in range(12):
    if it == 2:
        torch.cuda.synchronize(); timer.records.clear()
    s1, s2 = ops.smoothness_losses(img1, flow)
    p = ops.warp_photometric_error(img1, img2, flow, occ)
    (s1 + s2 + p).backward()
    with torch.no_grad():
        ops.occlusion_from_flow(flow)
torch.cuda.synchronize()
ops.set_kernel_timer(None)
for name, v in sorted(timer.summary().items()):
    print(f"{name:14s} {v['launches']:4d} launches  avg {1e3 * v['ms_total'] / v['launches']:8.1f} us   {v['bytes_total'] / v['ms_total'] / 1e6:8.0f} GB/s algorithmic")
