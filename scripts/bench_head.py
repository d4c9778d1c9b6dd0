#!/usr/bin/env python3
"""GPU box: the fused supervised head (ocf_flow_occ_head_fwd / _bwd) at 8 x 109 x 256 -> 436 x 1024 against the framework's chain, hipGraph replay."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import ocflow_amd  # noqa: F401
from ocflow_amd import ops

dev = "cuda:0"


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    del g
    return best


B, h, w = 8, 109, 256
flow = torch.randn(B, 2, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
occ = torch.sigmoid(torch.randn(B, 1, h, w, device=dev)).bfloat16().requires_grad_(True)
fgt = torch.randn(B, 2, 4 * h, 4 * w, device=dev)
ogt = (torch.rand(B, 1, 4 * h, 4 * w, device=dev) < 0.3).float()
up = torch.nn.Upsample(scale_factor=4, mode="bilinear")


def ours_f():
    with torch.no_grad():
        ops.flow_occ_head_loss(flow, occ, fgt, ogt)


def ours_fb():
    flow.grad = occ.grad = None
    a, b = ops.flow_occ_head_loss(flow, occ, fgt, ogt)
    (a + b).backward()


def lib_fb():
    flow.grad = occ.grad = None
    (F.l1_loss(up(flow.float()), fgt) + F.binary_cross_entropy(up(occ.float()), ogt)).backward()


print(f"head: forward only {timeit(ours_f):.1f} us | forward + backward ours {timeit(ours_fb):.1f} us, framework {timeit(lib_fb):.1f} us")
