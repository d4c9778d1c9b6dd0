#!/usr/bin/env python3
"""GPU box: ops.batchnorm_lrelu (csrc/bn.hip) against the framework chain nn.BatchNorm2d -> nn.LeakyReLU on the feature-pyramid shapes of
the benchmark (8 images at 436x1024, bf16 channels_last): forward and forward + backward time per site."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F
from ocflow_amd import ops

SHAPES = [(8, 16, 218, 512), (8, 32, 109, 256), (8, 64, 55, 128), (8, 96, 28, 64), (8, 128, 14, 32), (8, 196, 7, 16)]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for shape in SHAPES:
    bn = nn.BatchNorm2d(shape[1]).cuda()
    x = torch.randn(shape, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
    cot = torch.randn_like(x)

    def fwd_ours():
        with torch.no_grad():
            return ops.batchnorm_lrelu(x, bn, 0.1)

    def fwd_ref():
        with torch.no_grad():
            return F.leaky_relu(bn(x), 0.1)

    def both_ours():
        torch.autograd.grad(ops.batchnorm_lrelu(x, bn, 0.1), (x, bn.weight, bn.bias), cot)

    def both_ref():
        torch.autograd.grad(F.leaky_relu(bn(x), 0.1), (x, bn.weight, bn.bias), cot)

    # graph-replayed (what the benchmark runs): launch overhead of the eager path excluded
    res = []
    for fn in (fwd_ours, fwd_ref, both_ours, both_ref):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                fn()
        res.append(timeit(g.replay))
    mb = x.numel() * 2 / 1e6
    print(f"{shape}: {mb:6.1f} MB | fwd ours {res[0]:6.1f} us  framework {res[1]:6.1f} us | fwd+bwd ours {res[2]:6.1f} us  framework {res[3]:6.1f} us", flush=True)
