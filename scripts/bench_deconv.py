#!/usr/bin/env python3
"""GPU box: the small transposed-convolution kernels (ocf_deconv3x3s2_*) against the framework's conv_transpose2d + autograd at the
estimators' up-sampling layers, 8 pairs at 436 x 1024, bf16 channels_last, hipGraph replay of 10 calls."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import ocflow_amd  # noqa: F401
from ocflow_amd import conv, _lib
from ocflow_amd.ops import _t4
import ctypes as C
from ocflow_amd.runtime import use_repo_miopen_db

use_repo_miopen_db(None)
dev = "cuda:0"
B = 8
HW = {3: (55, 128), 4: (28, 64), 5: (14, 32), 6: (7, 16)}
OUT = {3: (109, 256), 4: (55, 128), 5: (28, 64), 6: (14, 32)}


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
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    del g
    return best


print(f"{'layer':28s} | fwd ours   lib | fwd+bwd ours   lib   (us)")
tot = [0.0] * 4
for lvl in (3, 4, 5, 6):
    H, W = HW[lvl]
    Ho, Wo = OUT[lvl]
    op = (Ho - (2 * H - 1), Wo - (2 * W - 1))
    for ci, co in ((2, 2), (32, 2), (16, 1), (1, 1)):
        x = torch.randn(B, ci, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = (torch.randn(ci, co, 3, 3, device=dev) * 0.1).requires_grad_(True)
        wb = w.detach().bfloat16()
        b = torch.randn(co, device=dev).requires_grad_(True)
        gy = torch.randn(B, co, Ho, Wo, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)

        def ours_f():
            with torch.no_grad():
                conv.deconv3x3s2_bias(x, w, b, op)

        def lib_f():
            with torch.no_grad():
                F.conv_transpose2d(x, wb, None, 2, 1, op).add_(b.view(1, -1, 1, 1))

        def ours_fb():
            x.grad = w.grad = b.grad = None
            conv.deconv3x3s2_bias(x, w, b, op).backward(gy)

        def lib_fb():
            x.grad = w.grad = b.grad = None
            (F.conv_transpose2d(x, w.bfloat16(), None, 2, 1, op) + b.bfloat16().view(1, -1, 1, 1)).backward(gy)

        gx = torch.empty_like(x.detach())
        gw = torch.zeros(ci, co, 3, 3, device=dev)
        gb = torch.zeros(co, device=dev)
        L = _lib.load()
        st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ws = w.stride()

        def raw(dx, dw):
            _lib.check(L.ocf_deconv3x3s2_bwd(_t4(x.detach()), C.c_void_p(w.data_ptr()), ws[0], ws[1], ws[2], ws[3], 1, _t4(gy), _t4(gx if dx else None),
                                             C.c_void_p(gw.data_ptr()) if dw else None, 9 * co, 9, 3, 1, C.c_void_p(gb.data_ptr()) if dw else None,
                                             None, B, ci, co, H, W, Ho, Wo, _lib.OCF_BF16, st()), "bwd")

        t = [timeit(f) for f in (ours_f, lib_f, ours_fb, lib_fb, lambda: raw(True, False), lambda: raw(False, True))]
        for i in range(4):
            tot[i] += t[i]
        print(f"{ci:3d}->{co} {H:3d}x{W:3d} -> {Ho:3d}x{Wo:3d}      | {t[0]:7.1f} {t[1]:7.1f} | {t[2]:7.1f} {t[3]:7.1f} | dgrad {t[4]:6.1f} wgrad {t[5]:6.1f}", flush=True)
print(f"{'sum':28s} | {tot[0]:7.1f} {tot[1]:7.1f} | {tot[2]:7.1f} {tot[3]:7.1f}")
