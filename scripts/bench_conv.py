#!/usr/bin/env python3
"""GPU box: the implicit-GEMM convolution kernel (ocf_conv3x3_igemm) against the framework's convolution (MIOpen / CK with the repo's
find-db) at the decoder's layer shapes, 8 pairs at 436 x 1024, bf16 channels_last, interleaved rounds in one process.
  forward : ours = conv + bias + LeakyReLU in one launch     theirs = conv, and conv + ocf_bias_act_fwd (what the model runs today)
  dgrad   : ours = the same kernel on the transposed weight  theirs = convolution_backward(input only)
usage: python scripts/bench_conv.py [--tw 0|16|32|64] [--levels 2,3] """
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import ocflow_amd  # noqa: F401
from ocflow_amd import conv, ops
from ocflow_amd.runtime import use_repo_miopen_db

ap = argparse.ArgumentParser()
ap.add_argument("--tw", type=int, default=0, help="policy of ocf_conv3x3_igemm: 0 auto, 16/32/64 patch width, +256 direct kernel, +512 staged kernel")
ap.add_argument("--levels", default="2,3,4,5,6")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--ours-only", action="store_true", help="skip the library timings (kernel A/B runs)")
args = ap.parse_args()
use_repo_miopen_db(None)
dev = "cuda:0"
B = 8
HW = {2: (109, 256), 3: (55, 128), 4: (28, 64), 5: (14, 32), 6: (7, 16)}
FLOW_IN = {2: 128, 3: 160, 4: 192, 5: 224, 6: 288}
OCC_IN = {2: 96, 3: 160, 4: 224, 5: 288, 6: 416}
layers = []
for lvl in [int(v) for v in args.levels.split(",")]:
    H, W = HW[lvl]
    chain = [(FLOW_IN[lvl], 128, 1), (128, 128, 1), (128, 96, 1), (96, 64, 1), (64, 32, 1), (32, 2, 1), (OCC_IN[lvl], 128, 1), (32, 16, 1), (16, 1, 1)]
    if lvl == 2:
        chain += [(64, 128, 1), (128, 128, 2), (128, 128, 4), (128, 96, 8), (96, 64, 16)]
    seen = set()
    for ci, co, d in chain:
        if (ci, co, d) not in seen:
            seen.add((ci, co, d))
            layers.append((ci, co, d, H, W))


_GRAPHS = {}


def timeit(fn, n):
    """n calls replayed from a hipGraph (the regime of the training step: no host launch path in the measurement)"""
    g = _GRAPHS.get(fn)  # keyed by the function OBJECT (kept alive by the dict): an id() can be re-used by a later lambda, and
    if g is None:        # replaying a graph whose tensors have been freed is a GPU memory fault
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
        _GRAPHS[fn] = g
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print(f"{'layer':34s} | fwd ours  lib   lib+epi  TF/s ours | dgrad ours  lib   TF/s ours")
tot = [0.0] * 5
for ci, co, d, H, W in layers:
    x = torch.randn(B, ci, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device=dev) * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device=dev)
    gy = torch.randn(B, co, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    pk = conv.PACKER.get(w, False)
    out = torch.empty_like(gy)
    f_ours = lambda: conv.igemm_raw(x, pk, co, b, None, d, 0.1, 1.0, out=out, policy=args.tw)
    f_lib = lambda: torch.ops.aten.convolution(x, w, None, (1, 1), (d, d), (d, d), False, (0, 0), 1)

    def f_lib_epi():
        y = torch.ops.aten.convolution(x, w, None, (1, 1), (d, d), (d, d), False, (0, 0), 1)
        with torch.no_grad():
            ops.bias_leaky_relu(y, b, 0.1)

    can_d = co % 8 == 0
    if can_d:
        pkt = conv.PACKER.get(w, True, ci)
        gx = torch.empty_like(x)
        d_ours = lambda: conv.igemm_raw(gy, pkt, ci, None, None, d, 1.0, 1.0, out=gx, policy=args.tw)
    d_lib = lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (d, d), (d, d), False, (0, 0), 1, [True, False, False])
    if args.ours_only:
        f_lib = f_lib_epi = d_lib = (lambda: None)
    fns = [f_ours, f_lib, f_lib_epi] + ([d_ours] if can_d else []) + [d_lib]
    for fn in fns:
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    best = [1e30] * len(fns)
    for _ in range(args.rounds):
        for i, fn in enumerate(fns):
            best[i] = min(best[i], timeit(fn, args.iters))
    fl = 2.0 * B * H * W * 9 * ci * co
    if not can_d:
        best = best[:3] + [float("nan")] + best[3:]
    print(f"{ci:4d}->{co:3d} d{d:2d} {H:3d}x{W:3d}            | {best[0]:7.1f} {best[1]:7.1f} {best[2]:7.1f}   {fl / best[0] / 1e6:6.0f}    | {best[3]:7.1f} {best[4]:7.1f}   "
          f"{fl / best[3] / 1e6:6.0f}", flush=True)
    for i in range(5):
        if best[i] == best[i]:
            tot[i] += best[i]
    torch.cuda.synchronize()
    _GRAPHS.clear()  # the graphs go before the tensors they were captured on
print(f"{'sum':34s} | {tot[0]:7.1f} {tot[1]:7.1f} {tot[2]:7.1f}             | {tot[3]:7.1f} {tot[4]:7.1f}")
