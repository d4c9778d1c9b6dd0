#!/usr/bin/env python3
"""GPU box: every convolution of the benchmark model (shape, count per step) timed standalone through the framework (MIOpen with the repo's
find-db): forward, input gradient, weight gradient, bf16 channels_last.  Sorted by total time per step: where the 16.6 ms go."""
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ocflow_amd.runtime import init_distributed, synthetic_batch, use_repo_miopen_db

args = bench.parse()
use_repo_miopen_db(None)
env = init_distributed()
stepper = bench.build_stepper(args, env)
batch = stepper.prepare(synthetic_batch(args.batch, args.height, args.width, env.device, seed=42, index_offset=0))
seen = OrderedDict()


from torch.utils._python_dispatch import TorchDispatchMode


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        if func is torch.ops.aten.convolution.default:
            x, w, bias, s, p, d, tr = args[:7]
            key = ("ConvTranspose2d" if tr else "Conv2d", x.shape[1], w.shape[1] if tr else w.shape[0], tuple(w.shape[2:]), tuple(s), tuple(p), tuple(d),
                   x.shape[2], x.shape[3])
            seen[key] = seen.get(key, 0) + 1
        return func(*args, **(kwargs or {}))


with Spy():
    stepper.step(batch, 0)
torch.cuda.synchronize()


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


rows = []
B = args.batch
for (kind, ci, co, k, s, p, d, H, W), cnt in seen.items():
    tr = kind == "ConvTranspose2d"
    x = torch.randn((B, ci, H, W), device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn((ci, co, *k) if tr else (co, ci, *k), device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    y = torch.ops.aten.convolution(x, w, None, s, p, d, tr, (0, 0), 1)
    gy = torch.randn_like(y)
    tf = timeit(lambda: torch.ops.aten.convolution(x, w, None, s, p, d, tr, (0, 0), 1))
    td = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, s, p, d, tr, (0, 0), 1, [True, False, False]))
    tw = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, s, p, d, tr, (0, 0), 1, [False, True, False]))
    fl = 2.0 * B * y.shape[2] * y.shape[3] * ci * co * k[0] * k[1]
    rows.append((cnt * (tf + td + tw), kind, ci, co, k[0], s[0], d[0], H, W, cnt, tf, td, tw, fl))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{len(rows)} distinct convolutions, {sum(r[9] for r in rows)} calls per step, standalone sum {tot / 1e3:.2f} ms per step (fwd {sum(r[9] * r[10] for r in rows) / 1e3:.2f}, "
      f"dgrad {sum(r[9] * r[11] for r in rows) / 1e3:.2f}, wgrad {sum(r[9] * r[12] for r in rows) / 1e3:.2f})")
for r in rows[:45]:
    t, kind, ci, co, k, s, d, H, W, cnt, tf, td, tw, fl = r
    print(f"{'T' if kind != 'Conv2d' else ' '} {ci:4d}->{co:3d} k{k} s{s} d{d:2d} {H:3d}x{W:4d} x{cnt} | fwd {tf:7.1f} us ({fl / tf / 1e6:5.0f} TF/s) dgrad {td:7.1f} ({fl / td / 1e6:5.0f}) "
          f"wgrad {tw:7.1f} ({fl / tw / 1e6:5.0f}) | {t / 1e3:6.2f} ms/step", flush=True)
