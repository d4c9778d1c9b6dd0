"""Helpers shared by the golden generator (tests/golden/make_golden.py) and the tests.

`rngfree_fill` is the RNG-free weight rule of SURVEY.md section 7 step 1: every tensor of a
state_dict is a closed-form function of (key string, element index), so whole-network
golden outputs can be reproduced from the key list alone without shipping weights.
"""
import math

import numpy as np
import torch


def _phase(name: str) -> float:
    h = 0
    for i, ch in enumerate(name):
        h = (h + ord(ch) * (i + 1)) % 1009
    return 2.0 * math.pi * h / 1009.0


def rngfree_tensor(name: str, shape) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.float64)
    s = np.sin(idx * 0.6180339887498949 + _phase(name))
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        v = 1.0 + 0.1 * s * s
    elif leaf == "running_mean":
        v = 0.05 * s
    elif len(shape) <= 1:
        v = 1.0 + 0.1 * s if leaf == "weight" else 0.05 * s
    else:
        fan_in = int(np.prod(shape[1:]))
        v = math.sqrt(3.0 / max(fan_in, 1)) * s
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


@torch.no_grad()
def rngfree_fill(module: torch.nn.Module) -> torch.nn.Module:
    sd = module.state_dict()
    for k, t in sd.items():
        t.copy_(rngfree_tensor(k, tuple(t.shape)).to(t.dtype))
    return module


def smooth_field(shape, seed=0, amp=1.0, waves=6) -> np.ndarray:
    """Deterministic smooth (image-like) field of the given [B,C,H,W] shape in about [-amp, amp]."""
    B, C, H, W = shape
    rs = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    out = np.zeros(shape, dtype=np.float64)
    for b in range(B):
        for c in range(C):
            acc = np.zeros((H, W))
            for _ in range(waves):
                fx, fy = rs.uniform(-0.35, 0.35, size=2)
                ph = rs.uniform(0, 2 * math.pi)
                acc += rs.uniform(0.3, 1.0) * np.sin(fx * xx + fy * yy + ph)
            out[b, c] = acc / waves * 2.0
    return (amp * out).astype(np.float32)
