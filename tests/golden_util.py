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


def _name_seed(name: str) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) % (1 << 63)
    return h


@torch.no_grad()
def seeded_fill(module: torch.nn.Module, seed: int = 0) -> torch.nn.Module:
    """Pseudo-random weights that depend only on (state_dict key, shape, seed): every tensor is drawn from its own CPU generator seeded by
    a hash of its key, so the reference and this build get identical weights whatever their construction order.  Convolutions / linear
    weights: uniform(-b, b) with b = sqrt(3 / fan_in) (unit-gain, keeps activations O(1) through the net); 1-D `weight`: 1 + 0.1 u;
    biases / running_mean: 0.05 u; running_var: 1 + 0.1 u^2.  Used by the TRAIN-mode fixtures: the sinusoidal `rngfree_fill` weights give
    nearly rank-deficient feature maps, whose batch statistics are ill-conditioned (the reference's own fp32 and fp64 runs then disagree
    by tens of per cent in the gradients -- make_golden.py::gold_train_fpn)."""
    sd = module.state_dict()
    for k, t in sd.items():
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t.zero_()
            continue
        g = torch.Generator().manual_seed((_name_seed(k) + 7919 * seed) % (1 << 62))
        u = torch.rand(tuple(t.shape), generator=g, dtype=torch.float64) * 2 - 1
        if leaf == "running_var":
            v = 1.0 + 0.1 * u * u
        elif leaf == "running_mean":
            v = 0.05 * u
        elif t.dim() <= 1:
            v = 1.0 + 0.1 * u if leaf == "weight" else 0.05 * u
        else:
            fan_in = int(np.prod(t.shape[1:]))
            v = math.sqrt(3.0 / max(fan_in, 1)) * u
        t.copy_(v.to(t.dtype))
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
