#!/usr/bin/env python3
"""GPU box: one bf16-autocast training step of FlowOccModel('flowoccnet') with the repo's convolution kernels against the same step
with the framework's convolutions (OCFLOW_NATIVE_CONV switched in-process): loss, outputs and every parameter gradient (relative L2).
usage: python scripts/cmp_native_grads.py [B H W] [--master]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ocflow_amd  # noqa: F401
from ocflow_amd.models.flow_occ_model import FlowOccModel
from ocflow_amd.runtime import synthetic_batch, use_repo_miopen_db

use_repo_miopen_db(None)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B, H, W = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (2, 128, 192)
dev = torch.device("cuda:0")
res = {}
for native in ("fp32", "0", "1"):
    os.environ["OCFLOW_NATIVE_CONV"] = "0" if native == "fp32" else native
    torch.manual_seed(0)
    m = FlowOccModel({"model": "flowoccnet", "learning_rate": 1e-4}).to(dev).to(memory_format=torch.channels_last)
    if "--master" in sys.argv:
        for mod in m.modules():
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                mod.weight.data = mod.weight.data.to(torch.bfloat16)
    batch = synthetic_batch(B, H, W, dev, seed=42)
    batch = tuple(t.contiguous(memory_format=torch.channels_last) if t.dim() == 4 and t.size(1) > 2 else t for t in batch)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=native != "fp32"):
        loss = m.training_step(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
    res[native] = (float(loss), {k: p.grad.detach().float().cpu() for k, p in m.named_parameters() if p.grad is not None})
print("loss fp32", res["fp32"][0], "library", res["0"][0], "native", res["1"][0])
ref = res["fp32"][1]
for tag in ("0", "1"):
    cs = []
    for k, g0 in ref.items():
        g1 = res[tag][1].get(k)
        if g1 is not None and g0.numel() >= 512:
            cs.append((float((g1 * g0).sum() / (g1.norm() * g0.norm() + 1e-30)), k))
    cs.sort()
    print("bf16", "native" if tag == "1" else "library", "vs fp32: lowest cosines", [(round(c, 3), k.replace("model.", "")) for c, k in cs[:6]], "median", round(cs[len(cs) // 2][0], 4))
worst = []
for k, g0 in res["0"][1].items():
    g1 = res["1"][1].get(k)
    if g1 is None:
        print("MISSING gradient with native kernels:", k)
        continue
    rel = float((g1 - g0).norm() / (g0.norm() + 1e-30))
    cos = float((g1 * g0).sum() / (g1.norm() * g0.norm() + 1e-30))
    worst.append((rel, cos, float(g1.norm() / (g0.norm() + 1e-30)), k))
worst.sort(reverse=True)
for rel, cos, ratio, k in worst[:25]:
    print(f"{rel:8.4f} cos {cos:7.4f} norm ratio {ratio:7.4f}  {k}")
print("median rel", sorted(w[0] for w in worst)[len(worst) // 2])
