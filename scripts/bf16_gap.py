"""bf16 throughput mode vs fp32 parity mode: flow EPE / occlusion L1 gap of the same FlowOccModel on the same synthetic pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocflow_amd.runtime import use_repo_miopen_db, synthetic_batch
use_repo_miopen_db()
from ocflow_amd.models.flow_occ_model import FlowOccModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
for model in ("flowoccnet", "pwoc"):
    H, W = (436, 1024) if model == "flowoccnet" else (448, 1024)
    m = FlowOccModel({"model": model, "learning_rate": 1e-4}).to(dev).eval()
    x, _, _ = synthetic_batch(2, H, W, dev, seed=1)
    with torch.no_grad():
        f32, o32 = m(x)
        mc = m.to(memory_format=torch.channels_last)
        xc = x.contiguous(memory_format=torch.channels_last)
        f32c, o32c = mc(xc)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            fbf, obf = mc(xc)
    epe = lambda a, b: float(torch.sqrt(((a.float() - b.float()) ** 2).sum(1)).mean())
    print(f"{model}: |flow| mean {float(f32.abs().mean()):.4f}  fp32 nchw vs fp32 channels_last EPE {epe(f32, f32c):.2e}  "
          f"fp32 vs bf16-autocast EPE {epe(f32, fbf):.3e}  occ L1 {float((o32 - obf.float()).abs().mean()):.3e}")
