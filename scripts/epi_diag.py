"""fused vs unfused conv epilogue inside the unsupervised pwc step (fp32): distance of a deep gradient to the golden value"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import ocflow_amd
from ocflow_amd import ops
from oracle import ocflow_oracle as O
from golden_util import rngfree_fill
from conftest import load_golden
from ocflow_amd.models.model import FlowStageModel
torch.backends.cudnn.allow_tf32 = False
g = load_golden("nets")
DEV = "cuda:0"
hp = dict(learning_rate=1e-5, model="pwc", displacement=4, photo_weight=4.0, smooth1_weight=0.5, smooth2_weight=0.0, with_occ=False, occ_aware=False)
fused = ops.bias_leaky_relu
res = {}
for name, fn in (("fused", fused), ("unfused", O.bias_leaky_relu), ("fused2", fused)):
    ops.bias_leaky_relu = fn
    m = rngfree_fill(FlowStageModel(hp)).to(DEV)
    batch = tuple(torch.from_numpy(g[k]).to(DEV) for k in ("stage_x", "stage_gt_flow", "stage_gt_occ"))
    loss = m.training_step(batch, 0)
    loss.backward()
    gr = m.flow_pred.predict_flow2.weight.grad[:, :16].detach().cpu().numpy()
    res[name] = gr
    print(name, "loss", float(loss), "max|grad - golden|", np.abs(gr - g["stage_plain_grad_predict_flow2"]).max(), flush=True)
print("fused vs unfused", np.abs(res["fused"] - res["unfused"]).max(), "fused vs fused2", np.abs(res["fused"] - res["fused2"]).max())
