import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocflow_amd, torch
from ocflow_amd.models.flow_occ_model import FlowOccModel
from ocflow_amd.runtime import DistEnv, TrainStepper, synthetic_batch
env = DistEnv(device=torch.device("cuda:0"))
batch = synthetic_batch(2, 128, 192, env.device, seed=3)
outs = {}
for mw in (False, True, False):
    torch.manual_seed(7)
    st = TrainStepper(FlowOccModel({"learning_rate": 1e-4, "model": "flowoccnet"}), env, amp_dtype=torch.bfloat16, channels_last=True,
                      fused_optimizer=True, master_weights=mw)
    b = st.prepare(batch)
    acts = {}
    def hook(name):
        def f(m, i, o):
            if torch.is_tensor(o): acts[name] = o.detach().float().clone()
        return f
    for n, m in st.raw.named_modules():
        if len(list(m.children())) == 0: m.register_forward_hook(hook(n))
    st.raw.train()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        loss = st.raw.training_step(b, 0)
    print("mw", mw, "loss", float(loss), "n conv masters", len(st._masters))
    outs[len(outs)] = acts
a, b_, c = outs[0], outs[1], outs[2]
first = None
for k in a:
    if k in b_ and a[k].shape == b_[k].shape:
        d = float((a[k] - b_[k]).abs().max()); d2 = float((a[k] - c[k]).abs().max())
        if d > 0 or d2 > 0:
            print("first differing leaf output:", k, "autocast vs master", d, " autocast vs autocast", d2, "absmax", float(a[k].abs().max()))
            break
