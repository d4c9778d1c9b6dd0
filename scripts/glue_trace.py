"""which autograd nodes / modules issue the framework's copy and add kernels of the headline step?  One eager step under torch.profiler;
for every aten::copy_ / aten::add / aten::add_ / aten::contiguous / aten::cat with device time, the chain of enclosing profiler ranges."""
import os
import sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from ocflow_amd.models.flow_occ_model import FlowOccModel
from ocflow_amd.runtime import DistEnv, TrainStepper, synthetic_batch

env = DistEnv(device=torch.device("cuda:0"))
torch.manual_seed(0)
st = TrainStepper(FlowOccModel({"model": "flowoccnet", "learning_rate": 1e-4}), env, amp_dtype=torch.bfloat16, channels_last=True, fused_optimizer=True,
                  master_weights=True)
batch = st.prepare(synthetic_batch(8, 436, 1024, env.device, seed=1))
for _ in range(3):
    st.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    st.step(batch)
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::clone")
agg = defaultdict(lambda: [0, 0.0, set()])
for ev in prof.events():
    if ev.name in want and ev.device_time_total > 0 and not any(c.name in want and c.device_time_total > 0 for c in ev.cpu_children):
        chain, p = [], ev.cpu_parent
        while p is not None and len(chain) < 6:
            chain.append(p.name)
            p = p.cpu_parent
        key = (ev.name, " < ".join(chain[:4]))
        agg[key][0] += 1
        agg[key][1] += ev.device_time_total
        agg[key][2].add(str(ev.input_shapes)[:80])
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"total device time of these ops: {tot:.0f} us in {sum(v[0] for _, v in rows)} calls")
for (name, chain), (n, t, shapes) in rows[:45]:
    print(f"{t:8.1f} us {n:3d} x {name:14s} | {chain[:150]} | {sorted(shapes)[:2]}")
sys.stdout.flush()
os._exit(0)
