"""Diagnostic: where does a training step spend its time (first-call JIT vs steady state)?  Writes progress lines."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocflow_amd.models.flow_occ_model import FlowOccModel
from ocflow_amd.runtime import DistEnv, TrainStepper, synthetic_batch

def log(*a):
    print(f"[{time.perf_counter()-T0:8.2f}s]", *a, flush=True)

T0 = time.perf_counter()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H = int(sys.argv[2]) if len(sys.argv) > 2 else 436
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
layout = sys.argv[5] if len(sys.argv) > 5 else "cl"
dev = torch.device("cuda:0")
env = DistEnv(device=dev)
torch.manual_seed(0)
m = FlowOccModel({"model": "flowoccnet", "learning_rate": 1e-4})
st = TrainStepper(m, env, amp_dtype=torch.bfloat16 if dtype == "bf16" else None, channels_last=(layout == "cl"))
log("model built", B, H, W, dtype, layout)
batch = st.prepare(synthetic_batch(B, H, W, dev))
torch.cuda.synchronize(); log("batch ready")
with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(dtype == "bf16")):
    with torch.no_grad():
        out = st.raw(batch[0])
torch.cuda.synchronize(); log("first no-grad forward done")
for i in range(4):
    t = time.perf_counter()
    loss = st.step(batch, i)
    torch.cuda.synchronize()
    log(f"step {i}: {1e3*(time.perf_counter()-t):.1f} ms loss {float(loss):.4f}")
# steady state breakdown with the torch profiler
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(3):
        st.step(batch, i)
    torch.cuda.synchronize()
log("profiled 3 steps")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=35, max_name_column_width=60), flush=True)
