"""Per-operator view of the eager training step (torch.profiler): which aten ops own the small-kernel time outside the convs.
  python scripts/op_profile.py [--model flowoccnet] [--steps 3]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocflow_amd  # noqa: F401  (runtime flags before HIP starts)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from ocflow_amd.runtime import DistEnv, TrainStepper, synthetic_batch, use_repo_miopen_db

ap = argparse.ArgumentParser(); ap.add_argument("--model", default="flowoccnet"); ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
use_repo_miopen_db()
torch.cuda.set_device(0)
env = DistEnv(device=torch.device("cuda", 0))
torch.manual_seed(42)
st = TrainStepper(bench.build_module(argparse.Namespace(model=a.model)), env, amp_dtype=torch.bfloat16, channels_last=True, fused_optimizer=True, master_weights=bool(int(os.environ.get("MW", "1"))))
batch = st.prepare(synthetic_batch(8, 436, 1024, env.device, seed=42))
for i in range(26):
    st.step(batch, i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], acc_events=True) as prof:
    for i in range(a.steps):
        st.step(batch, i)
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = []
for e in ka:
    cuda_self = getattr(e, "self_device_time_total", None)
    if cuda_self is None: cuda_self = e.self_cuda_time_total
    rows.append((cuda_self / a.steps, e.count / a.steps, e.key))
rows.sort(reverse=True)
print(f"{'self GPU us/step':>16} {'calls/step':>10}  name")
for t, c, k in rows[:70]:
    print(f"{t:16.1f} {c:10.1f}  {k[:150]}")
print("total self GPU us/step:", sum(r[0] for r in rows))
