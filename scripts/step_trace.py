"""Diagnostic: per-step wall time of the bench workload from a cold process (is the first run on a fresh box slower?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocflow_amd.runtime import use_repo_miopen_db
use_repo_miopen_db()
import torch
import bench
from ocflow_amd.runtime import init_distributed, synthetic_batch
args = bench.parse()
env = init_distributed()
torch.backends.cudnn.benchmark = True
st = bench.build_stepper(args, env)
batch = st.prepare(synthetic_batch(args.batch, args.height, args.width, env.device, seed=42))
ts = []
for i in range(args.steps):
    t = time.perf_counter(); st.step(batch, i); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
print("per-step ms:", " ".join(f"{t:.1f}" for t in ts))
import gc
gc.collect(); gc.freeze(); gc.disable()
ts = []
for i in range(args.steps):
    t = time.perf_counter(); st.step(batch, i); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
print("gc off per-step ms:", " ".join(f"{t:.1f}" for t in ts))
# unsynchronised throughput over the last 20
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(20): st.step(batch, i)
torch.cuda.synchronize(); print("async 20 steps: %.2f ms/step" % (1e3 * (time.perf_counter() - t) / 20))
