#!/usr/bin/env python3
"""GPU box: which lines of this repo issue the small framework kernels of a training step (fills, copies, casts, adds)?
A TorchDispatchMode counts aten ops of the watched kinds per innermost ocflow_amd / bench frame for one eager step (forward + backward)."""
import argparse, collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocflow_amd  # noqa: F401
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from ocflow_amd.runtime import DistEnv, TrainStepper, synthetic_batch, use_repo_miopen_db

WATCH = ("fill_", "zero_", "zeros", "copy_", "_to_copy", "clone", "add", "add_", "cat", "mul", "contiguous", "empty_like", "slice_backward",
         "sum", "constant_pad_nd", "select_backward")
counts = collections.Counter()


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            big = 0
            for a in args:
                if torch.is_tensor(a):
                    big = max(big, a.numel())
            where = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "/ocflow_amd/" in fr.filename or fr.filename.endswith("bench.py"):
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            counts[(name, where, "small" if big < 2_000_000 else "large")] += 1
        return func(*args, **(kwargs or {}))


use_repo_miopen_db()
env = DistEnv(device=torch.device("cuda", 0))
torch.manual_seed(42)
st = TrainStepper(bench.build_module(argparse.Namespace(model="flowoccnet")), env, amp_dtype=torch.bfloat16, channels_last=True, fused_optimizer=True,
                  master_weights=True)
batch = st.prepare(synthetic_batch(8, 436, 1024, env.device, seed=42))
for i in range(3):
    st.step(batch, i)
torch.cuda.synchronize()
with Mode():
    st.step(batch, 3)
torch.cuda.synchronize()
tot = collections.Counter()
for (name, where, size), n in counts.items():
    tot[name] += n
print("totals:", dict(tot))
for (name, where, size), n in sorted(counts.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d} {name:18s} {size:6s} {where}")
