"""Where do the ~400 small fp32 elementwise launches per step come from?  Checks parameter/gradient/Adam-state strides,
profiles optimizer.step() alone, and times the step with the stock and the fused optimizer (eager + graph).
  python scripts/opt_diag.py [--model flowoccnet]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ocflow_amd.runtime import DistEnv, TrainStepper, synthetic_batch, use_repo_miopen_db


def timed(stepper, batch, n=30):
    for i in range(5):
        stepper.step(batch, i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        stepper.step(batch, i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--model", default="flowoccnet"); a = ap.parse_args()
    use_repo_miopen_db()
    args = argparse.Namespace(model=a.model)
    env = DistEnv(device=torch.device("cuda", 0))
    torch.cuda.set_device(0)
    for fused in (False, True):
        torch.manual_seed(42)
        st = TrainStepper(bench.build_module(args), env, amp_dtype=torch.bfloat16, channels_last=True, fused_optimizer=fused)
        batch = st.prepare(synthetic_batch(8, 436, 1024, env.device, seed=42))
        for i in range(26):
            st.step(batch, i)
        torch.cuda.synchronize()
        if not fused:
            mism = 0; tot = 0
            for p in st.raw.parameters():
                if p.grad is None: continue
                tot += 1
                s = st.optimizer.state[p]
                if not (p.stride() == p.grad.stride() == s["exp_avg"].stride()):
                    mism += 1
                    if mism <= 6: print("stride mismatch", tuple(p.shape), p.stride(), p.grad.stride(), s["exp_avg"].stride(), flush=True)
            print(f"params with grads {tot}, stride mismatches {mism}", flush=True)
        print(f"fused={fused}: eager {timed(st, batch):.2f} ms/step", flush=True)
        st.enable_graph(batch)
        print(f"fused={fused}: graph {timed(st, batch):.2f} ms/step", flush=True)
        del st; torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
