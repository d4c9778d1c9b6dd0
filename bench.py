#!/usr/bin/env python3
"""bench.py -- headline benchmark of the OCFlow hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

metric (BASELINE.json): image-pairs/sec at 436x1024 bf16 + corr-kernel HBM GB/s.
A "step" = ONE full training step of `FlowOccModel({'model': 'flowoccnet'})` (BASELINE configs[1]: flow_occ_model,
436x1024 Sintel-style pairs, corr d=4, 5 pyramid levels) on one per-GPU shard of synthetic pairs: forward through the
FPN + 5 woc levels (HIP warp + fused mask/correlation kernels) + L1/BCE loss + backward + Adam, bf16 autocast,
channels_last, inputs resident in HBM before the timed region.  Per-GPU shard = `--batch` pairs (default 8 = the
per-rank share of configs[4], "batch 64 across 8 GPUs"); weak scaling: per-GPU work is fixed as N grows.

One JSON line on rank 0 with, besides the driver contract keys:
  roofline     -- the cost-volume kernel (ocf_corr_fwd): algorithmic bytes of its launches / their durations from HIP events
                  stamped by the dispatch itself on the launch stream (hipExtLaunchKernelGGL), vs the 8 TB/s HBM3E peak;
                  `traffic` = PMC bytes per launch from the committed rocprofv3 counter passes; `copy_ceiling_GBps` = a device
                  copy measured in this run; `kernels` = the same figures for every other hot-path kernel of the step (warp forward /
                  backward, correlation backward and -- from the unsupervised leg -- photometric, smoothness, range map).  With
                  graph replay (default) the launches of the timed region cannot carry events, so the SAME step is run eagerly
                  right after the timed loop for this leg.
  unsupervised -- (N = 1) the other half of the hot path: `FlowStageModel('flownet')` occlusion-aware training step (two forwards,
                  fused warp + photometric, smoothness, range-map occlusion) on the same shapes, timed the same way.
  cpu_baseline -- the oracle (CPU restatement of the reference's op structure, "port") on this box's host cores (N = 1 only,
                  bounded): the SAME training step at batch 1 fp32 (best / median), `compute_cost_volume` per pyramid level in
                  algorithmic GB/s, and one step of BASELINE configs[2] (pwc, 8 x 384 x 512) when the budget allows.
Launch modes: one GPU = the whole step in one hipGraph; several GPUs = forward+backward graph, one RCCL all-reduce of a flat
gradient buffer, optimizer graph (falls back, all ranks together, to eager DDP / the eager flat loop).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")  # before the HIP runtime starts; see ocflow_amd/__init__.py
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=40)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--preroll", type=int, default=24,
                   help="untimed lazy-initialisation steps BEFORE the W warm-up steps: the first step JIT-loads MIOpen/torch "
                        "kernels on a fresh box (~45 s) and the runtime has a one-time ~80 ms event at step 20 (measured, "
                        "scripts/step_trace.py); neither belongs to the steady state being measured")
    p.add_argument("--batch", type=int, default=8, help="image pairs per GPU per step")
    p.add_argument("--height", type=int, default=436)
    p.add_argument("--width", type=int, default=1024)
    p.add_argument("--model", default="flowoccnet", choices=["flowoccnet", "pwoc", "flownet", "pwc"],
                   help="flowoccnet/pwoc: FlowOccModel supervised step (headline); flownet/pwc: FlowStageModel unsupervised "
                        "occlusion-aware step (2 forwards + warp/photometric/smoothness/range-map loss kernels)")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--layout", default="cl", choices=["cl", "nchw"])
    p.add_argument("--cpu-seconds", type=float, default=60.0, help="budget of the cpu_baseline leg (0 disables it)")
    p.add_argument("--preload", type=int, default=1,
                   help="roofline legs: enqueue each event-timed eager step behind a spin kernel so that its kernels run back to back, the "
                        "regime of the timed graph replay (0: plain eager loop, the GPU idles between launches)")
    p.add_argument("--cfg3-steps", type=int, default=6,
                   help="timed steps of the extra BASELINE configs[2] object (`cfg3`: FlowStageModel('pwc') at 8 x 384 x 512) the default N = 1 "
                        "run appends to its line; 0 disables it")
    p.add_argument("--unsup-steps", type=int, default=10,
                   help="timed steps of the unsupervised FlowStageModel leg (N = 1 only; 0 disables it)")
    p.add_argument("--config", default=None, choices=[None, "cfg3"],
                   help="cfg3: BASELINE configs[2] -- FlowStageModel('pwc') occlusion-aware step, batch 8 at 384x512")
    p.add_argument("--fused-adam", type=int, default=1,
                   help="1 (default): run the module's Adam (models/flow_occ_model.py:77-78) through torch's fused multi-tensor "
                        "kernel -- same update rule, checked in tests/test_models_gpu.py::test_fused_adam_same_update; 0: foreach")
    p.add_argument("--master-weights", type=int, default=1,
                   help="1 (default): bf16 compute copies of the convolution weights refreshed from fp32 masters by one multi-tensor "
                        "kernel per step instead of autocast's per-layer casts (same values; not with --graph 0 under DDP); 0: plain autocast")
    p.add_argument("--graph", type=int, default=-1,
                   help="1: capture the training step in hipGraphs and replay it (one GPU: forward+loss+backward+Adam in one graph; "
                        "several: forward+loss+backward graph, one eager all-reduce of the flat gradient buffer, Adam graph); "
                        "0: eager loop (torch DDP when several ranks); -1 (default): graphs, falling back to eager if the capture "
                        "raises.  Same kernels either way; replay removes the host launch path (~3 ms of a 21 ms eager step)")
    return p.parse_args()


def build_module(args):
    if args.model in ("flowoccnet", "pwoc"):
        from ocflow_amd.models.flow_occ_model import FlowOccModel
        return FlowOccModel({"model": args.model, "learning_rate": 1e-4})
    from ocflow_amd.models.model import FlowStageModel
    # config/unsupervised_config.yml weights with network_type 'flow' 
    return FlowStageModel({"learning_rate": 1e-5, "model": args.model, "displacement": 4, "photo_weight": 4.0, "smooth1_weight": 0.5,
                           "smooth2_weight": 0.0, "with_occ": False, "occ_aware": True})


def build_stepper(args, env, dp_mode="ddp"):
    from ocflow_amd.runtime import TrainStepper
    torch.manual_seed(42)
    amp = torch.bfloat16 if args.dtype == "bf16" else None
    return TrainStepper(build_module(args), env, amp_dtype=amp, channels_last=(args.layout == "cl"),
                        fused_optimizer=bool(getattr(args, "fused_adam", 0)), dp_mode=dp_mode,
                        master_weights=bool(getattr(args, "master_weights", 0)))


def usable_cores() -> int:
    """host cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (the GPU box gives a
    one-GPU job a 16-core share of a much larger host; oversubscribing it makes the CPU leg crawl)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, int(q / int(f2.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, int(os.environ.get("OCFLOW_CPU_CORES", "32"))))


def cpu_model_name() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, budget_s):
    """Reference CPU path (oracle = port of the reference's ATen op structure) on the host cores, SURVEY 8d5: (i) the same model and
    training step at batch 1, fp32 -- first step is the probe, further steps only while they fit the budget, best and median
    reported; (ii) `compute_cost_volume` (pad-and-shift loop, correlation_layer.py:7-40) per pyramid level at batch 1 in
    algorithmic GB/s (fp32 byte model h*w*(8C + 324)); (iii) one step of BASELINE configs[2] (pwc, occlusion-aware, 8 x 384 x 512)
    if the remaining budget covers the survey's estimate for it.  Progress goes to stderr (a silent GPU box is taken for hung)."""
    import statistics
    from oracle.backend import oracle_ops
    from oracle import ocflow_oracle as O
    from ocflow_amd.runtime import DistEnv, TrainStepper, synthetic_batch
    cores = usable_cores()
    torch.set_num_threads(cores)
    env = DistEnv()
    times = []
    t_start = time.perf_counter()
    with oracle_ops():
        torch.manual_seed(42)
        stepper = TrainStepper(build_module(args), env, amp_dtype=None, channels_last=False)
        batch = synthetic_batch(1, args.height, args.width, "cpu", seed=42)
        while True:
            t0 = time.perf_counter()
            stepper.step(batch)
            times.append(time.perf_counter() - t0)
            print(f"[bench] cpu_baseline step {len(times)}: {times[-1]:.2f} s on {cores} threads", file=sys.stderr, flush=True)
            spent = time.perf_counter() - t_start
            if len(times) >= 6 or spent + times[-1] > 0.3 * budget_s:
                break
        del stepper
    timed = times[1:] if len(times) > 1 else times  # first step pays allocator / oneDNN primitive warm-up
    med, best = statistics.median(timed), min(timed)
    out = {"value": round(1.0 / med, 4), "unit": "image-pairs/s", "cores": cores, "kind": "port", "cpu_model": cpu_model_name(),
           "best": round(1.0 / best, 4), "median": round(1.0 / med, 4),
           "sample": f"{len(timed)} training step(s){' after 1 warm-up' if len(times) > 1 else ' (cold)'} of "
                     f"the same model ('{args.model}') at batch 1, {args.height}x{args.width}, fp32, oracle ops "
                     f"(pad-and-shift correlation, grid_sample warp); value = median ({med:.2f} s/step), best {best:.2f} s/step"}
    # (ii) the correlation alone, per level (FPN shapes at this resolution)
    levels = {}
    gen = torch.Generator().manual_seed(1)
    h, w = args.height, args.width
    dims = []
    for _ in range(6):
        h, w = (h + 1) // 2, (w + 1) // 2
        dims.append((h, w))
    for lvl, Cc, (hh, ww) in zip((6, 5, 4, 3, 2), (196, 128, 96, 64, 32), reversed(dims[1:])):
        f1 = torch.randn((1, Cc, hh, ww), generator=gen)
        f2 = torch.randn((1, Cc, hh, ww), generator=gen)
        O.compute_cost_volume(f1, f2, 4)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.compute_cost_volume(f1, f2, 4)
            ts.append(time.perf_counter() - t0)
        nbytes = hh * ww * (8 * Cc + 324)
        levels[f"C{Cc}x{hh}x{ww}"] = {"best_ms": round(1e3 * min(ts), 3), "GBps": round(nbytes / min(ts) / 1e9, 3)}
    out["compute_cost_volume_levels_b1_fp32"] = levels
    # (iii) configs[2]: the survey measured 11.8 s/step on 8 threads; run it only if that fits what is left
    # one warm-up + three timed steps, best and median -- while they fit the budget (the first step is the probe)
    remaining = budget_s - (time.perf_counter() - t_start)
    if remaining > 12.0 * 8 / max(cores, 1) * 1.3 + 3.0:
        try:
            from ocflow_amd.models.model import FlowStageModel
            ts3 = []
            with oracle_ops():
                torch.manual_seed(42)
                m = FlowStageModel({"learning_rate": 1e-5, "model": "pwc", "displacement": 4, "photo_weight": 4.0, "smooth1_weight": 0.5,
                                    "smooth2_weight": 0.0, "with_occ": False, "occ_aware": True})
                st3 = TrainStepper(m, env, amp_dtype=None, channels_last=False)
                b3 = synthetic_batch(8, 384, 512, "cpu", seed=42)
                while len(ts3) < 4:
                    t0 = time.perf_counter()
                    st3.step(b3)
                    ts3.append(time.perf_counter() - t0)
                    print(f"[bench] cpu_baseline configs[2] step {len(ts3)}: {ts3[-1]:.2f} s", file=sys.stderr, flush=True)
                    if budget_s - (time.perf_counter() - t_start) < 1.15 * ts3[-1]:
                        break
            timed3 = ts3[1:] if len(ts3) > 1 else ts3
            out["cfg3_pwc_b8_384x512"] = {"pairs_per_s": round(8.0 / statistics.median(timed3), 4), "best_pairs_per_s": round(8.0 / min(timed3), 4),
                                          "s_per_step": round(statistics.median(timed3), 2), "steps": len(timed3),
                                          "note": f"{len(timed3)} timed step(s)" + (" after 1 warm-up" if len(ts3) > 1 else " (cold: the budget allowed no more)")}
        except Exception as e:  # noqa: BLE001
            out["cfg3_pwc_b8_384x512"] = {"error": str(e)}
    return out


def copy_ceiling_gbps(device) -> float:
    """measured device-copy rate (read + write bytes / time) of a 1 GiB fp32 tensor: the practical HBM ceiling beside the 8 TB/s
    spec peak the roofline fraction is quoted against"""
    n = 1 << 28
    a = torch.empty(n, device=device, dtype=torch.float32).fill_(1.0)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return round(2.0 * 4 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)


MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 matrix-core peak of an MI355X (MI355X_MICROARCH.md; the 5 PF headline figure is 2:1 sparse)


def kernel_rooflines(summ: dict, peak: float) -> dict:
    """per-kernel {launches, avg_us, achieved GB/s, frac} from a KernelTimer summary; the convolution kernels (names conv3x3_*: the
    estimator / context / encoder layers on the repo's implicit-GEMM kernels) are matrix-core bound and carry achieved TFLOP/s against the
    dense bf16 MFMA peak instead (the encoder's two narrow level-1 layers, enc16_* / enc3_*, are HBM-bound and stay in GB/s)"""
    out = {}
    for name, v in summ.items():
        if v["ms_total"] <= 0:
            continue
        if name.startswith("conv3x3"):
            tf = v["flops_total"] / (v["ms_total"] * 1e-3) / 1e12
            out[name] = {"bound": "mfma", "launches": v["launches"], "avg_launch_us": round(1e3 * v["ms_total"] / v["launches"], 2),
                         "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                         "us_measured_total": round(1e3 * v["ms_total"], 1),
                         "per_shape": {k: {"launches": w["launches"], "avg_us": round(1e3 * w["ms_total"] / w["launches"], 2),
                                           "TFLOPs": round(w.get("flops", 0) / (1e-3 * w["ms_total"] / w["launches"]) / 1e12, 1)}
                                       for k, w in sorted(v["by_shape"].items(), key=lambda kv: -kv[1]["ms_total"])[:24]}}
            continue
        gbps = v["bytes_total"] / (v["ms_total"] * 1e-3) / 1e9
        out[name] = {"launches": v["launches"], "avg_launch_us": round(1e3 * v["ms_total"] / v["launches"], 2), "achieved": round(gbps, 1),
                     "frac": round(gbps / peak, 4), "algorithmic_bytes_per_launch": int(v["bytes_total"] / v["launches"]),
                     "per_shape": {k: {"launches": w["launches"], "avg_us": round(1e3 * w["ms_total"] / w["launches"], 2),
                                       "GBps": round(w["bytes"] / (1e-3 * w["ms_total"] / w["launches"]) / 1e9, 1)}
                                   for k, w in v["by_shape"].items()}}
    return out


LINE_LIMIT = 7800  # the driver keeps an ~8.7 KB tail of stdout: a longer line is cut at the FRONT and parses as nothing (round 4)


def compact_kernels(detail: dict) -> dict:
    """ONE number per kernel group for the stdout line: its roofline fraction (HBM or MFMA, see `kernel_rooflines`); the
    per-shape tables go to the side file / stderr (`write_kernel_detail`)"""
    return {name: v["frac"] for name, v in detail.items()}


def write_kernel_detail(detail: dict, extra: dict) -> str:
    """per-shape tables of every kernel group -> profiles/bench_kernels_latest.json (and stderr, for the driver's log)"""
    blob = dict(extra)
    blob["kernels"] = detail
    path = os.path.join(ROOT, "profiles", "bench_kernels_latest.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(blob, f, indent=1)
    except OSError as e:
        path = f"(not written: {e})"
    print("[bench] kernel detail: " + json.dumps(blob), file=sys.stderr, flush=True)
    return path


def fit_line(line: dict, limit: int = LINE_LIMIT) -> dict:
    """the ONE stdout line must stay parseable from the driver's tail: drop optional detail, least important first, until
    `json.dumps(line)` fits `limit` bytes.  The contract keys, `roofline` {bound, achieved, peak, unit, frac, traffic} and
    `cpu_baseline` {value, unit, cores, kind, sample} are never dropped."""
    drops = [("cpu_baseline", "compute_cost_volume_levels_b1_fp32"), ("roofline", "event_regime"), ("config", "network_execution"),
             ("roofline", "kernels"), ("roofline", "per_level"), ("cpu_baseline", "cfg3_pwc_b8_384x512"), ("roofline", "fused_op"),
             ("distributed", None), ("cfg3", None), ("unsupervised", None)]
    dropped = []
    for top, key in drops:
        if len(json.dumps(line)) <= limit:
            break
        if key is None:
            if line.get(top) is not None:
                line[top] = None
                dropped.append(top)
        elif isinstance(line.get(top), dict) and key in line[top]:
            del line[top][key]
            dropped.append(f"{top}.{key}")
    if dropped:
        line["dropped_for_length"] = dropped
    return line


_SPIN = {}


def preload_queue(ms: float):
    """Put a spin kernel of ~`ms` milliseconds on the current stream, so that the eager step enqueued behind it finds the GPU busy and its
    kernels run BACK TO BACK -- the regime of the timed hipGraph replay.  In a plain eager loop the host is the bottleneck (≈1100 launches
    per step), the GPU idles between launches and every timed kernel starts on cold instruction / data caches: the same correlation launches
    measured 10-15 % longer that way than `rocprofv3` sees them inside the replay.  The kernel timer's events are stamped by the dispatch
    itself, so the spin kernel is not part of any measured interval."""
    if ms <= 0:
        return
    if "cycles_per_ms" not in _SPIN:  # calibrate torch.cuda._sleep once (its unit is a device clock)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1_000_000)
        torch.cuda.synchronize()
        e0.record()
        torch.cuda._sleep(10_000_000)
        e1.record()
        torch.cuda.synchronize()
        _SPIN["cycles_per_ms"] = 10_000_000 / max(e0.elapsed_time(e1), 1e-3)
    torch.cuda._sleep(int(ms * _SPIN["cycles_per_ms"]))


def unsupervised_leg(args, env, steps):
    """the loss side of the hot path: FlowStageModel('flownet') occlusion-aware step (two forwards, fused warp + photometric,
    smoothness, range map) at the same shapes, hipGraph replay, then the same step eagerly with the kernel timer"""
    import copy
    from ocflow_amd import ops
    from ocflow_amd.runtime import KernelTimer, synthetic_batch
    a2 = copy.copy(args)
    a2.model = "flownet"
    stepper = build_stepper(a2, env, dp_mode="ddp")
    batch = stepper.prepare(synthetic_batch(args.batch, args.height, args.width, env.device, seed=43))
    stepper.step(batch)
    graphed = True
    try:
        stepper.enable_graph(batch)
        batch = stepper._static_batch
    except Exception as e:  # noqa: BLE001
        graphed = False
        print(f"[bench] unsupervised leg: hipGraph capture failed ({type(e).__name__}: {e}); eager", file=sys.stderr, flush=True)
    for _ in range(6):
        stepper.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = stepper.step(batch, i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    loss_val = float(loss)
    timer = KernelTimer(enabled=True)
    stepper.disable_graph()
    t_e = time.perf_counter()
    stepper.step(batch, 0)
    torch.cuda.synchronize()
    t_e = time.perf_counter() - t_e  # one eager step: how long the host needs to enqueue it
    ops.set_kernel_timer(timer)
    for i in range(4):
        preload_queue(1.3e3 * t_e if args.preload else 0.0)
        stepper.step(batch, i)
        torch.cuda.synchronize()
    ops.set_kernel_timer(None)
    res = {"metric": "image-pairs/sec (unsupervised occlusion-aware training step)", "value": round(args.batch / dt, 3), "unit": "image-pairs/s",
           "ms_per_step": round(1e3 * dt, 3), "steps": steps, "launch": "hipGraph replay" if graphed else "eager", "final_loss": round(loss_val, 5),
           "workload": f"FlowStageModel('flownet', occ_aware) full training step, {args.batch} x {args.height}x{args.width}, {args.dtype}"}
    if not (loss_val == loss_val and abs(loss_val) < 1e6):
        res["invalid"] = f"loss {loss_val}"
    summ = timer.summary()
    stepper.close()
    return res, summ


def fused_op_roofline(summ: dict, peak: float):
    """K1 + K2 as the operator north_star names -- the fused correlation + warp -- on SURVEY 8d4's byte model h*w*(4C + 170) per pair and
    level (4C + 162 without a flow): read f1 and f2 once, the flow, write the 81 taps; the warped tensor is NOT part of it.  Measured
    over every instance of the operator in the timed steps of `summ` (a KernelTimer summary): an instance served by ONE launch
    (`corr_warp_fwd`: the warp gathered inside the correlation kernel) counts that launch; an instance served by the warp kernel followed
    by the correlation kernel counts BOTH launches against the same fused bytes (the 2C*s bytes per pixel written and re-read in between
    are that form's overhead, not algorithmic bytes)."""
    one, corr, warp = summ.get("corr_warp_fwd"), summ.get("corr_fwd"), summ.get("warp_fwd")
    ms = sum(v["ms_total"] for v in (one, corr, warp) if v)
    if ms <= 0 or not corr:
        return None
    nbytes = sum(v["bytes_total"] for v in (one, corr) if v)
    flow_px = 0
    if warp:  # every warp launch precedes one correlation launch whose own byte model has no flow term: add its 8 bytes per pixel
        for shape, w in warp["by_shape"].items():
            dims = shape.replace("B", "").replace("C", "").split("x")
            flow_px += w["launches"] * int(dims[0]) * int(dims[2]) * int(dims[3])
    nbytes += 8 * flow_px
    gbps = nbytes / (ms * 1e-3) / 1e9
    n_one = one["launches"] if one else 0
    return {"bound": "hbm", "achieved": round(gbps, 1), "peak": peak, "unit": "GB/s", "frac": round(gbps / peak, 4),
            "operator": "fused correlation + warp (K1 + K2), bytes = h*w*(4C + 170) per pair and level (SURVEY 8d4)",
            "instances": {"one_launch": n_one, "warp_then_correlation": warp["launches"] if warp else 0,
                          "no_flow (top level)": corr["launches"] - (warp["launches"] if warp else 0)},
            "source": "unsupervised leg (FlowStageModel('flownet'): flow_net.py:110-112 and the no-grad second forward models/model.py:381-388)"}


def cfg3_leg(args, env, steps):
    """BASELINE configs[2] inside the default run: FlowStageModel('pwc') occlusion-aware training step, batch 8 at 384 x 512, hipGraph replay"""
    import copy
    from ocflow_amd.runtime import synthetic_batch
    a3 = copy.copy(args)
    a3.model, a3.batch, a3.height, a3.width = "pwc", 8, 384, 512
    stepper = build_stepper(a3, env, dp_mode="ddp")
    batch = stepper.prepare(synthetic_batch(8, 384, 512, env.device, seed=44))
    stepper.step(batch)
    graphed = True
    try:
        stepper.enable_graph(batch)
        batch = stepper._static_batch
    except Exception as e:  # noqa: BLE001
        graphed = False
        print(f"[bench] cfg3 leg: hipGraph capture failed ({type(e).__name__}: {e}); eager", file=sys.stderr, flush=True)
    for _ in range(4):
        stepper.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = stepper.step(batch, i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = {"metric": "image-pairs/sec (BASELINE configs[2]: unsupervised_config.yml photometric train step, batch 8 at 384x512)",
           "value": round(8 / dt, 3), "unit": "image-pairs/s", "ms_per_step": round(1e3 * dt, 3), "steps": steps,
           "launch": "hipGraph replay" if graphed else "eager", "final_loss": round(float(loss), 5),
           "workload": "FlowStageModel('pwc', occ_aware) full training step, 8 x 384x512, " + args.dtype}
    if not (float(loss) == float(loss) and abs(float(loss)) < 1e6):
        res["invalid"] = f"loss {float(loss)}"
    stepper.close()
    return res


def pmc_traffic_per_launch():
    """HBM bytes per launch of the correlation kernel from the committed rocprofv3 PMC passes of THIS command
    (scripts/profile_bench.sh -> profiles/latest_rocprof_ocf_kernels_summary.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, per
    MI355X_MICROARCH.md), averaged over the five pyramid-level launches exactly like `achieved`.  PMC counters cannot be read
    from inside the timed run, so this field is only as fresh as that file; None when it is absent."""
    path = os.path.join(ROOT, "profiles", "latest_rocprof_ocf_kernels_summary.json")
    try:
        with open(path) as f:
            kernels = json.load(f)["kernels"]
        total, launches, dur = 0, 0, 0.0
        for name, grids in kernels.items():
            if "corr_fwd" not in name:
                continue
            for g in grids.values():
                if "FETCH_SIZE_bytes_corrected" in g and "WRITE_SIZE_bytes_corrected" in g:
                    total += g["FETCH_SIZE_bytes_corrected"] + g["WRITE_SIZE_bytes_corrected"]
                    dur += g.get("avg_us", 0.0)
                    launches += 1
        if launches == 0:
            return {"traffic": None}
        return {"traffic": int(total / launches), "traffic_source": "profiles/latest_rocprof_ocf_kernels_summary.json (rocprofv3 --pmc "
                "FETCH_SIZE / WRITE_SIZE passes of this command)", "rocprof_avg_launch_us": round(dur / launches, 2)}
    except (OSError, ValueError, KeyError):
        return {"traffic": None}


def main():
    args = parse()
    if args.config == "cfg3":  # BASELINE configs[2]: unsupervised_config.yml photometric train step, batch 8 at 384x512
        args.model, args.batch, args.height, args.width = "pwc", 8, 384, 512
    # stdout carries exactly ONE line, the JSON result of rank 0: libraries that print to fd 1 (RCCL's version banner at
    # communicator creation, MIOpen notes) are sent to stderr for the whole run and the real stdout is restored for that line
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    from ocflow_amd import ops
    from ocflow_amd.runtime import (KernelTimer, all_reduce_max, barrier, init_distributed, shutdown_distributed,
                                    synthetic_batch)
    from ocflow_amd.runtime import use_repo_miopen_db
    use_repo_miopen_db(os.environ.get("OCFLOW_MIOPEN_CAPTURE"))  # before the first convolution
    env = init_distributed()
    if env.device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X: the OCFlow hot path has no CPU fallback")
    if args.gpus != env.world_size:
        if env.world_size == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run --nproc-per-node {args.gpus}")
    # MIOpen immediate mode + the find-db / kernel cache the repo ships for this workload (ocflow_amd/miopen_db, produced by a
    # find-mode run of this benchmark): the tuned solvers are picked without the 45 s search of a fresh box.  OCFLOW_MIOPEN_FIND=1
    # re-runs the search (same throughput, measured), OCFLOW_MIOPEN_CAPTURE=<dir> regenerates the db.
    torch.backends.cudnn.benchmark = os.environ.get("OCFLOW_MIOPEN_FIND", "0") == "1"

    if args.model in ("pwoc", "pwc"):
        # the PWC-style pyramids need sides divisible by 64; the reference's datasets centre-crop to that (datasets.py:146-150)
        from ocflow_amd.data.flow_io import render_size
        h, w = render_size(args.height, args.width)
        if (h, w) != (args.height, args.width):
            if env.is_main:
                print(f"[bench] model {args.model}: {args.height}x{args.width} cropped to the reference's render size {h}x{w}",
                      file=sys.stderr, flush=True)
            args.height, args.width = h, w
    use_graph = args.graph != 0
    # several ranks + graph replay: "flat" data parallelism (one all-reduce of a flat gradient buffer between two graphs);
    # eager: torch DDP (bucketed all-reduce overlapped with backward)
    stepper = build_stepper(args, env, dp_mode="flat" if use_graph else "ddp")
    B = args.batch
    batch = stepper.prepare(synthetic_batch(B, args.height, args.width, env.device, seed=42, index_offset=env.rank * B))

    t_w = time.perf_counter()
    def avg_step_seconds(n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            stepper.step(batch)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    if use_graph:
        stepper.step(batch)  # lazy initialisation (kernel JIT / db loads) must not happen inside a capture
        t_eager = avg_step_seconds(4)
        failure = None
        try:
            stepper.enable_graph(batch)
        except Exception as e:  # noqa: BLE001 - any capture problem: measure the eager loop instead
            if args.graph == 1:
                raise
            failure = f"{type(e).__name__}: {e}"
        # every rank takes the same branch: one rank replaying graphs while another waits in DDP's all-reduce would hang
        if all_reduce_max(env, 1.0 if failure else 0.0) > 0.0:
            print(f"[bench] rank {env.rank}: hipGraph capture failed ({failure or 'on another rank'}); using the eager loop",
                  file=sys.stderr, flush=True)
            use_graph = False
            stepper.close()
            del stepper
            torch.cuda.empty_cache()
            stepper = build_stepper(args, env, dp_mode="ddp")
            batch = stepper.prepare(synthetic_batch(B, args.height, args.width, env.device, seed=42, index_offset=env.rank * B))
        else:
            batch = stepper._static_batch
            # sanity gate: replay must not be slower than the eager loop it replaces (it is ~20 % faster on one GPU).  Multi-GPU
            # replay cannot be rehearsed on the one-GPU development box (two ranks sharing a GPU replay pathologically slowly),
            # so the decision is taken here, on the machine being measured, by all ranks together.
            t_graph = avg_step_seconds(8)
            if all_reduce_max(env, t_graph / t_eager) > 1.5:
                print(f"[bench] rank {env.rank}: hipGraph replay {1e3 * t_graph:.1f} ms/step vs eager {1e3 * t_eager:.1f} ms/step; "
                      "using the eager loop (same stepper, gradients through the flat buffer)", file=sys.stderr, flush=True)
                use_graph = False
                stepper.disable_graph()
    for i in range(args.preroll):
        stepper.step(batch)
        if env.is_main and i in (0, args.preroll - 1):  # the first step loads/JITs kernels on a fresh box: show progress
            torch.cuda.synchronize()
            print(f"[bench] init pre-roll step {i + 1}/{args.preroll} done at {time.perf_counter() - t_w:.1f} s", file=sys.stderr, flush=True)
    for i in range(args.warmup):
        stepper.step(batch)
    torch.cuda.synchronize()
    if env.is_main:
        print(f"[bench] {args.warmup} warm-up steps done at {time.perf_counter() - t_w:.1f} s", file=sys.stderr, flush=True)
    timer = KernelTimer(enabled=not use_graph)  # events cannot be read back from inside a replayed graph
    ops.set_kernel_timer(timer)
    barrier(env)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = stepper.step(batch, i)
    torch.cuda.synchronize()
    barrier(env)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    loss_val = float(loss)
    # a throughput number is only worth reporting if the steps it timed were real training steps: finite loss, sane gradients
    gmax = max(float(p.grad.detach().abs().max()) for p in stepper.raw.parameters() if p.grad is not None)
    if not (loss_val == loss_val and abs(loss_val) < 1e6 and gmax == gmax and gmax < 1e4):
        raise SystemExit(f"[bench] INVALID RUN: loss {loss_val} / max|grad| {gmax} after the timed steps")
    if use_graph:
        # roofline leg: the SAME step, eager, right after the timed replays (identical kernels and shapes), with the
        # correlation launches bracketed by HIP events on the launch stream
        stepper.disable_graph()
        timer.enabled = True
        for i in range(min(args.steps, 10)):
            preload_queue(1.3e3 * t_eager if args.preload else 0.0)  # (see preload_queue: kernels back to back, as in the replay)
            stepper.step(batch, i)
            torch.cuda.synchronize()
    ops.set_kernel_timer(None)
    elapsed = all_reduce_max(env, elapsed)
    weights_master = bool(stepper._masters)

    # ---- roofline of the dominant hot-path kernel, from the launches of the timed region itself
    all_k = kernel_rooflines(timer.summary(), HBM_PEAK_GBPS)
    kernel_detail = {}
    summ = timer.summary().get("corr_fwd", None)
    roofline = None
    if summ and summ["ms_total"] > 0:
        achieved = summ["bytes_total"] / (summ["ms_total"] * 1e-3) / 1e9
        per_level = {k: {"launches": v["launches"], "avg_us": round(1e3 * v["ms_total"] / v["launches"], 2),
                         "GBps": round(v["bytes"] / (1e-3 * v["ms_total"] / v["launches"]) / 1e9, 1)}
                     for k, v in summ["by_shape"].items()}
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                    "kernel": "ocf::corr_fwd_rows (levels 2-3) / corr_fwd_wave (levels 4-6) (ocf_corr_fwd)",
                    "launches": summ["launches"], "avg_launch_us": round(1e3 * summ["ms_total"] / summ["launches"], 2),
                    "algorithmic_bytes_per_launch": int(summ["bytes_total"] / summ["launches"]),
                    "tflops": round(summ["flops_total"] / (summ["ms_total"] * 1e-3) / 1e12, 2), "per_level": per_level,
                    "event_regime": ("eager steps behind a spin kernel (kernels back to back as in the timed replay; dispatch-stamped HIP events)"
                                     if (use_graph and args.preload) else ("plain eager loop" if use_graph else "the timed steps themselves"))}

    if roofline is not None:
        roofline.update(pmc_traffic_per_launch())
        kernel_detail = {k: v for k, v in all_k.items() if k != "corr_fwd"}

    # ---- several ranks: what RCCL saw, and what the gradient exchange costs by itself (the first multi-GPU run describes itself)
    dist_info = None
    if env.world_size > 1:
        import torch.distributed as dist
        nparam = sum(p.numel() for p in stepper.raw.parameters())
        buf = torch.zeros(nparam, device=env.device, dtype=torch.float32)
        for _ in range(3):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        t_ar = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        ar_ms = all_reduce_max(env, (time.perf_counter() - t_ar) / 10 * 1e3)
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "gradient_bytes": 4 * nparam,
                     "allreduce_ms": round(ar_ms, 4), "allreduce_fraction_of_step": round(ar_ms / (1e3 * elapsed / args.steps), 4),
                     "overlapped_with_backward": not use_graph}
        del buf

    unsup = cfg3 = None
    if env.is_main and env.world_size == 1 and args.unsup_steps > 0 and args.model in ("flowoccnet", "pwoc"):
        try:
            stepper.close()
            del stepper
            torch.cuda.empty_cache()
            unsup, usumm = unsupervised_leg(args, env, args.unsup_steps)
            if roofline is not None:
                for k, v in kernel_rooflines(usumm, HBM_PEAK_GBPS).items():
                    kernel_detail.setdefault(k if k not in all_k else k + "_unsupervised", v)
                roofline["fused_op"] = fused_op_roofline(usumm, HBM_PEAK_GBPS)
        except Exception as e:  # noqa: BLE001 - the headline numbers stay valid
            unsup = {"error": f"{type(e).__name__}: {e}"}
        stepper = None
    if env.is_main and env.world_size == 1 and args.cfg3_steps > 0 and args.config is None and args.model == "flowoccnet":
        try:
            torch.cuda.empty_cache()
            cfg3 = cfg3_leg(args, env, args.cfg3_steps)
        except Exception as e:  # noqa: BLE001
            cfg3 = {"error": f"{type(e).__name__}: {e}"}
    if roofline is not None and env.is_main:
        try:
            roofline["copy_ceiling_GBps"] = copy_ceiling_gbps(env.device)
        except Exception:  # noqa: BLE001
            roofline["copy_ceiling_GBps"] = None

    cpu = None
    if env.is_main and env.world_size == 1 and args.cpu_seconds > 0:
        try:
            cpu = cpu_baseline(args, args.cpu_seconds)
        except Exception as e:  # the GPU numbers stay valid without the baseline
            cpu = {"value": None, "unit": "image-pairs/s", "cores": usable_cores(), "kind": "port", "sample": f"failed: {e}"}

    if env.is_main:
        n = env.world_size
        pairs = B * n * args.steps
        line = {
            "metric": "image-pairs/sec at 436x1024 bf16 (training step)", "value": round(pairs / elapsed, 3),
            "unit": "image-pairs/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{'flow_occ_model FlowOccModel' if args.model in ('flowoccnet', 'pwoc') else 'unsupervised FlowStageModel'}"
                                   f"('{args.model}') full training step, {args.height}x{args.width} "
                                   f"Sintel-style pairs, corr d=4, 5 pyramid levels, {B} pairs/GPU (BASELINE configs[1] "
                                   f"shape; per-rank share of configs[4])",
                       "global_batch": B * n, "height": args.height, "width": args.width, "layout": args.layout,
                       "parallelism": f"dp{n}" + ("" if n == 1 else (" (flat all-reduce between graphs)" if use_graph else " (torch DDP)")),
                       "launch": "hipGraph replay of the whole step" if use_graph else "eager",
                       "final_loss": round(loss_val, 5),
                       "optimizer": "adam(fused)" if args.fused_adam else "adam(foreach)",
                       "weights": "fp32 masters + bf16 compute copies" if weights_master else "fp32 (autocast casts per use)",
                       "network_execution": ("reference layers / parameters; "
                                             + ("paired encoder (2 BatchNorm groups)" if os.environ.get("OCFLOW_UNPAIRED_FPN", "0") != "1"
                                                else "encoder per image")
                                             + ", " + ("decoder inputs padded to 32 channels" if os.environ.get("OCFLOW_UNPADDED_CAT", "0") != "1"
                                                       else "unpadded concatenations") + " (DESIGN.md 10)")},
            "roofline": roofline, "unsupervised": unsup, "cfg3": cfg3, "cpu_baseline": cpu,
        }
        if dist_info is not None:
            line["distributed"] = dist_info
        if roofline is not None:
            # ONE fraction per kernel group on the line; the per-shape tables go to the side file and stderr
            roofline["kernels"] = compact_kernels(kernel_detail)
            roofline["kernels_detail"] = os.path.relpath(
                write_kernel_detail(kernel_detail, {"value": line["value"], "ms_per_step": line["ms_per_step"], "corr_fwd": all_k.get("corr_fwd")}), ROOT)
        line = fit_line(line)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    shutdown_distributed()


if __name__ == "__main__":
    main()
