"""Process-level runtime around the hot path: one process per GPU (`torch.distributed`, backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU-only hosts), data-parallel sharding of image pairs, synthetic Sintel-shaped batches
generated on the device, a bf16 / channels_last training stepper, and a HIP-event kernel timer.

Sharding: image pairs are independent; a global batch is split into contiguous per-rank shards;
the only exchange is the gradient all-reduce (mean) that DistributedDataParallel overlaps with backward.  Batch-coupled
statistics (BatchNorm in the FPN, the feature-normalisation scalars of FlowNetCV) stay per-rank, as they would under
the reference's own (commented-out) Lightning DDP, test_unsupervised.py:123,134."""
from __future__ import annotations

import os
import time
from contextlib import nullcontext
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------ MIOpen tuning db
_MIOPEN_DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
_MIOPEN_STATUS = None


def _shipped_db_digest() -> str:
    """content hash of the shipped db (file names + bytes): a box that ran an earlier commit must not keep serving its stale copy"""
    import hashlib
    h = hashlib.sha256()
    for root, _dirs, files in sorted(os.walk(_MIOPEN_DB_DIR)):
        for f in sorted(files):
            p = os.path.join(root, f)
            h.update(os.path.relpath(p, _MIOPEN_DB_DIR).encode())
            with open(p, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def _running_miopen_tag() -> Optional[str]:
    """the version tag MIOpen puts into its db file names (e.g. 3_5_0_20250912-42-1199-g2584e35062), read from the libMIOpen.so this
    process will load (torch bundles its own); None when it cannot be determined"""
    import mmap
    import re
    cands = []
    try:
        import torch as _t
        cands.append(os.path.join(os.path.dirname(_t.__file__), "lib", "libMIOpen.so"))
    except Exception:  # pragma: no cover
        pass
    cands += ["/opt/rocm/lib/libMIOpen.so"]
    for p in cands:
        try:
            with open(p, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as m:
                r = re.search(rb"HIP\.(\d+_\d+_\d+_[0-9A-Za-z\-]+)\x00", m)
                if r:
                    return r.group(1).decode()
        except OSError:
            continue
    return None


def miopen_db_status() -> Dict[str, object]:
    """{'shipped_tag', 'running_tag', 'matched', 'digest'}: does the find-db shipped in ocflow_amd/miopen_db belong to the MIOpen build this
    process runs?  MIOpen only reads db files named after its own version tag; with any other build the shipped entries are IGNORED and
    every convolution goes through the library's heuristic solver choice -- the configuration in which 32-channel aligned decoder
    inputs once ended in a faulting library kernel, so the padding is switched off then (`_epilogue.cat_aligned`)."""
    global _MIOPEN_STATUS
    if _MIOPEN_STATUS is None:
        import re
        shipped = None
        cfg = os.path.join(_MIOPEN_DB_DIR, "config")
        for f in (os.listdir(cfg) if os.path.isdir(cfg) else ()):
            r = re.match(r"gfx950[0-9a-f]*\.HIP\.(.+)\.ufdb\.txt$", f)
            if r:
                shipped = r.group(1)
        running = _running_miopen_tag()
        _MIOPEN_STATUS = {"shipped_tag": shipped, "running_tag": running, "matched": shipped is not None and shipped == running,
                          "digest": _shipped_db_digest() if os.path.isdir(_MIOPEN_DB_DIR) else None}
    return _MIOPEN_STATUS


def use_repo_miopen_db(capture_dir: Optional[str] = None) -> str:
    """Point MIOpen's USER databases (find-db / perf-db text files and the compiled-kernel cache) at a writable copy of the
    ones shipped in `ocflow_amd/miopen_db/`.  A fresh MI355X box has no gfx950 find-db and no kernel cache (ROCm 7.2 ships only
    the gfx950 heuristics models), so the first step of this workload spends 50-120 s searching and JIT-compiling conv
    kernels; with the shipped db (1.6 MB, produced by this very benchmark) the same solvers are picked immediately.
    Must run before the first convolution.  `capture_dir`: use that directory instead (to (re)generate the shipped db).

    The copy lives in a per-user directory (mode 0700) whose name carries the content hash of the shipped db: a new db is a new
    directory, a stale copy from an earlier commit is never served.  `miopen_db_status()` says whether the db's version tag is the
    running MIOpen's; when it is not, a warning is printed once and the aligned concatenations fall back to plain ones."""
    import shutil
    import tempfile
    import warnings
    if "MIOPEN_USER_DB_PATH" in os.environ and capture_dir is None:
        return os.environ["MIOPEN_USER_DB_PATH"]
    src = _MIOPEN_DB_DIR
    if capture_dir is not None:
        dst = capture_dir
        os.makedirs(dst, exist_ok=True)
    else:
        st = miopen_db_status()
        if not st["matched"]:
            warnings.warn(f"ocflow_amd: the shipped MIOpen find-db is for MIOpen {st['shipped_tag']}, this process runs {st['running_tag']}: "
                          "its entries will be ignored (first steps search / JIT-compile their convolution kernels) and the 32-channel "
                          "aligned decoder inputs are switched off", RuntimeWarning, stacklevel=2)
        base = os.path.join(tempfile.gettempdir(), f"ocflow_miopen_{os.getuid()}")
        os.makedirs(base, mode=0o700, exist_ok=True)
        if os.stat(base).st_uid != os.getuid():  # (a directory of that name planted by someone else)
            base = tempfile.mkdtemp(prefix="ocflow_miopen_")
        dst = os.path.join(base, f"{st['digest']}_{os.environ.get('LOCAL_RANK', '0')}")
        if os.path.isdir(src) and not os.path.isdir(dst):
            tmp = dst + f".tmp{os.getpid()}"
            shutil.copytree(src, tmp)
            try:
                os.replace(tmp, dst)
            except OSError:  # another process of this rank got there first
                shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(dst, exist_ok=True)
    os.environ["MIOPEN_USER_DB_PATH"] = os.path.join(dst, "config")
    os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join(dst, "cache")
    os.makedirs(os.environ["MIOPEN_USER_DB_PATH"], exist_ok=True)
    os.makedirs(os.environ["MIOPEN_CUSTOM_CACHE_DIR"], exist_ok=True)
    return dst


# ------------------------------------------------------------------------------------------ distributed
@dataclass
class DistEnv:
    rank: int = 0
    local_rank: int = 0
    world_size: int = 1
    device: torch.device = torch.device("cpu")
    backend: Optional[str] = None

    @property
    def is_main(self):
        return self.rank == 0


def init_distributed(force_cpu: bool = False) -> DistEnv:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract) and join the process group."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_gpu = torch.cuda.is_available() and not force_cpu
    # OCFLOW_DIST_REHEARSAL=1: rehearse the multi-rank launch path on a ONE-GPU box -- every rank uses cuda:0 and the process
    # group is gloo (RCCL refuses two ranks on one device).  Exercises everything of an N-rank run except RCCL itself.
    rehearsal = use_gpu and os.environ.get("OCFLOW_DIST_REHEARSAL", "0") == "1"
    device = torch.device("cuda", 0 if rehearsal else local_rank) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    backend = None
    # OCFLOW_FORCE_DIST=1: join a process group even as a single rank (exercises the real RCCL code path -- stream ordering of
    # the gradient all-reduce against hipGraph replays -- on a one-GPU box)
    if world > 1 or os.environ.get("OCFLOW_FORCE_DIST", "0") == "1":
        backend = "nccl" if use_gpu and not rehearsal else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if not dist.is_initialized():
            kw = {"device_id": device} if backend == "nccl" else {}
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return DistEnv(rank, local_rank, world, device, backend)


def shutdown_distributed():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def barrier(env: DistEnv):
    if env.world_size > 1:
        if env.backend == "nccl":
            dist.barrier(device_ids=[env.device.index])
        else:
            dist.barrier()


def all_reduce_max(env: DistEnv, value: float) -> float:
    if env.world_size == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=env.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_range(global_batch: int, env: DistEnv) -> Tuple[int, int]:
    """contiguous shard [lo, hi) of a global batch for this rank (DistributedSampler-style, no padding)"""
    per, rem = divmod(global_batch, env.world_size)
    lo = env.rank * per + min(env.rank, rem)
    return lo, lo + per + (1 if env.rank < rem else 0)


def wrap_ddp(module: torch.nn.Module, env: DistEnv, bucket_cap_mb: int = 8) -> torch.nn.Module:
    """DistributedDataParallel with bucket views: gradients are all-reduced (averaged) in ~8 MB buckets while the rest of
    backward is still running.  The whole FlowOccNet gradient is 27.6 MB fp32: one 32 MB bucket would only start its
    all-reduce when backward has finished (nothing left to overlap with); four 8 MB buckets put all but the last behind the
    remaining ~6 ms of backward, and at 8 MB an xGMI ring step (2(N-1)/N * S per GPU over 7 point-to-point links) is still far
    above the ~30 us launch latency of a collective."""
    if env.world_size == 1:
        return module
    from torch.nn.parallel import DistributedDataParallel as DDP
    # static_graph: the reference nets own parameters that never reach the loss (FlowOccNet's finest-level up-convolutions, like
    # the reference's -- Lightning 1.1 ran them with find_unused_parameters=True); the step's autograd graph is the same every
    # iteration, so DDP learns the unused set once instead of walking the graph each step.
    if env.device.type == "cuda":
        return DDP(module, device_ids=[env.device.index], output_device=env.device.index, gradient_as_bucket_view=True,
                   bucket_cap_mb=bucket_cap_mb, broadcast_buffers=False, static_graph=True)
    return DDP(module, gradient_as_bucket_view=True, bucket_cap_mb=bucket_cap_mb, broadcast_buffers=False, static_graph=True)


# ------------------------------------------------------------------------------------------ synthetic data
def _lowpass(x: torch.Tensor, passes: int = 3) -> torch.Tensor:
    for _ in range(passes):
        x = F.avg_pool2d(F.pad(x, (2, 2, 2, 2), mode="reflect"), 5, stride=1)
    return x


def synthetic_batch(batch: int, height: int, width: int, device, seed: int = 42, index_offset: int = 0):
    """Sintel-shaped synthetic sample(s) generated on `device`: frame1 = low-pass filtered uniform noise in
    [-1,1], frame2 = frame1 displaced by a smooth flow field (sigma ~ 4 px) built from the same generator, plus the
    ground-truth flow [B,2,H,W] and a Bernoulli(0.3)-like blob occlusion mask [B,1,H,W].  Sample i of the GLOBAL batch
    depends only on (seed, index_offset + i), so shards are reproducible for any world size."""
    imgs, flows, occs = [], [], []
    for i in range(batch):
        g = torch.Generator(device="cpu").manual_seed(seed * 1000003 + index_offset + i)
        base = torch.rand((1, 3, height // 4 + 8, width // 4 + 8), generator=g) * 2 - 1
        coarse = torch.randn((1, 2, height // 32 + 2, width // 32 + 2), generator=g) * 4.0
        blob = torch.rand((1, 1, height // 16 + 2, width // 16 + 2), generator=g)
        base, coarse, blob = base.to(device), coarse.to(device), blob.to(device)
        f1 = F.interpolate(_lowpass(base, 1), size=(height, width), mode="bicubic", align_corners=False).clamp(-1, 1)
        flow = F.interpolate(coarse, size=(height, width), mode="bicubic", align_corners=False)
        ys, xs = torch.meshgrid(torch.arange(height, device=device, dtype=torch.float32),
                                torch.arange(width, device=device, dtype=torch.float32), indexing="ij")
        gx = 2.0 * (xs + flow[:, 0]) / max(width - 1, 1) - 1.0
        gy = 2.0 * (ys + flow[:, 1]) / max(height - 1, 1) - 1.0
        f2 = F.grid_sample(f1, torch.stack((gx, gy), dim=-1), mode="bilinear", padding_mode="border", align_corners=True)
        occ = (F.interpolate(blob, size=(height, width), mode="bilinear", align_corners=False) > 0.62).float()
        imgs.append(torch.cat((f1, f2), dim=1))
        flows.append(flow)
        occs.append(occ)
    return torch.cat(imgs), torch.cat(flows), torch.cat(occs)


# ------------------------------------------------------------------------------------------ training stepper
class TrainStepper:
    """One optimisation step of a Lightning-style module: forward + loss (module.training_step) + backward + Adam,
    under bf16 autocast when `amp_dtype` is given (convs on MFMA via MIOpen, HIP ops bf16-in / fp32-accumulate)."""

    def __init__(self, module: torch.nn.Module, env: DistEnv, amp_dtype: Optional[torch.dtype] = torch.bfloat16,
                 channels_last: bool = True, fused_optimizer: bool = False, dp_mode: str = "ddp", master_weights: bool = False):
        """dp_mode (world_size > 1 only): "ddp" = torch DistributedDataParallel (bucketed all-reduce overlapped with the eager
        backward); "flat" = no DDP wrapper, gradients are averaged with ONE all-reduce of a flat fp32 buffer between backward
        and the optimizer -- the mode `enable_graph` needs, because then nothing collective sits inside a captured graph.

        master_weights (bf16 autocast only, not with "ddp" across ranks): the convolution weights the model computes with are
        kept as bf16 tensors, refreshed from fp32 master copies (which the optimizer updates) by one multi-tensor kernel per
        step.  Values are exactly autocast's -- it rounds the fp32 weight to bf16 at every use and widens the bf16 weight
        gradient back to fp32 -- but as 2 multi-tensor launches instead of ~245 per-layer cast kernels (1.0 ms of a 17.6 ms
        step).  `module_state_dict()` / `restore_fp32_weights()` give the fp32 view back."""
        if dp_mode not in ("ddp", "flat"):
            raise ValueError(f"dp_mode must be 'ddp' or 'flat', got {dp_mode!r}")
        self.env = env
        self.dp_mode = dp_mode if (env.world_size > 1 or env.backend is not None) else "ddp"
        self.amp_dtype = amp_dtype if env.device.type == "cuda" else None
        self.channels_last = channels_last
        module = module.to(env.device)
        if channels_last:
            module = module.to(memory_format=torch.channels_last)
        self.raw = module
        self.optimizer = module.configure_optimizers()
        if fused_optimizer and env.device.type == "cuda":
            # same Adam update (models/flow_model.py:105-106 returns a plain Adam), one multi-tensor kernel instead of the
            # per-tensor fallback torch's foreach path takes when parameter / gradient strides differ
            for g in self.optimizer.param_groups:
                g["fused"], g["foreach"] = True, False
        if self.dp_mode == "flat":
            self.module = _StepAdapter(module)
            with torch.no_grad():  # what DDP's constructor does: every replica starts from rank 0's parameters and buffers
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, src=0)
        else:
            self.module = wrap_ddp(_StepAdapter(module), env)
        self._masters = {}  # id(model parameter) -> fp32 master Parameter
        if master_weights and self.amp_dtype == torch.bfloat16 and not (self.dp_mode == "ddp" and env.world_size > 1):
            self._setup_master_weights()
        self._manual = self.dp_mode == "flat" or bool(self._masters)  # gradients go through the flat buffer
        # flat mode all-reduces whenever a process group exists (a single rank under OCFLOW_FORCE_DIST included: that is how
        # the RCCL path -- stream ordering of the all-reduce between the two graph replays -- is exercised on a one-GPU box)
        self._reduce = self.dp_mode == "flat" and dist.is_available() and dist.is_initialized()
        self._graph = None
        self._graph_opt = None
        self._graph_pool = None
        self._flat = None
        self._flat_adam = None

    # ---- fp32 masters / bf16 compute weights
    def _setup_master_weights(self):
        weights = [m.weight for m in self.raw.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d))]
        for p in weights:
            if p.dtype != torch.float32 or id(p) in self._masters:
                continue
            self._masters[id(p)] = torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format))
            p.data = p.data.to(torch.bfloat16)  # same strides (channels_last weights stay channels_last)
        self._mw_model = [p for p in self.raw.parameters() if id(p) in self._masters]
        self._mw_master = [self._masters[id(p)] for p in self._mw_model]
        for g in self.optimizer.param_groups:  # the optimizer updates the masters
            g["params"] = [self._masters.get(id(p), p) for p in g["params"]]

    def module_state_dict(self):
        """state_dict of the wrapped module with the fp32 master values in place of the bf16 compute weights"""
        sd = self.raw.state_dict()
        if self._masters:
            by_id = {id(p): k for k, p in self.raw.named_parameters()}
            for pid, m in self._masters.items():
                sd[by_id[pid]] = m.detach().clone()
        return sd

    def restore_fp32_weights(self):
        """put the fp32 masters back into the module (end of training with master_weights)"""
        for p, m in zip(getattr(self, "_mw_model", []), getattr(self, "_mw_master", [])):
            p.data = m.detach().clone(memory_format=torch.preserve_format)

    def prepare(self, batch):
        out = []
        for t in batch:
            t = t.to(self.env.device, non_blocking=True)
            if self.channels_last and t.dim() == 4 and t.size(1) > 2:
                t = t.contiguous(memory_format=torch.channels_last)
            out.append(t)
        return tuple(out)

    def step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        if self._graph is not None:
            return self._replay(batch)
        return self._eager_step(batch, batch_idx)

    # ---- the step in two halves: forward + loss + backward, then the optimizer; the "manual" path (flat data parallelism
    # and/or master weights) moves the gradients through one flat fp32 buffer between the two
    def _forward_backward(self, batch, batch_idx: int = 0) -> torch.Tensor:
        # gradients are dropped, not zero-filled: backward then writes them instead of accumulating (one fill and one add
        # per parameter saved); inside a graph capture they are re-allocated from the graph's pool, so replays reuse them
        self.raw.zero_grad(set_to_none=True)
        ctx = torch.autocast(self.env.device.type, dtype=self.amp_dtype) if self.amp_dtype is not None else nullcontext()
        with ctx:
            loss = self.module(batch, batch_idx)
        if self.dp_mode == "flat" and self.env.world_size > 1:
            (loss / self.env.world_size).backward()  # sum over ranks of grad(loss / N) = DDP's averaged gradient
        else:
            loss.backward()
        from ocflow_amd import conv as nconv
        nconv.join_wgrad_stream()  # weight gradients launched beside the input gradients (ocflow_amd.conv.wgrad_bias_async)
        return loss.detach()

    def _grad_params(self):
        """model parameters that received a gradient, in a fixed order"""
        return [p for p in self.raw.parameters() if p.grad is not None]

    def _flat_views(self, params):
        """one fp32 buffer + per-parameter views with the parameters' own (dense, possibly channels_last) strides, so the
        multi-tensor copy and the fused optimizer take their fast paths"""
        pad = lambda n: (n + 63) & ~63  # 256-byte aligned views: the multi-tensor kernels only vectorise aligned operands
        flat = torch.zeros(sum(pad(p.numel()) for p in params), device=self.env.device, dtype=torch.float32)
        views, off = [], 0
        for p in params:
            views.append(flat[off:off + p.numel()].as_strided(p.shape, p.stride()))
            off += pad(p.numel())
        return flat, views

    @staticmethod
    def _gather(views, grads):
        """views[i] <- grads[i] (fp32 <- fp32 or bf16): one multi-tensor copy per source dtype"""
        for dt in (torch.float32, torch.bfloat16):
            idx = [i for i, g in enumerate(grads) if g.dtype == dt]
            if idx:
                torch._foreach_copy_([views[i] for i in idx], [grads[i] for i in idx])

    def _attach(self, params, views):
        for p, v in zip(params, views):
            self._masters.get(id(p), p).grad = v  # the optimizer's tensor for this parameter

    def _setup_flat_adam(self, params):
        """Re-home what Adam touches -- the fp32 parameters it updates (masters where there are any), both moments and the bf16 compute
        copies -- into FOUR flat buffers laid out like the flat gradient buffer, every tensor becoming a view with its own shape and
        strides, so that ocf_adam_flat updates the whole network in one launch (csrc/adam.hip).  The optimizer object keeps working on
        the same storage (eager steps, state_dict, checkpoints).  Returns None when the optimizer is not a plain single-group Adam."""
        opt = self.optimizer
        if os.environ.get("OCFLOW_FLAT_ADAM", "1") == "0" or type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
            return None
        g = opt.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or torch.is_tensor(g["lr"]):
            return None
        targets = [self._masters.get(id(p), p) for p in params]
        in_group = {id(t) for t in g["params"]}
        if any(t.dtype != torch.float32 or id(t) not in in_group or t not in opt.state or "exp_avg" not in opt.state[t] for t in targets):
            return None
        dev = self.env.device
        # (every check BEFORE the first storage is re-homed: a None from here must leave the optimizer and the packer's tables untouched)
        if any(not torch.is_tensor(opt.state[t].get("step")) or opt.state[t]["step"].device != dev or opt.state[t]["step"].dtype != torch.float32
               for t in targets):
            return None
        pad = lambda n: (n + 63) & ~63  # (the layout of _flat_views)
        total = sum(pad(t.numel()) for t in targets)
        fp, fm, fv = (torch.zeros(total, device=dev, dtype=torch.float32) for _ in range(3))
        fb = torch.zeros(total, device=dev, dtype=torch.bfloat16) if self._masters else None
        steps, off = [], 0
        with torch.no_grad():
            for p, t in zip(params, targets):
                n = t.numel()
                view = lambda flat: flat[off:off + n].as_strided(t.shape, t.stride())
                st = opt.state[t]
                for key, flat in (("exp_avg", fm), ("exp_avg_sq", fv)):
                    v = view(flat)
                    v.copy_(st[key])
                    st[key] = v
                v = view(fp)
                v.copy_(t.data)
                t.data = v
                if t is not p:  # bf16 compute copy of a master
                    v = view(fb)
                    v.copy_(p.data)
                    p.data = v
                steps.append(st["step"])
                off += pad(n)
        return {"p": fp, "m": fm, "v": fv, "b": fb, "steps": steps, "n": total, "lr": float(g["lr"]), "betas": tuple(map(float, g["betas"])),
                "eps": float(g["eps"]), "wd": float(g["weight_decay"])}

    def _optimizer_step(self):
        fa = getattr(self, "_flat_adam", None)
        if fa is not None and self._flat is not None and torch.cuda.is_current_stream_capturing():
            # (captured: the gradients are in self._flat at every replay; eager steps gather into a fresh buffer and take the optimizer's
            # own step below, on the same storage)
            from . import _lib
            import ctypes as C
            torch._foreach_add_(fa["steps"], 1)
            ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
            with torch.cuda.device(self.env.device):
                _lib.check(_lib.load().ocf_adam_flat(ptr(fa["p"]), ptr(self._flat), ptr(fa["m"]), ptr(fa["v"]), ptr(fa["b"]), fa["n"], fa["lr"],
                                                     fa["betas"][0], fa["betas"][1], fa["eps"], fa["wd"], ptr(fa["steps"][0]),
                                                     C.c_void_p(torch.cuda.current_stream(self.env.device).cuda_stream)), "ocf_adam_flat")
            return
        self.optimizer.step()
        if self._masters:
            with torch.no_grad():
                torch._foreach_copy_(self._mw_model, self._mw_master)  # bf16 compute weights <- updated fp32 masters

    def _eager_step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        loss = self._forward_backward(batch, batch_idx)
        if self._manual:
            params = self._grad_params()
            flat, views = self._flat_views(params)
            self._gather(views, [p.grad for p in params])
            if self._reduce:
                dist.all_reduce(flat)
            self._attach(params, views)
        self._optimizer_step()
        self.raw.global_step = getattr(self.raw, "global_step", 0) + 1
        return loss

    # ---- whole-step hipGraph: the eager step issues ~1700 kernel launches and is host-bound; capturing it once and replaying
    # removes the launch path.  One GPU: a single graph (forward + loss + backward + Adam).  Several GPUs ("flat" mode): graph A
    # = forward + loss + backward + gather of the gradients into one flat fp32 buffer, then ONE eager all-reduce of that buffer
    # (27.6 MB for FlowOccNet) on the process group, then graph B = Adam reading the reduced gradients through views of the flat
    # buffer.  No collective is ever captured, so the same code runs over RCCL and (in tests) gloo; the price is that the
    # all-reduce is not overlapped with backward (~0.3 ms of a 17.7 ms step at 8 GPUs over xGMI).
    def enable_graph(self, batch, warmup: int = 3):
        if self.env.world_size != 1 and self.dp_mode != "flat":  # (a single rank in "ddp" mode has no DDP wrapper at all)
            raise RuntimeError("TrainStepper.enable_graph: with several ranks construct the stepper with dp_mode='flat' "
                               "(a DDP-wrapped step has its all-reduces inside backward and is not captured)")
        if self.env.device.type != "cuda":
            raise RuntimeError("TrainStepper.enable_graph needs a HIP device")
        if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1") != "0" and os.environ.get("OCFLOW_UNSAFE_GRAPH", "0") != "1":
            # OCFLOW_UNSAFE_GRAPH=1 exists only to reproduce the corruption (scripts/long_run.py)
            raise RuntimeError("TrainStepper.enable_graph: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 must be set before the HIP runtime "
                               "initialises (import ocflow_amd first); the packet-capture replay path of ROCm 7 corrupts this graph")
        import ocflow_amd
        if ocflow_amd.GRAPH_ENV_LATE and os.environ.get("OCFLOW_UNSAFE_GRAPH", "0") != "1":
            raise RuntimeError("TrainStepper.enable_graph: ocflow_amd was imported after the HIP runtime had initialised, so "
                               "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is not in effect in this process; import ocflow_amd (or export the "
                               "variable) before the first torch.cuda call")
        # Adam must keep its step counter on the device to be capturable
        for g in self.optimizer.param_groups:
            g["capturable"] = True
        for st in self.optimizer.state.values():  # steps taken before the capture left host-side counters behind
            if "step" in st and torch.is_tensor(st["step"]) and st["step"].device != self.env.device:
                st["step"] = st["step"].to(self.env.device)
        self._static_batch = tuple(t.clone() for t in batch)
        self._graph_pool = torch.cuda.graph_pool_handle()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(warmup):  # lazy initialisation (kernel attribute queries, allocator growth) outside the capture
                self._eager_step(self._static_batch, i)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # With a process group alive, ProcessGroupNCCL's watchdog thread polls the events of recently issued collectives; under
        # the default "global" capture mode such a query from ANOTHER thread while this thread captures is an error that
        # terminates the process ("operation not permitted when stream is capturing", seen with the single-rank RCCL rehearsal).
        # "thread_local" restricts the check to the capturing thread (kernels other threads -- autograd's -- launch on the
        # capturing stream are captured in every mode); the pause lets the watchdog retire the warm-up collectives first.
        mode = "global"
        if dist.is_available() and dist.is_initialized():
            mode = "thread_local"
            time.sleep(1.0)
        graph = torch.cuda.CUDAGraph()
        from ocflow_amd import conv as nconv
        if not self._manual:
            try:
                with nconv.PACKER.capture_owner(self), torch.cuda.graph(graph, pool=self._graph_pool, capture_error_mode=mode):
                    self._static_loss = self._eager_step(self._static_batch, 0)
            except BaseException:
                nconv.PACKER.release_owner(self)
                nconv.weights_changed()
                raise
            self._graph = graph
            return self
        split = self._reduce  # an all-reduce sits between backward and the optimizer
        nconv.clear_grad_sinks()
        nconv.record_native(True)   # one more eager step: which parameters get their gradients from the repo's convolution kernels?
        try:
            with torch.cuda.stream(side):
                self._eager_step(self._static_batch, warmup)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
        finally:
            nconv.record_native(False)
        params = self._grad_params()  # after the warm-up steps: exactly the parameters that reach the loss
        self._flat, views = self._flat_views(params)
        # those kernels accumulate straight into the flat fp32 buffer (zeroed at the top of every replay): autograd produces no gradient
        # tensor for these parameters, nothing is cast to bf16 and back, nothing is copied per parameter
        self._flat_adam = self._setup_flat_adam(params)
        if self._flat_adam is not None:
            nconv.PACKER.refresh()  # the bf16 weights moved: rebuild the packer's descriptor table outside the capture
            torch.cuda.synchronize()
        opt_graph = None
        try:
            # the sinks (process-global in ocflow_amd.conv, owned by this stepper) exist inside this block only: a capture that raises
            # must not leave them armed for the next model of the process
            with nconv.grad_sinks(params, views, owner=self, enabled=os.environ.get("OCFLOW_GRAD_SINKS", "1") != "0") as sinks, \
                    nconv.PACKER.capture_owner(self):
                self._sink_ids = sinks
                with torch.cuda.graph(graph, pool=self._graph_pool, capture_error_mode=mode):
                    if sinks:
                        self._flat.zero_()
                    self._static_loss = self._forward_backward(self._static_batch, 0)
                    got = [p for p in self.raw.parameters() if p.grad is not None or id(p) in sinks]
                    if len(got) != len(params) or any(a is not b for a, b in zip(got, params)):
                        raise RuntimeError("TrainStepper.enable_graph: the set of parameters with gradients changed between steps")
                    rest = [i for i, p in enumerate(params) if id(p) not in sinks]
                    self._gather([views[i] for i in rest], [params[i].grad for i in rest])
                    for i, p in enumerate(params):  # a sink parameter that ALSO received a framework gradient (another call site): add it
                        if id(p) in sinks and p.grad is not None:
                            views[i].add_(p.grad)
                    if not split:
                        self._attach(params, views)
                        self._optimizer_step()
            # (the captured launches keep their pointers; eager steps outside the graph use autograd's own gradients)
            if split:
                # (no collective in here: a rank whose capture raised must not leave the others waiting inside one)
                self._attach(params, views)
                opt_graph = torch.cuda.CUDAGraph()
                with nconv.PACKER.capture_owner(self), torch.cuda.graph(opt_graph, pool=self._graph_pool, capture_error_mode=mode):
                    self._optimizer_step()
        except BaseException:
            # nothing half-captured survives: no graph, no flat buffer a stale sink could have pointed at
            self._graph = self._graph_opt = None
            self._flat = None
            self._sink_ids = set()
            nconv.PACKER.release_owner(self)
            nconv.weights_changed()
            raise
        self._graph = graph
        self._graph_opt = opt_graph
        return self

    def disable_graph(self):
        """back to the eager loop on the same parameters / optimizer state (the flat buffers the optimizer was re-homed into stay)"""
        from ocflow_amd import conv as nconv
        self._graph = self._graph_opt = None
        nconv.clear_grad_sinks(owner=self)
        nconv.PACKER.release_owner(self)
        nconv.weights_changed()
        return self

    def close(self):
        """release what this stepper owns outside itself: its graphs and their pool, the flat buffers, its gradient sinks (if a capture
        of its own is what installed them) and the packer's descriptor tables only its graphs could replay.  Idempotent; also run by
        the destructor."""
        try:
            from ocflow_amd import conv as nconv
            nconv.clear_grad_sinks(owner=self)
            self._graph = self._graph_opt = None
            self._static_batch = self._static_loss = None
            self._flat = None
            self._graph_pool = None
            nconv.PACKER.release_owner(self)
        except Exception:  # noqa: BLE001 - interpreter shutdown: modules may already be gone
            pass

    def __del__(self):
        self.close()

    @staticmethod
    def _weights_changed():
        from ocflow_amd import conv as nconv
        nconv.weights_changed()

    def _replay(self, batch) -> torch.Tensor:
        if batch is not self._static_batch:
            for dst, src in zip(self._static_batch, batch):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
        self._graph.replay()
        if self._graph_opt is not None:
            dist.all_reduce(self._flat)
            self._graph_opt.replay()
        # the captured optimizer updated the weights without moving a version counter or running an optimizer hook: the packed copies
        # the NEXT eager forward (validation, an eager step) would read are one update behind (the replay itself re-packs at its top)
        self._weights_changed()
        self.raw.global_step = getattr(self.raw, "global_step", 0) + 1
        return self._static_loss


class _StepAdapter(torch.nn.Module):
    """DDP hooks fire on `forward`; route it to the module's `training_step` so the reference's step code is what runs."""

    def __init__(self, module):
        super().__init__()
        self.inner = module

    def forward(self, batch, batch_idx):
        return self.inner.training_step(batch, batch_idx)


# ------------------------------------------------------------------------------------------ kernel timer
@dataclass
class KernelTimer:
    """HIP-event timing of individual kernel launches on the stream they are launched on (the ops launch on torch's
    current stream, so torch.cuda.Event brackets exactly the kernel).  Installed with ops.set_kernel_timer()."""
    enabled: bool = True
    records: List[tuple] = field(default_factory=list)

    def launch(self, name: str, meta: Dict, fn):
        if not self.enabled:
            return fn()
        from . import _lib
        import ctypes
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # torch creates the hipEvent lazily at the first record(): materialise both, then let the C launcher re-record them
        # immediately around the kernel launch (ocf_profile_next_launch) -- host-language event records sit tens of
        # microseconds away from the launch and would count GPU-waits-for-host gaps as kernel time
        e0.record(); e1.record()
        _lib.load().ocf_profile_next_launch(ctypes.c_void_p(e0.cuda_event), ctypes.c_void_p(e1.cuda_event))
        r = fn()
        self.records.append((name, meta, e0, e1))
        return r

    def summary(self) -> Dict[str, Dict]:
        """name -> {launches, ms_total, bytes_total, flops_total, by_shape}; call after torch.cuda.synchronize()"""
        out: Dict[str, Dict] = {}
        for name, meta, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            s = out.setdefault(name, {"launches": 0, "ms_total": 0.0, "bytes_total": 0, "flops_total": 0, "by_shape": {}})
            s["launches"] += 1
            s["ms_total"] += ms
            s["bytes_total"] += meta.get("bytes", 0)
            s["flops_total"] += meta.get("flops", 0)
            key = meta.get("shape", "")
            b = s["by_shape"].setdefault(key, {"launches": 0, "ms_total": 0.0, "bytes": meta.get("bytes", 0), "flops": meta.get("flops", 0)})
            b["launches"] += 1
            b["ms_total"] += ms
        return out
