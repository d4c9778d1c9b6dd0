"""Functional operators of the OCFlow hot path on MI355X.

Each operator keeps the name, argument meaning and error behaviour of the reference function it replaces
(file:line under the OCFlow tree) and runs the hand-written HIP kernels of libocflow_hip.so through the C ABI
(include/ocflow_hip.h).  Tensors must live on a HIP device: there is deliberately NO CPU / PyTorch fallback --
a CPU tensor raises, a missing library raises.

Layout: inputs may be NCHW-contiguous, channels_last, or arbitrary strided views (x[:, :3], channel slices of a
concat buffer); strides are passed to the kernels, nothing is copied.  Features may be fp32 or bf16 (fp32
accumulation); flow / occlusion / images of the losses are fp32.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib

__all__ = [
    "compute_cost_volume", "corr_warp", "warp", "normalize_features", "robust_l1", "photometric_error",
    "warp_photometric_error", "gradient", "first_order_smoothness_loss", "second_order_smoothness_loss",
    "smoothness_losses", "compute_range_map", "occlusion_from_flow", "ssim", "psnr", "set_deterministic", "masked_l1", "conv3x3", "conv3x3_wgrad",
]

MAX_DISPLACEMENT = 4
# OCFLOW_FUSED_WARP: "1" = always ONE launch (the bilinear warp gathered inside the correlation kernel, the warped tensor never exists in
# HBM: 4C + 170 bytes per pixel instead of 8C + 170); "gather" = one launch whenever no gradient is recorded and a kernel with the
# in-kernel gather serves the request (ocf_corr_fwd_variant in _lib.CORR_VARIANTS_WITH_GATHER: the setting for memory-bound inference at
# sizes where the warped pyramid does not fit); "0" / "auto" (default) = warp kernel + correlation kernel.  "auto"
# is two launches because that is what measured faster at every level of the benchmark pyramid (profiles/r3_f_kbench.txt, 8 pairs,
# levels 2..6: fused 31.4 / 18.2 / 17.2 / 9.0 / 21.1 us against 29.1 / 16.2 / 11.0 / 8.2 / 9.9 us): a staged second-operand unit costs
# four 16-byte gathers instead of one load, and the texture path, not HBM, is what the staging loops are bound by.
_FUSED_WARP = os.environ.get("OCFLOW_FUSED_WARP", "auto")

_KERNEL_TIMER = None
_DETERMINISTIC = os.environ.get("OCFLOW_DETERMINISTIC", "0") == "1"


class KernelPolicy:
    """Which kernel serves a request that several can serve (same results): the value of each field is handed to the C ABI PER CALL
    (include/ocflow_hip.h: enum ocf_corr_policy / ocf_corr_bwd_policy / ocf_warp_bwd_policy / ocf_range_policy; 0 = automatic).  The
    library itself keeps no state; this object is the host-side default, initialised from OCFLOW_{CORR_FWD,CORR_BWD,WARP_BWD,
    RANGE_MAP,WGRAD}_POLICY for whole-model A/B runs, and `kernel_policy(...)` scopes an override for tests and measurements."""
    __slots__ = ("corr_fwd", "corr_bwd", "warp_bwd", "range_map", "wgrad")

    def __init__(self):
        env = os.environ.get
        self.corr_fwd = int(env("OCFLOW_CORR_FWD_POLICY", "0"))
        self.corr_bwd = int(env("OCFLOW_CORR_BWD_POLICY", "0"))
        self.warp_bwd = int(env("OCFLOW_WARP_BWD_POLICY", "0"))
        self.range_map = int(env("OCFLOW_RANGE_MAP_POLICY", "0"))
        self.wgrad = int(env("OCFLOW_WGRAD_POLICY", "0"))


POLICY = KernelPolicy()


class kernel_policy:
    """`with ops.kernel_policy(corr_fwd=_lib.CORR_BAND_ONLY): ...` -- the launches issued inside the block (forward AND the backward
    passes recorded inside it) carry these policies; the previous values come back on exit.  Host-side convenience only."""

    def __init__(self, **kw):
        for k in kw:
            if k not in KernelPolicy.__slots__:
                raise TypeError(f"unknown kernel policy field {k!r}")
        self.kw = kw

    def __enter__(self):
        self.old = {k: getattr(POLICY, k) for k in self.kw}
        for k, v in self.kw.items():
            setattr(POLICY, k, int(v))
        return POLICY

    def __exit__(self, *exc):
        for k, v in self.old.items():
            setattr(POLICY, k, v)
        return False


def set_deterministic(flag: bool = True):
    """Run-to-run bit-identical results for the two scatter kernels (warp backward, range map): they accumulate through 64-bit
    fixed-point integer atomics instead of float atomics (ocf_warp_bwd_det / ocf_range_map_det).  Slower; off by default
    (OCFLOW_DETERMINISTIC=1 turns it on at import).  Every other kernel is deterministic by construction except the fp64 / fp32
    block-partial atomics of the reductions, whose arrival-order effect is below fp32 resolution of the loss."""
    global _DETERMINISTIC
    _DETERMINISTIC = bool(flag)


def _workspace(nbytes_fn, *args, device):
    n = C.c_int64(0)
    _lib.check(nbytes_fn(*args, C.byref(n)), "workspace size query")
    return torch.empty(n.value, device=device, dtype=torch.uint8)


def set_kernel_timer(timer):
    """Install (or remove with None) a runtime.KernelTimer: the cost-volume launches are then bracketed by HIP events
    on the launch stream and tagged with their algorithmic bytes / FLOPs (bench.py's roofline leg)."""
    global _KERNEL_TIMER
    _KERNEL_TIMER = timer


def _launch(name, meta, fn):
    """run one C-ABI launcher; with a KernelTimer installed its kernels are bracketed by dispatch-stamped HIP events and tagged
    with their algorithmic bytes (`meta()` is only evaluated then)"""
    if _KERNEL_TIMER is not None:
        return _KERNEL_TIMER.launch(name, meta(), fn)
    return fn()


# algorithmic bytes per pixel of the streaming kernels (SURVEY 8d4 / DESIGN.md section 4); es = feature element size
def _meta(name_shape, nbytes, flops=0):
    return {"bytes": int(nbytes), "flops": int(flops), "shape": name_shape}


def corr_fwd_algorithmic(B, C, H, W, d, elem_size, has_flow, has_occ):
    """Bytes one fused correlation launch must move: read f1 and f2 once (2*C*s), the flow (8 B/px) and
    occlusion (4 B/px) fields when present, write (2d+1)^2 outputs (s each); FLOPs = 2*C per output."""
    n = (2 * d + 1) ** 2
    px = B * H * W
    return px * (2 * C * elem_size + (8 if has_flow else 0) + (4 if has_occ else 0) + n * elem_size), px * n * 2 * C


# ------------------------------------------------------------------------------------------------ plumbing
def _require_hip(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ocflow_amd.ops: tensors must be on a HIP (cuda) device; the MI355X kernels have no "
                               "CPU fallback")


def _t4(t: Optional[torch.Tensor]):
    if t is None:
        return None
    sb, sc, sh, sw = t.stride()
    return C.byref(_lib.Tensor4(t.data_ptr(), sb, sc, sh, sw))


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return _lib.OCF_F32
    if t.dtype == torch.bfloat16:
        return _lib.OCF_BF16
    raise TypeError(f"ocflow_amd.ops: unsupported feature dtype {t.dtype}")


def _is_cl(t: torch.Tensor) -> bool:
    # (a [B, C, 1, 1] map is both layouts at once: its results are allocated with the plain contiguous strides, which the library's
    # convolutions then take with either weight format -- channels_last STRIDES on such a tensor made MIOpen demand channels_last weights
    # of an NCHW model: the 1 x 1 top level of a 64 x 64 input)
    return t.stride(1) == 1 and t.size(1) > 1 and (t.size(2) * t.size(3) > 1 or not t.is_contiguous())


def _empty_like_layout(ref: torch.Tensor, shape, dtype=None) -> torch.Tensor:
    fmt = torch.channels_last if _is_cl(ref) else torch.contiguous_format
    return torch.empty(shape, device=ref.device, dtype=dtype or ref.dtype, memory_format=fmt)


def _zeros_like_layout(ref: torch.Tensor, shape, dtype=None) -> torch.Tensor:
    fmt = torch.channels_last if _is_cl(ref) else torch.contiguous_format
    return torch.empty(shape, device=ref.device, dtype=dtype or ref.dtype, memory_format=fmt).zero_()


def _features(*ts):
    """Bring feature operands to one kernel dtype (bf16 only if all are bf16, else fp32)."""
    dt = torch.bfloat16 if all(t.dtype == torch.bfloat16 for t in ts) else torch.float32
    return [t if t.dtype == dt else t.to(dt) for t in ts]


def _f32(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else (t if t.dtype == torch.float32 else t.float())


def _check4(t, name, C_=None):
    if t.dim() != 4:
        raise ValueError(f"{name} must be [B,C,H,W], got {tuple(t.shape)}")
    if C_ is not None and t.size(1) != C_:
        raise ValueError(f"{name} must have {C_} channels, got {t.size(1)}")


# ------------------------------------------------------------------------------------- raw launch wrappers
def _warp_fwd_raw(img, flow, flow_scale, align_corners):
    B, C_, H, W = img.shape
    out = _empty_like_layout(img, img.shape)

    def launch():
        with torch.cuda.device(img.device):
            _lib.check(_lib.load().ocf_warp_fwd(_t4(img), _t4(flow), _t4(out), B, C_, H, W, float(flow_scale),
                                                int(bool(align_corners)), _dtype_code(img), _stream(img)), "ocf_warp_fwd")

    _launch("warp_fwd", lambda: _meta(f"B{B}xC{C_}x{H}x{W}", B * H * W * (2 * C_ * img.element_size() + 8)), launch)
    return out


def _warp_bwd_raw(img, flow, gout, flow_scale, align_corners, need_img, need_flow, policy=None):
    B, C_, H, W = img.shape
    if gout.dtype != img.dtype:
        gout = gout.to(img.dtype)
    if _DETERMINISTIC:
        L = _lib.load()
        gimg = _empty_like_layout(img, img.shape, torch.float32) if need_img else None
        gflow = torch.empty((B, 2, H, W), device=img.device, dtype=torch.float32) if need_flow else None
        ws = _workspace(L.ocf_warp_bwd_det_workspace_bytes, B, C_, H, W, device=img.device)
        with torch.cuda.device(img.device):
            _lib.check(L.ocf_warp_bwd_det(_t4(img), _t4(flow), _t4(gout), _t4(gimg), _t4(gflow), C.c_void_p(ws.data_ptr()), B, C_, H, W,
                                          float(flow_scale), int(bool(align_corners)), _dtype_code(img), _stream(img)), "ocf_warp_bwd_det")
        if gimg is not None and gimg.dtype != img.dtype:
            gimg = gimg.to(img.dtype)
        return gimg, gflow
    gimg = _empty_like_layout(img, img.shape, torch.float32) if need_img else None
    gflow = torch.empty((B, 2, H, W), device=img.device, dtype=torch.float32) if need_flow else None
    # the scatter kernel accumulates into grad_img with float atomics (zero it first); the gather kernels overwrite it.  Plan and
    # launch receive the SAME policy value and evaluate one decision function in the library.
    policy = POLICY.warp_bwd if policy is None else policy
    plan = 0 if gimg is None else _lib.load().ocf_warp_bwd_plan(_t4(img), _t4(gout), _t4(gimg), _t4(gflow), B, C_, H, W, _dtype_code(img), policy)
    if plan < 0:
        _lib.check(plan, "ocf_warp_bwd_plan")
    zero_first = bool(plan & 1)
    if zero_first:
        gimg.zero_()
    if plan & 2 and img.dtype != torch.float32:
        # the single-launch small-map kernel writes grad_img with plain stores: take it in the feature dtype (no cast launch afterwards)
        gimg = _empty_like_layout(img, img.shape, img.dtype)
        policy |= _lib.WARP_BWD_GIMG_NATIVE

    # scratch of the binned gather (only allocated where that path is taken: plan == 0)
    ws = None
    if gimg is not None and policy in (_lib.WARP_BWD_AUTO, _lib.WARP_BWD_BINNED) and plan == 0:
        ws = _workspace(_lib.load().ocf_warp_bwd_workspace_bytes, B, C_, H, W, device=img.device)

    def launch():
        with torch.cuda.device(img.device):
            _lib.check(_lib.load().ocf_warp_bwd(_t4(img), _t4(flow), _t4(gout), _t4(gimg), _t4(gflow), B, C_, H, W,
                                                float(flow_scale), int(bool(align_corners)), _dtype_code(img), policy,
                                                None if ws is None else C.c_void_p(ws.data_ptr()), _stream(img)),
                       "ocf_warp_bwd")

    # read gout + img (es each), flow 8; write (atomics) fp32 grad_img 4C, grad_flow 8
    _launch("warp_bwd", lambda: _meta(f"B{B}xC{C_}x{H}x{W}", B * H * W * (2 * C_ * img.element_size() + 4 * C_ + 16)), launch)
    if gimg is not None and gimg.dtype != img.dtype:
        gimg = gimg.to(img.dtype)
    return gimg, gflow


# --------------------------------------------------------------------------------------------- cost volume
def _fuse_warp(f1, f2, flow, occ, out, d, needs_grad) -> bool:
    """one fused launch (True) or warp kernel + correlation kernel (False) for a correlation with a flow: see _FUSED_WARP"""
    if _FUSED_WARP == "1":
        return True  # (with a gradient recorded the backward then re-creates the warped tensor)
    if _FUSED_WARP != "gather" or needs_grad or f1.dtype != torch.bfloat16:
        return False
    B, C_, H, W = f1.shape
    v = _lib.load().ocf_corr_fwd_variant(_t4(f1), _t4(f2), _t4(flow), _t4(occ), _t4(out), B, C_, H, W, d, _dtype_code(f1), POLICY.corr_fwd)
    return v in _lib.CORR_VARIANTS_WITH_GATHER


def _corr_warp_fwd_raw(f1, f2, flow, occ, d, flow_scale, align_corners, out, needs_grad, leaky_slope: float = 0.0):
    """-> (cost volume, warped second operand or None).  Stateless launch sequence shared by the autograd Function below and by the
    registered operator torch.ops.ocflow.corr_warp (torch_ops.py).  leaky_slope != 0: LeakyReLU applied in the kernel's store epilogue."""
    B, C_, H, W = f1.shape
    n = 2 * d + 1
    if out is None:
        out = _empty_like_layout(f1, (B, n * n, H, W))
    # with a flow: ocf_corr_fwd can gather the warped operand inside the correlation kernel (no warped tensor in HBM; same
    # values: the fused kernels round the gathered operand to the feature dtype before the contraction, exactly what the
    # materialised tensor holds).  A training forward keeps the two-launch form -- backward consumes the warped tensor, which
    # is then kept instead of being recomputed; no-grad forwards (evaluation, the second forward of the occlusion-aware step)
    # take the single launch where a gather kernel serves the shape (_fuse_warp).
    w2 = None
    if flow is not None and not _fuse_warp(f1, f2, flow, occ, out, d, needs_grad):
        w2 = _warp_fwd_raw(f2, flow, flow_scale, align_corners)
    op2, opflow = (f2, flow) if w2 is None else (w2, None)
    policy = POLICY.corr_fwd

    def launch():
        with torch.cuda.device(f1.device):
            _lib.check(_lib.load().ocf_corr_fwd(_t4(f1), _t4(op2), _t4(opflow), _t4(occ), _t4(out), B, C_, H, W, d,
                                                float(flow_scale), int(bool(align_corners)), float(leaky_slope), _dtype_code(f1), policy,
                                                _stream(f1)), "ocf_corr_fwd")

    if _KERNEL_TIMER is not None:
        nbytes, flops = corr_fwd_algorithmic(B, C_, H, W, d, f1.element_size(), opflow is not None, occ is not None)
        name = "corr_warp_fwd" if opflow is not None else "corr_fwd"  # (the fused K1+K2 launch is accounted separately: SURVEY 8d4)
        _KERNEL_TIMER.launch(name, {"bytes": nbytes, "flops": flops, "shape": f"B{B}xC{C_}x{H}x{W}"}, launch)
    else:
        launch()
    return out, w2


def _corr_warp_bwd_raw(f1, f2, flow, occ, w2_saved, gout, d, fs, ac, needs, bwd_policy, warp_bwd_policy):
    """gradients (g1, g2, gflow, gocc) of corr_warp; `needs` = which of (f1, f2, flow, occ) want one.  Stateless: every tensor the
    kernels read is an argument (the saved warped operand `w2_saved`, or None to recompute it)."""
    B, C_, H, W = f1.shape
    need1, need2, needf, needo = needs
    if gout.dtype != f1.dtype:
        gout = gout.to(f1.dtype)
    # autograd hands the cost-volume gradient over as a channel slice of the decoder-input gradient (cat backward): 81
    # contiguous values per pixel at the concat buffer's pixel pitch.  The backward kernels read per-pixel tap windows at any
    # pixel pitch, so that view is consumed as is; only a gradient whose taps are not contiguous per pixel (NCHW) is brought to
    # channels_last first.
    if _is_cl(f1) and gout.stride(1) != 1:
        gout = gout.contiguous(memory_format=torch.channels_last)
    # second operand of the plain correlation (kept by the two-launch forward; recomputed after a fused forward)
    if w2_saved is not None:
        w2 = w2_saved
    else:
        w2 = _warp_fwd_raw(f2, flow, fs, ac) if flow is not None else f2
    need_w2 = need2 or needf or needo
    g1 = _empty_like_layout(f1, f1.shape) if need1 else None
    gw = gocc = None
    fused_occ = False
    if occ is not None and (need1 or need_w2):
        # occlusion folded into the kernels: w2*occ is formed while staging, gw comes back already multiplied by occ
        # and the occlusion gradient is a by-product of the same pass (6 elementwise launches per level otherwise)
        gw = _empty_like_layout(w2, w2.shape) if need_w2 else None
        gocc = torch.empty((B, 1, H, W), device=f1.device, dtype=torch.float32) if (needo and need_w2) else None
        occ32 = _f32(occ)
        es = f1.element_size()

        def launch_occ():
            with torch.cuda.device(f1.device):
                return _lib.load().ocf_corr_occ_bwd(_t4(f1), _t4(w2), _t4(occ32), _t4(gout), _t4(g1), _t4(gw), _t4(gocc), B, C_, H, W,
                                                    d, _dtype_code(f1), bwd_policy, _stream(f1))

        # read grad_out n*es, f1, w2 (C*es each), occ 4; write g1, g2 (C*es each), gocc 4
        st = _launch("corr_bwd", lambda: _meta(f"B{B}xC{C_}x{H}x{W}", B * H * W * (4 * C_ * es + (2 * d + 1) ** 2 * es + 8),
                                               2 * 2 * C_ * (2 * d + 1) ** 2 * B * H * W), launch_occ)
        if st == _lib.OCF_EUNSUPPORTED:
            gw = gocc = None  # layout the tiled kernels do not take: compose it from the plain launches below
        else:
            _lib.check(st, "ocf_corr_occ_bwd")
            fused_occ = True
            if gocc is not None and occ.dtype != torch.float32:
                gocc = gocc.to(occ.dtype)
    if not fused_occ:
        w2o = w2 * occ.to(w2.dtype) if occ is not None else w2
        gw = _empty_like_layout(w2o, w2o.shape) if need_w2 else None
        if need1 or need_w2:
            es = f1.element_size()

            def launch_plain():
                with torch.cuda.device(f1.device):
                    _lib.check(_lib.load().ocf_corr_bwd(_t4(f1), _t4(w2o), _t4(gout), _t4(g1), _t4(gw), B, C_, H, W, d,
                                                        _dtype_code(f1), bwd_policy, _stream(f1)), "ocf_corr_bwd")

            _launch("corr_bwd", lambda: _meta(f"B{B}xC{C_}x{H}x{W}", B * H * W * (4 * C_ * es + (2 * d + 1) ** 2 * es),
                                              2 * 2 * C_ * (2 * d + 1) ** 2 * B * H * W), launch_plain)
        if need_w2 and occ is not None:
            if needo:
                gocc = (gw.float() * w2.float()).sum(dim=1, keepdim=True)
            gw = gw * occ.to(gw.dtype)
    g2 = gflow = None
    if need_w2:
        if flow is not None:
            if need2 or needf:
                g2, gflow = _warp_bwd_raw(f2, flow, gw, fs, ac, need2, needf, warp_bwd_policy)
        else:
            g2 = gw if need2 else None
    return g1, g2, gflow, gocc


def _leaky_mask(gout, y, slope: float):
    """gout * (y > 0 ? 1 : slope): the gradient of LeakyReLU taken at its OUTPUT y (same sign as its input for slope > 0), one HIP pass
    (ocf_bias_act_bwd; its bias-gradient by-product goes to a pooled scratch row)"""
    B, C_, H, W = y.shape
    if gout.dtype != y.dtype:
        gout = gout.to(y.dtype)
    dz = torch.empty_like(y)
    gb = _take_zeros(C_, y.device)
    with torch.cuda.device(y.device):
        st = _lib.load().ocf_bias_act_bwd(_t4(y), _t4(gout), _t4(dz), C.c_void_p(gb.data_ptr()), B, C_, H, W, float(slope), 1, _dtype_code(y),
                                          _stream(y))
    if st == _lib.OCF_EUNSUPPORTED:  # a layout the epilogue kernel does not take
        return gout * torch.where(y > 0, 1.0, float(slope)).to(gout.dtype)
    _lib.check(st, "ocf_bias_act_bwd")
    return dz


class _CorrWarpFn(Function):
    @staticmethod
    def forward(ctx, f1, f2, flow, occ, d, flow_scale, align_corners, out, leaky_slope=0.0, grad_premasked=False):
        out, w2 = _corr_warp_fwd_raw(f1, f2, flow, occ, d, flow_scale, align_corners, out, any(ctx.needs_input_grad[:4]), leaky_slope)
        # (grad_premasked: the consumer hands the gradient over already multiplied by the activation's derivative --
        # conv.conv3x3_dense_block's input-gradient epilogue --, so the activated volume need not be kept)
        ctx.leaky = float(leaky_slope) if (leaky_slope and not grad_premasked) else 0.0
        ctx.save_for_backward(f1, f2, flow, occ, w2, out if ctx.leaky else None)
        ctx.cfg = (d, float(flow_scale), bool(align_corners))
        ctx.bwd_policy = POLICY.corr_bwd
        ctx.warp_bwd_policy = POLICY.warp_bwd
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        f1, f2, flow, occ, w2_saved, y = ctx.saved_tensors
        d, fs, ac = ctx.cfg
        if ctx.leaky:
            gout = _leaky_mask(gout, y, ctx.leaky)
        g1, g2, gflow, gocc = _corr_warp_bwd_raw(f1, f2, flow, occ, w2_saved, gout, d, fs, ac, ctx.needs_input_grad[:4], ctx.bwd_policy,
                                                 ctx.warp_bwd_policy)
        return g1, g2, gflow, gocc, None, None, None, None, None, None


def corr_warp(features1, features2, flow=None, occ=None, max_displacement=4, flow_scale=1.0, align_corners=False, out=None, *,
              leaky_slope=0.0, grad_premasked=False):
    """Cost volume of features1 against  warp(features2, flow*flow_scale) [* occ].

    leaky_slope != 0: `LeakyReLU(leaky_slope)` of the volume, applied in the kernel's store epilogue (cost_volume_flow_net.py:173,189:
    `self.leakyRELU(corr)` as a separate pass over the 81-channel volume).  Its backward multiplies the incoming gradient by the
    activation's derivative in one HIP pass -- unless `grad_premasked`, a promise of the consumer that the gradient it sends back is
    already so multiplied (conv.conv3x3_dense_block does that in the epilogue of its input-gradient kernels).

    With a flow: a forward that records a gradient is TWO launches -- ocf_warp_fwd materialises the warped operand (backward consumes
    it), then ocf_corr_fwd with the mask applied to the band outputs; so is a no-grad forward by default (measured faster); with
    OCFLOW_FUSED_WARP=gather a no-grad forward is ONE launch where a kernel with the in-kernel gather serves the shape (the warped
    tensor then never exists in HBM; same values bit for bit).  OCFLOW_FUSED_WARP=0 / 1
    force either form.

    `out`: optional pre-allocated [B, (2d+1)^2, H, W] tensor of the features' dtype with unit channel stride and ANY pixel pitch -- e.g.
    the first 81 channels of the decoder-input buffer (flow_occ_net.py:171-176 concatenates the cost volume in front of the features):
    the kernel writes there and the concatenation copy disappears.  Forward only (no gradient may be recorded: a view of a wider
    buffer cannot carry this operator's backward; the differentiable form assembles the whole buffer in one autograd node).

    Replaces the per-level chain  warp -> [* occ] -> correlation  of flow_net.py:110-112 and
    cost_volume_flow_occ_net.py:204-206 (with flow=None, occ given: flow_occ_net.py:167-168)."""
    _check4(features1, "features1")
    _check4(features2, "features2")
    if features1.shape != features2.shape:
        raise ValueError(f"feature shapes differ: {tuple(features1.shape)} vs {tuple(features2.shape)}")
    d = int(max_displacement)
    if d < 1 or d > MAX_DISPLACEMENT:
        raise ValueError(f"max_displacement must be in [1, {MAX_DISPLACEMENT}] for the MI355X kernel, got {d}")
    _require_hip(features1, features2, flow, occ)
    B, _, H, W = features1.shape
    if flow is not None:
        _check4(flow, "flow", 2)
        if flow.shape[0] != B or flow.shape[2:] != features1.shape[2:]:
            raise ValueError("flow must be [B,2,H,W] matching the features")
    if occ is not None:
        _check4(occ, "occ", 1)
        if occ.shape[0] != B or occ.shape[2:] != features1.shape[2:]:
            raise ValueError("occ must be [B,1,H,W] matching the features")
    f1, f2 = _features(features1, features2)
    if f1.numel() == 0:  # empty batch / empty map: what the reference's pad-and-shift loop returns, without a launch
        n = 2 * d + 1
        return (f1.sum() + f2.sum()) * 0 + f1.new_zeros((B, n * n, H, W))
    if out is not None:
        n = 2 * d + 1
        if out.shape != (B, n * n, H, W) or out.dtype != f1.dtype or out.device != f1.device or out.stride(1) != 1:
            raise ValueError(f"out must be a [{B},{n * n},{H},{W}] {f1.dtype} tensor with unit channel stride on {f1.device}")
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (f1, f2, flow, occ, out)):
            raise RuntimeError("corr_warp(out=...) is a forward-only form: call it under torch.no_grad() or on detached tensors")
        return _corr_warp_fwd_raw(f1, f2, _f32(flow), _f32(occ), d, flow_scale, align_corners, out, False, leaky_slope)[0]
    if _tracing() and out is None:
        from . import torch_ops  # noqa: F401
        cv = torch.ops.ocflow.corr_warp(f1, f2, _f32(flow), _f32(occ), d, float(flow_scale), bool(align_corners))[0]
        return torch.nn.functional.leaky_relu(cv, float(leaky_slope)) if leaky_slope else cv
    return _CorrWarpFn.apply(f1, f2, _f32(flow), _f32(occ), d, flow_scale, align_corners, out, float(leaky_slope), bool(grad_premasked))


def compute_cost_volume(features1, features2, max_displacement=4, *, leaky_slope=0.0, grad_premasked=False):
    """Drop-in for models/networks/correlation_layer.py:7-40.

    out[b, (dy+d)(2d+1)+(dx+d), y, x] = mean_c features1[b,c,y,x] * features2[b,c,y+dy,x+dx]   (zero outside).
    leaky_slope / grad_premasked: see corr_warp (the activation the PWC-style nets apply right after it, fused into the launch)."""
    return corr_warp(features1, features2, None, None, max_displacement, leaky_slope=leaky_slope, grad_premasked=grad_premasked)


# ---------------------------------------------------------------------------------------------------- warp
class _WarpFn(Function):
    @staticmethod
    def forward(ctx, img, flow, flow_scale, align_corners):
        ctx.save_for_backward(img, flow)
        ctx.cfg = (float(flow_scale), bool(align_corners))
        ctx.bwd_policy = POLICY.warp_bwd
        return _warp_fwd_raw(img, flow, flow_scale, align_corners)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        img, flow = ctx.saved_tensors
        fs, ac = ctx.cfg
        gimg, gflow = _warp_bwd_raw(img, flow, gout, fs, ac, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.bwd_policy)
        return gimg, gflow, None, None


def warp(img, flow, align_corners=False, flow_scale=1.0):
    """Backward bilinear warp of `img` [B,C,H,W] by `flow` [B,2,H,W] (zeros padding).

    align_corners=False: the in-network feature warp (cost_volume_flow_net.py:121-151 and clones) -- the grid is
    normalised with 2x/(W-1)-1 but sampled with align_corners=False, i.e. at ((x+u)W/(W-1)-0.5, ...).
    align_corners=True : the loss-side image warp (models/model.py:191-221, flow_model.py:49-79), samples at (x+u, y+v)."""
    _check4(img, "img")
    _check4(flow, "flow", 2)
    if flow.shape[0] != img.shape[0] or flow.shape[2:] != img.shape[2:]:
        raise ValueError("flow must be [B,2,H,W] matching img")
    _require_hip(img, flow)
    (x,) = _features(img)
    if x.numel() == 0:  # empty batch: grid_sample of nothing
        return x * 1
    if _tracing():
        from . import torch_ops  # noqa: F401
        return torch.ops.ocflow.warp(x, _f32(flow), float(flow_scale), bool(align_corners))
    return _WarpFn.apply(x, _f32(flow), flow_scale, align_corners)


# ------------------------------------------------------------------------------------ feature normalisation
def _moments(f):
    B, C_, H, W = f.shape
    sums = torch.empty((B, 2), device=f.device, dtype=torch.float64)
    with torch.cuda.device(f.device):
        _lib.check(_lib.load().ocf_moments(_t4(f), C.c_void_p(sums.data_ptr()), B, C_, H, W, _dtype_code(f), _stream(f)),
                   "ocf_moments")
    return sums


def _norm_kernels_take(feats) -> bool:
    """ocf_norm_*: dense (NCHW-contiguous or channels_last) maps of one stride pattern with whole 16-byte units per sample"""
    f0 = feats[0]
    if not all(f.stride() == f0.stride() and (f.is_contiguous() or f.is_contiguous(memory_format=torch.channels_last)) for f in feats):
        return False
    return (f0[0].numel() * f0.element_size()) % 16 == 0 and all(f.data_ptr() % 16 == 0 for f in feats) and f0.shape[0] <= 4096


class _NormalizeKernelFn(Function):
    """normalize_features on the HIP kernels of csrc/norm.hip: forward = ocf_moments + ONE pass per map, backward = ONE reduction pass +
    ONE elementwise pass per map (the composed form below takes ~10 framework passes per direction)."""

    @staticmethod
    def forward(ctx, *feats):
        S, (B, C_, H, W) = len(feats), feats[0].shape
        L = _lib.load()
        sums = torch.empty((S, B, 2), device=feats[0].device, dtype=torch.float64)
        outs = []
        with torch.cuda.device(feats[0].device):
            for s, f in enumerate(feats):
                _lib.check(L.ocf_moments(_t4(f), C.c_void_p(sums[s].data_ptr()), B, C_, H, W, _dtype_code(f), _stream(f)), "ocf_moments")
            for s, f in enumerate(feats):
                y = torch.empty_like(f)
                _lib.check(L.ocf_norm_fwd(_t4(f), _t4(y), C.c_void_p(sums.data_ptr()), S, s, B, C_, H, W, _dtype_code(f), _stream(f)), "ocf_norm_fwd")
                outs.append(y)
        ctx.save_for_backward(*feats, sums)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gouts):
        *feats, sums = ctx.saved_tensors
        S, (B, C_, H, W) = len(feats), feats[0].shape
        L = _lib.load()
        gs = []
        for g, f in zip(gouts, feats):
            g = g if g.dtype == f.dtype else g.to(f.dtype)
            gs.append(g if g.stride() == f.stride() else torch.empty_like(f).copy_(g))
        bsums = torch.empty(2, device=feats[0].device, dtype=torch.float64)
        grads = []
        with torch.cuda.device(feats[0].device):
            for s, (g, f) in enumerate(zip(gs, feats)):
                _lib.check(L.ocf_norm_bwd_sums(_t4(f), _t4(g), C.c_void_p(sums.data_ptr()), C.c_void_p(bsums.data_ptr()), int(s == 0), S, B, C_, H, W,
                                               _dtype_code(f), _stream(f)), "ocf_norm_bwd_sums")
            for s, (g, f) in enumerate(zip(gs, feats)):
                gx = torch.empty_like(f)
                _lib.check(L.ocf_norm_bwd(_t4(f), _t4(g), _t4(gx), C.c_void_p(sums.data_ptr()), C.c_void_p(bsums.data_ptr()), S, s, B, C_, H, W,
                                          _dtype_code(f), _stream(f)), "ocf_norm_bwd")
                grads.append(gx)
        return tuple(grads)


class _NormalizeFn(Function):
    """(f_s - mu)/sqrt(var+1e-16) with ONE scalar mu/var over {maps, batch} (correlation_layer.py:59-82);
    gradients flow through mu and var as in the reference.  Composed form (ocf_moments + framework elementwise passes) for layouts the
    kernels of csrc/norm.hip do not take (strided views)."""

    @staticmethod
    def forward(ctx, *feats):
        n = float(feats[0][0].numel())
        sums = torch.stack([_moments(f) for f in feats])  # [S,B,2] fp64
        m1 = sums[..., 0] / n
        var_sb = sums[..., 1] / n - m1 * m1
        mu = m1.mean()
        var = var_sb.mean()
        inv_std = torch.rsqrt(var + 1e-16)
        outs = tuple(((f.float() - mu.float()) * inv_std.float()).to(f.dtype) for f in feats)
        ctx.save_for_backward(*feats, m1.float(), mu.float(), inv_std.float())
        ctx.n = n
        return outs

    @staticmethod
    @once_differentiable
    def backward(ctx, *gouts):
        *feats, m1, mu, inv_std = ctx.saved_tensors
        S, B = m1.shape
        n = ctx.n
        gsum = sum(g.float().sum() for g in gouts)
        gdot = sum((g.float() * (f.float() - mu)).sum() for g, f in zip(gouts, feats))
        dmu = -inv_std * gsum / (S * B * n)
        dvar = -0.5 * gdot * inv_std ** 3
        grads = []
        for s, (g, f) in enumerate(zip(gouts, feats)):
            centred = f.float() - m1[s].view(B, 1, 1, 1)
            grads.append((g.float() * inv_std + dmu + dvar * 2.0 * centred / (S * B * n)).to(f.dtype))
        return tuple(grads)


def _normalize_flags(feats, normalize, center, across_channels, across_images):
    """the non-default flag combinations of correlation_layer.py:42-82: per-(sample[, channel]) biased moments, optionally averaged over
    maps and batch; centre and / or scale.  No call site of the reference reaches this; plain tensor operations (fp32 statistics)."""
    for f in feats:
        _check4(f, "feature")
    _require_hip(*feats)
    dim = (1, 2, 3) if across_channels else (2, 3)
    moments = [torch.var_mean(f.float(), dim=dim, keepdim=True, unbiased=False) for f in feats]
    means, variances = [m for _, m in moments], [v for v, _ in moments]
    if across_images:
        means = [torch.stack(means).mean()] * len(feats)
        variances = [torch.stack(variances).mean()] * len(feats)
    out = [f.float() for f in feats]
    if center:
        out = [f - m for f, m in zip(out, means)]
    if normalize:
        out = [f / torch.sqrt(v + 1e-16) for f, v in zip(out, variances)]
    return [o.to(f.dtype) for o, f in zip(out, feats)]


def normalize_features(feature_list: Sequence[torch.Tensor], normalize=True, center=True, moments_across_channels=True,
                       moments_across_images=True) -> List[torch.Tensor]:
    """Drop-in for models/networks/correlation_layer.py:42-82.  The defaults -- what every call site of the reference uses -- run on the
    HIP kernels (csrc/norm.hip); the other flag combinations are composed from framework reductions on the same device."""
    if not (normalize and center and moments_across_channels and moments_across_images):
        return _normalize_flags(list(feature_list), normalize, center, moments_across_channels, moments_across_images)
    feats = list(feature_list)
    for f in feats:
        _check4(f, "feature")
    _require_hip(*feats)
    if any(f.shape != feats[0].shape for f in feats):
        raise ValueError("normalize_features: all feature maps must have the same shape")
    feats = _features(*feats)
    if _norm_kernels_take(feats) and os.environ.get("OCFLOW_COMPOSED_NORMALIZE", "0") != "1":
        return list(_NormalizeKernelFn.apply(*feats))
    return list(_NormalizeFn.apply(*feats))


# ------------------------------------------------------------------------------------------- photometric
def robust_l1(x, alpha=0.001):
    """Charbonnier penalty (models/model.py:27-35); plain elementwise helper."""
    return (x ** 2 + alpha ** 2) ** 0.5


def _photo_fwd_raw(img1, img2, flow, occ):
    """fp64 [2]: sum e * (1 - occ), sum (1 - occ)   (ocf_photo_loss_fwd)"""
    B, C_, H, W = img1.shape
    sums = torch.empty(2, device=img1.device, dtype=torch.float64)

    def launch():
        with torch.cuda.device(img1.device):
            _lib.check(_lib.load().ocf_photo_loss_fwd(_t4(img1), _t4(img2), _t4(flow), _t4(occ), C.c_void_p(sums.data_ptr()),
                                                      B, C_, H, W, _stream(img1)), "ocf_photo_loss_fwd")

    px_bytes = 8 * C_ + (8 if flow is not None else 0) + (4 if occ is not None else 0)  # 36 for RGB + flow + mask
    _launch("photo_fwd", lambda: _meta(f"B{B}xC{C_}x{H}x{W}", B * H * W * px_bytes), launch)
    return sums


def _photo_bwd_raw(img1, img2, flow, occ, scale, want_flow, want_pred):
    """(grad_flow [B,2,H,W] | None, dLoss/dP [B,C,H,W] | None) for dLoss/d sums[0] = `scale` (device fp32 scalar)"""
    B, C_, H, W = img1.shape
    scale = scale.float().reshape(1).contiguous()
    gflow = torch.empty((B, 2, H, W), device=img1.device, dtype=torch.float32) if (flow is not None and want_flow) else None
    gpred = torch.empty((B, C_, H, W), device=img1.device, dtype=torch.float32) if (flow is None and want_pred) else None
    if gflow is not None or gpred is not None:
        def launch():
            with torch.cuda.device(img1.device):
                _lib.check(_lib.load().ocf_photo_loss_bwd(_t4(img1), _t4(img2), _t4(flow), _t4(occ), C.c_void_p(scale.data_ptr()),
                                                          _t4(gflow), _t4(gpred), B, C_, H, W, _stream(img1)),
                           "ocf_photo_loss_bwd")

        px_bytes = 8 * C_ + (16 if flow is not None else 4 * C_) + (4 if occ is not None else 0)  # 44 for RGB + flow + mask
        _launch("photo_bwd", lambda: _meta(f"B{B}xC{C_}x{H}x{W}", B * H * W * px_bytes), launch)
    return gflow, gpred


class _PhotoFn(Function):
    @staticmethod
    def forward(ctx, img1, img2, flow, occ):
        sums = _photo_fwd_raw(img1, img2, flow, occ)
        ctx.save_for_backward(img1, img2, flow, occ)
        num, den = sums[0].float(), sums[1].float()
        ctx.mark_non_differentiable(den)
        return num, den

    @staticmethod
    @once_differentiable
    def backward(ctx, gnum, _gden):
        img1, img2, flow, occ = ctx.saved_tensors
        B, C_, H, W = img1.shape
        n1, n2, nf, no = ctx.needs_input_grad
        if no:
            raise NotImplementedError("photometric_error: gradient w.r.t. the occlusion mask is not implemented "
                                      "(the reference computes the mask under no_grad, models/model.py:381-391)")
        if flow is not None and (n1 or n2):
            raise NotImplementedError("warp_photometric_error: image gradients through the fused warp are not "
                                      "implemented; use ops.warp + ops.photometric_error")
        gflow, gpred = _photo_bwd_raw(img1, img2, flow, occ, gnum, nf, n1 or n2)
        g1 = -gpred if (gpred is not None and n1) else None
        g2 = gpred if (gpred is not None and n2) else None
        return g1, g2, gflow, None


def _tracing() -> bool:
    """inside torch.compile / make_fx: the operators go through their dispatcher identities (torch.ops.ocflow.*, torch_ops.py) -- a
    ctypes launch cannot be traced; eager calls keep the direct path (no dispatcher overhead per launch)"""
    return torch.compiler.is_compiling()


def _photo(img1, img2, flow, occ):
    _check4(img1, "img")
    _check4(img2, "img_pred")
    if img1.shape != img2.shape:
        raise ValueError("photometric_error: image shapes differ")
    _require_hip(img1, img2, flow, occ)
    if occ is not None:
        _check4(occ, "occ", 1)
    if _tracing():
        from . import torch_ops  # noqa: F401  (registers torch.ops.ocflow.*)
        sums = torch.ops.ocflow.photo_loss(_f32(img1), _f32(img2), _f32(flow), _f32(occ))
        num, den = sums[0], sums[1].detach()
    else:
        num, den = _PhotoFn.apply(_f32(img1), _f32(img2), _f32(flow), _f32(occ))
    if occ is not None:
        return num / (den * 3 + 1e-16)  # the reference's literal 3 (models/model.py:43), whatever the channel count
    return num / float(img1.numel())


def photometric_error(img_pred, img, occ=None):
    """Drop-in for models/model.py:37-46: mean Charbonnier error, or sum(e*(1-occ)) / (3*sum(1-occ) + 1e-16)
    (the 3 is the reference's literal: its images are RGB; occ: 1 occluded, 0 visible)."""
    return _photo(img, img_pred, None, occ)


def warp_photometric_error(img1, img2, flow, occ=None):
    """photometric_error(warp(img2, flow, align_corners=True), img1, occ) in one pass: the warped image is never
    written (models/model.py:379+394 / :327-330)."""
    _check4(flow, "flow", 2)
    return _photo(img1, img2, flow, occ)


# --------------------------------------------------------------------------------------------- masked L1
class _MaskedL1Fn(Function):
    @staticmethod
    def forward(ctx, pred, target, mask):
        B, C_, H, W = pred.shape
        sums = torch.empty(2, device=pred.device, dtype=torch.float64)
        with torch.cuda.device(pred.device):
            _lib.check(_lib.load().ocf_masked_l1_fwd(_t4(pred), _t4(target), _t4(mask), C.c_void_p(sums.data_ptr()), B, C_, H, W,
                                                     _stream(pred)), "ocf_masked_l1_fwd")
        ctx.save_for_backward(pred, target, mask)
        num, den = sums[0].float(), sums[1].float()
        ctx.mark_non_differentiable(den)
        return num, den

    @staticmethod
    @once_differentiable
    def backward(ctx, gnum, _gden):
        pred, target, mask = ctx.saved_tensors
        n_p, n_t, n_m = ctx.needs_input_grad
        if n_m:
            raise NotImplementedError("masked_l1: gradient w.r.t. the mask is not implemented (the reference's mask is data, "
                                      "models/inpainting_model.py:63)")
        if not (n_p or n_t):
            return None, None, None
        B, C_, H, W = pred.shape
        scale = gnum.float().reshape(1).contiguous()
        g = torch.empty((B, C_, H, W), device=pred.device, dtype=torch.float32)
        with torch.cuda.device(pred.device):
            _lib.check(_lib.load().ocf_masked_l1_bwd(_t4(pred), _t4(target), _t4(mask), C.c_void_p(scale.data_ptr()), _t4(g), B, C_, H, W,
                                                     _stream(pred)), "ocf_masked_l1_bwd")
        return (g if n_p else None), (-g if n_t else None), None


def masked_l1(pred, target, mask):
    """`(torch.abs(pred - target) * mask).sum() / (C * mask.sum() + 1e-16)` -- the completion loss of the inpainting head over the
    occluded pixels only (models/inpainting_model.py:71; C = 3 there).  pred / target [B,C,H,W], mask [B,1,H,W]; one reduction pass
    forward and one elementwise pass backward instead of sub, abs, mul, two sums and their five backward kernels."""
    _check4(pred, "pred")
    _check4(target, "target")
    _check4(mask, "mask", 1)
    if pred.shape != target.shape or mask.shape[0] != pred.shape[0] or mask.shape[2:] != pred.shape[2:]:
        raise ValueError("masked_l1: pred / target [B,C,H,W] and mask [B,1,H,W] must agree")
    _require_hip(pred, target, mask)
    num, den = _MaskedL1Fn.apply(_f32(pred), _f32(target), _f32(mask))
    return num / (den * 3 + 1e-16)  # the reference's literal 3 (inpainting_model.py:73)


# -------------------------------------------------------------------------------------------- smoothness
def gradient(img, stride=1):
    """Forward differences (models/model.py:53-66); plain slicing helper."""
    assert img.dim() == 4
    return img[:, :, :, stride:] - img[:, :, :, :-stride], img[:, :, stride:, :] - img[:, :, :-stride, :]


def _smooth_fwd_raw(img, flow, alpha):
    """fp64 [4]: the four weighted Charbonnier sums of ocf_smooth_loss_fwd"""
    B, C_, H, W = img.shape
    sums = torch.empty(4, device=img.device, dtype=torch.float64)

    def launch():
        with torch.cuda.device(img.device):
            _lib.check(_lib.load().ocf_smooth_loss_fwd(_t4(img), _t4(flow), C.c_void_p(sums.data_ptr()), B, C_, H, W,
                                                       float(alpha), _stream(img)), "ocf_smooth_loss_fwd")

    _launch("smooth_fwd", lambda: _meta(f"B{B}xC{C_}x{H}x{W}", B * H * W * (4 * C_ + 8)), launch)
    return sums


def _smooth_bwd_raw(img, flow, coef, alpha):
    """grad_flow for dLoss/d sums = `coef` (device fp32 [4])"""
    B, C_, H, W = img.shape
    coef = coef.float().contiguous()
    gflow = torch.empty((B, 2, H, W), device=img.device, dtype=torch.float32)

    def launch():
        with torch.cuda.device(img.device):
            _lib.check(_lib.load().ocf_smooth_loss_bwd(_t4(img), _t4(flow), C.c_void_p(coef.data_ptr()), _t4(gflow), B, C_, H, W,
                                                       float(alpha), _stream(img)), "ocf_smooth_loss_bwd")

    _launch("smooth_bwd", lambda: _meta(f"B{B}xC{C_}x{H}x{W}", B * H * W * (4 * C_ + 16)), launch)
    return gflow


class _SmoothFn(Function):
    @staticmethod
    def forward(ctx, img, flow, alpha):
        sums = _smooth_fwd_raw(img, flow, alpha)
        ctx.save_for_backward(img, flow)
        ctx.alpha = float(alpha)
        return sums.float()

    @staticmethod
    @once_differentiable
    def backward(ctx, gsums):
        img, flow = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("smoothness loss: gradient w.r.t. the image is not implemented")
        return None, _smooth_bwd_raw(img, flow, gsums, ctx.alpha), None


def smoothness_losses(img, flow, alpha=100.0):
    """(first_order, second_order) edge-aware smoothness of models/model.py:93-114 from ONE pass over img/flow."""
    _check4(img, "img")
    _check4(flow, "flow", 2)
    if img.shape[0] != flow.shape[0] or img.shape[2:] != flow.shape[2:]:
        raise ValueError("smoothness: img and flow must share B,H,W")
    _require_hip(img, flow)
    B, _, H, W = img.shape
    if _tracing():
        from . import torch_ops  # noqa: F401
        s = torch.ops.ocflow.smooth_loss(_f32(img), _f32(flow), float(alpha))
    else:
        s = _SmoothFn.apply(_f32(img), _f32(flow), alpha)
    nan = float("nan")  # torch.mean over an empty tensor is nan in the reference too
    n = [B * 2 * H * (W - 1), B * 2 * (H - 1) * W, B * 2 * H * (W - 2), B * 2 * (H - 2) * W]
    inv = [1.0 / v if v > 0 else nan for v in n]
    first = 0.5 * (s[0] * inv[0] + s[1] * inv[1])
    second = 0.5 * (s[2] * inv[2] + s[3] * inv[3])
    return first, second


def first_order_smoothness_loss(img, flow, alpha=100.0):
    """Drop-in for models/model.py:93-101."""
    return smoothness_losses(img, flow, alpha)[0]


def second_order_smoothness_loss(img, flow, alpha=100.0):
    """Drop-in for models/model.py:103-114."""
    return smoothness_losses(img, flow, alpha)[1]


# --------------------------------------------------------------------------------------------- range map
def _range_map(flow, want_occ):
    _check4(flow, "flow", 2)
    _require_hip(flow)
    f = _f32(flow.detach())
    if _tracing():
        from . import torch_ops  # noqa: F401
        rng, occ = torch.ops.ocflow.range_map(f, bool(want_occ))
        return rng, (occ if want_occ else None)
    return _range_map_raw(f, want_occ)


def _range_map_raw(f, want_occ):
    B, _, H, W = f.shape
    rng = torch.empty((B, 1, H, W), device=f.device, dtype=torch.float32)
    occ = torch.empty_like(rng) if want_occ else None
    occ_p = C.c_void_p(occ.data_ptr()) if want_occ else None
    with torch.cuda.device(f.device):
        if _DETERMINISTIC:
            L = _lib.load()
            ws = _workspace(L.ocf_range_map_det_workspace_bytes, B, H, W, device=f.device)
            _lib.check(L.ocf_range_map_det(_t4(f), C.c_void_p(rng.data_ptr()), occ_p, C.c_void_p(ws.data_ptr()), B, H, W, _stream(f)),
                       "ocf_range_map_det")
        else:
            _launch("range_map", lambda: _meta(f"B{B}x{H}x{W}", B * H * W * 16),
                    lambda: _lib.check(_lib.load().ocf_range_map(_t4(f), C.c_void_p(rng.data_ptr()), occ_p, B, H, W, POLICY.range_map, _stream(f)),
                                       "ocf_range_map"))
    return rng, occ


def compute_range_map(flow):
    """Drop-in for FlowStageModel.compute_range_map (models/model.py:243-305): forward splat of the bilinear weights
    of (grid + flow); non-differentiable (the reference calls it under no_grad and scatter_add_ has no grad to flow)."""
    return _range_map(flow, False)[0]


def occlusion_from_flow(flow):
    """occ = 1 - clamp(range_map(flow), 0, 1)  (models/model.py:388-391) in the same launch sequence."""
    return _range_map(flow, True)[1]


# -------------------------------------------------------------------------------------------------- SSIM
class _SsimFn(Function):
    """per-sample sums of the SSIM map; differentiable w.r.t. BOTH images (the gradient w.r.t. the second is the same kernel pair with
    the images swapped: the formula is symmetric)"""

    @staticmethod
    def forward(ctx, a, b, ws):
        B, C_, H, W = a.shape
        sums = torch.empty(B, device=a.device, dtype=torch.float64)
        with torch.cuda.device(a.device):
            _lib.check(_lib.load().ocf_ssim_fwd(_t4(a), _t4(b), C.c_void_p(sums.data_ptr()), B, C_, H, W, ws, _stream(a)), "ocf_ssim_fwd")
        ctx.save_for_backward(a, b)
        ctx.ws = ws
        return sums.float()

    @staticmethod
    @once_differentiable
    def backward(ctx, gsums):
        a, b = ctx.saved_tensors
        B, C_, H, W = a.shape
        L = _lib.load()
        coef = gsums.float().contiguous()
        grads = []
        for need, x, y in ((ctx.needs_input_grad[0], a, b), (ctx.needs_input_grad[1], b, a)):
            if not need:
                grads.append(None)
                continue
            g = torch.empty((B, C_, H, W), device=a.device, dtype=torch.float32)
            ws_buf = _workspace(L.ocf_ssim_bwd_workspace_bytes, B, C_, H, W, ctx.ws, device=a.device)
            with torch.cuda.device(a.device):
                _lib.check(L.ocf_ssim_bwd(_t4(x), _t4(y), C.c_void_p(coef.data_ptr()), _t4(g), C.c_void_p(ws_buf.data_ptr()), B, C_, H, W,
                                          ctx.ws, _stream(a)), "ocf_ssim_bwd")
            grads.append(g)
        return grads[0], grads[1], None


def ssim(img1, img2, window_size=11, size_average=True):
    """Drop-in for inpainting_metrics/ssim/ssim.py:65-75: mean SSIM (Gaussian window sigma 1.5, zero padding window_size // 2), or the
    per-sample means when size_average is False.  One kernel instead of five depthwise convolutions and an elementwise chain;
    DIFFERENTIABLE w.r.t. both images (two kernels per image instead of autograd through the chain), which is what the
    `photo_mode='ssim'` option of FlowStageModel trains with."""
    _check4(img1, "img1")
    _check4(img2, "img2")
    if img1.shape != img2.shape:
        raise ValueError("ssim: image shapes differ")
    ws = int(window_size)
    if ws < 1 or ws > 16:
        raise ValueError(f"ssim: window_size must be in [1, 16] for the MI355X kernel, got {ws}")
    _require_hip(img1, img2)
    a, b = _f32(img1), _f32(img2)
    B, C_, H, W = a.shape
    sums = _SsimFn.apply(a, b, ws)
    pad = ws // 2
    n = C_ * (H + 2 * pad - ws + 1) * (W + 2 * pad - ws + 1)
    per_sample = sums / n
    return per_sample.mean() if size_average else per_sample


def psnr(img1, img2, pixel_max=255.0, per_sample=True):
    """Drop-in for inpainting_metrics/psnr/psnr.py:4-11 applied image by image as calculate_psnr does
    (inpainting_metrics/__init__.py:76-78: `psnr((img+1)*127.5, (complete+1)*127.5)` on numpy copies of each image): here one
    kernel over the batch, on the device.  `img1`, `img2`: [B,C,H,W] already in the 0..pixel_max scale.  Returns the per-image
    values [B] (100 where the images are identical, like the reference), or their mean when per_sample is False."""
    _check4(img1, "img1")
    _check4(img2, "img2")
    if img1.shape != img2.shape:
        raise ValueError("psnr: image shapes differ")
    _require_hip(img1, img2)
    a, b = _f32(img1.detach()), _f32(img2.detach())
    B, C_, H, W = a.shape
    sums = torch.empty(B, device=a.device, dtype=torch.float64)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().ocf_sqdiff(_t4(a), _t4(b), C.c_void_p(sums.data_ptr()), B, C_, H, W, _stream(a)), "ocf_sqdiff")
    mse = sums / (C_ * H * W)
    val = torch.where(mse == 0, torch.full_like(mse, 100.0), 20.0 * torch.log10(float(pixel_max) / torch.sqrt(mse.clamp_min(1e-300))))
    return val if per_sample else val.mean()


# ------------------------------------------------------------------------- channel concatenation of the decoder inputs
def concat_raw(parts, out, cout: int):
    """ocf_concat_channels into `out` = a [B, cout, H, W] bf16 channels_last tensor OR a channel slice of a wider one (16-byte aligned
    pixels): out[:, :sum C_k] = cat(parts), out[:, sum C_k:] = 0"""
    B, _, H, W = parts[0].shape
    n = len(parts)
    t4s = [_lib.Tensor4(t.data_ptr(), *t.stride()) for t in parts]
    arr = (C.POINTER(_lib.Tensor4) * n)(*[C.pointer(t) for t in t4s])
    chans = (C.c_int * n)(*[int(t.shape[1]) for t in parts])
    with torch.cuda.device(out.device):
        fn = lambda: _lib.check(_lib.load().ocf_concat_channels(arr, chans, n, _t4(out), int(cout), B, H, W, _lib.OCF_BF16, _stream(out)),
                                "ocf_concat_channels")
        _launch("concat", lambda: _meta(f"B{B}x{'+'.join(str(int(t.shape[1])) for t in parts)}->{cout}x{H}x{W}",
                                        2 * B * H * W * (sum(int(t.shape[1]) for t in parts) + int(cout))), fn)
    return out


class _ConcatFn(Function):
    @staticmethod
    def forward(ctx, cout, *parts):
        B, _, H, W = parts[0].shape
        out = torch.empty((B, cout, H, W), device=parts[0].device, dtype=torch.bfloat16, memory_format=torch.channels_last)
        concat_raw(parts, out, cout)
        ctx.sizes = [int(t.shape[1]) for t in parts]
        return out

    @staticmethod
    def backward(ctx, gy):
        grads, c0 = [], 0
        for i, c in enumerate(ctx.sizes):
            grads.append(gy.narrow(1, c0, c) if ctx.needs_input_grad[1 + i] else None)
            c0 += c
        return (None, *grads)


CONCAT_MAX_CHANNELS = 768  # ocf_concat_channels assembles 32-pixel rows in 48 KB of LDS (csrc/concat.hip)


def concat_channels_eligible(parts, cout=None) -> bool:
    """operands ocf_concat_channels takes: up to six bf16 pixel-major [B, C_k, H, W] tensors on one HIP device, at most 768 output
    channels (`cout`: the padded channel count the caller will ask for; default the sum rounded up to 32)"""
    if not (1 <= len(parts) <= 6):
        return False
    total = sum(int(t.shape[1]) for t in parts)
    if max((total + 31) // 32 * 32, int(cout or 0)) > CONCAT_MAX_CHANNELS:
        return False
    x0 = parts[0]
    for t in parts:
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 4 and t.shape[0] == x0.shape[0] and t.shape[2:] == x0.shape[2:]):
            return False
        B, C_, H, W = t.shape
        sb, sc, sh, sw = t.stride()
        p = sw if W > 1 else (sh if H > 1 else (sb if B > 1 else C_))
        if not ((C_ == 1 or sc == 1) and p >= C_ and (H == 1 or sh == p * W) and (B == 1 or sb == p * W * H)):
            return False
    return True


def concat_channels(parts, cout=None):
    """`torch.cat(parts, 1)` zero-padded to `cout` channels (a multiple of 8; default: the sum rounded up to 8) as one HIP launch into
    a dense channels_last tensor; gradients are channel slices of the incoming gradient (views, as autograd's own cat backward)"""
    parts = list(parts)
    total = sum(int(t.shape[1]) for t in parts)
    cout = (total + 7) // 8 * 8 if cout is None else int(cout)
    if cout % 8 or cout < total:
        raise ValueError(f"cout must be a multiple of 8 and >= {total}, got {cout}")
    _require_hip(*parts)
    return _ConcatFn.apply(cout, *parts)


class _UnpadChannelsFn(Function):
    """x[:, :c] of a zero-padded channels_last tensor as a DENSE channels_last tensor (one copy); backward: the incoming gradient
    zero-padded back to x's channel count by one ocf_concat_channels launch (autograd's own slice backward is a fill of an
    NCHW-contiguous tensor plus a strided copy, which the consumer then has to re-layout)"""

    @staticmethod
    def forward(ctx, x, c):
        ctx.cp = int(x.shape[1])
        return x[:, :c].contiguous(memory_format=torch.channels_last)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        if gy.dtype == torch.bfloat16 and ctx.cp % 8 == 0 and concat_channels_eligible([gy], ctx.cp):
            return _ConcatFn.apply(ctx.cp, gy), None
        gx = torch.zeros((gy.shape[0], ctx.cp, gy.shape[2], gy.shape[3]), device=gy.device, dtype=gy.dtype).contiguous(memory_format=torch.channels_last)
        gx[:, :gy.shape[1]] = gy
        return gx, None


def unpad_channels(x, c: int):
    """the first `c` channels of `x` as a dense channels_last tensor (the inverse of `concat_channels([y], cpad)`)"""
    return x if x.shape[1] == c else _UnpadChannelsFn.apply(x, int(c))


class _HeadLossFn(Function):
    """(mean L1 of the x4 up-sampled flow, mean BCE of the x4 up-sampled occlusion probability) and their gradients w.r.t. the
    quarter-resolution tensors: ocf_flow_occ_head_fwd / _bwd (csrc/head_loss.hip)"""

    @staticmethod
    def forward(ctx, flow_lr, occ_lr, flow_gt, occ_gt):
        B, _, h, w = flow_lr.shape
        dev = flow_lr.device
        sums = torch.empty(2, device=dev, dtype=torch.float64)
        losses = torch.empty(2, device=dev, dtype=torch.float32)
        need = ctx.needs_input_grad[0] or (occ_lr is not None and ctx.needs_input_grad[1])
        part = torch.empty((B, h, w, 27), device=dev, dtype=torch.float32) if need else None
        with torch.cuda.device(dev):
            _lib.check(_lib.load().ocf_flow_occ_head_fwd(_t4(flow_lr), _t4(occ_lr), _t4(flow_gt), _t4(occ_gt), _ptr(sums), _ptr(losses), _ptr(part),
                                                         B, h, w, _dtype_code(flow_lr), _dtype_code(occ_lr) if occ_lr is not None else 0,
                                                         _stream(flow_lr)), "ocf_flow_occ_head_fwd")
        ctx.save_for_backward(part, flow_lr, occ_lr)  # (the low-resolution tensors only for their shapes / strides / dtypes)
        return losses[0], losses[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_flow, g_occ):
        part, flow_lr, occ_lr = ctx.saved_tensors
        B, _, h, w = flow_lr.shape
        gflow = torch.empty_like(flow_lr) if ctx.needs_input_grad[0] else None
        gocc = torch.empty_like(occ_lr) if (occ_lr is not None and ctx.needs_input_grad[1]) else None
        scal = lambda g: None if g is None else g.to(torch.float32).contiguous()
        g_flow, g_occ = scal(g_flow), scal(g_occ)
        with torch.cuda.device(flow_lr.device):
            _lib.check(_lib.load().ocf_flow_occ_head_bwd(_ptr(part), _ptr(g_flow), _ptr(g_occ), _t4(gflow), _t4(gocc), B, h, w,
                                                         _dtype_code(flow_lr), _dtype_code(occ_lr) if occ_lr is not None else 0,
                                                         _stream(flow_lr)), "ocf_flow_occ_head_bwd")
        return gflow, gocc, None, None


def flow_occ_head_eligible(flow_lr, occ_lr, flow_gt, occ_gt) -> bool:
    """quarter-resolution predictions on a HIP device, fp32 ground truth at exactly four times their size"""
    if os.environ.get("OCFLOW_FUSED_HEAD", "1") == "0" or torch.compiler.is_compiling():
        return False
    ts = (flow_lr, occ_lr, flow_gt, occ_gt)
    if not all(torch.is_tensor(t) and t.is_cuda and t.dim() == 4 for t in ts):
        return False
    B, _, h, w = flow_lr.shape
    return (flow_lr.shape[1] == 2 and tuple(occ_lr.shape) == (B, 1, h, w) and tuple(flow_gt.shape) == (B, 2, 4 * h, 4 * w)
            and tuple(occ_gt.shape) == (B, 1, 4 * h, 4 * w) and flow_gt.dtype == torch.float32 and occ_gt.dtype == torch.float32
            and flow_lr.dtype in (torch.float32, torch.bfloat16) and occ_lr.dtype in (torch.float32, torch.bfloat16)
            and not flow_gt.requires_grad and not occ_gt.requires_grad)


def flow_occ_head_loss(flow_lr, occ_lr, flow_gt, occ_gt):
    """`(F.l1_loss(up4(flow_lr).float(), flow_gt), F.binary_cross_entropy(up4(occ_lr).float(), occ_gt))` with
    `up4 = nn.Upsample(scale_factor=4, mode="bilinear")`, in two launches forward + one backward"""
    _require_hip(flow_lr, occ_lr, flow_gt, occ_gt)
    return _HeadLossFn.apply(flow_lr, occ_lr, flow_gt, occ_gt)


def resize_bilinear(x, size, align_corners: bool = True):
    """`F.interpolate(x, size=size, mode="bilinear", align_corners=align_corners)` for an fp32 device tensor that needs no gradient, any
    strides (ocf_resize_bilinear: one launch, one thread per output element)"""
    _check4(x, "x")
    _require_hip(x)
    if x.dtype != torch.float32 or x.requires_grad:
        raise ValueError("resize_bilinear: fp32 tensors without gradient only (use F.interpolate)")
    B, C_, H, W = x.shape
    h, w = int(size[0]), int(size[1])
    y = torch.empty((B, C_, h, w), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().ocf_resize_bilinear(_t4(x), _t4(y), B, C_, H, W, h, w, int(bool(align_corners)), _stream(x)), "ocf_resize_bilinear")
    return y


def pair_input_eligible(x1, x2) -> bool:
    """x1, x2 = channels 0-2 / 3-5 of ONE fp32 [B, 6, H, W] device tensor that needs no gradient (the reference's `inputs[:, 0:3]`,
    `inputs[:, 3:6]`), with bf16 autocast active -- the case `pair_input` serves"""
    if not (torch.is_tensor(x1) and torch.is_tensor(x2) and x1.is_cuda and x1.dtype == torch.float32 and x2.dtype == torch.float32):
        return False
    if x1.dim() != 4 or x1.shape != x2.shape or x1.shape[1] != 3 or x1.stride() != x2.stride() or x1.requires_grad or x2.requires_grad:
        return False
    if not (torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16):
        return False
    return x2.data_ptr() - x1.data_ptr() == 3 * x1.stride(1) * 4 and os.environ.get("OCFLOW_PAIR_INPUT", "1") != "0"


def pair_input(x1, x2):
    """`torch.cat((x1, x2), 0).to(bfloat16, channels_last)` for the two halves of a 6-channel pair tensor, one launch (ocf_pair_input)"""
    B, _, H, W = x1.shape
    out = torch.empty((2 * B, 3, H, W), device=x1.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    six = x1.as_strided((B, 6, H, W), x1.stride())  # (the parent's channels 0-5: x2 starts three channel strides after x1)
    with torch.cuda.device(x1.device):
        _lib.check(_lib.load().ocf_pair_input(_t4(six), C.c_void_p(out.data_ptr()), B, H, W, _stream(x1)), "ocf_pair_input")
    return out


# ------------------------------------------------------------------------- 3x3 convolution, weight gradient
def conv3x3_wgrad(x, gy, dilation=1, out=None):
    """Weight gradient of a 3x3 / stride-1 / padding == dilation convolution (the estimator / context network layers: flow_net.py:11-37,
    flow_occ_net.py:11-67, context_net.py:11-17) on the matrix cores: x [B,Ci,H,W], gy [B,Co,H,W] dense channels_last bf16 ->
    fp32 [Co,Ci,3,3] in channels_last weight strides.  `out`: an fp32 tensor of that shape to ACCUMULATE into (split-K float atomics);
    None allocates a zeroed one.  Raises for operands the kernel does not take (the callers check `conv3x3_wgrad_eligible` first)."""
    _require_hip(x, gy)
    B, Ci, H, W = x.shape
    Co = gy.shape[1]
    if out is None:
        out = torch.zeros((Co, Ci, 3, 3), device=x.device, dtype=torch.float32).contiguous(memory_format=torch.channels_last)
    so, si, sky, skx = out.stride()
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().ocf_conv3x3_wgrad(_t4(x), _t4(gy), C.c_void_p(out.data_ptr()), so, si, sky, skx, B, Ci, Co, H, W, int(dilation),
                                                 _dtype_code(x), POLICY.wgrad, _stream(x)), "ocf_conv3x3_wgrad")
    return out


def conv3x3_wgrad_eligible(x, weight, stride, padding, dilation, groups) -> bool:
    """operands ocf_conv3x3_wgrad takes: bf16, dense channels_last activations, 3x3, stride 1, padding == dilation <= 16, one group,
    channel counts that are multiples of 8 (16-byte pixels)"""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.dim() == 4):
        return False
    if tuple(weight.shape[2:]) != (3, 3) or groups != 1 or tuple(stride) != (1, 1):
        return False
    d = tuple(dilation)
    if d[0] != d[1] or tuple(padding) != d or d[0] > 16:
        return False
    Ci, Co = weight.shape[1], weight.shape[0]
    return Ci % 8 == 0 and Co % 8 == 0 and x.is_contiguous(memory_format=torch.channels_last) and x.size(1) > 1


class _Conv3x3Fn(Function):
    """`F.conv2d(x, w, None, 1, d, d)` whose backward takes the INPUT gradient from the framework (MIOpen) and the WEIGHT gradient from
    ocf_conv3x3_wgrad: one launch per layer instead of MIOpen's workspace fill + split-K kernel + cast."""

    @staticmethod
    def forward(ctx, x, w, d):
        ctx.save_for_backward(x, w)
        ctx.d = d
        return torch.ops.aten.convolution(x, w, None, (1, 1), (d, d), (d, d), False, (0, 0), 1)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        d = ctx.d
        gx = gw = None
        if not gy.is_contiguous(memory_format=torch.channels_last):
            gy = gy.contiguous(memory_format=torch.channels_last)
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (d, d), (d, d), False, (0, 0), 1, [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            acc = _take_zeros(w.numel(), gy.device).as_strided(w.shape, w.stride())
            gw = conv3x3_wgrad(x, gy.to(x.dtype), d, out=acc)
        return gx, gw, None


def conv3x3(x, weight, dilation=1):
    """3x3 / stride-1 / same convolution without bias (the epilogue kernels add it) with the hand-written weight gradient"""
    return _Conv3x3Fn.apply(x, weight, int(dilation))


# ---------------------------------------------------------------------------------- convolution epilogue
_ZERO_POOL = {}  # device index -> [fp32 buffer of zeros, next free element, graph-capture id it was filled in (0 = eager)]
_ZERO_POOL_ELEMS = 1 << 22  # 16 MB: bias gradients and (ocf_conv3x3_wgrad) whole fp32 weight gradients of a step
_CAPTURE_ID = C.c_ulonglong(0)


def _capture_id(device) -> int:
    """identifier of the graph capture the current stream of `device` is recording into (0: not capturing)"""
    _lib.check(_lib.load().ocf_stream_capture_id(C.c_void_p(torch.cuda.current_stream(device).cuda_stream), C.byref(_CAPTURE_ID)),
               "ocf_stream_capture_id")
    return _CAPTURE_ID.value


def _take_zeros(n: int, device) -> torch.Tensor:
    """n zero-initialised fp32 elements for a kernel that accumulates into them: slices of one 256 KB buffer zeroed by a single
    fill, instead of one memset per layer (~60 per step, 4 us each).  A slice is handed out once; the buffer is replaced when
    it runs out (the slices keep the old one alive for as long as the gradients they became are referenced)."""
    n_pad = (n + 3) & ~3
    # a buffer zeroed outside the running graph capture (or in an earlier one) would not be re-zeroed by this graph's replays
    _lib.check(_lib.load().ocf_stream_capture_id(C.c_void_p(torch.cuda.current_stream(device).cuda_stream), C.byref(_CAPTURE_ID)),
               "ocf_stream_capture_id")
    cid = _CAPTURE_ID.value
    entry = _ZERO_POOL.get(device.index)
    if entry is None or entry[2] != cid or entry[1] + n_pad > entry[0].numel():
        entry = [torch.zeros(max(_ZERO_POOL_ELEMS, n_pad), device=device, dtype=torch.float32), 0, cid]
        _ZERO_POOL[device.index] = entry
    out = entry[0][entry[1]:entry[1] + n]
    entry[1] += n_pad
    return out


class _BiasActFn(Function):
    @staticmethod
    def forward(ctx, x, bias, negative_slope):
        B, C_, H, W = x.shape
        b32 = _f32(bias.detach()).contiguous()
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().ocf_bias_act_fwd(_t4(x), C.c_void_p(b32.data_ptr()), B, C_, H, W, float(negative_slope),
                                                    _dtype_code(x), _stream(x)), "ocf_bias_act_fwd")
        ctx.mark_dirty(x)
        ctx.slope = float(negative_slope)
        ctx.bias_dtype = bias.dtype
        if ctx.slope != 1.0:
            ctx.save_for_backward(x)
        else:
            ctx.fmt = torch.channels_last if _is_cl(x) else torch.contiguous_format
        return x

    @staticmethod
    def backward(ctx, gy):
        B, C_, H, W = gy.shape
        passthrough = ctx.slope == 1.0

        def pixel_major(t):
            """channel-contiguous with pixels in dense (b, y, x) order at ANY pixel pitch: dense channels_last, or a channel
            slice of the gradient of a concatenation (the decoders' dense blocks) -- the kernel reads either without a copy"""
            sb, sc, sh, sw = t.stride()
            pitch = sw if W > 1 else (sh if H > 1 else (sb if B > 1 else C_))  # size-1 dimensions have arbitrary strides
            return ((sc == 1 or C_ == 1) and pitch >= C_ and (sh == W * pitch or H == 1 or W == 1)
                    and (sb == H * W * pitch or B == 1 or (H == 1 and W == 1)))

        densify = False
        if passthrough:
            y, gg = None, gy
            if not (gg.is_contiguous() or pixel_major(gg)):
                gg = gg.contiguous(memory_format=ctx.fmt)  # reduction operand only; the gradient returned stays `gy`
            elif ctx.fmt == torch.channels_last and not gg.is_contiguous(memory_format=torch.channels_last) and gy.dim() == 4 and C_ > 1 \
                    and os.environ.get("OCFLOW_NO_DENSIFY", "0") != "1":
                # a channel slice of a concatenation's gradient (the up-sampling layers feed torch.cat): the consumer of the returned
                # gradient -- the transposed convolution's backward -- would copy it to a dense tensor first; the reduction pass
                # reads every element anyway, so it writes that dense copy itself (slope 1: gx = gy) and one launch per site is saved
                densify = True
        else:
            (y,) = ctx.saved_tensors
            gg = gy if gy.dtype == y.dtype else gy.to(y.dtype)
            if gg.stride() != y.stride() and not (_is_cl(y) and pixel_major(gg)):
                gg = gg.contiguous(memory_format=torch.channels_last if _is_cl(y) else torch.contiguous_format)
                if gg.stride() != y.stride():
                    gg = torch.empty_like(y).copy_(gg)
        if densify:
            gx = torch.empty(gy.shape, dtype=gy.dtype, device=gy.device, memory_format=torch.channels_last)
            y = gx  # (the kernel only looks at the sign of y to pick gy or gy * slope: with slope 1 both are gy)
        else:
            gx = None if passthrough else torch.empty_like(y)
        gb = _take_zeros(C_, gy.device) if ctx.needs_input_grad[1] else None
        if gb is not None or not passthrough or densify:
            if gb is None:
                gb = _take_zeros(C_, gy.device)
            with torch.cuda.device(gy.device):
                st = _lib.load().ocf_bias_act_bwd(_t4(y), _t4(gg), _t4(gx), C.c_void_p(gb.data_ptr()), B, C_, H, W, ctx.slope, 1,
                                                  _dtype_code(gg), _stream(gg))
            if densify and st == _lib.OCF_EUNSUPPORTED:  # a layout the copying form does not take: the pass-through form as before
                densify, gx = False, None
                with torch.cuda.device(gy.device):
                    st = _lib.load().ocf_bias_act_bwd(_t4(None), _t4(gg), _t4(None), C.c_void_p(gb.data_ptr()), B, C_, H, W, ctx.slope, 1,
                                                      _dtype_code(gg), _stream(gg))
            _lib.check(st, "ocf_bias_act_bwd")
        if gb is not None and ctx.bias_dtype != torch.float32:
            gb = gb.to(ctx.bias_dtype)
        return (gx if (densify or not passthrough) else gy), (gb if ctx.needs_input_grad[1] else None), None


def bias_leaky_relu(x, bias, negative_slope=0.1):
    """`F.leaky_relu(x + bias[None, :, None, None], negative_slope)` for the bias-free output `x` of a convolution, IN PLACE
    on `x` -- the epilogue of the reference's `conv()` blocks (nn.Conv2d(bias=True) + nn.LeakyReLU(0.1):
    cost_volume_flow_net.py:19-21, simple_flow_net.py:55-57, ...).  One pass forward instead of add_ + leaky_relu, one pass
    backward instead of leaky_relu_backward + sum((0,2,3)) + dtype casts; forward values are bit-identical to the unfused
    chain in fp32 and under bf16 autocast (bias rounded to bf16, sum rounded, product rounded).  `x` must be a dense
    channels_last or NCHW tensor this call may overwrite (a fresh convolution output)."""
    _check4(x, "x")
    if bias.dim() != 1 or bias.numel() != x.size(1):
        raise ValueError(f"bias must have {x.size(1)} elements, got {tuple(bias.shape)}")
    _require_hip(x, bias)
    _dtype_code(x)
    return _BiasActFn.apply(x, bias, float(negative_slope))


_BN_WS = {}  # device -> scratch of the batch-norm kernels (per-block partial sums between their two passes)


def _bn_workspace(C_, device, groups=1):
    n = C.c_int64(0)
    _lib.check(_lib.load().ocf_bn_act_workspace_bytes(int(C_), int(groups), C.byref(n)), "ocf_bn_act_workspace_bytes")
    ws = _BN_WS.get(device)
    if ws is None or ws.numel() < n.value:
        ws = torch.empty(max(n.value, 1 << 20), device=device, dtype=torch.uint8)
        _BN_WS[device] = ws
    return ws


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# ---- replaying the running-statistics update of a BatchNorm forward WITHOUT running the forward again.  The occlusion-aware training step
# calls the network a second time on the swapped pair without gradient (models/model.py:384); for the FPN nets that second encoder pass
# computes exactly the features of the first one (per-image statistics groups), so networks/flow_net.py re-uses them -- and what the second
# pass would have done to the BatchNorm buffers (momentum update with the same batch statistics, groups in swapped order;
# num_batches_tracked) is applied from the statistics the first pass saved.
_BN_RECORD = None  # list of (running_mean, running_var, num_batches_tracked, mean [G,C], invstd [G,C], rows per group, momentum, eps) while recording


class bn_record_stats:
    """`with bn_record_stats() as rec:` -- every training-mode ocf_bn_act_fwd inside notes its buffers and saved statistics in `rec`"""

    def __enter__(self):
        global _BN_RECORD
        self.prev, _BN_RECORD = _BN_RECORD, []
        return _BN_RECORD

    def __exit__(self, *exc):
        global _BN_RECORD
        _BN_RECORD = self.prev
        return False


def bn_replay_running_update(records, order):
    """apply to the recorded BatchNorm buffers what one more forward over the same batch would: for g in `order` (statistics groups),
    running = (1 - momentum) * running + momentum * stat_g (variance unbiased by n / (n - 1)), num_batches_tracked += len(order).
    A few multi-tensor launches for the whole network."""
    if not records:
        return
    with torch.no_grad():
        by_mom = {}
        for r in records:
            by_mom.setdefault(r[6], []).append(r)
        for mom, recs in by_mom.items():
            for g in order:
                rms, rvs, means, vars_ = [], [], [], []
                for rm, rv, nbt, mean, invstd, n, _mo, eps in recs:
                    rms.append(rm)
                    rvs.append(rv)
                    means.append(mean[g])
                    unb = n / (n - 1.0) if n > 1 else 1.0
                    vars_.append(((invstd[g].double().pow(-2) - eps) * unb).float())
                torch._foreach_lerp_(rms, means, mom)
                torch._foreach_lerp_(rvs, vars_, mom)
        nbts = [r[2] for r in records if r[2] is not None]
        if nbts:
            torch._foreach_add_(nbts, len(order))


class _BnActFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, momentum, eps, slope, groups):
        B, C_, H, W = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((groups, C_), device=x.device, dtype=torch.float32)
        invstd = torch.empty((groups, C_), device=x.device, dtype=torch.float32)
        ws = _bn_workspace(C_, x.device, groups)
        with torch.cuda.device(x.device):
            st = _lib.load().ocf_bn_act_fwd(_t4(x), _t4(y), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var), _ptr(nbt),
                                            _ptr(mean), _ptr(invstd), _ptr(ws), B, C_, H, W, float(eps), float(momentum), float(slope), 1,
                                            int(groups), _dtype_code(x), _stream(x))
        _lib.check(st, "ocf_bn_act_fwd")
        if _BN_RECORD is not None:
            _BN_RECORD.append((running_mean, running_var, nbt, mean, invstd, (B // groups) * H * W, float(momentum), float(eps)))
        # the backward recomputes the LeakyReLU mask from x (ocf_bn_act_bwd_x): y is not kept for it.  OCFLOW_BN_BWD_Y=1: the first form,
        # which reads the saved output (A/B; same results)
        ctx.with_y = os.environ.get("OCFLOW_BN_BWD_Y", "0") == "1"
        ctx.save_for_backward(x, y if ctx.with_y else None, weight, bias, mean, invstd)
        ctx.slope = float(slope)
        ctx.groups = int(groups)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, y, weight, bias, mean, invstd = ctx.saved_tensors
        B, C_, H, W = x.shape
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        if gy.stride() != x.stride():
            gy = torch.empty_like(x).copy_(gy)
        gx = torch.empty_like(x)
        dgamma = torch.empty(C_, device=x.device, dtype=torch.float32)
        dbeta = torch.empty(C_, device=x.device, dtype=torch.float32)
        ws = _bn_workspace(C_, x.device, ctx.groups)
        with torch.cuda.device(x.device):
            if ctx.with_y:
                _lib.check(_lib.load().ocf_bn_act_bwd(_t4(x), _t4(y), _t4(gy), _t4(gx), _ptr(weight), _ptr(mean), _ptr(invstd), _ptr(dgamma),
                                                      _ptr(dbeta), _ptr(ws), B, C_, H, W, ctx.slope, ctx.groups, _dtype_code(x), _stream(x)),
                           "ocf_bn_act_bwd")
            else:
                _lib.check(_lib.load().ocf_bn_act_bwd_x(_t4(x), _t4(gy), _t4(gx), _ptr(weight), _ptr(bias), _ptr(mean), _ptr(invstd),
                                                        _ptr(dgamma), _ptr(dbeta), _ptr(ws), B, C_, H, W, ctx.slope, ctx.groups,
                                                        _dtype_code(x), _stream(x)), "ocf_bn_act_bwd_x")
        return gx, dgamma, dbeta, None, None, None, None, None, None, None


def batchnorm_lrelu_supported(x, bn) -> bool:
    """does `batchnorm_lrelu` take this call with the HIP kernels?  (dense channels_last fp32 / bf16 on the GPU, affine fp32 parameters,
    tracked running statistics with a fixed momentum, C a multiple of 4; training mode, or evaluation without a gradient)"""
    if not (torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    if x.shape[1] % 4 != 0 or x.numel() == 0 or not x.is_contiguous(memory_format=torch.channels_last) or x.shape[1] == 1:
        return False
    if bn.weight is None or bn.bias is None or bn.weight.dtype != torch.float32 or bn.bias.dtype != torch.float32:
        return False
    if not bn.track_running_stats or bn.running_mean is None or bn.momentum is None or bn.running_mean.dtype != torch.float32:
        return False
    if not bn.training and (x.requires_grad or bn.weight.requires_grad) and torch.is_grad_enabled():
        return False
    return x.shape[0] * x.shape[2] * x.shape[3] > 1 or not bn.training


def batchnorm_lrelu(x, bn, negative_slope=0.1, groups=1):
    """`F.leaky_relu(bn(x), negative_slope)` for an nn.BatchNorm2d `bn` (groups=2: of `bn` applied to the two halves of the batch in turn) -- the reference's conv -> BatchNorm2d -> LeakyReLU(0.1) blocks
    of the feature pyramid (models/networks/feature_pyramid_net.py:5-95).  Two HIP passes per direction instead of the framework's nine
    launches per site (statistics with a last-block fp64 reduction, normalise + activation; the running statistics and
    num_batches_tracked are updated by the statistics pass).  Same module, same parameters and buffers; layouts / modes the kernels do
    not take (NCHW, evaluation with gradients, momentum=None ...) go through the framework's own operators."""
    if groups not in (1, 2) or x.shape[0] % groups:
        raise ValueError(f"groups must be 1 or 2 and divide the batch, got {groups} for a batch of {x.shape[0]}")
    if not batchnorm_lrelu_supported(x, bn) or (groups == 2 and not bn.training):
        if groups == 1:
            return torch.nn.functional.leaky_relu(bn(x), negative_slope)
        # `groups` consecutive parts of the batch, each normalised by its own statistics: the module called on each part in turn
        return torch.cat([torch.nn.functional.leaky_relu(bn(part), negative_slope) for part in x.chunk(groups, 0)], 0)
    _require_hip(x)
    if bn.training:
        return _BnActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.momentum), float(bn.eps),
                              float(negative_slope), int(groups))
    B, C_, H, W = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().ocf_bn_act_fwd(_t4(x), _t4(y), _ptr(bn.weight), _ptr(bn.bias), _ptr(bn.running_mean), _ptr(bn.running_var), None,
                                              None, None, None, B, C_, H, W, float(bn.eps), 0.0, float(negative_slope), 0, 1, _dtype_code(x),
                                              _stream(x)), "ocf_bn_act_fwd")
    return y


def bias_add(x, bias):
    """`x + bias[None, :, None, None]` IN PLACE on the bias-free output of an nn.Conv2d / nn.ConvTranspose2d without activation
    (prediction layers, up-convolutions: flow_net.py:38-40, flow_occ_net.py:32-34, feature_pyramid_net.py:31).  Backward is one
    reduction of the incoming gradient (which passes through untouched) instead of sum((0,2,3)) + two dtype casts."""
    return bias_leaky_relu(x, bias, 1.0)
