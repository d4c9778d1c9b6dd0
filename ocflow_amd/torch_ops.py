"""torch.library registration of the hot-path operators:

    torch.ops.ocflow.corr_warp(f1, f2, flow?, occ?, d, flow_scale, align_corners) -> (cost volume, warped f2 kept for backward)
                                                   reference: flow_net.py:110-112, flow_occ_net.py:167-168, correlation_layer.py:7-40
    torch.ops.ocflow.cost_volume(f1, f2, d)        reference: compute_cost_volume, correlation_layer.py:7-40
    torch.ops.ocflow.warp(img, flow, flow_scale, align_corners)
                                                   reference: warp() of flow_net.py:57-87 (AC False), models/model.py:191-221 (AC True)
    torch.ops.ocflow.photo_loss(img1, img2, flow?, occ?) -> fp32 [2] (sum e (1 - occ), sum (1 - occ))
                                                   reference: FlowStageModel.warp + photometric_error, models/model.py:191-221, 27-46
    torch.ops.ocflow.smooth_loss(img, flow, alpha) -> fp32 [4]    reference: models/model.py:53-66, 93-114
    torch.ops.ocflow.range_map(flow, want_occ) -> (range, occ)    reference: models/model.py:223-305, 391

The kernels stay behind the C ABI and `ctypes` (ocflow_amd/ops.py); this module gives them dispatcher identities -- schema, fake
(meta) implementations for shape propagation, autograd formulas as registered backward OPERATORS whose arguments are exactly the saved
tensors (stateless: no forward is re-run) -- so that they pass `torch.library.opcheck`, trace under `torch.compile` / `make_fx` without
graph breaks and show up by name in profiler traces.  The reference-named entry points of `ocflow_amd.ops` call these identities
whenever they run under a tracer (`torch.compiler.is_compiling()`), and the ctypes launchers directly in eager mode: same launches.
"""
from typing import Optional, Tuple

import torch
from torch.library import custom_op

from . import ops


def _cl(t: torch.Tensor) -> bool:
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()


def _empty_cv(f1: torch.Tensor, d: int) -> torch.Tensor:
    n = (2 * d + 1) ** 2
    return torch.empty((f1.shape[0], n, f1.shape[2], f1.shape[3]), dtype=f1.dtype, device=f1.device,
                       memory_format=torch.channels_last if _cl(f1) else torch.contiguous_format)


def _none_if_empty(t: torch.Tensor) -> Optional[torch.Tensor]:
    return t if t.numel() else None


# ------------------------------------------------------------------------------------------------------------- corr_warp (K1 + K2)
@custom_op("ocflow::corr_warp", mutates_args=())
def corr_warp(f1: torch.Tensor, f2: torch.Tensor, flow: Optional[torch.Tensor], occ: Optional[torch.Tensor], max_displacement: int,
              flow_scale: float, align_corners: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    # the warped operand is always produced when there is a flow: the op cannot know whether a backward will follow, and an empty
    # second result (no flow) keeps the schema fixed
    out, w2 = ops._corr_warp_fwd_raw(f1, f2, flow, occ, max_displacement, flow_scale, align_corners, None, flow is not None)
    return out, (w2 if w2 is not None else f1.new_empty(0))


@corr_warp.register_fake
def _(f1, f2, flow, occ, max_displacement, flow_scale, align_corners):
    return _empty_cv(f1, max_displacement), (torch.empty_like(f2) if flow is not None else f1.new_empty(0))


@custom_op("ocflow::corr_warp_backward", mutates_args=())
def corr_warp_backward(f1: torch.Tensor, f2: torch.Tensor, flow: Optional[torch.Tensor], occ: Optional[torch.Tensor], warped: torch.Tensor,
                       grad_out: torch.Tensor, max_displacement: int, flow_scale: float, align_corners: bool, need_f1: bool, need_f2: bool,
                       need_flow: bool, need_occ: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    g1, g2, gflow, gocc = ops._corr_warp_bwd_raw(f1, f2, flow, occ, _none_if_empty(warped), grad_out, max_displacement, flow_scale, align_corners,
                                                 (need_f1, need_f2, need_flow and flow is not None, need_occ and occ is not None),
                                                 ops.POLICY.corr_bwd, ops.POLICY.warp_bwd)
    e = lambda dt: torch.empty(0, dtype=dt, device=f1.device)  # (a fresh tensor per slot: results of a custom operator may not alias)
    return (g1 if g1 is not None else e(f1.dtype), g2 if g2 is not None else e(f1.dtype), gflow if gflow is not None else e(torch.float32),
            gocc if gocc is not None else e(torch.float32))


@corr_warp_backward.register_fake
def _(f1, f2, flow, occ, warped, grad_out, max_displacement, flow_scale, align_corners, need_f1, need_f2, need_flow, need_occ):
    e = lambda dt: torch.empty(0, dtype=dt, device=f1.device)
    return (torch.empty_like(f1) if need_f1 else e(f1.dtype), torch.empty_like(f2) if need_f2 else e(f1.dtype),
            torch.empty_like(flow) if (need_flow and flow is not None) else e(torch.float32),
            torch.empty_like(occ) if (need_occ and occ is not None) else e(torch.float32))


def _cw_setup(ctx, inputs, output):
    f1, f2, flow, occ, d, fs, ac = inputs
    ctx.save_for_backward(f1, f2, flow, occ, output[1])
    ctx.cfg = (d, fs, ac)
    ctx.set_materialize_grads(False)


def _cw_backward(ctx, grad_out, _grad_warped):
    f1, f2, flow, occ, warped = ctx.saved_tensors
    if grad_out is None:
        return (None,) * 7
    n1, n2, nf, no = ctx.needs_input_grad[:4]
    g1, g2, gflow, gocc = torch.ops.ocflow.corr_warp_backward(f1, f2, flow, occ, warped, grad_out, *ctx.cfg, n1, n2, nf, no)
    return (g1 if n1 else None, g2 if n2 else None, gflow if (nf and flow is not None) else None,
            gocc if (no and occ is not None) else None, None, None, None)


corr_warp.register_autograd(_cw_backward, setup_context=_cw_setup)


# ------------------------------------------------------------------------------------------------------------- cost_volume (K1)
@custom_op("ocflow::cost_volume", mutates_args=())
def cost_volume(f1: torch.Tensor, f2: torch.Tensor, max_displacement: int) -> torch.Tensor:
    return ops._corr_warp_fwd_raw(f1, f2, None, None, max_displacement, 1.0, False, None, False)[0]


@cost_volume.register_fake
def _(f1, f2, max_displacement):
    return _empty_cv(f1, max_displacement)


@custom_op("ocflow::cost_volume_backward", mutates_args=())
def cost_volume_backward(f1: torch.Tensor, f2: torch.Tensor, grad_out: torch.Tensor, max_displacement: int) -> Tuple[torch.Tensor, torch.Tensor]:
    # exactly the saved tensors and the cotangent: ONE backward launch sequence (ocf_corr_bwd), nothing is recomputed
    g1, g2, _, _ = ops._corr_warp_bwd_raw(f1, f2, None, None, None, grad_out, max_displacement, 1.0, False, (True, True, False, False),
                                          ops.POLICY.corr_bwd, ops.POLICY.warp_bwd)
    return g1, g2


@cost_volume_backward.register_fake
def _(f1, f2, grad_out, max_displacement):
    return torch.empty_like(f1), torch.empty_like(f2)


def _cv_setup(ctx, inputs, output):
    f1, f2, d = inputs
    ctx.save_for_backward(f1, f2)
    ctx.d = d


def _cv_backward(ctx, grad_out):
    f1, f2 = ctx.saved_tensors
    g1, g2 = torch.ops.ocflow.cost_volume_backward(f1, f2, grad_out, ctx.d)
    return g1, g2, None


cost_volume.register_autograd(_cv_backward, setup_context=_cv_setup)


# ------------------------------------------------------------------------------------------------------------- warp (K2)
@custom_op("ocflow::warp", mutates_args=())
def warp(img: torch.Tensor, flow: torch.Tensor, flow_scale: float, align_corners: bool) -> torch.Tensor:
    return ops._warp_fwd_raw(img.detach(), flow.detach().float(), flow_scale, align_corners)


@warp.register_fake
def _(img, flow, flow_scale, align_corners):
    return torch.empty_like(img)


@custom_op("ocflow::warp_backward", mutates_args=())
def warp_backward(img: torch.Tensor, flow: torch.Tensor, grad_out: torch.Tensor, flow_scale: float,
                  align_corners: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    gimg, gflow = ops._warp_bwd_raw(img.detach(), flow.detach().float(), grad_out, flow_scale, align_corners, True, True)
    return gimg, gflow.to(flow.dtype)


@warp_backward.register_fake
def _(img, flow, grad_out, flow_scale, align_corners):
    return torch.empty_like(img), torch.empty_like(flow)


def _warp_setup(ctx, inputs, output):
    img, flow, fs, ac = inputs
    ctx.save_for_backward(img, flow)
    ctx.cfg = (fs, ac)


def _warp_backward(ctx, grad_out):
    img, flow = ctx.saved_tensors
    gimg, gflow = torch.ops.ocflow.warp_backward(img, flow, grad_out.contiguous(), *ctx.cfg)
    return gimg, gflow, None, None


warp.register_autograd(_warp_backward, setup_context=_warp_setup)


# ------------------------------------------------------------------------------------------------------------- photo_loss (K4)
@custom_op("ocflow::photo_loss", mutates_args=())
def photo_loss(img1: torch.Tensor, img2: torch.Tensor, flow: Optional[torch.Tensor], occ: Optional[torch.Tensor]) -> torch.Tensor:
    return ops._photo_fwd_raw(img1, img2, flow, occ).float()


@photo_loss.register_fake
def _(img1, img2, flow, occ):
    return img1.new_empty((2,), dtype=torch.float32)


@custom_op("ocflow::photo_loss_backward", mutates_args=())
def photo_loss_backward(img1: torch.Tensor, img2: torch.Tensor, flow: Optional[torch.Tensor], occ: Optional[torch.Tensor],
                        grad_num: torch.Tensor) -> torch.Tensor:
    """grad_flow (with a flow) or dLoss/d(prediction) = dLoss/d img2 (without)"""
    gflow, gpred = ops._photo_bwd_raw(img1, img2, flow, occ, grad_num, True, True)
    return gflow if flow is not None else gpred


@photo_loss_backward.register_fake
def _(img1, img2, flow, occ, grad_num):
    return torch.empty_like(flow) if flow is not None else torch.empty_like(img2)


def _photo_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _photo_backward(ctx, gsums):
    img1, img2, flow, occ = ctx.saved_tensors
    n1, n2, nf, no = ctx.needs_input_grad
    if no:
        raise NotImplementedError("ocflow::photo_loss: gradient w.r.t. the occlusion mask is not implemented (the reference computes the "
                                  "mask under no_grad, models/model.py:381-391)")
    if flow is not None and (n1 or n2):
        raise NotImplementedError("ocflow::photo_loss: image gradients through the fused warp are not implemented; use ocflow::warp first")
    g = torch.ops.ocflow.photo_loss_backward(img1, img2, flow, occ, gsums[0])
    if flow is not None:
        return None, None, (g if nf else None), None
    return (-g if n1 else None), (g if n2 else None), None, None


photo_loss.register_autograd(_photo_backward, setup_context=_photo_setup)


# ------------------------------------------------------------------------------------------------------------- smooth_loss (K5)
@custom_op("ocflow::smooth_loss", mutates_args=())
def smooth_loss(img: torch.Tensor, flow: torch.Tensor, alpha: float) -> torch.Tensor:
    return ops._smooth_fwd_raw(img, flow, alpha).float()


@smooth_loss.register_fake
def _(img, flow, alpha):
    return img.new_empty((4,), dtype=torch.float32)


@custom_op("ocflow::smooth_loss_backward", mutates_args=())
def smooth_loss_backward(img: torch.Tensor, flow: torch.Tensor, coef: torch.Tensor, alpha: float) -> torch.Tensor:
    return ops._smooth_bwd_raw(img, flow, coef, alpha)


@smooth_loss_backward.register_fake
def _(img, flow, coef, alpha):
    return torch.empty_like(flow)


def _smooth_setup(ctx, inputs, output):
    img, flow, alpha = inputs
    ctx.save_for_backward(img, flow)
    ctx.alpha = alpha


def _smooth_backward(ctx, gsums):
    img, flow = ctx.saved_tensors
    if ctx.needs_input_grad[0]:
        raise NotImplementedError("ocflow::smooth_loss: gradient w.r.t. the image is not implemented")
    return None, torch.ops.ocflow.smooth_loss_backward(img, flow, gsums, ctx.alpha), None


smooth_loss.register_autograd(_smooth_backward, setup_context=_smooth_setup)


# ------------------------------------------------------------------------------------------------------------- range_map (K6)
@custom_op("ocflow::range_map", mutates_args=())
def range_map(flow: torch.Tensor, want_occ: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """non-differentiable (the reference runs it under no_grad; scatter_add_ has no gradient to the flow)"""
    rng, occ = ops._range_map_raw(flow, want_occ)
    return rng, (occ if occ is not None else rng.new_empty(0))


@range_map.register_fake
def _(flow, want_occ):
    shape = (flow.shape[0], 1, flow.shape[2], flow.shape[3])
    return flow.new_empty(shape, dtype=torch.float32), (flow.new_empty(shape, dtype=torch.float32) if want_occ else flow.new_empty(0))
