"""torch.library registration of the two core operators: `torch.ops.ocflow.cost_volume` and `torch.ops.ocflow.warp`.

The kernels stay behind the C ABI and `ctypes` (ocflow_amd/ops.py); this module gives them dispatcher identities -- schema, fake
(meta) implementations for shape propagation, autograd formulas as registered backward operators -- so that they pass
`torch.library.opcheck`, trace under `torch.compile` / `make_fx` without graph breaks and show up by name in profiler traces.
`ocflow_amd.ops.compute_cost_volume` / `warp` remain the reference-named entry points; these are the same launches.

    torch.ops.ocflow.cost_volume(f1, f2, max_displacement)      reference: models/networks/correlation_layer.py:7-40
    torch.ops.ocflow.warp(img, flow, flow_scale, align_corners)  reference: warp() of flow_net.py:57-87 (AC False), models/model.py:191-221 (AC True)
"""
from typing import Tuple

import torch
from torch.library import custom_op

from . import ops


@custom_op("ocflow::cost_volume", mutates_args=())
def cost_volume(f1: torch.Tensor, f2: torch.Tensor, max_displacement: int) -> torch.Tensor:
    return ops.compute_cost_volume(f1.detach(), f2.detach(), max_displacement)


@cost_volume.register_fake
def _(f1, f2, max_displacement):
    n = (2 * max_displacement + 1) ** 2
    cl = f1.dim() == 4 and f1.is_contiguous(memory_format=torch.channels_last) and not f1.is_contiguous()
    return torch.empty((f1.shape[0], n, f1.shape[2], f1.shape[3]), dtype=f1.dtype, device=f1.device,
                       memory_format=torch.channels_last if cl else torch.contiguous_format)


@custom_op("ocflow::cost_volume_backward", mutates_args=())
def cost_volume_backward(f1: torch.Tensor, f2: torch.Tensor, grad_out: torch.Tensor, max_displacement: int) -> Tuple[torch.Tensor, torch.Tensor]:
    a, b = f1.detach().requires_grad_(), f2.detach().requires_grad_()
    with torch.enable_grad():
        out = ops.compute_cost_volume(a, b, max_displacement)
    g1, g2 = torch.autograd.grad(out, (a, b), grad_out)  # (ocf_corr_bwd; the forward launch above is the price of a stateless op)
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
    g1, g2 = torch.ops.ocflow.cost_volume_backward(f1, f2, grad_out.contiguous(memory_format=torch.channels_last)
                                                   if f1.is_contiguous(memory_format=torch.channels_last) and not f1.is_contiguous()
                                                   else grad_out.contiguous(), ctx.d)
    return g1, g2, None


cost_volume.register_autograd(_cv_backward, setup_context=_cv_setup)


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
