"""TEST INFRASTRUCTURE (like everything under oracle/): route `ocflow_amd.ops` through the CPU oracle so that
  * tests/ can check the HOST logic (network wiring, state_dict keys, step functions, DDP plumbing) without a GPU, and
  * bench.py's `cpu_baseline` leg can time the reference's CPU op structure on the GPU box's host cores.
Never imported by anything under ocflow_amd/: the product's ops raise on CPU tensors and have no fallback."""
from contextlib import contextmanager

from oracle import ocflow_oracle as O


def _lrelu(cv, leaky_slope):
    """the activation the PWC-style nets apply to the cost volume (cost_volume_flow_net.py:173,189): the product fuses it into the launch"""
    import torch.nn.functional as F
    return F.leaky_relu(cv, float(leaky_slope)) if leaky_slope else cv


def _corr_warp(f1, f2, flow=None, occ=None, max_displacement=4, flow_scale=1.0, align_corners=False, out=None, *, leaky_slope=0.0,
               grad_premasked=False):
    return _lrelu(O.corr_warp(f1, f2, flow, occ, max_displacement, flow_scale, align_corners), leaky_slope)


def _compute_cost_volume(features1, features2, max_displacement=4, *, leaky_slope=0.0, grad_premasked=False):
    return _lrelu(O.compute_cost_volume(features1, features2, max_displacement), leaky_slope)


def _warp(img, flow, align_corners=False, flow_scale=1.0):
    return O.warp(img, flow * flow_scale, align_corners)


def _normalize(feature_list, *a, **k):
    return O.normalize_features(list(feature_list))


def _warp_photo(img1, img2, flow, occ=None):
    return O.photometric_error(O.warp(img2, flow, True), img1, occ)


def _smooth(img, flow, alpha=100.0):
    return O.first_order_smoothness_loss(img, flow, alpha), O.second_order_smoothness_loss(img, flow, alpha)


def _occ_from_flow(flow):
    return O.occlusion_from_range(O.compute_range_map(flow))


PATCHES = {
    "corr_warp": _corr_warp,
    "compute_cost_volume": _compute_cost_volume,
    "warp": _warp,
    "normalize_features": _normalize,
    "warp_photometric_error": _warp_photo,
    "photometric_error": O.photometric_error,
    "smoothness_losses": _smooth,
    "first_order_smoothness_loss": O.first_order_smoothness_loss,
    "second_order_smoothness_loss": O.second_order_smoothness_loss,
    "compute_range_map": O.compute_range_map,
    "occlusion_from_flow": _occ_from_flow,
    "bias_leaky_relu": O.bias_leaky_relu,
    "bias_add": O.bias_add,
    "masked_l1": O.masked_l1,
}


def patch_ops(monkeypatch=None):
    import ocflow_amd.ops as ops
    for name, fn in PATCHES.items():
        if monkeypatch is not None:
            monkeypatch.setattr(ops, name, fn)
        else:  # subprocess workers of the gloo test
            setattr(ops, name, fn)


@contextmanager
def oracle_ops():
    """temporarily patch ocflow_amd.ops with the oracle (restores the HIP operators on exit)"""
    import ocflow_amd.ops as ops
    saved = {name: getattr(ops, name) for name in PATCHES}
    try:
        patch_ops(None)
        yield
    finally:
        for name, fn in saved.items():
            setattr(ops, name, fn)
