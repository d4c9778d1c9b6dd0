"""`FlowNetCV` (model 'pwc'): PWC-Net-style 6-level siamese encoder, per level {normalise -> warp -> cost volume ->
LeakyReLU -> 5 dense-connected convs -> flow}, dilated context net (reference:
models/networks/cost_volume_flow_net.py:22-246).  Convolutions stay in PyTorch-ROCm; warp / cost volume /
feature-moment kernels are HIP (ocflow_amd.ops).  Parameter names are the reference's (checkpoint compatible)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from ocflow_amd import conv as nconv
from ocflow_amd import ops
from ._epilogue import ConvLReLU, cat_aligned, conv_bias, conv_lrelu_padded, deconv_bias, deconv_bias_range, padded_channels_ok

ENC_WIDTHS = (3, 16, 32, 64, 96, 128, 196)
DENSE_WIDTHS = (128, 128, 96, 64, 32)
DENSE_TOTAL = int(np.sum(DENSE_WIDTHS))
# level -> scale applied to the up-sampled flow before warping (20 / 2^level; cost_volume_flow_net.py:186,201,216,231)
WARP_SCALE = {5: 0.625, 4: 1.25, 3: 2.5, 2: 5.0}


def conv(in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1):
    return ConvLReLU(nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding,
                               dilation=dilation, bias=True), nn.LeakyReLU(0.1))


def predict_flow(in_planes):
    return nn.Conv2d(in_planes, 2, kernel_size=3, stride=1, padding=1, bias=True)


def deconv(in_planes, out_planes, kernel_size=4, stride=2, padding=1):
    return nn.ConvTranspose2d(in_planes, out_planes, kernel_size, stride, padding, bias=True)


def build_encoder(net):
    """conv{L}a (stride 2), conv{L}aa, conv{L}b for L=1..5; level 6 is registered as conv6aa (stride 2), conv6a, conv6b"""
    for lvl in range(1, 7):
        cin, cout = ENC_WIDTHS[lvl - 1], ENC_WIDTHS[lvl]
        names = ("a", "aa", "b") if lvl < 6 else ("aa", "a", "b")
        for j, suffix in enumerate(names):
            setattr(net, f"conv{lvl}{suffix}", conv(cin if j == 0 else cout, cout, stride=2 if j == 0 else 1))


def encode(net, im):
    """c1..c6 of one image batch (cost_volume_flow_net.py:158-169).  A level whose channel count is no multiple of 8 (level 6: 196) keeps
    its activations in tensors rounded up to 8 channels, the extra ones zero, where the repo's kernels consume them (_epilogue.
    conv_lrelu_padded); the level's feature map is the channel range of the last one"""
    feats = []
    x = im
    for lvl in range(1, 7):
        names = ("a", "aa", "b") if lvl < 6 else ("aa", "a", "b")
        for suffix in names:
            layer = getattr(net, f"conv{lvl}{suffix}")
            c = layer[0].out_channels
            if c % 8 and isinstance(layer, ConvLReLU) and padded_channels_ok(x):
                x = conv_lrelu_padded(layer[0], x, layer[1].negative_slope)
            else:
                x = layer(x[:, :layer[0].in_channels] if x.shape[1] != layer[0].in_channels else x)
        x = ops.unpad_channels(x, c)  # (dense: the feature normalisation and the correlation kernels want whole pixels)
        feats.append(x)
    return feats  # c1..c6


def build_context(net, in_planes):
    plan = ((128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1))
    cin = in_planes
    for i, (cout, dil) in enumerate(plan, start=1):
        setattr(net, f"dc_conv{i}", conv(cin, cout, padding=dil, dilation=dil))
        cin = cout
    net.dc_conv7 = predict_flow(32)


def context(net, x):
    for i in range(1, 7):
        x = getattr(net, f"dc_conv{i}")(x)
    return net.dc_conv7(x)


def dense_block(net, lvl, x):
    """DenseNet connectivity, newest features FIRST: x = cat(conv(x), x)  (cost_volume_flow_net.py:176-180)"""
    for j in range(5):
        x = torch.cat((getattr(net, f"conv{lvl}_{j}")(x), x), 1)
    return x


def dense_convs(net, lvl):
    return [getattr(net, f"conv{lvl}_{j}")[0] for j in range(5)]  # (ConvLReLU = Sequential(conv, LeakyReLU(0.1)))


def level_features(net, lvl, corr_fn, others, native, like, dense=None):
    """decoder features of one pyramid level: LeakyReLU(cost volume), concatenated in front of `others`, through the level's dense block
    (cost_volume_flow_net.py:172-180,188-195; cost_volume_flow_occ_net.py:188-213).  `corr_fn(leaky_slope=..., grad_premasked=...)`
    computes the cost volume of features shaped like `like`.  In the bf16 channels_last configuration the whole level is ONE buffer:
    ocf_corr_fwd applies the LeakyReLU in its store epilogue, ocf_concat_channels writes [corr | others | zeros] into the buffer's tail and
    every dense layer reads / writes channel ranges of it (conv.conv3x3_dense_block) -- no torch.cat, no separate activation pass.
    Otherwise: the reference's op chain (with the concatenation aligned to 32 channels when `native`)."""
    convs = dense_convs(net, lvl)
    others = [t if t.dtype == like.dtype else t.to(like.dtype) for t in others]
    dense = native if dense is None else dense
    if dense and nconv.dense_block_eligible(others, convs, lead_channels=(2 * int(getattr(net, 'displacement', 4)) + 1) ** 2, like=like):
        corr = corr_fn(leaky_slope=0.1, grad_premasked=True)
        parts = [corr] + others
        if not nconv.dense_block_eligible(parts, convs):  # (cannot happen for the dense channels_last volume ocf_corr_fwd returns)
            raise RuntimeError(f"level_features: cost volume layout {tuple(corr.shape)} / {corr.stride()} / {corr.dtype}")
        return nconv.conv3x3_dense_block(parts, convs, 0.1, lead_mask_channels=int(corr.shape[1]))
    corr = corr_fn(leaky_slope=0.1, grad_premasked=False)
    if others:
        parts = [corr] + [t.to(corr.dtype) for t in others]
        feats = cat_aligned(parts) if native else torch.cat(parts, 1)
    else:
        feats = cat_aligned([corr]) if native else corr
    return dense_block(net, lvl, feats)


class FlowNetCV(nn.Module):
    def __init__(self, displacement=4):
        super().__init__()
        build_encoder(self)
        self.normalize = ops.normalize_features
        self.leakyRELU = nn.LeakyReLU(0.1)
        self.displacement = displacement
        nd = (2 * displacement + 1) ** 2
        for lvl in range(6, 1, -1):
            od = nd if lvl == 6 else nd + ENC_WIDTHS[lvl] + 4
            cin = od
            for j, w in enumerate(DENSE_WIDTHS):
                setattr(self, f"conv{lvl}_{j}", conv(cin, w))
                cin += w
            setattr(self, f"predict_flow{lvl}", predict_flow(cin))
            setattr(self, f"deconv{lvl}", deconv(2, 2))
            if lvl > 2:
                setattr(self, f"upfeat{lvl}", deconv(cin, 2))
        build_context(self, nd + ENC_WIDTHS[2] + 4 + DENSE_TOTAL)
        # default PyTorch init, as in the reference (its kaiming block is commented out, :114-118)

    def warp(self, img, flow):
        """feature warp, align_corners=False variant (cost_volume_flow_net.py:121-151)"""
        return ops.warp(img, flow, align_corners=False)

    def pyramids(self, x):
        """(c1, c2): the siamese encoder on both images (cost_volume_flow_net.py:158-169).  The encoder has no batch-coupled layer, so the two
        calls of the reference are ONE call on the 2B images of the pair (same values per image; half the launches of 18 layers, twice
        the work per launch) in the bf16 throughput configuration, where ops.pair_input assembles that batch in one launch; every
        other configuration (fp32 parity runs, CPU) calls the encoder per image, as the reference does"""
        x1, x2 = x[:, :3], x[:, 3:]
        if os.environ.get("OCFLOW_UNPAIRED_PWC", "0") != "1" and not torch.compiler.is_compiling() and ops.pair_input_eligible(x1, x2):
            both = ops.pair_input(x1, x2)
            B = x.shape[0]
            feats = encode(self, both)
            return [f[:B] for f in feats], [f[B:] for f in feats]
        return encode(self, x1), encode(self, x2)

    def forward(self, x):
        return self.decode(x, *self.pyramids(x))

    def forward_two_way(self, x):
        """(forward(x), forward(cat(x[:, 3:], x[:, :3])) without gradient): what the occlusion-aware step asks for in two calls
        (models/model.py:376,384) -- the second call's pyramids ARE the first call's, swapped (same weights, no batch-coupled layer), so
        the encoder runs once"""
        c1, c2 = self.pyramids(x)
        out = self.decode(x, c1, c2)
        with torch.no_grad():
            back = self.decode(x, [t.detach() for t in c2], [t.detach() for t in c1])
        return out, back

    def decode(self, x, c1, c2):
        # Aligned (32-channel) concatenations: with the repo's own convolution kernels consuming them (round 4: every dense-block layer,
        # the prediction layers and the context network run on ocf_conv3x3_igemm / ocf_conv3x3_wgrad_bias when their input has a multiple of
        # 8 channels) this is the default in the bf16 channels_last configuration.  The ONE library kernel class that met padded operands
        # in round 2 -- the transposed convolution `upfeat` on a padded [in + pad, 2, 4, 4] weight, the suspect of the GPU memory fault at
        # 8 x 384 x 512 -- is handed the unpadded channel range instead.  OCFLOW_PAD_PWC=1 forces the round-2
        # configuration (padded operands into the library: scripts/diag_fault2.py), OCFLOW_PAD_PWC=0 the plain concatenations.
        env = os.environ.get("OCFLOW_PAD_PWC", "auto")
        native = env == "auto" and nconv.enabled() and x.is_cuda and torch.is_autocast_enabled() and not torch.compiler.is_compiling()
        pad = env == "1" or native
        up_flow = up_feat = flow = feats = None
        for lvl in range(6, 1, -1):
            f1, f2 = c1[lvl - 1], c2[lvl - 1]
            if lvl < 6:
                f2 = ops.warp(f2, up_flow, False, flow_scale=WARP_SCALE[lvl])
            f1, f2 = self.normalize([f1, f2])
            # LeakyReLU(cost volume) [| normalised c1 | up_flow | up_feat] through the dense block (the NORMALISED c1 is what enters the
            # decoder: SURVEY F6, :190)
            corr_fn = lambda leaky_slope, grad_premasked, a=f1, b=f2: ops.compute_cost_volume(
                a, b, max_displacement=self.displacement, leaky_slope=leaky_slope, grad_premasked=grad_premasked)
            feats = level_features(self, lvl, corr_fn, [] if lvl == 6 else [f1, up_flow, up_feat], pad, f1, dense=native)
            if pad:
                flow = conv_bias(getattr(self, f"predict_flow{lvl}"), feats)
                if lvl > 2:
                    up_flow = deconv_bias(getattr(self, f"deconv{lvl}"), flow)
                    upf = getattr(self, f"upfeat{lvl}")
                    up_feat = deconv_bias_range(upf, feats) if native else deconv_bias(upf, feats)
                continue
            flow = getattr(self, f"predict_flow{lvl}")(feats)
            if lvl > 2:
                up_flow = getattr(self, f"deconv{lvl}")(flow)
                up_feat = getattr(self, f"upfeat{lvl}")(feats)
        flow2 = flow + context(self, feats)
        flow1 = F.interpolate(flow2.float(), scale_factor=4, mode="bilinear", align_corners=True) * 20
        return flow1, flow2 * 5.0
