"""`InpaintSANet` -- the gated-convolution completion generator with self-attention, the `inpainting_stage='gated'` second stage of the
two-stage trainers and what the reference's shipped `config/unsupervised_config.yml` selects
(reference: models/networks/gated_conv_inpainting_net.py:9-71 `get_pad` / `Conv2dWithProj` / `GatedProjConv2dWithActivation`,
:111-126 `GatedProjDeConv2dWithActivation`, :284-317 `Self_Attn`, :319-407 `InpaintSANet`).

Two encoder / dilated-bottleneck / decoder stacks (coarse, refine) of GATED layers: a layer is two 1x1 -> kxk -> 1x1 projection
convolutions on the same input, `act(BatchNorm(f(x))) * sigmoid(g(x))`; the refine stack has one self-attention block at quarter
resolution.  Written as a table of layer specifications instead of the reference's literal `nn.Sequential`s; module names, parameter
shapes, registration order and the Kaiming-normal initialisation are the reference's, so its checkpoints (`generator.*` of an
`InpaintingStageModel` checkpoint) load unchanged.

MI355X notes: BatchNorm + LeakyReLU(0.2) of a gated layer goes through the two-pass HIP kernels of the encoder (`ops.batchnorm_lrelu`,
channels_last tensors on the GPU); the N x N attention of `Self_Attn` (N = (H/4)(W/4): 768 at the reference's 96 x 128 training size, 27,904
at Sintel size -- 3.1 GB of fp32 scores per sample if materialised as the reference does, `:306-310`) is evaluated by
`scaled_dot_product_attention` with unit scale, which never writes the score matrix; `with_attn=True` (the reference can return the
attention map) materialises it and is only meant for small inputs."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ocflow_amd import ops


def get_pad(in_, ksize, stride, dilation=1):
    """'same'-style padding of gated_conv_inpainting_net.py:9-18 for a size or an (h, w) pair"""
    def one(n):
        out = math.ceil(float(n) / stride)
        return int(((out - 1) * stride + dilation * (ksize - 1) + 1 - n) / 2)
    if isinstance(in_, (list, tuple)):
        return (one(in_[0]), one(in_[1]))
    return one(in_)


class Conv2dWithProj(nn.Module):
    """1x1 projection -> k x k convolution -> 1x1 expansion, no activation in between (gated_conv_inpainting_net.py:20-36)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=False, proj_ratio=4):
        super().__init__()
        inter = max(in_channels // proj_ratio, 1)
        self.conv1 = nn.Conv2d(in_channels, inter, 1, bias=bias)
        self.conv2 = nn.Conv2d(inter, inter, kernel_size, stride=stride, padding=padding, dilation=dilation, bias=bias)
        self.conv3 = nn.Conv2d(inter, out_channels, 1, bias=bias)

    def forward(self, x):
        return self.conv3(self.conv2(self.conv1(x)))


class GatedProjConv2dWithActivation(nn.Module):
    """act(BatchNorm(f(x))) * sigmoid(g(x)) with f, g = Conv2dWithProj (gated_conv_inpainting_net.py:38-74)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=False, batch_norm=True,
                 activation="lrelu", proj_ratio=4):
        super().__init__()
        if not batch_norm:
            raise NotImplementedError("batch_norm=False: the reference's forward reads self.batch_norm, which only exists with batch_norm=True "
                                      "(gated_conv_inpainting_net.py:55-68)")
        self.activation = nn.LeakyReLU(0.2, inplace=True) if activation == "lrelu" else activation
        self.conv2d = Conv2dWithProj(in_channels, out_channels, kernel_size, stride, padding, dilation, bias, proj_ratio)
        self.mask_conv2d = Conv2dWithProj(in_channels, out_channels, kernel_size, stride, padding, dilation, bias, proj_ratio)
        self.batch_norm = nn.BatchNorm2d(out_channels)
        self.sigmoid = nn.Sigmoid()
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        f, g = self.conv2d(x), self.mask_conv2d(x)
        if isinstance(self.activation, nn.LeakyReLU):
            f = ops.batchnorm_lrelu(f, self.batch_norm, self.activation.negative_slope)
        else:
            f = self.batch_norm(f)
            if self.activation is not None:
                f = self.activation(f)
        return f * torch.sigmoid(g)


class GatedProjDeConv2dWithActivation(nn.Module):
    """nearest x2 resize + gated layer (gated_conv_inpainting_net.py:111-126)"""

    def __init__(self, scale_factor, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True, batch_norm=True,
                 activation="lrelu", proj_ratio=4):
        super().__init__()
        self.conv2d = GatedProjConv2dWithActivation(in_channels, out_channels, kernel_size, stride, padding, dilation, bias, batch_norm, activation,
                                                    proj_ratio)
        self.scale_factor = scale_factor

    def forward(self, x):
        return self.conv2d(F.interpolate(x, scale_factor=2))  # (the reference resizes by 2 whatever `scale_factor` says, :125)


class Self_Attn(nn.Module):
    """SAGAN self-attention over all positions: out = gamma * V softmax(Q^T K)^T + x (gated_conv_inpainting_net.py:284-317)"""

    def __init__(self, in_dim, with_attn=False):
        super().__init__()
        self.in_channels = in_dim
        self.with_attn = with_attn
        self.query_conv = nn.Conv2d(in_dim, in_dim // 8, kernel_size=1)
        self.key_conv = nn.Conv2d(in_dim, in_dim // 8, kernel_size=1)
        self.value_conv = nn.Conv2d(in_dim, in_dim, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x):
        B, C, H, W = x.shape
        n = H * W
        q = self.query_conv(x).reshape(B, -1, n).transpose(1, 2)  # B x N x C/8
        k = self.key_conv(x).reshape(B, -1, n).transpose(1, 2)    # B x N x C/8
        v = self.value_conv(x).reshape(B, -1, n).transpose(1, 2)  # B x N x C
        if self.with_attn:
            attention = self.softmax(torch.bmm(q, k.transpose(1, 2)))  # B x N x N
            out = torch.bmm(attention, v)
        else:  # softmax(q k^T) v without the N x N matrix in memory; unit scale (the reference does not divide by sqrt(d))
            attention = None
            out = F.scaled_dot_product_attention(q.unsqueeze(1), k.unsqueeze(1), v.unsqueeze(1), scale=1.0).squeeze(1)
        out = out.transpose(1, 2).reshape(B, C, H, W)
        out = self.gamma * out + x
        return (out, attention) if self.with_attn else out


# (kind, in, out, kernel, stride, dilation, resolution divisor of the padding rule, extra) in units of cnum = 32
_C = 32
_COARSE = (("g", 4, _C, 5, 1, 1, 1, {"proj_ratio": 1}), ("g", _C, 2 * _C, 4, 2, 1, 1, {}), ("g", 2 * _C, 2 * _C, 3, 1, 1, 2, {}),
           ("g", 2 * _C, 4 * _C, 4, 2, 1, 2, {}), ("g", 4 * _C, 4 * _C, 3, 1, 1, 4, {}), ("g", 4 * _C, 4 * _C, 3, 1, 1, 4, {}),
           ("g", 4 * _C, 4 * _C, 3, 1, 2, 4, {}), ("g", 4 * _C, 4 * _C, 3, 1, 4, 4, {}), ("g", 4 * _C, 4 * _C, 3, 1, 8, 4, {}),
           ("g", 4 * _C, 4 * _C, 3, 1, 16, 4, {}), ("g", 4 * _C, 4 * _C, 3, 1, 1, 4, {}), ("g", 4 * _C, 4 * _C, 3, 1, 1, 4, {}),
           ("d", 4 * _C, 2 * _C, 3, 1, 1, 2, {}), ("g", 2 * _C, 2 * _C, 3, 1, 1, 2, {}), ("d", 2 * _C, _C, 3, 1, 1, 1, {}),
           ("g", _C, _C // 2, 3, 1, 1, 1, {}), ("g", _C // 2, 3, 3, 1, 1, 1, {"activation": "tanh"}))
_REFINE_CONV = (("g", 4, _C, 5, 1, 1, 1, {"proj_ratio": 1}), ("g", _C, _C, 4, 2, 1, 1, {}), ("g", _C, 2 * _C, 3, 1, 1, 2, {}),
                ("g", 2 * _C, 2 * _C, 4, 2, 1, 2, {}), ("g", 2 * _C, 4 * _C, 3, 1, 1, 4, {}), ("g", 4 * _C, 4 * _C, 3, 1, 1, 4, {}),
                ("g", 4 * _C, 4 * _C, 3, 1, 1, 4, {}), ("g", 4 * _C, 4 * _C, 3, 1, 2, 4, {}), ("g", 4 * _C, 4 * _C, 3, 1, 4, 4, {}),
                ("g", 4 * _C, 4 * _C, 3, 1, 8, 4, {}), ("g", 4 * _C, 4 * _C, 3, 1, 16, 4, {}))
_REFINE_UP = (("g", 4 * _C, 4 * _C, 3, 1, 1, 4, {}), ("g", 4 * _C, 4 * _C, 3, 1, 1, 4, {}), ("d", 4 * _C, 2 * _C, 3, 1, 1, 2, {}),
              ("g", 2 * _C, 2 * _C, 3, 1, 1, 2, {}), ("d", 2 * _C, _C, 3, 1, 1, 1, {}), ("g", _C, _C // 2, 3, 1, 1, 1, {}),
              ("g", _C // 2, 3, 3, 1, 1, 1, {"activation": "tanh"}))


def _stack(spec, img_size, n_in_channel):
    h, w = img_size
    layers = []
    for i, (kind, cin, cout, k, stride, dil, div, extra) in enumerate(spec):
        kw = dict(extra)
        if kw.get("activation") == "tanh":
            kw["activation"] = nn.Tanh()
        pad = get_pad((h // div, w // div), k, stride, dil)
        cin = n_in_channel if (i == 0 and cin == 4) else cin
        if kind == "g":
            layers.append(GatedProjConv2dWithActivation(cin, cout, k, stride, padding=pad, dilation=dil, **kw))
        else:
            layers.append(GatedProjDeConv2dWithActivation(2, cin, cout, k, stride, padding=pad, dilation=dil, **kw))
    return nn.Sequential(*layers)


class InpaintSANet(nn.Module):
    """generator: (imgs [B,3,H,W] in [-1,1], masks [B,1,H,W], 1 = pixel to complete) -> (coarse, refined) completions
    (gated_conv_inpainting_net.py:319-407); `img_size` fixes the paddings of the strided layers as in the reference"""

    def __init__(self, n_in_channel=4, img_size=(64, 128)):
        super().__init__()
        self.coarse_net = _stack(_COARSE, img_size, n_in_channel)
        self.refine_conv_net = _stack(_REFINE_CONV, img_size, n_in_channel)
        self.refine_attn = Self_Attn(4 * _C, with_attn=False)
        self.refine_upsample_net = _stack(_REFINE_UP, img_size, n_in_channel)

    def forward(self, imgs, masks):
        coarse_x = self.coarse_net(torch.cat([imgs * (1 - masks), masks], dim=1))
        x = self.refine_conv_net(torch.cat([imgs * (1 - masks) + coarse_x * masks, masks], dim=1))
        x = self.refine_upsample_net(self.refine_attn(x))
        return coarse_x, x
