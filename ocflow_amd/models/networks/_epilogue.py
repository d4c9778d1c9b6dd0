"""conv -> bias -> LeakyReLU blocks of the estimator nets with the bias add and the activation in one HIP pass.

The reference writes either `F.leaky_relu(self.convN(x), 0.1)` or `nn.Sequential(nn.Conv2d(..., bias=True), nn.LeakyReLU(0.1))`
(cost_volume_flow_net.py:19-21 `conv`, simple_flow_net.py:55-57).  Module structure, parameter names and values are unchanged:
the convolution runs without its bias and `ops.bias_leaky_relu` applies bias + activation in place on its output."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ocflow_amd import conv as nconv
from ocflow_amd import ops

CHANNEL_ALIGN = 32  # channels; see cat_aligned


def cat_aligned(parts, dim: int = 1):
    """torch.cat(parts, 1) for a tensor that feeds convolutions (the decoder inputs: cost volume + features + flow ...; reference
    flow_occ_net.py:153-176, flow_net.py:104-121).  Those concatenations have 117, 66, 149, 34 ... channels, and on 2-byte-aligned pixels
    the framework's bf16 channels_last convolutions fall to slow solvers: 117->128 at 8x109x256 takes 285 / 475 / 256 us (forward /
    input gradient / weight gradient) against 112 / 123 / 144 us at 120 channels (profiles/r2_f_pad_channels.txt) -- 1.6 ms of the
    16.6 ms step over the ten such layers.  In the 16-bit channels_last configuration the concatenation is therefore rounded up to a
    multiple of 32 channels with ZERO channels at the end (8 is enough for the speed; 32 keeps every implicit-GEMM tile of the
    library whole -- a find-mode search over the multiple-of-8 shapes ran into a faulting vector-1 input-gradient kernel of the
    library); `_conv_nobias` pads the weight's input channels with zeros to match, so
    every output and every parameter gradient is the same sum with zero terms added.  Parameters keep the reference's shapes.
    OCFLOW_UNPADDED_CAT=1 switches it off (A/B runs).  Used by FlowNet / FlowOccNet (the benchmark models).  The PWC-style nets
    (FlowNetCV, FlowOccNetCV) keep plain torch.cat: with their dense blocks padded, the library's heuristic solver choice for the new
    shapes ended in a GPU memory fault inside a library kernel at 8x384x512 (no find-db entry exists for them); not pursued."""
    x0 = parts[0]
    if dim == 1 and x0.is_cuda and x0.dim() == 4 and x0.dtype in (torch.bfloat16, torch.float16) and not _unpadded_cat():
        ctot = sum(int(t.shape[1]) for t in parts)
        pad = -ctot % CHANNEL_ALIGN
        native = nconv.enabled() and os.environ.get("OCFLOW_NATIVE_CAT", "1") != "0" and not torch.compiler.is_compiling()
        if native and x0.dtype == torch.bfloat16:
            # with the repo's own convolution kernels behind it (ocflow_amd.conv) the aligned tensor never meets a library solver, so it
            # is always formed -- by ONE launch that gathers the parts and writes the alignment zeros (ops.concat_channels)
            ps = [t if t.dtype == x0.dtype else t.to(x0.dtype) for t in parts]
            if ops.concat_channels_eligible(ps):
                return ops.concat_channels(ps, ctot + pad)
        if pad and not _padded_shape_is_known(int(x0.shape[0]), ctot + pad, int(x0.shape[2]), int(x0.shape[3])):
            pad = 0  # no find-db entry for the aligned shape: the library would choose its kernel heuristically -- keep the plain tensor
        if pad:
            cl = x0.is_contiguous(memory_format=torch.channels_last)
            z = _zero_channels(x0, pad, cl)
            parts = list(parts) + [z]
    return torch.cat(list(parts), dim)


_ZEROS = {}  # (device, dtype, B, pad, H, W, channels_last) -> read-only zero block shared by every concatenation of that shape


def _zero_channels(x0, pad, cl):
    """a [B, pad, H, W] block of zeros in x0's layout.  Never written, so one tensor per shape serves every step (a fill launch per
    concatenation and step otherwise); it stays allocated for the life of the process (a few MB per pyramid level)."""
    key = (x0.device, x0.dtype, x0.shape[0], pad, x0.shape[2], x0.shape[3], cl)
    z = _ZEROS.get(key)
    if z is None:
        if cl:  # (torch.zeros takes no memory_format: a channels_last view of an NHWC allocation)
            z = x0.new_zeros((x0.shape[0], x0.shape[2], x0.shape[3], pad)).permute(0, 3, 1, 2)
        else:
            z = x0.new_zeros((x0.shape[0], pad, x0.shape[2], x0.shape[3]))
        if not torch.cuda.is_current_stream_capturing():
            _ZEROS[key] = z  # (a tensor created during a graph capture lives in the graph's pool: not shared)
    return z


def _unpadded_cat() -> bool:
    return os.environ.get("OCFLOW_UNPADDED_CAT", "0") == "1"


_DB_INPUTS = None  # {(N, C, H, W)} of the forward convolutions the shipped find-db holds an entry for


def _padded_shape_is_known(n: int, c: int, h: int, w: int) -> bool:
    """Aligned concatenations are only formed for inputs whose convolution the SHIPPED find-db covers AND only while that db is the
    running MIOpen's (runtime.miopen_db_status): there the solver is the one this repo measured and ran for hours.  Any other shape or
    library build would send a never-seen (N, C + pad, H, W) problem through the library's heuristic solver choice; round 2 met two GPU
    memory faults on that path, so such inputs stay unpadded (slower convolutions, known-good kernels).
    OCFLOW_PAD_ALWAYS=1 restores unconditional padding (block-level equivalence tests, measurements)."""
    if os.environ.get("OCFLOW_PAD_ALWAYS", "0") == "1":
        return True
    global _DB_INPUTS
    if _DB_INPUTS is None:
        from ocflow_amd import runtime
        shapes = set()
        st = runtime.miopen_db_status()
        if st["matched"]:
            cfg = os.path.join(os.path.dirname(os.path.abspath(runtime.__file__)), "miopen_db", "config")
            for f in os.listdir(cfg):
                if f.endswith(".ufdb.txt"):
                    with open(os.path.join(cfg, f)) as fh:
                        for line in fh:
                            key = line.split("=", 1)[0].split("-")
                            # C-H-W-khxkw-K-Ho-Wo-N-pad-stride-dilation-...-BF16-F
                            if len(key) >= 12 and key[-1] == "F" and key[-2] == "BF16":
                                try:
                                    shapes.add((int(key[7]), int(key[0]), int(key[1]), int(key[2])))
                                except ValueError:
                                    pass
        _DB_INPUTS = shapes
    return (n, c, h, w) in _DB_INPUTS


_WZEROS = {}  # (device, dtype, shape, channels_last) -> read-only zero block appended to weights of that shape


def _zero_weight_block(weight, shape, cl):
    key = (weight.device, weight.dtype, tuple(shape), cl)
    z = _WZEROS.get(key)
    if z is None:
        z = weight.new_zeros(shape)
        if cl:
            z = z.contiguous(memory_format=torch.channels_last)
        if not (weight.is_cuda and torch.cuda.is_current_stream_capturing()):
            _WZEROS[key] = z
    return z


class _PadInChannels(torch.autograd.Function):
    """weight with zero input channels appended, in the weight's own memory format: ONE launch forward (a concatenation with a shared
    read-only zero block), one copy backward (F.pad + .contiguous(channels_last) and their autograd took 5-6 launches per layer and step)"""

    @staticmethod
    def forward(ctx, weight, cin, dim):
        shape = list(weight.shape)
        shape[dim] = cin - weight.shape[dim]
        cl = weight.dim() == 4 and weight.is_contiguous(memory_format=torch.channels_last) and not weight.is_contiguous()
        fmt = torch.channels_last if cl else torch.contiguous_format
        out = torch.cat((weight, _zero_weight_block(weight, shape, cl)), dim)
        if not out.is_contiguous(memory_format=fmt):
            out = out.contiguous(memory_format=fmt)
        ctx.cfg = (dim, weight.shape[dim], fmt)
        return out

    @staticmethod
    def backward(ctx, grad):
        dim, n, fmt = ctx.cfg
        return grad.narrow(dim, 0, n).contiguous(memory_format=fmt), None, None


def _match_in_channels(weight, cin: int, dim: int = 1):
    """the weight with zero input channels appended up to `cin` (the input came from `cat_aligned`); autograd slices the gradient back"""
    extra = cin - weight.shape[dim]
    if extra == 0:
        return weight
    if extra < 0 or extra >= CHANNEL_ALIGN:
        raise RuntimeError(f"convolution input has {cin} channels, the weight expects {weight.shape[dim]}")
    return _PadInChannels.apply(weight, cin, dim)


def _conv_nobias(conv: nn.Conv2d, x):
    """the convolution proper.  3x3 / stride-1 / same layers in the bf16 channels_last throughput configuration take their weight
    gradient from the hand-written MFMA kernel (ops.conv3x3: ocf_conv3x3_wgrad); forward and input gradient stay MIOpen's.
    Opt-in (OCFLOW_NATIVE_WGRAD=1): the first version of that kernel is parity-green and level with MIOpen's kernel + helper launches
    on the large layers (profiles/r2_*_wgrad.txt) but not yet a net win for the whole step."""
    w = _match_in_channels(conv.weight, x.shape[1]) if conv.groups == 1 else conv.weight
    mode = _native_wgrad()
    if mode and (mode == "1" or x.shape[0] * x.shape[2] * x.shape[3] <= _SMALL_PIXELS) and \
            ops.conv3x3_wgrad_eligible(x, w, conv.stride, conv.padding, conv.dilation, conv.groups):
        return ops.conv3x3(x, w, conv.dilation[0])
    return F.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups)


_SMALL_PIXELS = 4096  # B*H*W up to which "small" mode uses the hand-written weight gradient (the two coarsest pyramid levels at B = 8)


def _native_wgrad() -> str:
    """OCFLOW_NATIVE_WGRAD: "0" (default) never, "1" every eligible layer, "small" only layers of at most _SMALL_PIXELS pixels"""
    v = os.environ.get("OCFLOW_NATIVE_WGRAD", "0")
    return "" if v == "0" else v


NATIVE_MIN_PIXELS = int(os.environ.get("OCFLOW_NATIVE_CONV_MIN_PIXELS", "0"))


def _native(conv: nn.Conv2d, x) -> bool:
    """does this layer run on the repo's implicit-GEMM kernel (ocflow_amd.conv)?  3x3 / stride-1 / padding == dilation layers on bf16
    channels_last inputs of at least NATIVE_MIN_PIXELS pixels (0: every pyramid level -- the launcher switches to its barrier-free
    small-map kernel at the two coarsest levels; OCFLOW_NATIVE_CONV_MIN_PIXELS=32768 keeps the library there for A/B runs)"""
    return (nconv.enabled() and conv.padding_mode == "zeros" and x.dim() == 4 and x.shape[0] * x.shape[2] * x.shape[3] >= NATIVE_MIN_PIXELS
            and nconv.eligible(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups))


def conv_lrelu(conv: nn.Conv2d, x, negative_slope: float = 0.1):
    # OCFLOW_UNFUSED_EPILOGUE=1: the reference's own op chain (A/B switch for measurements and bisecting, not a fallback)
    if conv.bias is None or conv.padding_mode != "zeros" or _unfused():
        return F.leaky_relu(_conv_module(conv, x), negative_slope)
    if _native(conv, x):
        return nconv.conv3x3_bias_act(x, conv.weight, conv.bias, negative_slope, conv.dilation[0])
    return ops.bias_leaky_relu(_conv_nobias(conv, x), conv.bias, negative_slope)


def padded_channels_ok(x) -> bool:
    """may a layer whose channel count is no multiple of 8 (the PWC encoders' 196) hand on a zero-padded output?  Only where the repo's
    own kernels consume it: bf16 autocast on a HIP device, channels_last activations (OCFLOW_PAD_ENCODER=0: never)"""
    return (nconv.enabled() and os.environ.get("OCFLOW_PAD_ENCODER", "1") != "0" and torch.is_tensor(x) and x.is_cuda and x.dim() == 4
            and x.dtype == torch.bfloat16 and torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16
            and not torch.compiler.is_compiling() and not _unfused())


def conv_lrelu_padded(conv: nn.Conv2d, x, negative_slope: float = 0.1):
    """conv_lrelu whose result carries conv.out_channels rounded up to 8 channels, the extra ones ZERO -- so that the next 3x3 layer (and
    this layer's own backward) run on ocf_conv3x3_igemm / ocf_conv3x3_wgrad_bias, which read pixels in 16-byte chunks.  With 196 channels
    the library serves these layers through its slowest solvers: 142 us forward, 170 us input gradient per layer at 16 x 6 x 8 against
    ~10 us (profiles/r5_cfg3_step_breakdown.txt).  `x` may itself be such a padded tensor.  Falls back to conv_lrelu (plain channels)
    where the padded form cannot be produced."""
    cw = int(conv.out_channels)
    cp = (cw + 7) // 8 * 8
    if cp == cw or conv.bias is None or conv.padding_mode != "zeros":
        return conv_lrelu(conv, x, negative_slope)
    if tuple(conv.stride) == (1, 1) and _native(conv, x):
        return nconv.conv3x3_bias_act(x, conv.weight, conv.bias, negative_slope, conv.dilation[0], out_channels=cp)
    y = conv_lrelu(conv, x[:, :conv.in_channels] if x.shape[1] != conv.in_channels else x, negative_slope)
    if ops.concat_channels_eligible([y], cp):
        return ops.concat_channels([y], cp)  # (one launch: the copy and the zero channels)
    return y


def conv_lrelu_chain(convs, x, negative_slope: float = 0.1):
    """`for conv in convs: x = F.leaky_relu(conv(x), negative_slope)` for layers whose outputs feed only the next one (conv1..conv5 of the
    estimators, flow_net.py:25-37 / flow_occ_net.py:33-45; conv1..conv6 of the context network, context_net.py:20-27).  On the
    implicit-GEMM kernels the chain is ONE autograd node (ocflow_amd.conv.conv3x3_chain): the activation derivative of every inner
    layer is applied in the epilogue of the next layer's input-gradient kernel.  Anything else: layer by layer."""
    convs = list(convs)
    if not _unfused() and x.dim() == 4 and x.shape[0] * x.shape[2] * x.shape[3] >= NATIVE_MIN_PIXELS and len(convs) > 1 \
            and nconv.chain_eligible(x, convs):
        return nconv.conv3x3_chain(x, [c.weight for c in convs], [c.bias for c in convs], [negative_slope] * len(convs),
                                   [c.dilation[0] for c in convs])
    for c in convs:
        x = conv_lrelu(c, x, negative_slope)
    return x


def _conv_module(conv: nn.Conv2d, x):
    """conv(x) with the module's own bias / padding mode, for an input that may carry the alignment channels of `cat_aligned`"""
    if x.shape[1] == conv.weight.shape[1] or conv.groups != 1:
        return conv(x)
    return conv._conv_forward(x, _match_in_channels(conv.weight, x.shape[1]), conv.bias)


def _unfused() -> bool:
    # under torch.compile / make_fx the reference's own op chain is traced (the compiler fuses bias + activation itself; the in-place
    # ctypes launch cannot be traced) -- the correlation / warp / loss operators trace as torch.ops.ocflow.* (ocflow_amd/torch_ops.py)
    return os.environ.get("OCFLOW_UNFUSED_EPILOGUE", "0") == "1" or torch.compiler.is_compiling()


def conv_bias(conv: nn.Conv2d, x):
    """nn.Conv2d(bias=True) without activation (prediction layers): bias-free convolution + in-place bias kernel"""
    if conv.bias is None or conv.padding_mode != "zeros" or _unfused():
        return _conv_module(conv, x)
    if _native(conv, x):
        return nconv.conv3x3_bias_act(x, conv.weight, conv.bias, 1.0, conv.dilation[0])
    return ops.bias_add(_conv_nobias(conv, x), conv.bias)


def deconv_bias(deconv: nn.ConvTranspose2d, x, output_size=None, sigmoid: bool = False):
    """nn.ConvTranspose2d(bias=True)(x, output_size=...) as a bias-free transposed convolution + in-place bias kernel; `sigmoid=True`:
    torch.sigmoid of it (flow_occ_net.py:62-66), inside the same launches where the repo's small-channel kernels take the layer"""
    act = torch.sigmoid if sigmoid else (lambda t: t)
    w = _match_in_channels(deconv.weight, x.shape[1], dim=0) if deconv.groups == 1 else deconv.weight  # (weight is [in, out, kh, kw])
    output_padding = deconv._output_padding(x, output_size, deconv.stride, deconv.padding, deconv.kernel_size, 2, deconv.dilation)
    if deconv.bias is None or deconv.padding_mode != "zeros" or _unfused():
        return act(F.conv_transpose2d(x, w, deconv.bias, deconv.stride, deconv.padding, output_padding, deconv.groups, deconv.dilation))
    if nconv.deconv_eligible(x, deconv.weight, deconv.stride, deconv.padding, deconv.dilation, deconv.groups, output_padding):
        return nconv.deconv3x3s2_bias(x, deconv.weight, deconv.bias, output_padding, sigmoid)
    if not sigmoid and nconv.deconv4x4_eligible(x, deconv.weight, deconv.stride, deconv.padding, deconv.dilation, deconv.groups, output_padding):
        return nconv.deconv4x4s2_bias(x, deconv.weight, deconv.bias)  # the PWC-style `upfeat` layers (forward; backward: the framework's)
    y = F.conv_transpose2d(x, w, None, deconv.stride, deconv.padding, output_padding, deconv.groups, deconv.dilation)
    return act(ops.bias_add(y, deconv.bias))


def deconv_bias_range(deconv: nn.ConvTranspose2d, x):
    """deconv_bias on the first deconv.in_channels channels of `x` (the PWC-style `upfeat` layers read the level's dense-block buffer
    without its alignment channels).  On ocf_deconv4x4s2_fwd the layer takes the whole buffer, so that its gradient comes back as one
    channels_last tensor of the buffer's shape instead of autograd's NCHW slice backward"""
    ci = int(deconv.weight.shape[0])
    if x.shape[1] == ci:
        return deconv_bias(deconv, x)
    xs = x[:, :ci]
    if deconv.bias is not None and deconv.padding_mode == "zeros" and not _unfused() and deconv.output_padding == (0, 0) \
            and nconv.deconv4x4_eligible(xs, deconv.weight, deconv.stride, deconv.padding, deconv.dilation, deconv.groups, (0, 0)):
        return nconv.deconv4x4s2_bias(x, deconv.weight, deconv.bias)
    return deconv_bias(deconv, xs)


class ConvLReLU(nn.Sequential):
    """nn.Sequential(nn.Conv2d, nn.LeakyReLU) with the same children / state_dict keys ('0.weight', '0.bias')"""

    def forward(self, x):
        return conv_lrelu(self[0], x, self[1].negative_slope)
