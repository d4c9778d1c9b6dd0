"""The estimator / context-network convolutions on this repo's own matrix-core kernels.

`conv3x3_bias_act(x, weight, bias, negative_slope, dilation)` is `F.leaky_relu(F.conv2d(x, weight, bias, 1, d, d), slope)` for the 3x3 /
stride-1 / padding == dilation layers of flow_net.py:11-37, flow_occ_net.py:11-67 and context_net.py:11-17 in the bf16 channels_last
throughput configuration:

  forward          ocf_conv3x3_igemm        convolution + bias + LeakyReLU in one launch
  input gradient   ocf_conv3x3_igemm        the same kernel on grad_out with the flipped / transposed weight
  weight gradient  ocf_conv3x3_wgrad        (fp32, split-K)

The kernels read PACKED weights (bf16 [rows][9][k], zero padded): `WeightPacker` keeps one packed copy per (weight, orientation) and
refreshes every copy of a network with ONE launch per step (ocf_conv3x3_pack_weights) whenever the weights have changed.
There is no CPU path: CPU tensors raise (ops._require_hip)."""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Dict, List, Optional, Tuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from . import ops
from .ops import _t4, _stream, _require_hip, _ptr


def plan(cout: int, b: int, h: int, w: int, dilation: int) -> Tuple[int, int, int]:
    """(16-row tiles per workgroup, patch width, rows of the packed weight it reads) ocf_conv3x3_igemm uses for this output"""
    mt, tw, rows = C.c_int(0), C.c_int(0), C.c_int(0)
    _lib.check(_lib.load().ocf_conv3x3_plan(int(cout), int(b), int(h), int(w), int(dilation), C.byref(mt), C.byref(tw), C.byref(rows)),
               "ocf_conv3x3_plan")
    return mt.value, tw.value, rows.value


def wgrad_plan(b: int, ci: int, co: int, h: int, w: int, dilation: int = 1, policy: int = 0) -> Dict[str, int]:
    """launch geometry of ocf_conv3x3_wgrad_bias for a request: {workgroups, chunks, tiles, segments_per_chunk, chunks_capped}"""
    out = [C.c_int(0) for _ in range(5)]
    _lib.check(_lib.load().ocf_conv3x3_wgrad_plan(int(b), int(ci), int(co), int(h), int(w), int(dilation), int(policy), *[C.byref(o) for o in out]),
               "ocf_conv3x3_wgrad_plan")
    return dict(zip(("workgroups", "chunks", "tiles", "segments_per_chunk", "chunks_capped"), (o.value for o in out)))


def _rows_pad(n: int) -> int:
    """rows of a packed weight: enough for every split of the output channels into equal blocks of <= 8 tiles of 16 rows that
    ocf_conv3x3_plan may choose (it depends on the image size; the packed copy does not)"""
    return ((n + 15) // 16 + 7) * 16


class _Packed:
    """packed copy of one weight in one orientation + its device descriptor"""
    __slots__ = ("dst", "rows", "k", "desc_host", "transpose", "version", "epoch", "src_ref")


class WeightPacker:
    """Packed bf16 copies [rows_pad][9][k_pad] of convolution weights for ocf_conv3x3_igemm, refreshed by one launch.

    `get(weight, transpose)` returns the packed tensor for a weight (registering it on first use); `refresh()` re-packs every registered
    weight whose storage may have changed -- ONE launch for all of them.  `get` refreshes lazily when a weight's version counter moved or
    while a graph is being captured (the pack must then be a node of that graph)."""

    def __init__(self):
        self.entries: Dict[Tuple[int, int], _Packed] = {}
        self._order: List[_Packed] = []
        self._dev_descs: Optional[torch.Tensor] = None
        # descriptor tables a captured graph may still replay a pack launch on: (table, owners) -- owners = weak references to the
        # objects that captured such a launch (`capture_owner`); None = captured by someone unknown: kept for the life of the process.
        # A table no capture ever saw is dropped as soon as it is replaced.
        self._retired: List[Tuple[torch.Tensor, Optional[list]]] = []
        self._table_owners: Optional[list] = []   # owners of the CURRENT table ([]: never captured, None: captured, owner unknown)
        self._capturing_owner = None
        self._total = 0
        self._stamp = -1          # ops-level "weights changed" counter seen by the last refresh
        self._capture_seen = 0

    def _register(self, w: torch.Tensor, transpose: bool, min_rows: int) -> _Packed:
        co, ci = int(w.shape[0]), int(w.shape[1])
        rows, k = (ci, co) if transpose else (co, ci)
        e = _Packed()
        e.rows, e.k = _rows_pad(max(rows, min_rows)), (k + 7) // 8 * 8
        e.dst = torch.zeros((e.rows, 9, e.k), device=w.device, dtype=torch.bfloat16)
        e.transpose = bool(transpose)
        e.version = -1
        e.epoch = -1
        e.src_ref = weakref.ref(w)  # (weak: the packer must not keep the weights of every model a process ever built alive)
        self.entries[(id(w), int(transpose), int(min_rows))] = e
        self._order.append(e)
        self._retire_table()  # descriptor table is rebuilt on the next refresh
        return e

    # ---- lifetime of descriptor tables (`_retired` grew for the life of the process)
    def _retire_table(self):
        if self._dev_descs is not None:
            owners = self._table_owners
            if owners is None:
                self._retired.append((self._dev_descs, None))
            else:
                live = [r for r in owners if r() is not None]
                if live:
                    self._retired.append((self._dev_descs, live))
        self._dev_descs = None
        self._table_owners = []

    def _sweep_retired(self):
        self._retired = [(t, o) for t, o in self._retired if o is None or any(r() is not None for r in o)]

    class _OwnerScope:
        def __init__(self, packer, owner):
            self.packer, self.owner = packer, owner

        def __enter__(self):
            self.prev = self.packer._capturing_owner
            self.packer._capturing_owner = weakref.ref(self.owner)
            return self

        def __exit__(self, *exc):
            self.packer._capturing_owner = self.prev
            return False

    def capture_owner(self, owner):
        """`with PACKER.capture_owner(obj):` around a graph capture: the descriptor tables the captured pack launches read stay alive
        until `obj` is released (`release_owner`) or garbage-collected"""
        return WeightPacker._OwnerScope(self, owner)

    def release_owner(self, owner):
        """the graphs `owner` captured are gone: tables only they could replay are freed"""
        def strip(owners):
            return None if owners is None else [r for r in owners if r() is not None and r() is not owner]
        self._retired = [(t, strip(o)) for t, o in self._retired]
        self._table_owners = strip(self._table_owners)
        self._sweep_retired()

    def _note_capture(self):
        """a pack launch reading the current table is being captured"""
        if self._table_owners is None:
            return
        if self._capturing_owner is None or self._capturing_owner() is None:
            self._table_owners = None
        elif not any(r() is self._capturing_owner() for r in self._table_owners):
            self._table_owners.append(self._capturing_owner)

    def _purge(self):
        """drop the entries whose weight has been garbage-collected (a later tensor may re-use the id)"""
        if any(e.src_ref() is None for e in self._order):
            self._order = [e for e in self._order if e.src_ref() is not None]
            self.entries = {k: e for k, e in self.entries.items() if e.src_ref() is not None}
            self._retire_table()
        self._sweep_retired()

    def _build_table(self, device):
        live = [(e, e.src_ref()) for e in self._order]
        n = len(live)
        nbytes = C.c_int64(0)
        lib = _lib.load()
        _lib.check(lib.ocf_conv3x3_pack_bytes(n, C.byref(nbytes)), "ocf_conv3x3_pack_bytes")
        host = (C.c_char * nbytes.value)()
        first = C.c_int64(0)
        for i, (e, w) in enumerate(live):
            so, si, sky, skx = w.stride()
            nxt = C.c_int64(0)
            _lib.check(lib.ocf_conv3x3_pack_describe(host, i, C.c_void_p(w.data_ptr()), so, si, sky, skx, int(w.shape[0]), int(w.shape[1]),
                                                     1 if w.dtype == torch.float32 else 0, C.c_void_p(e.dst.data_ptr()), e.rows, e.k,
                                                     1 if e.transpose else 0, first, C.byref(nxt)), "ocf_conv3x3_pack_describe")
            first = nxt
        self._total = first.value
        t = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).clone()
        self._retire_table()  # a captured graph may still replay a launch that reads the old table
        self._dev_descs = t.to(device)
        self._ptrs = [(w.data_ptr(), w.dtype) for _, w in live]

    def refresh(self, device=None):
        self._purge()
        if not self._order:
            return
        device = device or self._order[0].dst.device
        live = [e.src_ref() for e in self._order]
        if self._dev_descs is None or len(live) != len(self._ptrs) or any(w is None or (w.data_ptr(), w.dtype) != p for w, p in zip(live, self._ptrs)):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ocflow_amd.conv.WeightPacker: a weight was registered or re-allocated during a graph capture; run the "
                                   "step once eagerly first")
            self._build_table(device)
            live = [e.src_ref() for e in self._order]
        if torch.cuda.is_current_stream_capturing():
            self._note_capture()
        with torch.cuda.device(device):
            _lib.check(_lib.load().ocf_conv3x3_pack_weights(C.c_void_p(self._dev_descs.data_ptr()), len(self._order), self._total,
                                                            C.c_void_p(torch.cuda.current_stream(device).cuda_stream)), "ocf_conv3x3_pack_weights")
        for e, w in zip(self._order, live):
            e.version = w._version
            e.epoch = _OPT_EPOCH[0]

    def get(self, w: torch.Tensor, transpose: bool, min_rows: int = 0) -> _Packed:
        """min_rows: output channels the kernel will be asked to write (the input gradient of a layer fed by a zero-padded
        concatenation has more channels than the weight has inputs: the extra rows are zero)"""
        if w.dtype not in (torch.bfloat16, torch.float32):
            raise TypeError(f"ocflow_amd.conv: weight dtype {w.dtype}")
        e = self.entries.get((id(w), int(transpose), int(min_rows)))
        if e is not None and e.src_ref() is not w:  # the id of a dead tensor, re-used
            self._purge()
            e = self.entries.get((id(w), int(transpose), int(min_rows)))
            if e is not None and e.src_ref() is not w:
                e = None
        if e is None:
            e = self._register(w, transpose, int(min_rows))
        # a weight changes (a) through in-place tensor ops, which move its version counter, or (b) under an optimizer step -- torch's FUSED
        # Adam updates parameters WITHOUT touching the version counter (measured: three eager steps on three-step-old packed weights), so
        # every optimizer.step() in the process moves an epoch (global post-step hook below)
        stale = e.version != w._version or e.epoch != _OPT_EPOCH[0]
        if torch.cuda.is_current_stream_capturing():
            cid = ops._capture_id(w.device)
            stale = stale or self._capture_seen != cid
            if stale:
                self._capture_seen = cid
        if stale:
            self.refresh(w.device)
        return e


_OPT_EPOCH = [0]


def _optimizer_stepped(*_args, **_kw):
    _OPT_EPOCH[0] += 1


def weights_changed():
    """tell the packer that weights may have changed in a way it cannot see (a custom in-place update that bypasses both the tensor
    version counter and torch.optim): the next use re-packs"""
    _OPT_EPOCH[0] += 1


try:  # torch >= 2.0
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_hook
    _reg_hook(_optimizer_stepped)
except ImportError:  # pragma: no cover
    pass

PACKER = WeightPacker()
# kernel choice of ocf_conv3x3_igemm (include/ocflow_hip.h): 0 automatic, 256 force the barrier-free small-map kernel, 512 forbid it,
# + 16 / 32 / 64 force the staged kernel's patch width.  Module default for whole-model A/B runs and for the parity tests.
IGEMM_POLICY = int(os.environ.get("OCFLOW_CONV_POLICY", "0"))


def igemm_raw(x, packed: _Packed, cout: int, bias=None, act=None, dilation: int = 1, negative_slope: float = 1.0, mask_slope: float = 1.0,
              out=None, policy: Optional[int] = None, res=None, res_premask: bool = False, act_channels: Optional[int] = None):
    """ocf_conv3x3_igemm on a channels_last bf16 view `x` [B,Cin,H,W] with a packed weight; returns `out` (allocated channels_last when None).
    res / res_premask / act_channels: the residual epilogue of ocf_conv3x3_igemm_ex (`res` may be `out` itself)"""
    _require_hip(x)
    B, cin, H, W = x.shape
    cin = min(cin, packed.k)  # channels past the packed weight's k (alignment channels of a concatenation) meet no weight
    if policy is None:
        policy = IGEMM_POLICY
    if out is None:
        out = torch.empty((B, cout, H, W), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    b32 = None
    if bias is not None:
        b32 = bias.detach()
        if b32.dtype != torch.float32:
            b32 = b32.float()
    with torch.cuda.device(x.device):
        if res is None and act_channels is None:
            fn = lambda: _lib.check(_lib.load().ocf_conv3x3_igemm(_t4(x), C.c_void_p(packed.dst.data_ptr()), packed.rows, packed.k, _t4(out), _ptr(b32),
                                                                   _t4(act), B, cin, cout, H, W, int(dilation), float(negative_slope),
                                                                   float(mask_slope), _lib.OCF_BF16, int(policy), _stream(x)), "ocf_conv3x3_igemm")
        else:
            ach = int(cout if act_channels is None else act_channels)
            fn = lambda: _lib.check(_lib.load().ocf_conv3x3_igemm_ex(_t4(x), C.c_void_p(packed.dst.data_ptr()), packed.rows, packed.k, _t4(out),
                                                                      _ptr(b32), _t4(act), _t4(res), int(bool(res_premask)), ach, B, cin, cout, H, W,
                                                                      int(dilation), float(negative_slope), float(mask_slope), _lib.OCF_BF16,
                                                                      int(policy), _stream(x)), "ocf_conv3x3_igemm_ex")
        ops._launch("conv3x3_igemm", lambda: ops._meta(f"B{B}x{cin}->{cout}x{H}x{W}d{dilation}", B * H * W * 2 * (cin + cout),
                                                       2 * B * H * W * 9 * cin * cout), fn)
    return out


# ---- gradient sinks: a training loop that keeps ONE flat fp32 gradient buffer (runtime.TrainStepper) hands the views of that buffer in;
# the weight / bias gradient kernels then accumulate straight into them and autograd sees no gradient for those parameters at all (no
# bf16 round trip, no per-parameter copy into the flat buffer).  The owner zeroes the buffer at the start of every step.
# The sinks belong to ONE owner (the stepper that is capturing) and exist only inside its `grad_sinks(...)` block; an entry is keyed by
# id(parameter) but carries a weak reference to the parameter and is honoured only for THAT tensor with THAT shape, so neither a stale
# entry nor a re-used id can route a gradient into someone else's buffer (ADVICE medium).
GRAD_SINK: Dict[int, Tuple["weakref.ref", torch.Tensor]] = {}   # id(parameter) -> (weakref(parameter), fp32 view: its shape and strides)
SEEN_NATIVE: Dict[int, "weakref.ref"] = {}  # parameters whose gradients the kernels of this module produced (see record_native)
_RECORD = [False]
_SINK_OWNER = [None]  # weakref to the owner of the installed sinks


def record_native(flag: bool):
    """while on, the backward passes note which parameters they serve (the stepper asks once, after its warm-up steps)"""
    _RECORD[0] = bool(flag)
    if flag:
        SEEN_NATIVE.clear()


def _seen(p) -> bool:
    r = SEEN_NATIVE.get(id(p))
    return r is not None and r() is p


def install_grad_sinks(params, views, owner=None) -> set:
    """register `views[i]` as the accumulation target of `params[i]` for every parameter seen by the native backward; returns their ids"""
    GRAD_SINK.clear()
    for p, v in zip(params, views):
        if _seen(p) and v.dtype == torch.float32 and v.shape == p.shape and v.device == p.device:
            GRAD_SINK[id(p)] = (weakref.ref(p), v)
    _SINK_OWNER[0] = weakref.ref(owner) if owner is not None else None
    return set(GRAD_SINK)


def clear_grad_sinks(owner=None):
    """drop the sinks (`owner` given: only if they are that owner's)"""
    if owner is not None and _SINK_OWNER[0] is not None and _SINK_OWNER[0]() is not owner:
        return
    GRAD_SINK.clear()
    _SINK_OWNER[0] = None


class grad_sinks:
    """`with grad_sinks(params, views, owner):` -- the sinks exist inside the block only, whatever leaves it (a capture that raises
    must not leave them armed for the next model of the process)"""

    def __init__(self, params, views, owner=None, enabled=True):
        self.args, self.owner, self.enabled, self.ids = (params, views), owner, enabled, set()

    def __enter__(self):
        if self.enabled:
            self.ids = install_grad_sinks(*self.args, owner=self.owner)
        return self.ids

    def __exit__(self, *exc):
        clear_grad_sinks(self.owner)
        return False


def _grad_target(p, device):
    """(fp32 tensor to accumulate the gradient of parameter `p` into, is_sink)"""
    if _RECORD[0]:
        SEEN_NATIVE[id(p)] = weakref.ref(p)
    ent = GRAD_SINK.get(id(p))
    if ent is not None:
        ref, sink = ent
        if ref() is p and sink.shape == p.shape and sink.device == p.device:
            return sink, True
        del GRAD_SINK[id(p)]  # the id of a dead parameter re-used by this one: not its sink
    return ops._take_zeros(p.numel(), device).as_strided(p.shape, p.stride()), False


def wgrad_bias_raw(x, dz, gw, gbias=None, dilation: int = 1, policy: int = 0):
    """ocf_conv3x3_wgrad_bias: gw (fp32 [Co,Ci,3,3], any strides) += dW, gbias (fp32 [Co] or None) += sum dz; both must be zeroed by the caller"""
    _require_hip(x, dz)
    B, _, H, W = x.shape
    co, ci = int(gw.shape[0]), int(gw.shape[1])
    so, si, sky, skx = gw.stride()
    with torch.cuda.device(x.device):
        fn = lambda: _lib.check(_lib.load().ocf_conv3x3_wgrad_bias(_t4(x), _t4(dz), C.c_void_p(gw.data_ptr()), so, si, sky, skx, _ptr(gbias), B, ci, co,
                                                                    H, W, int(dilation), _lib.OCF_BF16, int(policy), _stream(x)),
                                "ocf_conv3x3_wgrad_bias")
        ops._launch("conv3x3_wgrad", lambda: ops._meta(f"B{B}x{ci}->{co}x{H}x{W}d{dilation}", B * H * W * 2 * (ci + co), 2 * B * H * W * 9 * ci * co), fn)
    return gw


# ---- weight gradients beside the input gradients (OFF by default: OCFLOW_WGRAD_STREAM=1 turns it on).  Inside a layer's backward the two
# are independent (both read dz; one reads the saved input, the other the weight), and with gradient sinks armed the weight gradient has
# no consumer before the optimizer: it can be launched on a second stream that forks from the current one (dz is ready) and is joined
# once, after the whole backward (`join_wgrad_stream`, called by TrainStepper); inside a hipGraph capture the fork / join become graph
# edges.  Measured (profiles/r5_wgrad_side_stream_not_default.txt): a replayed graph with ANY such branch is 0.65 - 0.8 ms per step SLOWER
# than the single-stream graph on this runtime (8.95 -> 9.61 ms with only the layers of <= 2048 pixels forked, 9.72 ms with all of them):
# the multi-branch graph loses more in its replay than the concurrency of two matrix-core kernels returns.
_WGRAD_SIDE: Dict[int, "torch.cuda.Stream"] = {}
_WGRAD_PENDING: Dict[int, bool] = {}


def _wgrad_async() -> bool:
    return os.environ.get("OCFLOW_WGRAD_STREAM", "0") == "1" and ops._KERNEL_TIMER is None


def wgrad_bias_async(x, dz, gw, gbias, dilation: int, sinks_only: bool):
    """`wgrad_bias_raw`, on the side stream when every output is a gradient sink (nothing of this backward reads it)"""
    if not (sinks_only and _wgrad_async() and x.shape[0] * x.shape[2] * x.shape[3] <= int(os.environ.get("OCFLOW_WGRAD_STREAM_MAX_PIXELS", str(1 << 30)))):
        return wgrad_bias_raw(x, dz, gw, gbias, dilation)
    dev = x.device.index
    side = _WGRAD_SIDE.get(dev)
    if side is None:
        side = _WGRAD_SIDE[dev] = torch.cuda.Stream(device=x.device)
    side.wait_stream(torch.cuda.current_stream(x.device))
    with torch.cuda.stream(side):
        wgrad_bias_raw(x, dz, gw, gbias, dilation)
    for t in (x, dz):  # their memory must not be handed out again (on the main stream) while the side stream reads it
        t.record_stream(side)
    _WGRAD_PENDING[dev] = True
    return gw


def join_wgrad_stream():
    """the current stream waits for the weight gradients launched beside it (no-op when there are none)"""
    for dev, pending in list(_WGRAD_PENDING.items()):
        if pending:
            torch.cuda.current_stream(torch.device("cuda", dev)).wait_stream(_WGRAD_SIDE[dev])
            _WGRAD_PENDING[dev] = False


_MAX_ELEMS = 1 << 30  # the kernels index a tensor with 32-bit in-image offsets: B*H*W*pitch below 2^30 elements (csrc/conv_igemm.hip,
                      # conv_wgrad3.hip return OCF_EUNSUPPORTED beyond; the predicates below say False first, so callers fall back)


def _pitch(t) -> int:
    B, C_, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    return sw if W > 1 else (sh if H > 1 else (sb if B > 1 else C_))


def _offsets_fit(t, other_channels: int = 0) -> bool:
    B, _, H, W = t.shape
    return B * H * W * max(_pitch(t), int(other_channels)) < _MAX_ELEMS


def wgrad_eligible(x, dz, ci: int, co: int) -> bool:
    """operands ocf_conv3x3_wgrad_bias takes: bf16 pixel-major views with 16-byte aligned pixels that hold the channel counts rounded up to 8"""
    if not (x.dtype == torch.bfloat16 and dz.dtype == torch.bfloat16 and _pixel_major(x, 16) and _pixel_major(dz, 16)):
        return False
    if not (_offsets_fit(x) and _offsets_fit(dz)):
        return False

    def pitch(t):
        B, C_, H, W = t.shape
        sb, sc, sh, sw = t.stride()
        return sw if W > 1 else (sh if H > 1 else (sb if B > 1 else C_))
    return pitch(x) >= (ci + 7) // 8 * 8 and pitch(dz) >= (co + 7) // 8 * 8


def eligible(x, weight, stride, padding, dilation, groups) -> bool:
    """layers ocf_conv3x3_igemm takes: bf16 channels_last activations with a multiple of 8 channels, 3x3, stride 1, padding == dilation <= 16"""
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4):
        return False
    if weight.dtype not in (torch.bfloat16, torch.float32) or tuple(weight.shape[2:]) != (3, 3) or groups != 1 or tuple(stride) != (1, 1):
        return False
    d = tuple(dilation)
    if d[0] != d[1] or tuple(padding) != d or d[0] > 16:
        return False
    extra = x.shape[1] - weight.shape[1]  # alignment channels of `cat_aligned` (zeros)
    if x.shape[1] % 8 != 0 or extra < 0 or extra >= 32:
        return False
    return _pixel_major(x, 16) and _offsets_fit(x, weight.shape[0])


def _pixel_major(t, align: int) -> bool:
    """channel-contiguous with pixels in dense (b, y, x) order at one pixel pitch -- dense channels_last tensors and channel slices of
    them (size-1 dimensions carry arbitrary strides)"""
    B, C_, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    p = sw if W > 1 else (sh if H > 1 else (sb if B > 1 else C_))
    return ((C_ == 1 or sc == 1) and p >= C_ and (H == 1 or sh == p * W) and (B == 1 or sb == p * W * H) and (p * 2) % align == 0
            and t.data_ptr() % align == 0)


def _shape_like(weight, cin, dtype=torch.bfloat16):
    """operand for the framework's convolution backward: the weight in the activations' dtype (plain autocast keeps fp32 weights) and, for
    an input fed by a zero-padded concatenation, with zero input channels appended"""
    if weight.shape[1] == cin:
        return weight if weight.dtype == dtype else weight.to(dtype)
    wp = weight.new_zeros((weight.shape[0], cin, 3, 3), dtype=dtype).contiguous(memory_format=torch.channels_last)
    wp[:, :weight.shape[1]] = weight
    return wp


class _ConvBiasActFn(Function):
    """forward: one launch (convolution + bias + LeakyReLU).  backward: LeakyReLU derivative + bias gradient (ocf_bias_act_bwd), input
    gradient by the same implicit-GEMM kernel on the transposed weight, weight gradient by ocf_conv3x3_wgrad or the framework.
    `cpad` > out_channels: the output is a [B, cpad, H, W] tensor whose channels behind the layer's own are ZERO (one fill + the launch
    into the channel range) -- the alignment channels the next layer's kernels want when the layer has e.g. 196 channels; the incoming
    gradient then has cpad channels too and its tail is ignored."""

    @staticmethod
    def forward(ctx, x, weight, bias, slope, dilation, cpad=0):
        pk = PACKER.get(weight, False)
        co = int(weight.shape[0])
        if cpad > co:
            B, _, H, W = x.shape
            y = torch.zeros((B, cpad, H, W), device=x.device, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
            igemm_raw(x, pk, co, bias, None, dilation, slope, 1.0, out=y[:, :co])
        else:
            y = igemm_raw(x, pk, co, bias, None, dilation, slope, 1.0)
        ctx.save_for_backward(x, weight, y, *([bias] if bias is not None else []))
        ctx.cfg = (float(slope), int(dilation), bias is not None, None if bias is None else bias.dtype)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors[:3]
        slope, d, has_bias, bias_dtype = ctx.cfg
        bias = ctx.saved_tensors[3] if has_bias else None
        B, cy, H, W = y.shape            # cy: channels of the output tensor (> co when it carries alignment zeros)
        co = int(weight.shape[0])
        padded = cy != co
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]
        gg = gy if gy.dtype == y.dtype else gy.to(y.dtype)
        if not _pixel_major(gg, 16 if padded else 2):
            gg = gg.contiguous(memory_format=torch.channels_last)
        # dz = gy * lrelu'(y): one elementwise pass (ocf_bias_act_bwd, which also reduces the bias gradient); a layer without activation
        # needs no pass at all when the weight-gradient kernel takes the bias gradient along
        use_wg3 = need_w and _native_wgrad() and wgrad_eligible(x, gg, weight.shape[1], co)
        gb = None
        if slope != 1.0 or (need_b and (padded or not use_wg3)):
            gb = ops._take_zeros(cy, gy.device)
            dz = torch.empty_like(y) if slope != 1.0 else None
            with torch.cuda.device(gy.device):
                _lib.check(_lib.load().ocf_bias_act_bwd(_t4(y if slope != 1.0 else None), _t4(gg), _t4(dz), C.c_void_p(gb.data_ptr()), B, cy, H, W,
                                                        slope, 1, _lib.OCF_BF16, _stream(gg)), "ocf_bias_act_bwd")
            if dz is None:
                dz = gg
            if padded:
                gb = gb[:co]
        else:
            dz = gg
        gx = gw = None
        if co < 8 and _native_wgrad():
            # prediction layers (2 flow channels, 1 occlusion logit): the kernels read channels in 16-byte chunks, so the gradient goes
            # into a zero-padded 8-channel copy (two small launches) instead of the framework's vector-1 / naive convolution backward
            if ops.concat_channels_eligible([dz]):
                with torch.no_grad():
                    dz = ops.concat_channels([dz], 8)  # (one launch: the copy and the zero channels together)
            else:
                dzp = torch.zeros((B, 8, H, W), device=dz.device, dtype=dz.dtype).contiguous(memory_format=torch.channels_last)
                dzp[:, :co] = dz
                dz = dzp
            use_wg3 = need_w and wgrad_eligible(x, dz, weight.shape[1], co)

        def for_library(t):  # the framework's convolution backward wants exactly the layer's channels, dense
            t = t[:, :co] if t.shape[1] != co else t
            return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)
        if need_x:
            if (cy % 8 == 0 or co < 8) and _pixel_major(dz, 16) and dz.shape[1] % 8 == 0:
                pkt = PACKER.get(weight, True, x.shape[1])
                gx = igemm_raw(dz, pkt, x.shape[1], None, None, d, 1.0, 1.0)
            else:
                gx = torch.ops.aten.convolution_backward(for_library(dz), x, _shape_like(weight, x.shape[1]), None, (1, 1), (d, d), (d, d), False,
                                                         (0, 0), 1, [True, False, False])[0]
        w_sink = b_sink = False
        if need_w and use_wg3 and wgrad_eligible(x, dz, weight.shape[1], co):
            acc, w_sink = _grad_target(weight, gy.device)
            gb2 = None
            if need_b and gb is None:
                gb2, b_sink = _grad_target(bias, gy.device)
                gb = gb2
            gw = wgrad_bias_async(x, dz, acc, gb2, d, w_sink and (gb2 is None or b_sink))
            if w_sink:
                gw = None
            if b_sink:
                gb = None
        elif need_w:
            xc = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            gw = torch.ops.aten.convolution_backward(for_library(dz), xc, _shape_like(weight, x.shape[1]), None, (1, 1), (d, d), (d, d), False,
                                                     (0, 0), 1, [False, True, False])[1]
            if gw.shape[1] != weight.shape[1]:
                gw = gw[:, :weight.shape[1]].contiguous(memory_format=torch.channels_last)
            if gw.dtype != weight.dtype:
                gw = gw.to(weight.dtype)
        if need_b and gb is None and not b_sink:
            gb = for_library(dz).float().sum((0, 2, 3))
        if gb is not None and need_b and bias_dtype != torch.float32:
            gb = gb.to(bias_dtype)
        return gx, gw, (gb if need_b else None), None, None, None


class _ConvChainFn(Function):
    """conv -> bias -> LeakyReLU, n times, the output of each layer feeding only the next (the estimators' conv1..conv5, the context
    network's conv1..conv6: flow_net.py:25-37, flow_occ_net.py:33-45, context_net.py:20-27) as ONE autograd node.
    forward : n launches of ocf_conv3x3_igemm (bias + activation fused).
    backward: per layer ONE ocf_conv3x3_wgrad_bias (weight + bias gradient into fp32) and ONE ocf_conv3x3_igemm whose epilogue applies the
    PREVIOUS layer's LeakyReLU derivative (`act` = that layer's saved output) -- the elementwise leaky_relu_backward / bias-sum pass
    of the per-layer form runs once per chain (for the last layer, whose output has other consumers) instead of once per layer."""

    @staticmethod
    def forward(ctx, x, slopes, dils, *params):
        n = len(slopes)
        ws, bs = params[:n], params[n:]
        ys = []
        h = x
        for i in range(n):
            h = igemm_raw(h, PACKER.get(ws[i], False), ws[i].shape[0], bs[i], None, dils[i], slopes[i], 1.0)
            ys.append(h)
        ctx.save_for_backward(x, *ws, *bs, *ys)
        ctx.cfg = (tuple(slopes), tuple(dils))
        return h

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        slopes, dils = ctx.cfg
        n = len(slopes)
        saved = ctx.saved_tensors
        x, ws, bs, ys = saved[0], saved[1:1 + n], saved[1 + n:1 + 2 * n], saved[1 + 2 * n:]
        dev = gy.device
        B, _, H, W = x.shape
        gg = gy if gy.dtype == torch.bfloat16 else gy.to(torch.bfloat16)
        if not _pixel_major(gg, 16):
            gg = gg.contiguous(memory_format=torch.channels_last)
        gws, gbs = [None] * n, [None] * n
        need_w = [ctx.needs_input_grad[3 + i] for i in range(n)]
        need_b = [ctx.needs_input_grad[3 + n + i] for i in range(n)]
        last_bias_done = False
        # the last layer's activation derivative: the one elementwise pass of the chain (it also reduces that layer's bias gradient)
        if slopes[-1] != 1.0:
            y = ys[-1]
            co = y.shape[1]
            dz = torch.empty_like(y)
            gb_t, b_sink = _grad_target(bs[-1], dev) if need_b[-1] else (ops._take_zeros(co, dev), False)
            with torch.cuda.device(dev):
                _lib.check(_lib.load().ocf_bias_act_bwd(_t4(y), _t4(gg), _t4(dz), C.c_void_p(gb_t.data_ptr()), B, co, H, W, slopes[-1], 1,
                                                        _lib.OCF_BF16, _stream(gg)), "ocf_bias_act_bwd")
            if need_b[-1]:
                gbs[-1] = None if b_sink else gb_t
            last_bias_done = True
        else:
            dz = gg
        gx = None
        for i in range(n - 1, -1, -1):
            xin = ys[i - 1] if i > 0 else x
            w = ws[i]
            if need_w[i] or need_b[i]:
                gw_t, w_sink = _grad_target(w, dev)
                gb_t, b_sink = None, False
                if need_b[i] and not (i == n - 1 and last_bias_done):
                    gb_t, b_sink = _grad_target(bs[i], dev)
                wgrad_bias_async(xin, dz, gw_t, gb_t, dils[i], w_sink and (gb_t is None or b_sink))
                gws[i] = None if w_sink else gw_t
                if gb_t is not None:
                    gbs[i] = None if b_sink else gb_t
            if i > 0:
                masked = slopes[i - 1] != 1.0
                dz = igemm_raw(dz, PACKER.get(w, True, xin.shape[1]), xin.shape[1], None, xin if masked else None, dils[i], 1.0,
                               slopes[i - 1] if masked else 1.0)
            elif ctx.needs_input_grad[0]:
                gx = igemm_raw(dz, PACKER.get(w, True, x.shape[1]), x.shape[1], None, None, dils[i], 1.0, 1.0)
        gbs = [g if (g is None or g.dtype == b.dtype) else g.to(b.dtype) for g, b in zip(gbs, bs)]
        return (gx, None, None, *gws, *gbs)


def conv3x3_chain(x, weights, biases, slopes, dilations):
    """n x (3x3 convolution + bias + LeakyReLU) as one autograd node; the caller checks `chain_eligible` first"""
    return _ConvChainFn.apply(x, tuple(float(s) for s in slopes), tuple(int(d) for d in dilations), *weights, *biases)


def chain_eligible(x, convs) -> bool:
    """every layer of the chain runs on the implicit-GEMM kernels forward AND backward: `eligible` inputs, output channels a multiple of 8
    (the next layer's input and this layer's gradient operand), biases present"""
    if not (enabled() and _native_wgrad()):
        return False
    cin = x.shape[1]
    for k, c in enumerate(convs):
        w = c.weight
        if c.bias is None or c.padding_mode != "zeros" or w.shape[0] % 8 != 0 or tuple(w.shape[2:]) != (3, 3) or c.groups != 1 \
                or tuple(c.stride) != (1, 1) or c.dilation[0] != c.dilation[1] or tuple(c.padding) != tuple(c.dilation) or c.dilation[0] > 16:
            return False
        if k == 0:
            if not eligible(x, w, c.stride, c.padding, c.dilation, c.groups):
                return False
        elif w.shape[1] != cin:
            return False
        cin = w.shape[0]
    return True


class _DenseBlockFn(Function):
    """DenseNet block `for j: x = cat(lrelu(conv_j(x)), x)` (cost_volume_flow_net.py:58-75,176-180 and the pwoc twin) on ONE channels_last
    buffer  [conv_{n-1} | ... | conv_0 | part_0 | part_1 | ... | zeros to a multiple of 32]  as ONE autograd node:

    forward : ocf_concat_channels writes the parts (+ alignment zeros) into the buffer's tail, then layer j reads the channel range
              behind its own output and writes its output in front of it (ocf_conv3x3_igemm on slices): no torch.cat, no copy.
    backward: ONE gradient buffer G of the same layout.  Layer n-1's output gradient = incoming[:, :w] * lrelu'  (one small elementwise
              pass, which also reduces that layer's bias gradient).  Then for j = n-1 .. 0: weight + bias gradient
              (ocf_conv3x3_wgrad_bias) and the input gradient by ocf_conv3x3_igemm_ex, whose epilogue ADDS into G[:, off_j:] -- the first
              of them takes the incoming gradient as its residual -- and multiplies the channels that are activations of this block
              (and the first `lead_mask_channels` of part 0: a cost volume whose LeakyReLU was fused into ocf_corr_fwd) by the
              activation's derivative.  autograd's five cat-backward slicings and gradient-accumulation adds are gone."""

    @staticmethod
    def forward(ctx, n_parts, slope, lead_mask, *args):
        parts, rest = args[:n_parts], args[n_parts:]
        n = len(rest) // 2
        ws, bs = rest[:n], rest[n:]
        B, _, H, W = parts[0].shape
        widths = [int(w.shape[0]) for w in ws]
        tot = sum(widths)
        cin = sum(int(p.shape[1]) for p in parts)
        cin_al = (cin + 31) // 32 * 32
        buf = torch.empty((B, tot + cin_al, H, W), device=parts[0].device, dtype=torch.bfloat16, memory_format=torch.channels_last)
        ops.concat_raw(parts, buf[:, tot:], cin_al)
        off = tot
        for j in range(n):
            igemm_raw(buf[:, off:], PACKER.get(ws[j], False), widths[j], bs[j], None, 1, slope, 1.0, out=buf[:, off - widths[j]:off])
            off -= widths[j]
        ctx.save_for_backward(buf, *ws, *bs)
        ctx.cfg = (n_parts, float(slope), int(lead_mask), [int(p.shape[1]) for p in parts], widths)
        return buf

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        n_parts, slope, lead_mask, part_ch, widths = ctx.cfg
        n = len(widths)
        saved = ctx.saved_tensors
        buf, ws, bs = saved[0], saved[1:1 + n], saved[1 + n:]
        dev = gy.device
        B, ctot, H, W = buf.shape
        tot = sum(widths)
        g0 = gy if gy.dtype == torch.bfloat16 else gy.to(torch.bfloat16)
        if not (_pixel_major(g0, 16) and g0.shape[1] == ctot):
            g0 = g0.contiguous(memory_format=torch.channels_last)
        G = torch.empty_like(buf)
        need_w = [ctx.needs_input_grad[3 + n_parts + j] for j in range(n)]
        need_b = [ctx.needs_input_grad[3 + n_parts + n + j] for j in range(n)]
        gws, gbs = [None] * n, [None] * n
        # the newest layer's output gradient: nothing but the consumers of the block adds into it.  ocf_bias_act_bwd wants dense
        # operands beside the (sliced) incoming gradient: a 32-channel copy of that layer's output, a dense dz
        wl = widths[-1]
        y_last = buf[:, :wl].contiguous(memory_format=torch.channels_last)
        dz_last = torch.empty_like(y_last)
        gb_t, b_sink = _grad_target(bs[-1], dev) if need_b[-1] else (ops._take_zeros(wl, dev), False)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().ocf_bias_act_bwd(_t4(y_last), _t4(g0[:, :wl]), _t4(dz_last), C.c_void_p(gb_t.data_ptr()), B, wl, H, W,
                                                    slope, 1, _lib.OCF_BF16, _stream(g0)), "ocf_bias_act_bwd")
        if need_b[-1]:
            gbs[-1] = None if b_sink else gb_t
        off = 0  # start of layer j's OUTPUT range; its input range starts at off + widths[j]
        for j in range(n - 1, -1, -1):
            o_in = off + widths[j]
            x_j, dz_j = buf[:, o_in:], (dz_last if j == n - 1 else G[:, off:o_in])
            if need_w[j] or need_b[j]:
                gw_t, w_sink = _grad_target(ws[j], dev)
                gb2, b_sink = (None, False)
                if need_b[j] and j != n - 1:
                    gb2, b_sink = _grad_target(bs[j], dev)
                wgrad_bias_async(x_j, dz_j, gw_t, gb2, 1, w_sink and (gb2 is None or b_sink))
                gws[j] = None if w_sink else gw_t
                if gb2 is not None:
                    gbs[j] = None if b_sink else gb2
            cout = ctot - o_in
            first = j == n - 1
            igemm_raw(dz_j, PACKER.get(ws[j], True, cout), cout, None, buf[:, o_in:], 1, 1.0, slope, out=G[:, o_in:],
                      res=(g0[:, o_in:] if first else G[:, o_in:]), res_premask=first, act_channels=(tot - o_in) + lead_mask)
            off = o_in
        gbs = [g if (g is None or g.dtype == b.dtype) else g.to(b.dtype) for g, b in zip(gbs, bs)]
        gparts, c0 = [], tot
        for i, c in enumerate(part_ch):
            gparts.append(G[:, c0:c0 + c] if ctx.needs_input_grad[3 + i] else None)
            c0 += c
        return (None, None, None, *gparts, *gws, *gbs)


def dense_block_eligible(parts, convs, lead_channels: int = 0, like=None) -> bool:
    """the block runs on the slice kernels forward AND backward: bf16 pixel-major parts on a HIP device, 3x3 / stride-1 / padding-1
    layers with bias and a multiple of 8 output channels whose input widths follow the dense pattern.  lead_channels: channels of a
    part 0 that is not computed yet (the cost volume: a dense channels_last tensor with the batch / map size of `like`) and will be put
    in front of `parts` (which may then be empty: level 6)"""
    if not (enabled() and _native_wgrad() and os.environ.get("OCFLOW_NATIVE_DENSE", "1") != "0"):
        return False
    parts = list(parts)
    ref = parts[0] if parts else like
    if ref is None or not (torch.is_tensor(ref) and ref.is_cuda and ref.dim() == 4) or (not parts and not lead_channels):
        return False
    if lead_channels and not (like is not None and like.is_cuda and like.dtype == torch.bfloat16 and like.shape[0] == ref.shape[0]
                              and like.shape[2:] == ref.shape[2:]):
        return False
    if len(parts) + (1 if lead_channels else 0) > 6 or (parts and not ops.concat_channels_eligible(parts)):
        return False
    cin = sum(int(p.shape[1]) for p in parts) + int(lead_channels)
    if (cin + 31) // 32 * 32 > ops.CONCAT_MAX_CHANNELS:
        return False
    tot = sum(int(c.weight.shape[0]) for c in convs)
    B, _, H, W = ref.shape
    if B * H * W * (tot + (cin + 31) // 32 * 32) >= _MAX_ELEMS:
        return False
    for c in convs:
        w = c.weight
        if c.bias is None or c.padding_mode != "zeros" or w.shape[0] % 8 != 0 or tuple(w.shape[2:]) != (3, 3) or c.groups != 1 \
                or tuple(c.stride) != (1, 1) or tuple(c.dilation) != (1, 1) or tuple(c.padding) != (1, 1) or w.shape[1] != cin \
                or w.dtype not in (torch.bfloat16, torch.float32):
            return False
        cin += int(w.shape[0])
    return True


def conv3x3_dense_block(parts, convs, negative_slope: float = 0.1, lead_mask_channels: int = 0):
    """[conv_{n-1}(..) | ... | conv_0(cat(parts)) | cat(parts) | alignment zeros] -- the value of `for c in convs: x = cat(lrelu(c(x)), x)`
    with zero channels appended up to a multiple of 32 -- as one autograd node on one buffer; the caller checks `dense_block_eligible`.
    lead_mask_channels: the first that many channels of parts[0] are the output of a LeakyReLU (same slope) whose derivative this
    node applies to their gradient (ops.compute_cost_volume(..., leaky_slope, grad_premasked=True))."""
    return _DenseBlockFn.apply(len(parts), float(negative_slope), int(lead_mask_channels), *parts, *[c.weight for c in convs],
                               *[c.bias for c in convs])


def conv3x3_bias_act(x, weight, bias=None, negative_slope: float = 1.0, dilation: int = 1, out_channels: int = 0):
    """`F.leaky_relu(F.conv2d(x, weight, bias, 1, dilation, dilation), negative_slope)` (negative_slope = 1: no activation) on the
    implicit-GEMM kernel; the caller checks `eligible` first.  out_channels > weight.shape[0]: the result carries that many channels, the
    ones behind the layer's own zero (see _ConvBiasActFn)"""
    return _ConvBiasActFn.apply(x, weight, bias, float(negative_slope), int(dilation), int(out_channels))


def _native_wgrad() -> bool:
    """OCFLOW_NATIVE_WGRAD3: "1" (default) takes the weight gradient from ocf_conv3x3_wgrad_bias, "0" from the framework"""
    return os.environ.get("OCFLOW_NATIVE_WGRAD3", "1") != "0"


def enabled() -> bool:
    """OCFLOW_NATIVE_CONV: "1" (default) routes eligible layers through these kernels, "0" keeps the framework's convolutions"""
    return os.environ.get("OCFLOW_NATIVE_CONV", "1") != "0"


# ---- the estimators' up-sampling layers (ConvTranspose2d 3x3, stride 2, padding 1, <= 4 output channels): csrc/deconv_small.hip

def deconv_eligible(x, weight, stride, padding, dilation, groups, output_padding) -> bool:
    """layers ocf_deconv3x3s2_* take: bf16 pixel-major x [B, Ci <= 64, H, W], weight [Ci, Co <= 4, 3, 3], stride 2, padding 1"""
    if not (enabled() and os.environ.get("OCFLOW_NATIVE_DECONV", "1") != "0"):
        return False
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and _pixel_major(x, 2)):
        return False
    if weight.dtype not in (torch.bfloat16, torch.float32) or tuple(weight.shape[2:]) != (3, 3) or groups != 1:
        return False
    if tuple(stride) != (2, 2) or tuple(padding) != (1, 1) or tuple(dilation) != (1, 1) or any(p not in (0, 1) for p in output_padding):
        return False
    return weight.shape[0] == x.shape[1] and weight.shape[0] <= 64 and weight.shape[1] <= 4


class _DeconvBiasFn(Function):
    """ConvTranspose2d(3x3, stride 2, padding 1) + bias: forward one launch, backward two (input gradient; weight + bias gradient into fp32)"""

    @staticmethod
    def forward(ctx, x, weight, bias, out_pad, act):
        B, ci, H, W = x.shape
        co = int(weight.shape[1])
        Ho, Wo = 2 * H - 1 + out_pad[0], 2 * W - 1 + out_pad[1]
        y = torch.empty((B, co, Ho, Wo), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        bf = None if bias is None else (bias if bias.dtype == torch.float32 else bias.float())
        si, so, sy, sx = weight.stride()
        with torch.cuda.device(x.device):
            fn = lambda: _lib.check(_lib.load().ocf_deconv3x3s2_fwd(_t4(x), C.c_void_p(weight.data_ptr()), si, so, sy, sx,
                                                                     int(weight.dtype == torch.float32), _ptr(bf), _t4(y), B, ci, co, H, W, Ho, Wo,
                                                                     int(act), _lib.OCF_BF16, _stream(x)), "ocf_deconv3x3s2_fwd")
            ops._launch("deconv3x3s2_fwd", lambda: ops._meta(f"B{B}x{ci}->{co}x{H}x{W}", 2 * B * (H * W * ci + Ho * Wo * co), 0), fn)
        ctx.save_for_backward(x, weight, y if act else None, *([bias] if bias is not None else []))
        ctx.geom = (Ho, Wo, bias is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight, y_act = ctx.saved_tensors[:3]
        Ho, Wo, has_bias = ctx.geom
        bias = ctx.saved_tensors[3] if has_bias else None
        B, ci, H, W = x.shape
        co = int(weight.shape[1])
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]
        gg = gy if gy.dtype == x.dtype else gy.to(x.dtype)
        if not _pixel_major(gg, 2):
            gg = gg.contiguous(memory_format=torch.channels_last)
        gx = torch.empty(x.shape, device=x.device, dtype=x.dtype, memory_format=torch.channels_last) if need_x else None
        gw = gb = None
        w_sink = b_sink = False
        if need_w:
            gw, w_sink = _grad_target(weight, gy.device)
            if need_b:
                gb, b_sink = _grad_target(bias, gy.device)
        si, so, sy, sx = weight.stride()
        gs = gw.stride() if gw is not None else (0, 0, 0, 0)
        with torch.cuda.device(x.device):
            fn = lambda: _lib.check(_lib.load().ocf_deconv3x3s2_bwd(_t4(x), C.c_void_p(weight.data_ptr()), si, so, sy, sx,
                                                                     int(weight.dtype == torch.float32), _t4(gg), _t4(gx), _ptr(gw), gs[0], gs[1], gs[2],
                                                                     gs[3], _ptr(gb), _t4(y_act), B, ci, co, H, W, Ho, Wo, _lib.OCF_BF16,
                                                                     _stream(x)),
                                    "ocf_deconv3x3s2_bwd")
            ops._launch("deconv3x3s2_bwd", lambda: ops._meta(f"B{B}x{ci}->{co}x{H}x{W}", 2 * B * (2 * H * W * ci + 2 * Ho * Wo * co), 0), fn)
        if need_b and gb is None:
            dz = gg.float() if y_act is None else (gg.float() * (1.0 - y_act.float()) * y_act.float()).to(gg.dtype).float()
            gb = dz.sum((0, 2, 3))
        if gb is not None and not b_sink and bias.dtype != torch.float32:
            gb = gb.to(bias.dtype)
        return gx, (None if w_sink else gw), (None if (b_sink or not need_b) else gb), None, None


def deconv3x3s2_bias(x, weight, bias, output_padding, sigmoid: bool = False):
    """`F.conv_transpose2d(x, weight, bias, 2, 1, output_padding)` -- with `sigmoid`: `torch.sigmoid(...)` of it, forward and backward in the
    same launches -- on the small-channel kernels; the caller checks `deconv_eligible` first"""
    return _DeconvBiasFn.apply(x, weight, bias, (int(output_padding[0]), int(output_padding[1])), bool(sigmoid))


# ---- the PWC-style decoders' `upfeat` layers (ConvTranspose2d(Ci ~ 550, 2, 4, 2, 1) on a channel range of the dense-block buffer):
# forward on csrc/deconv4x4.hip (GEMM on the matrix cores + gather), backward through the framework

def deconv4x4_eligible(x, weight, stride, padding, dilation, groups, output_padding) -> bool:
    """layers ocf_deconv4x4s2_fwd takes: bf16 pixel-major x [B, Ci >= 32, H, W] with 16-byte aligned pixels whose pitch covers Ci rounded up
    to 16, weight [Ci, Co <= 2, 4, 4], stride 2, padding 1, no output padding"""
    if not (enabled() and os.environ.get("OCFLOW_NATIVE_DECONV4", "1") != "0"):
        return False
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and _pixel_major(x, 16)):
        return False
    if weight.dtype not in (torch.bfloat16, torch.float32) or tuple(weight.shape[2:]) != (4, 4) or groups != 1:
        return False
    if tuple(stride) != (2, 2) or tuple(padding) != (1, 1) or tuple(dilation) != (1, 1) or any(p != 0 for p in output_padding):
        return False
    ci, co = int(weight.shape[0]), int(weight.shape[1])
    B, _, H, W = x.shape
    return (ci == x.shape[1] and 32 <= ci <= 1000 and co <= 2 and _pitch(x) >= (ci + 15) // 16 * 16 and B * H * W * _pitch(x) < (1 << 31))


class _Deconv4x4Fn(Function):
    """x = channels [0, Ci) of `xfull` (Ci = weight.shape[0]; xfull: the level's dense-block buffer).  The Function takes the WHOLE buffer so
    that its gradient is a channels_last tensor of the buffer's shape (zero behind Ci, one ocf_concat_channels launch): autograd's own
    slice backward fills an NCHW-contiguous tensor and copies into it, and every consumer of that sum then re-lays it out -- 65 - 72 us
    per pass over the 30 MB buffer of level 3, three passes per step (profiles/r5_cfg3_step_breakdown.txt)."""

    @staticmethod
    def forward(ctx, xfull, weight, bias):
        ci, co = int(weight.shape[0]), int(weight.shape[1])
        x = xfull[:, :ci]
        B, _, H, W = x.shape
        y = torch.empty((B, co, 2 * H, 2 * W), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        nb = C.c_int64(0)
        _lib.check(_lib.load().ocf_deconv4x4s2_fwd_workspace_bytes(B, ci, co, H, W, C.byref(nb)), "ocf_deconv4x4s2_fwd_workspace_bytes")
        ws = torch.empty(nb.value, device=x.device, dtype=torch.uint8)
        bf = None if bias is None else (bias if bias.dtype == torch.float32 else bias.float())
        si, so, sy, sx = weight.stride()
        with torch.cuda.device(x.device):
            fn = lambda: _lib.check(_lib.load().ocf_deconv4x4s2_fwd(_t4(x), C.c_void_p(weight.data_ptr()), si, so, sy, sx,
                                                                     int(weight.dtype == torch.float32), _ptr(bf), _t4(y), _ptr(ws), B, ci, co, H, W,
                                                                     _lib.OCF_BF16, _stream(x)), "ocf_deconv4x4s2_fwd")
            ops._launch("deconv4x4s2_fwd", lambda: ops._meta(f"B{B}x{ci}->{co}x{H}x{W}", 2 * B * H * W * (ci + 4 * co), 2 * B * H * W * ci * 16 * co), fn)
        ctx.save_for_backward(xfull, weight)
        ctx.has_bias = bias is not None
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xfull, weight = ctx.saved_tensors
        ci = int(weight.shape[0])
        x = xfull[:, :ci]
        need = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        gg = gy if gy.dtype == x.dtype else gy.to(x.dtype)
        gg = gg if gg.is_contiguous(memory_format=torch.channels_last) else gg.contiguous(memory_format=torch.channels_last)
        xc = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
        wb = weight if weight.dtype == x.dtype else weight.to(x.dtype)
        gx, gw, gb = torch.ops.aten.convolution_backward(gg, xc, wb, [int(weight.shape[1])] if ctx.has_bias else None, (2, 2), (1, 1), (1, 1), True,
                                                         (0, 0), 1, [need[0], need[1], bool(need[2])])
        if gx is not None and xfull.shape[1] != ci:
            cf = int(xfull.shape[1])
            if cf % 8 == 0 and ops.concat_channels_eligible([gx], cf):
                gx = ops._ConcatFn.apply(cf, gx)
            else:
                full = torch.zeros(xfull.shape, device=gx.device, dtype=gx.dtype).contiguous(memory_format=torch.channels_last)
                full[:, :ci] = gx
                gx = full
        if gw is not None and gw.dtype != weight.dtype:
            gw = gw.to(weight.dtype)
        if gb is not None and ctx.bias_dtype is not None and gb.dtype != ctx.bias_dtype:
            gb = gb.to(ctx.bias_dtype)
        return gx, gw, (gb if need[2] else None)


def deconv4x4s2_bias(x, weight, bias):
    """`F.conv_transpose2d(x[:, :weight.shape[0]], weight, bias, 2, 1)` for a 4 x 4 kernel and <= 2 output channels; `x` may be wider than
    the layer (the dense-block buffer it reads a channel range of); the caller checks `deconv4x4_eligible` on that range first"""
    return _Deconv4x4Fn.apply(x, weight, bias)


# ---- the encoder's full-resolution 16 -> 16 layer (csrc/conv_c16.hip): forward and input gradient on the direct matrix-core kernel, weight
# gradient from the framework

def c16_eligible(x, conv) -> bool:
    w = conv.weight
    if not (enabled() and os.environ.get("OCFLOW_NATIVE_C16", "1") != "0" and conv.bias is None):
        return False
    if tuple(w.shape) != (16, 16, 3, 3) or tuple(conv.stride) != (1, 1) or tuple(conv.padding) != (1, 1) or tuple(conv.dilation) != (1, 1) \
            or conv.groups != 1 or conv.padding_mode != "zeros" or w.dtype not in (torch.bfloat16, torch.float32):
        return False
    return (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] == 16 and _pixel_major(x, 16))


def _c16_raw(x, weight, transpose: bool):
    B, _, H, W = x.shape
    y = torch.empty((B, 16, H, W), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    so, si, sky, skx = weight.stride()
    with torch.cuda.device(x.device):
        fn = lambda: _lib.check(_lib.load().ocf_conv3x3_c16(_t4(x), C.c_void_p(weight.data_ptr()), so, si, sky, skx,
                                                            int(weight.dtype == torch.float32), _t4(y), B, H, W, int(transpose), _lib.OCF_BF16,
                                                            _stream(x)), "ocf_conv3x3_c16")
        ops._launch("enc16_conv", lambda: ops._meta(f"B{B}x16->16x{H}x{W}", B * H * W * 64, 2 * B * H * W * 9 * 256), fn)
    return y


class _ConvC16Fn(Function):
    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return _c16_raw(x, weight, False)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gg = gy if gy.dtype == torch.bfloat16 else gy.to(torch.bfloat16)
        if not _pixel_major(gg, 16):
            gg = gg.contiguous(memory_format=torch.channels_last)
        gx = _c16_raw(gg, weight, True) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1] and os.environ.get("OCFLOW_NATIVE_C16_WGRAD", "1") != "0" and _pixel_major(x, 16):
            acc, sink = _grad_target(weight, gy.device)  # fp32, zeroed (or the stepper's flat gradient buffer)
            B, _, H, W = x.shape
            so, si, sky, skx = acc.stride()
            nb = C.c_int64(0)
            _lib.check(_lib.load().ocf_conv3x3_c16_wgrad_workspace_bytes(B, H, W, C.byref(nb)), "ocf_conv3x3_c16_wgrad_workspace_bytes")
            ws = torch.empty(nb.value, device=x.device, dtype=torch.uint8)
            with torch.cuda.device(x.device):
                fn = lambda: _lib.check(_lib.load().ocf_conv3x3_c16_wgrad(_t4(x), _t4(gg), C.c_void_p(acc.data_ptr()), so, si, sky, skx,
                                                                          C.c_void_p(ws.data_ptr()), B, H, W, _lib.OCF_BF16, _stream(x)),
                                        "ocf_conv3x3_c16_wgrad")
                ops._launch("enc16_wgrad", lambda: ops._meta(f"B{B}x16->16x{H}x{W}", B * H * W * 64, 2 * B * H * W * 9 * 256), fn)
            gw = None if sink else acc
        elif ctx.needs_input_grad[1]:
            ggc = gg if gg.is_contiguous(memory_format=torch.channels_last) else gg.contiguous(memory_format=torch.channels_last)
            xc = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            gw = torch.ops.aten.convolution_backward(ggc, xc, _shape_like(weight, 16), None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                     [False, True, False])[1]
            if gw.dtype != weight.dtype:
                gw = gw.to(weight.dtype)
        return gx, gw


def conv3x3_c16(x, weight):
    """`F.conv2d(x, weight, None, 1, 1)` for the 16 -> 16 layer; the caller checks `c16_eligible` first"""
    return _ConvC16Fn.apply(x, weight)


# ---- the encoder's first layer (3 -> 16, stride 2): forward on csrc/conv_c16.hip's c3 kernel, weight gradient from the framework

def c3_eligible(x, conv) -> bool:
    w = conv.weight
    if not (enabled() and os.environ.get("OCFLOW_NATIVE_C3", "1") != "0" and conv.bias is None):
        return False
    if tuple(w.shape) != (16, 3, 3, 3) or tuple(conv.stride) != (2, 2) or tuple(conv.padding) != (1, 1) or tuple(conv.dilation) != (1, 1) \
            or conv.groups != 1 or conv.padding_mode != "zeros" or w.dtype not in (torch.bfloat16, torch.float32):
        return False
    return (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] == 3 and not x.requires_grad
            and x.is_contiguous(memory_format=torch.channels_last))


class _ConvC3Fn(Function):
    @staticmethod
    def forward(ctx, x, weight):
        B, _, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, 16, Ho, Wo), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
        so, si, sky, skx = weight.stride()
        with torch.cuda.device(x.device):
            fn = lambda: _lib.check(_lib.load().ocf_conv3x3s2_c3(_t4(x), C.c_void_p(weight.data_ptr()), so, si, sky, skx,
                                                                 int(weight.dtype == torch.float32), _t4(y), B, H, W, _lib.OCF_BF16, _stream(x)),
                                    "ocf_conv3x3s2_c3")
            ops._launch("enc3_conv_s2", lambda: ops._meta(f"B{B}x3->16x{H}x{W}", B * (H * W * 6 + Ho * Wo * 32), 2 * B * Ho * Wo * 27 * 16), fn)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gw = None
        if ctx.needs_input_grad[1] and os.environ.get("OCFLOW_NATIVE_C3_WGRAD", "1") != "0":
            gg = gy if gy.dtype == torch.bfloat16 else gy.to(torch.bfloat16)
            if not _pixel_major(gg, 16):
                gg = gg.contiguous(memory_format=torch.channels_last)
            acc, sink = _grad_target(weight, gy.device)
            B, _, H, W = x.shape
            so, si, sky, skx = acc.stride()
            nb = C.c_int64(0)
            _lib.check(_lib.load().ocf_conv3x3s2_c3_wgrad_workspace_bytes(B, H, W, C.byref(nb)), "ocf_conv3x3s2_c3_wgrad_workspace_bytes")
            ws = torch.empty(nb.value, device=x.device, dtype=torch.uint8)
            with torch.cuda.device(x.device):
                fn = lambda: _lib.check(_lib.load().ocf_conv3x3s2_c3_wgrad(_t4(x), _t4(gg), C.c_void_p(acc.data_ptr()), so, si, sky, skx,
                                                                           C.c_void_p(ws.data_ptr()), B, H, W, _lib.OCF_BF16, _stream(x)),
                                        "ocf_conv3x3s2_c3_wgrad")
                ops._launch("enc3_wgrad", lambda: ops._meta(f"B{B}x3->16x{H}x{W}", B * (H * W * 6 + gg.shape[2] * gg.shape[3] * 32), 0), fn)
            gw = None if sink else acc
        elif ctx.needs_input_grad[1]:
            gg = gy if gy.dtype == torch.bfloat16 else gy.to(torch.bfloat16)
            gg = gg if gg.is_contiguous(memory_format=torch.channels_last) else gg.contiguous(memory_format=torch.channels_last)
            gw = torch.ops.aten.convolution_backward(gg, x, _shape_like(weight, 3), None, (2, 2), (1, 1), (1, 1), False, (0, 0), 1,
                                                     [False, True, False])[1]
            if gw.dtype != weight.dtype:
                gw = gw.to(weight.dtype)
        return None, gw


def conv3x3s2_c3(x, weight):
    """`F.conv2d(x, weight, None, 2, 1)` for the encoder's first layer; the caller checks `c3_eligible` first"""
    return _ConvC3Fn.apply(x, weight)
