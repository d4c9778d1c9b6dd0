/* ocflow_hip.h -- C ABI of libocflow_hip.so: the MI355X (gfx950) hot path of OCFlow.
 *
 * Every entry point is a stream-ordered launcher: plain device pointers + explicit element
 * strides + sizes + a hipStream_t passed as void*.  No torch types, no allocation, no host
 * synchronisation, RE-ENTRANT (no process-wide state: where a request can be served by several
 * kernels the choice is a per-call `policy` argument, 0 = automatic), graph-capturable.  Return value: 0 = OCF_OK, negative = error
 * (never throws).  Each function names the reference interface (file:line under the OCFlow
 * tree) it replaces; the reference-side binding is shown in INTEGRATION.md.
 *
 * Conventions
 *   - tensors are logical NCHW [B,C,H,W] described by ocf_tensor4 {data, stride_b, stride_c,
 *     stride_h, stride_w} with strides in ELEMENTS, so NCHW-contiguous tensors, channels_last
 *     tensors, channel slices of a wider buffer and views such as x[:, :3] are all passed
 *     without a copy.  Fast paths: stride_c == 1 (channels_last) with 16-byte aligned pixels,
 *     or stride_w == 1 (NCHW).
 *   - feature tensors (f1, f2, img of the warp, cost volume, their gradients) have element
 *     type `dtype` (OCF_F32 or OCF_BF16); accumulation is always fp32.
 *   - flow  [B,2,H,W] (ch0 = u, x-displacement; ch1 = v, y-displacement), occ [B,1,H,W],
 *     loss-side images and all loss gradients are fp32.
 *   - reductions return fp64 partial sums in device memory (`double*`), zeroed by the
 *     launcher with hipMemsetAsync on `stream`.
 */
#ifndef OCFLOW_HIP_H
#define OCFLOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCF_ABI_VERSION 3 /* 2: per-call kernel policies replace the process-wide ocf_*_select switches of version 1;
                             3: ocf_corr_fwd takes `leaky_slope`; ocf_conv3x3_igemm_ex, ocf_conv3x3_wgrad_plan added */
#define OCF_MAX_DISPLACEMENT 4

enum ocf_dtype { OCF_F32 = 0, OCF_BF16 = 1 };

enum ocf_status {
  OCF_OK = 0,
  OCF_EINVAL = -1,       /* null pointer / non-positive size / bad enum                      */
  OCF_EUNSUPPORTED = -2, /* valid request this build has no kernel for (e.g. displacement>4) */
  OCF_ELAUNCH = -3,      /* HIP reported an error at launch (hipGetLastError)                */
  OCF_ENODEVICE = -4     /* no HIP device / wrong architecture                               */
};

typedef struct ocf_tensor4 {
  void* data;
  int64_t sb, sc, sh, sw; /* element strides of the logical [B,C,H,W] tensor */
} ocf_tensor4;

int ocf_abi_version(void);
const char* ocf_status_string(int status);
/* device name + CU count of the current device; returns OCF_ENODEVICE when there is none. */
int ocf_device_info(char* name, int name_len, int* compute_units);

/* Profiling aid: the NEXT launcher call on this thread records `start_event` immediately before and `stop_event`
 * immediately after its kernel launch(es) on the launch stream (hipEvent_t handles passed as void*; one-shot).  Events
 * recorded from the host language sit tens of microseconds of interpreter time away from the launch and, when the GPU is
 * waiting for the host, count that idle gap as kernel time. */
int ocf_profile_next_launch(void* start_event, void* stop_event);

/* *id = identifier of the graph capture `stream` is currently recording into, 0 when it is not capturing.  Host-side caches of
 * pre-initialised device buffers (the zeroed accumulators handed to ocf_bias_act_bwd) must not outlive the capture they were
 * filled in: the fill is a node of THAT graph and is not replayed by another one. */
int ocf_stream_capture_id(void* stream, unsigned long long* id);

/* ---- K1+K2  local cost volume, optionally fused with the backward warp of f2 -------------
 * out[b,(dy+d)(2d+1)+(dx+d),y,x] = 1/C * sum_c f1[b,c,y,x] * w2[b,c,y+dy,x+dx]   (0 outside)
 *   w2 = f2                                            when flow == NULL
 *   w2 = warp(f2, flow*flow_scale, align_corners)      when flow != NULL
 *   w2 = w2 * occ  (occ broadcast over channels)       when occ  != NULL
 * replaces: compute_cost_volume        models/networks/correlation_layer.py:7-40
 *           CostVolumeLayer.forward    models/networks/cost_volume_net.py (missing module; call
 *                                      sites flow_occ_net.py:168, cost_volume_flow_occ_net.py:188)
 *           warp -> [*occ] -> corr     flow_net.py:110-112, cost_volume_flow_occ_net.py:204-206
 *           ... -> LeakyReLU(0.1)      cost_volume_flow_net.py:173,189, cost_volume_flow_occ_net.py:189, pwc_net.py:158
 * d <= OCF_MAX_DISPLACEMENT.  out has (2d+1)^2 channels and element type `dtype`.
 * leaky_slope: 0 = none; otherwise out = LeakyReLU(out, leaky_slope) applied in the kernel's store epilogue (0 < leaky_slope <= 1).  The
 * gradient of such a volume w.r.t. the un-activated one is grad * (out > 0 ? 1 : leaky_slope): ocf_bias_act_bwd does that in one pass, or the
 * consumer's input-gradient epilogue does it for free (ocf_conv3x3_igemm_ex, `act_channels`). */
int ocf_corr_fwd(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ,
                 const ocf_tensor4* out, int B, int C, int H, int W, int d, float flow_scale, int align_corners,
                 float leaky_slope, int dtype, int policy /* enum ocf_corr_policy */, void* stream);

/* Kernel selection of ocf_corr_fwd for bf16, PER CALL (results are the same under every policy, only the kernel differs -- for A/B
 * measurements and for testing the kernels the automatic choice would not pick). */
enum ocf_corr_policy {
  OCF_CORR_POLICY_AUTO = 0,
  OCF_CORR_POLICY_BAND_ONLY = 1, /* the band-GEMM kernels for everything                                      */
  OCF_CORR_POLICY_NO_WAVE = 2,   /* automatic choice without the wave kernel (row / direct / band kernels)   */
  OCF_CORR_POLICY_WAVE = 3       /* the wave kernel whenever its preconditions hold (also for C <= 64)       */
};
/* Diagnostics: which kernel ocf_corr_fwd launches for a request (same arguments, nothing is launched).  The parity tests assert
 * that every instantiation the benchmark times is the one they value-check. */
enum ocf_corr_variant {
  OCF_CORR_ROWS_K1 = 1,         /* bf16 row kernel, one 32-channel block   (dense channels_last, d = 4, C <= 32)  */
  OCF_CORR_ROWS_K2 = 2,         /* bf16 row kernel, two blocks resident    (dense channels_last, d = 4, C <= 64)  */
  OCF_CORR_DIRECT = 3,          /* bf16 direct kernel: fragments straight from global memory (d = 4, 64 < C <= 128, or <= 224 for W <= 16) */
  OCF_CORR_WAVE = 4,            /* bf16 wave kernel: one wave per (row, strip, dy), no LDS; small levels (C > 64), any output pixel pitch,
                                   optional in-register warp of the second operand (d = 4, C <= 224)                                   */
  OCF_CORR_BAND_TH8 = 11,       /* bf16 band-GEMM kernel, C <= 32, 32x8 tiles, persistent multi-tile walk         */
  OCF_CORR_BAND_MULTI_TH4 = 12, /* bf16 band-GEMM kernel, C > 32, 4-row tiles                                      */
  OCF_CORR_BAND_MULTI_TH2 = 13, /*   ... 2-row tiles (fewer 4-row tiles than CUs)                                  */
  OCF_CORR_BAND_MULTI_TH1 = 14, /*   ... 1-row tiles (fewer than a quarter of that)                                */
  OCF_CORR_VECTOR_F32 = 20,     /* fp32 vector-ALU kernel                                                          */
  OCF_CORR_VECTOR_BF16 = 21     /* the same kernel on bf16 operands whose layout gives no 8- or 16-byte channel runs (NCHW, odd strides) */
};
int ocf_corr_fwd_variant(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ,
                         const ocf_tensor4* out, int B, int C, int H, int W, int d, int dtype, int policy);

/* gradient of the plain cost volume w.r.t. both operands (w2 materialised by the caller):
 *   g1[b,c,y,x] = 1/C sum_k gout[b,k,y,x]       * w2[b,c,y+dy,x+dx]
 *   g2[b,c,y,x] = 1/C sum_k gout[b,k,y-dy,x-dx] * f1[b,c,y-dy,x-dx]
 * replaces: autograd of correlation_layer.py:33-39 (81 x MulBackward/MeanBackward + CatBackward) */
enum ocf_corr_bwd_policy {
  OCF_CORR_BWD_POLICY_AUTO = 0,
  OCF_CORR_BWD_POLICY_VECTOR = 1, /* the vector-ALU kernels only                                                                          */
  OCF_CORR_BWD_POLICY_MFMA = 2    /* the matrix-core band-GEMM kernel whenever its preconditions hold (bf16, d = 4, channel-contiguous
                                     operands, 8-byte aligned pixels) -- what the automatic choice does as well                            */
};
enum ocf_corr_bwd_variant {
  OCF_CORR_BWD_MFMA = 1,    /* corr_bwd_mfma.hip: band GEMMs on the matrix cores */
  OCF_CORR_BWD_TILED16 = 2, /* corr.hip LDS-tiled vector kernel, 16-byte units   */
  OCF_CORR_BWD_TILED8 = 3,  /*   ... 8-byte units (C = 196 bf16)                 */
  OCF_CORR_BWD_GATHER = 4   /* global-memory gather kernel (strided layouts)     */
};
int ocf_corr_bwd(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* gout, const ocf_tensor4* g1,
                 const ocf_tensor4* g2, int B, int C, int H, int W, int d, int dtype, int policy, void* stream);
/* which kernel (enum ocf_corr_bwd_variant) ocf_corr_bwd / ocf_corr_occ_bwd launch for a request; nothing is launched */
int ocf_corr_bwd_variant(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* gout, const ocf_tensor4* g1,
                         const ocf_tensor4* g2, int B, int C, int H, int W, int d, int dtype, int policy);

/* the same with the occlusion mask of the forward (out = corr(f1, w2 * occ), occ fp32 [B,1,H,W]) folded in: `w2` is the UNMASKED
 * operand, the product w2*occ is formed while staging, g2 is the gradient w.r.t. w2 (= d/d(w2*occ) * occ) and
 * gocc [B,1,H,W] fp32 dense (optional, zeroed by the launcher) receives sum_c d/d(w2*occ)[c] * w2[c].
 * Only for layouts the LDS-tiled kernels take (channels_last, pixels a multiple of 8 bytes); otherwise OCF_EUNSUPPORTED and the
 * caller composes it from ocf_corr_bwd and elementwise ops.
 * replaces: autograd of `self.corr(f1, warped * occ)` flow_occ_net.py:167-168, cost_volume_flow_occ_net.py:204-206 */
int ocf_corr_occ_bwd(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* occ, const ocf_tensor4* gout,
                     const ocf_tensor4* g1, const ocf_tensor4* g2, const ocf_tensor4* gocc, int B, int C, int H, int W, int d,
                     int dtype, int policy, void* stream);

/* ---- K2  backward bilinear warp (zeros padding), both align_corners variants --------------
 * out[b,c,y,x] = bilinear(img[b,c], px, py), neighbours outside the image contribute 0
 *   align_corners=0: px = (x+s*u)*W/max(W-1,1) - 0.5, py likewise     (in-network feature warp)
 *   align_corners=1: px = (x+s*u)*(W-1)/max(W-1,1),   py likewise     (loss-side image warp)
 * replaces: warp()  cost_volume_flow_net.py:121-151, cost_volume_flow_occ_net.py:137-167,
 *           flow_net.py:57-87, flow_occ_net.py:96-126 (AC-False); models/model.py:191-221,
 *           flow_model.py:49-79 (AC-True)  [grid build + F.grid_sample] */
int ocf_warp_fwd(const ocf_tensor4* img, const ocf_tensor4* flow, const ocf_tensor4* out, int B, int C, int H, int W,
                 float flow_scale, int align_corners, int dtype, void* stream);
/* gimg  : fp32 [B,C,H,W]; the scatter kernel accumulates into it with float atomics and needs it ZEROED by the caller, the gather
 *         kernel pair (channels_last, >= 1M (pixel, channel) pairs) overwrites it: ocf_warp_bwd_plan() says which (bit 0 of its
 *         result = "zero grad_img first"; a caller that always zeroes is always correct);
 * gflow : fp32 [B,2,H,W], written (gradient w.r.t. the UNSCALED flow, i.e. includes s). */
/* Kernel selection of ocf_warp_bwd for channels_last operands, per call (same results up to float summation order; the SAME policy must
 * be given to ocf_warp_bwd_plan and to ocf_warp_bwd -- both evaluate one decision function, grid limits included):
 * AUTO = the gather kernel (per-destination-pixel lists in LDS, plain stores; sources whose taps lie within 6 pixels) + the per-source kernel
 *     (grad_flow, float atomics for the remaining far sources) when the tensor has >= 1M (pixel, channel) pairs, else the scatter kernel;
 * SCATTER = scatter kernel (global float atomics) always;  GATHER = gather pair whenever the layout allows -- for A/B measurements and
 * for testing both paths at any size. */
enum ocf_warp_bwd_policy {
  OCF_WARP_BWD_POLICY_AUTO = 0,
  OCF_WARP_BWD_POLICY_SCATTER = 1,
  OCF_WARP_BWD_POLICY_GATHER = 2, /* window gather (near sources) + float atomics for far sources, no workspace          */
  OCF_WARP_BWD_POLICY_BINNED = 3, /* binned gather (needs `workspace`; falls back to the window form when it is NULL)  */
  OCF_WARP_BWD_POLICY_SMALL = 4,  /* behaves as AUTO (named for tests and A/B runs): see the note on small maps below              */
  OCF_WARP_BWD_GIMG_NATIVE = 0x100 /* flag, OR-ed to the policy: grad_img is given (and written) in `dtype` instead of fp32; only valid
                                     when ocf_warp_bwd_plan() returned bit 1 for the same request */
};
/* AUTO additionally: channels_last maps of at most 1792 pixels per image (the three coarsest pyramid levels) take the single-launch
 * kernel at ANY channel count (grad_img by list gather over the whole image, grad_flow by further workgroups of the same grid).
 * ocf_warp_bwd_plan() result: bit 0 = zero grad_img first; bit 1 = grad_img may be passed in `dtype` (with OCF_WARP_BWD_GIMG_NATIVE). */
/* Scratch of the BINNED gather of ocf_warp_bwd (any contents, caller-allocated, at least this many bytes): a binning pass lists, per 32 x 8
 * destination tile, the sources whose bilinear taps land in it; the gather kernel walks that list instead of a fixed +-6 pixel window, so
 * sources with ANY displacement are gathered with plain stores (the window form sent 25-44 % of the sources of an untrained model through
 * float atomics).  With a workspace the automatic policy uses this form wherever it used the window form. */
int ocf_warp_bwd_workspace_bytes(int B, int C, int H, int W, int64_t* bytes);
int ocf_warp_bwd_plan(const ocf_tensor4* img, const ocf_tensor4* gout, const ocf_tensor4* gimg, const ocf_tensor4* gflow, int B, int C, int H,
                      int W, int dtype, int policy);
int ocf_warp_bwd(const ocf_tensor4* img, const ocf_tensor4* flow, const ocf_tensor4* gout, const ocf_tensor4* gimg,
                 const ocf_tensor4* gflow, int B, int C, int H, int W, float flow_scale, int align_corners, int dtype,
                 int policy, void* workspace /* may be NULL */, void* stream);

/* Deterministic variant: the same contributions accumulated through 64-bit fixed-point integer atomics
 * (2^-36 resolution, |sum| < 1.3e8) in a caller-provided workspace of ocf_warp_bwd_det_workspace_bytes() bytes (zeroed by the
 * launcher), then converted to fp32: results are bit-identical from run to run.  gimg need NOT be zeroed; any strides. */
int ocf_warp_bwd_det_workspace_bytes(int B, int C, int H, int W, int64_t* bytes);
int ocf_warp_bwd_det(const ocf_tensor4* img, const ocf_tensor4* flow, const ocf_tensor4* gout, const ocf_tensor4* gimg,
                     const ocf_tensor4* gflow, void* workspace, int B, int C, int H, int W, float flow_scale, int align_corners,
                     int dtype, void* stream);

/* ---- K4  Charbonnier photometric error, optionally fused with the AC-True image warp ------
 * e = sqrt((P - img1)^2 + 1e-6),  P = img2 (flow==NULL) or warp_ac1(img2, flow)
 * sums[0] = sum e*(1-occ)   sums[1] = sum (1-occ)   (occ==NULL: occ := 0, sums[1] = B*H*W)
 * replaces: FlowStageModel.warp + robust_l1 + photometric_error   models/model.py:191-221, 27-46
 *           (call sites model.py:327-330, 355-358, 379-394) */
int ocf_photo_loss_fwd(const ocf_tensor4* img1, const ocf_tensor4* img2, const ocf_tensor4* flow,
                       const ocf_tensor4* occ, double* sums, int B, int C, int H, int W, void* stream);
/* scale: device fp32 scalar = dLoss/d(sums[0]);  writes gflow [B,2,H,W] (flow != NULL) and/or
 * gpred [B,C,H,W] = dLoss/dP (flow == NULL only).  Either output pointer may be NULL. */
int ocf_photo_loss_bwd(const ocf_tensor4* img1, const ocf_tensor4* img2, const ocf_tensor4* flow,
                       const ocf_tensor4* occ, const float* scale, const ocf_tensor4* gflow, const ocf_tensor4* gpred,
                       int B, int C, int H, int W, void* stream);

/* ---- masked L1 of the inpainting head (BASELINE configs[3]; "next" row 8f-4, first slice) ------------------------------------
 * sums[0] = sum_{b,c,y,x} |pred - target| * mask[b,0,y,x]   sums[1] = sum mask     (fp64, zeroed by the launcher)
 * loss = sums[0] / (C * sums[1] + 1e-16) is formed by the caller.  pred / target [B,C,H,W], mask [B,1,H,W], all fp32, any strides.
 * backward: gpred = scale * sign(pred - target) * mask  with scale a device fp32 scalar (dLoss/d sums[0]).
 * replaces: `(torch.abs(img_completed-img1) * occ).sum() / (3*occ.sum() + 1e-16)`  models/inpainting_model.py:71 */
int ocf_masked_l1_fwd(const ocf_tensor4* pred, const ocf_tensor4* target, const ocf_tensor4* mask, double* sums, int B, int C, int H,
                      int W, void* stream);
int ocf_masked_l1_bwd(const ocf_tensor4* pred, const ocf_tensor4* target, const ocf_tensor4* mask, const float* scale,
                      const ocf_tensor4* gpred, int B, int C, int H, int W, void* stream);

/* ---- K5  edge-aware first + second order smoothness, one pass ----------------------------
 * sums[0] = sum wx *charb(dxF)   over [B,2,H,W-1]     sums[1] = sum wy *charb(dyF)  over [B,2,H-1,W]
 * sums[2] = sum wxx*charb(dxxF)  over [B,2,H,W-2]     sums[3] = sum wyy*charb(dyyF) over [B,2,H-2,W]
 * w = exp(-mean_c (alpha*dI)^2), stride-1 image differences for order 1, stride-2 for order 2.
 * replaces: gradient, first_order_smoothness_loss, second_order_smoothness_loss  models/model.py:53-66, 93-114 */
int ocf_smooth_loss_fwd(const ocf_tensor4* img, const ocf_tensor4* flow, double* sums, int B, int C, int H, int W,
                        float alpha, void* stream);
/* coef: device fp32[4] = dLoss/d(sums[i]); writes gflow [B,2,H,W] */
int ocf_smooth_loss_bwd(const ocf_tensor4* img, const ocf_tensor4* flow, const float* coef, const ocf_tensor4* gflow,
                        int B, int C, int H, int W, float alpha, void* stream);

/* ---- K6  range map (forward splat of bilinear weights) + occlusion mask -------------------
 * range[b,0,ty,tx] += w for the 4 integer neighbours of (x+u, y+v) that lie inside the image;
 * occ = 1 - clamp(range, 0, 1) (written when occ != NULL).  range/occ: contiguous fp32 [B,1,H,W];
 * range is zeroed by the launcher.
 * replaces: flow_to_warp + compute_range_map  models/model.py:223-305 (dup flow_model.py:81-163), model.py:391 */
/* policy (per call, same results up to float summation order): AUTO / SPLAT = LDS-window splat (99 us at 8 x 436 x 1024), GATHER =
 * per-pixel gather through LDS lists + float atomics for sources displaced by more than 8 pixels (137 us; its near-source part is
 * bit-reproducible) */
enum ocf_range_policy { OCF_RANGE_POLICY_AUTO = 0, OCF_RANGE_POLICY_SPLAT = 1, OCF_RANGE_POLICY_GATHER = 2 };
int ocf_range_map(const ocf_tensor4* flow, float* range, float* occ, int B, int H, int W, int policy, void* stream);
/* deterministic variant (fixed-point integer atomics, see ocf_warp_bwd_det); workspace: ocf_range_map_det_workspace_bytes() */
int ocf_range_map_det_workspace_bytes(int B, int H, int W, int64_t* bytes);
int ocf_range_map_det(const ocf_tensor4* flow, float* range, float* occ, void* workspace, int B, int H, int W, void* stream);

/* ---- K3  moments for normalize_features ---------------------------------------------------
 * sums[2*b+0] = sum over (C,H,W) of f[b],  sums[2*b+1] = sum of squares  (fp64, B pairs)
 * replaces: torch.var_mean(dim=(1,2,3)) in models/networks/correlation_layer.py:59-64 */
int ocf_moments(const ocf_tensor4* f, double* sums, int B, int C, int H, int W, int dtype, void* stream);

/* ---- K3  normalize_features as kernels --------------------------------------
 * `sums`: fp64 [S][B][2], the ocf_moments results of the S maps of the call (the reference normalises the two feature maps of a level
 * together: ONE mean and ONE variance for both maps and the whole batch).  With m1 = sum / n, m2 = sumsq / n, n = C*H*W:
 *   mu = mean_{s,b} m1[s][b],  var = mean_{s,b} (m2[s][b] - m1[s][b]^2),  inv_std = 1 / sqrt(var + 1e-16)
 * forward (per map):  y = (x - mu) * inv_std
 * backward: ocf_norm_bwd_sums accumulates bsums[0] += sum g, bsums[1] += sum g * (x - mu) (call it for every map; zero_first != 0 on the
 *   first), then per map  gx = g * inv_std + dmu + 2 dvar / (S B n) * (x - m1[which][b]),  dmu = -inv_std * bsums[0] / (S B n),
 *   dvar = -0.5 * bsums[1] * inv_std^3   -- the exact gradient through mu and var, as autograd of the reference computes it.
 * x / y / gy / gx: dense NCHW-contiguous or channels_last tensors of `dtype` with identical strides, 16-byte aligned, C*H*W*elem % 16 == 0
 * (anything else: OCF_EUNSUPPORTED, the caller composes it from ocf_moments and elementwise ops).
 * replaces: normalize_features  models/networks/correlation_layer.py:59-82 (2 x var_mean, stack, mean, sub, div) and its autograd */
int ocf_norm_fwd(const ocf_tensor4* x, const ocf_tensor4* y, const double* sums, int S, int which, int B, int C, int H, int W, int dtype,
                 void* stream);
int ocf_norm_bwd_sums(const ocf_tensor4* x, const ocf_tensor4* gy, const double* sums, double* bsums, int zero_first, int S, int B, int C, int H,
                      int W, int dtype, void* stream);
int ocf_norm_bwd(const ocf_tensor4* x, const ocf_tensor4* gy, const ocf_tensor4* gx, const double* sums, const double* bsums, int S, int which,
                 int B, int C, int H, int W, int dtype, void* stream);

/* ---- SSIM metric (first "next" row) --------------------------------------------
 * Gaussian window (sigma 1.5) of `window_size` <= 16 taps, zero padding window_size/2 (even windows give an (H+1)x(W+1) map),
 * C1 = 0.01^2, C2 = 0.03^2.  sums[b] = sum over (C, Ho, Wo) of the SSIM map of sample b (fp64).  img1/img2 fp32.
 * replaces: ssim / _ssim  inpainting_metrics/ssim/ssim.py:17-75 (5 depthwise F.conv2d + elementwise chain) */
int ocf_ssim_fwd(const ocf_tensor4* img1, const ocf_tensor4* img2, double* sums, int B, int C, int H, int W, int window_size,
                 void* stream);

/* SSIM as a differentiable photometric term: gradient of  sum_b coef[b] * sums[b]  w.r.t. img1 (the first argument;
 * by the symmetry of the formula the gradient w.r.t. img2 is the same call with the images swapped).  Two launches: the forward pass
 * again, storing the derivatives of the quotient w.r.t. the three windowed moments that involve img1 in `workspace`
 * (ocf_ssim_bwd_workspace_bytes), then the transposed separable Gaussian filter of those planes combined with the images.
 * coef: device fp32 [B]; gimg1: fp32 [B,C,H,W], any strides, written.
 * replaces: autograd of _ssim  inpainting_metrics/ssim/ssim.py:17-37 (5 conv backward passes + the elementwise chain) */
int ocf_ssim_bwd_workspace_bytes(int B, int C, int H, int W, int window_size, int64_t* bytes);
int ocf_ssim_bwd(const ocf_tensor4* img1, const ocf_tensor4* img2, const float* coef, const ocf_tensor4* gimg1, void* workspace, int B,
                 int C, int H, int W, int window_size, void* stream);

/* ---- PSNR metric (sibling of SSIM in the reference's inpainting_metrics) ------------------------------------------------------
 * sums[b] = sum over (C,H,W) of (img1 - img2)^2 of sample b (fp64, zeroed by the launcher); img1 / img2 fp32, any strides.
 * replaces: psnr()  inpainting_metrics/psnr/psnr.py:4-11  (numpy mean((a-b)^2) per image after a device-to-host copy,
 *           call site inpainting_metrics/__init__.py:76-78) */
int ocf_sqdiff(const ocf_tensor4* img1, const ocf_tensor4* img2, double* sums, int B, int C, int H, int W, void* stream);

/* ---- convolution epilogue of the estimator nets (second "next" row: the callers either side of the correlation) ----------
 * forward, IN PLACE on x (the bias-free convolution output):   t = T(x + T(bias[c]));  y = t > 0 ? t : T(t * negative_slope)
 *   negative_slope == 1: plain bias add (prediction layers, transposed convolutions).
 * backward:  gx = y > 0 ? gy : T(gy * negative_slope);  gbias[c] (+)= sum over (B,H,W) of gx   (fp32)
 *   gx == NULL (only with negative_slope == 1, where gx is gy itself): reduction only, y is not read and may be NULL.
 *   accumulate == 0: the launcher zeroes gbias first; accumulate != 0: gbias already holds the value to add to (callers hand
 *   out slices of one pre-zeroed buffer instead of paying a memset per layer).
 * x / y / gy / gx: dense channels_last or dense NCHW tensors of type `dtype` (anything else: OCF_EUNSUPPORTED); bias, gbias fp32 [C].
 * replaces: nn.Sequential(nn.Conv2d(bias=True), nn.LeakyReLU(0.1)) and the bias of nn.Conv2d / nn.ConvTranspose2d --
 *           conv(), predict_flow(), deconv() of cost_volume_flow_net.py:19-30, the estimators of flow_net.py / flow_occ_net.py,
 *           context_net.py, feature_pyramid_net.py upsample blocks
 *           [aten add_(bias) + leaky_relu | leaky_relu_backward + sum((0,2,3)) + dtype casts] */
int ocf_bias_act_fwd(const ocf_tensor4* x, const float* bias, int B, int C, int H, int W, float negative_slope, int dtype,
                     void* stream);
int ocf_bias_act_bwd(const ocf_tensor4* y, const ocf_tensor4* gy, const ocf_tensor4* gx, float* gbias, int B, int C, int H, int W,
                     float negative_slope, int accumulate, int dtype, void* stream);

/* ---- weight gradient of the 3x3, stride-1, padding == dilation convolutions of the estimator / context networks ------
 * gw[co][ci][ky][kx] += sum_{b,y,x} gy[b,co,y,x] * x[b,ci, y + (ky-1)*dilation, x + (kx-1)*dilation]   (zero outside the image)
 * x [B,Ci,H,W], gy [B,Co,H,W]: dense channels_last bf16, Ci % 8 == 0, Co % 8 == 0, dilation <= 16 (anything else: OCF_EUNSUPPORTED,
 * the caller keeps the framework's convolution backward).  gw: fp32, element strides (gw_so, gw_si, gw_sky, gw_skx), ACCUMULATED into
 * with float atomics (split-K over image rows): the caller zeroes it -- one buffer for every layer of a step instead of MIOpen's
 * per-layer fp32 workspace fill + bf16 cast.
 * replaces: the weight half of autograd's convolution_backward for conv() / nn.Conv2d(k=3, s=1, p=d, dilation=d) in
 *           flow_net.py:11-37, flow_occ_net.py:11-67, context_net.py:11-17, cost_volume_flow_net.py:58-112 */
/* policy (per call; same results up to float summation order): 0 = automatic, 1 = register-staged kernel, 2 = LDS-DMA three-stage ring
 * kernel (eight-wave workgroups, two Ci blocks per workgroup) */
int ocf_conv3x3_wgrad(const ocf_tensor4* x, const ocf_tensor4* gy, float* gw, int64_t gw_so, int64_t gw_si, int64_t gw_sky, int64_t gw_skx,
                      int B, int Ci, int Co, int H, int W, int dilation, int dtype, int policy, void* stream);

/* ---- the 3x3, stride-1, padding == dilation convolutions of the estimator / context networks as ONE implicit-GEMM kernel on the matrix
 * cores, forward and input gradient, bf16 channels_last:
 *   y[b,m,y,x] = epi( sum_{ky,kx,k} Wp[m][ky*3+kx][k] * x[b,k, y+(ky-1)*dilation, x+(kx-1)*dilation] )        (zero outside the image)
 *   epi(v)     = lrelu(v + bias[m], negative_slope) * (act[b,m,y,x] > 0 ? 1 : mask_slope)       bias == NULL: no bias; act == NULL: no mask
 * forward:        Wp = pack(weight, transpose = 0), bias + LeakyReLU fused (negative_slope = 1: plain bias)
 * input gradient: x := grad_out, Wp = pack(weight, transpose = 1), act = the convolution's own INPUT when that is the LeakyReLU output of
 *                 the previous layer (its gradient then leaves the kernel already multiplied by that activation's derivative)
 * x, y, act: channels_last views (channel stride 1; x with 16-byte aligned pixels, y / act 8-byte aligned when Cout % 4 == 0) -- channel
 * slices of wider buffers included, which is how the decoder's concatenations are read and written in place.  Cin % 8 == 0.
 * w_packed: [w_rows][9][w_k] bf16 from ocf_conv3x3_pack_weights, w_rows >= the rows_needed ocf_conv3x3_plan reports (at most 16 * (ceil(Cout / 16) + 7) for any image), w_k % 8 == 0, zero padded.
 * policy: 0 = automatic; 16 / 32 / 64 = force the staged kernel's patch width; + 256 = force the barrier-free small-map kernel
 * (csrc/conv_direct.hip: automatic up to 4096 pixels), + 512 = forbid it (A/B).  dilation <= 16.
 * replaces: conv() / nn.Conv2d(k=3, s=1, p=d, dilation=d) [+ bias + nn.LeakyReLU(0.1)] and the input half of their autograd in
 *           flow_net.py:11-37, flow_occ_net.py:11-67, context_net.py:11-17 */
int ocf_conv3x3_plan(int Cout, int B, int H, int W, int dilation, int* m_tiles, int* patch_width, int* rows_needed);
int ocf_conv3x3_igemm(const ocf_tensor4* x, const void* w_packed, int w_rows, int w_k, const ocf_tensor4* y, const float* bias,
                      const ocf_tensor4* act, int B, int Cin, int Cout, int H, int W, int dilation, float negative_slope, float mask_slope,
                      int dtype, int policy, void* stream);
/* The same launch with a RESIDUAL in the epilogue, for layers whose output is added into a buffer other layers also add into (the input
 * gradients of a DenseNet block, cost_volume_flow_net.py:176-180: every layer reads a channel range of ONE buffer, so every input gradient
 * adds into a channel range of ONE gradient buffer):
 *     m = (act && c < act_channels && act[b,c,y,x] <= 0) ? mask_slope : 1
 *     y = res_premask ? (lrelu(conv + bias) + res) * m : lrelu(conv + bias) * m + res
 * res: [B, Cout, H, W] like y (may BE y: in-place accumulation); act_channels: the mask applies to output channels below it (the block's
 * own activations; the channels behind them are the block's input, whose gradient passes unmasked).  Needs Cout % 8 == 0 and 16-byte
 * aligned pixels of y / act / res; res == NULL && act_channels >= Cout is exactly ocf_conv3x3_igemm. */
int ocf_conv3x3_igemm_ex(const ocf_tensor4* x, const void* w_packed, int w_rows, int w_k, const ocf_tensor4* y, const float* bias,
                         const ocf_tensor4* act, const ocf_tensor4* res, int res_premask, int act_channels, int B, int Cin, int Cout, int H,
                         int W, int dilation, float negative_slope, float mask_slope, int dtype, int policy, void* stream);
/* Weight packing for ocf_conv3x3_igemm, every layer of a network in ONE launch: descriptor i says "src (any-strided [Co][Ci][3][3], fp32 or
 * bf16) -> dst (bf16 [rows_pad][9][k_pad]; the kernel writes the valid region, the caller zeroes dst ONCE when it allocates it)"; transpose = 0: rows = Co, k = Ci; transpose = 1: rows = Ci, k = Co, taps flipped.
 * ocf_conv3x3_pack_describe fills descriptor `index` of a HOST array (ocf_conv3x3_pack_bytes(n) bytes) and chains the flat element index
 * (`first` of descriptor 0 is 0, `*next_first` feeds the next call; the last one is `total_elements`); the caller copies the array to the
 * device once and calls ocf_conv3x3_pack_weights(device array) whenever the source weights have changed. */
int ocf_conv3x3_pack_bytes(int n, int64_t* desc_bytes);
int ocf_conv3x3_pack_describe(void* host_descs, int index, const void* src, int64_t so, int64_t si, int64_t sky, int64_t skx, int Co, int Ci,
                              int src_is_f32, void* dst, int rows_pad, int k_pad, int transpose, int64_t first, int64_t* next_first);
int ocf_conv3x3_pack_weights(const void* device_descs, int n, int64_t total_elements, void* stream);

/* Weight AND bias gradient of the same layers in one launch, accumulated (float atomics) into the caller's zeroed fp32 buffers:
 *   gw[co][ci][ky][kx] += sum_{b,y,x} dz[b,co,y,x] * x[b,ci, y+(ky-1)*dilation, x+(kx-1)*dilation]      gbias[co] += sum dz[b,co,y,x]
 * dz = the gradient w.r.t. the convolution's OUTPUT before bias / after the activation's derivative has been applied (what
 * ocf_conv3x3_igemm's `act` epilogue or ocf_bias_act_bwd produce).  x, dz: channels_last views with 16-byte aligned pixels (channel
 * slices of wider buffers included) whose pixel pitch covers the channel count rounded up to 8, dilation <= 16; gw any element strides (the flat fp32 gradient buffer of a
 * training step, in the parameter's own layout); gbias may be NULL.  policy: 0 = automatic; otherwise
 * chunks + 1000 * ci_waves + 10000 * co_waves (tuning).  Third version of the kernel: one tap row per workgroup, one row segment of each
 * operand per stage, pixel slices summed through LDS before the atomics (csrc/conv_wgrad3.hip).
 * replaces: the weight and bias halves of autograd's convolution_backward + sum((0,2,3)) for the layers listed above. */
int ocf_conv3x3_wgrad_bias(const ocf_tensor4* x, const ocf_tensor4* dz, float* gw, int64_t gw_so, int64_t gw_si, int64_t gw_sky, int64_t gw_skx,
                           float* gbias, int B, int Ci, int Co, int H, int W, int dilation, int dtype, int policy, void* stream);
/* The launch geometry ocf_conv3x3_wgrad_bias uses for a request (host only, no GPU touched): `workgroups` = chunks x tiles, `tiles` = 3 tap
 * rows x (co, ci) blocks, `chunks` = pixel chunks (each adds a full copy of its tiles with float atomics), `segments_per_chunk` row segments
 * per chunk, `chunks_capped` = 1 when the map is so small that the chunk count was cut below one workgroup per compute unit (the two
 * regimes the parity tests assert).  Any output pointer may be NULL. */
int ocf_conv3x3_wgrad_plan(int B, int Ci, int Co, int H, int W, int dilation, int policy, int* workgroups, int* chunks, int* tiles,
                           int* segments_per_chunk, int* chunks_capped);

/* ---- forward of the PWC-style decoders' feature up-sampling layers: nn.ConvTranspose2d(Ci, Co <= 2, kernel 4, stride 2, padding 1) on a channel
 * range of the level's decoder buffer (round 5; csrc/deconv4x4.hip):
 *   y[b,co,oy,ox] = bias[co] + sum_{ci,ky,kx : oy = 2 iy - 1 + ky, ox = 2 ix - 1 + kx} x[b,ci,iy,ix] * w[ci,co,ky,kx]
 * x: bf16 pixel-major [B, Ci, H, W] view, 16-byte aligned pixels whose pitch covers Ci rounded up to 16 (the channels behind Ci must exist:
 * they meet zero weights); w: [Ci][Co][4][4] with element strides, fp32 or bf16; y: bf16 pixel-major [B, Co, 2H, 2W]; workspace:
 * ocf_deconv4x4s2_fwd_workspace_bytes() bytes, 16-byte aligned, any contents.  Three launches (weight pack, GEMM on the matrix cores, gather), no atomics.
 * replaces: `upfeat6..3` of cost_volume_flow_net.py:63-65,182-183,197-198 / cost_volume_flow_occ_net.py (forward; the backward stays the
 *           framework's convolution_backward). */
int ocf_deconv4x4s2_fwd_workspace_bytes(int B, int Ci, int Co, int H, int W, int64_t* bytes);
int ocf_deconv4x4s2_fwd(const ocf_tensor4* x, const void* w, int64_t w_si, int64_t w_so, int64_t w_sy, int64_t w_sx, int w_is_f32,
                        const float* bias, const ocf_tensor4* y, void* workspace, int B, int Ci, int Co, int H, int W, int dtype, void* stream);

/* ---- channel concatenation of the decoder inputs: out[:, c0_k : c0_k + channels[k]] = parts[k], out[:, sum(channels):Cout] = 0, one launch.
 * parts: n <= 6 pixel-major (channel stride 1, one pixel pitch) bf16 views of [B, channels[k], H, W]; out: channels_last
 * [B, Cout, H, W] or a channel slice of a wider channels_last tensor (pixel pitch a multiple of 8), Cout % 8 == 0, 16-byte aligned.  Anything else: OCF_EUNSUPPORTED (the caller keeps the framework's cat).
 * replaces: torch.cat((corr, f1, flow_up, feat_up), 1) / torch.cat((f1, warped2, occ_feat_up, occ_up), 1) of flow_net.py:104-121,
 *           flow_occ_net.py:153-176 [aten cat's generic gather kernel + the zero block of the channel alignment] */
int ocf_concat_channels(const ocf_tensor4* const* parts, const int* channels, int n, const ocf_tensor4* out, int Cout, int B, int H, int W, int dtype,
                        void* stream);

/* ---- the estimators' up-sampling layers: ConvTranspose2d(Ci, Co, kernel 3, stride 2, padding 1), Co <= 4, Ci <= 64, bf16 pixel-major views
 * x [B, Ci, H, W] and y [B, Co, Ho, Wo] with Ho in {2H - 1, 2H}, Wo in {2W - 1, 2W} (output_padding 0 / 1 per axis);
 *   y[b,co,oy,ox] = bias[co] + sum x[b,ci,iy,ix] * w[ci,co,ky,kx] over oy = 2 iy - 1 + ky, ox = 2 ix - 1 + kx.
 * w: the module's [Ci, Co, 3, 3] weight (fp32 when w_is_f32, else bf16) with element strides w_s*; bias fp32 or NULL.
 * _bwd: gx (bf16 pixel-major, may be NULL) = input gradient; gw (fp32, element strides g_s*, ACCUMULATED with atomics, may be NULL) and
 * gbias (fp32 [Co], accumulated, may be NULL) = parameter gradients.  Anything else: OCF_EUNSUPPORTED.
 * replaces: `upconv1` / `upconv2` of flow_net.py:38-48 and flow_occ_net.py:46-66 (F.conv_transpose2d + bias and their autograd: the
 *           library's vector-1 implicit GEMM / naive_conv kernels, a workspace fill and a cast per weight gradient). */
/* act = 1: y = sigmoid(bf16(result)) (`torch.sigmoid(deconv(...))` of flow_occ_net.py:62-66 in one launch); the backward then takes the
 * forward's output as y_act and applies the sigmoid derivative to gy on the fly (bf16-rounded, as the framework's sigmoid_backward);
 * act = 0 / y_act = NULL: the plain layer. */
int ocf_deconv3x3s2_fwd(const ocf_tensor4* x, const void* w, int64_t w_si, int64_t w_so, int64_t w_sy, int64_t w_sx, int w_is_f32,
                        const float* bias, const ocf_tensor4* y, int B, int Ci, int Co, int H, int W, int Ho, int Wo, int act, int dtype,
                        void* stream);
int ocf_deconv3x3s2_bwd(const ocf_tensor4* x, const void* w, int64_t w_si, int64_t w_so, int64_t w_sy, int64_t w_sx, int w_is_f32,
                        const ocf_tensor4* gy, const ocf_tensor4* gx, float* gw, int64_t g_si, int64_t g_so, int64_t g_sy, int64_t g_sx,
                        float* gbias, const ocf_tensor4* y_act, int B, int Ci, int Co, int H, int W, int Ho, int Wo, int dtype, void* stream);

/* ---- 3x3 / stride-1 / padding-1 convolution with 16 input and 16 output channels (bias-free: `layer1.double_conv.3` of the feature pyramid,
 * feature_pyramid_net.py:21-33), bf16 pixel-major x / y [B,16,H,W] (16-byte aligned pixels), w = the module's [16,16,3,3] weight (element
 * strides, fp32 or bf16).  transpose = 0: y = conv(x, w); transpose = 1: the input gradient (x = grad_out, y = grad_in: the same kernel
 * with the weight read transposed and tap-flipped).  B fragments straight from global memory, no LDS (csrc/conv_c16.hip).
 * replaces: F.conv2d / the input half of its autograd for that layer [the library: 54 us forward, 57 + 17 us input gradient at 16 x 218 x 512] */
int ocf_conv3x3_c16(const ocf_tensor4* x, const void* w, int64_t w_so, int64_t w_si, int64_t w_sky, int64_t w_skx, int w_is_f32,
                    const ocf_tensor4* y, int B, int H, int W, int transpose, int dtype, void* stream);
/* weight gradient of that layer: gw (fp32 [16][16][3][3], element strides, ADDED to: zeroed by the caller; block sums through `workspace`,
 * no atomics) += sum over
 * pixels of gy[px][o] * x[px + tap][i]; x, gy as above.  replaces: the weight half of the layer's autograd [the library: 106 us + a fill
 * and a cast launch] */
int ocf_conv3x3_c16_wgrad_workspace_bytes(int B, int H, int W, int64_t* bytes);
int ocf_conv3x3_c16_wgrad(const ocf_tensor4* x, const ocf_tensor4* gy, float* gw, int64_t g_so, int64_t g_si, int64_t g_sky, int64_t g_skx,
                          void* workspace /* ocf_conv3x3_c16_wgrad_workspace_bytes(), any contents */, int B, int H, int W, int dtype,
                          void* stream);

/* ---- the encoder's first layer: Conv2d(3, 16, 3, stride 2, padding 1, bias=False) (feature_pyramid_net.py:21-33 `layer1.double_conv.0`),
 * forward.  x: DENSE channels_last bf16 [B,3,H,W] (6-byte pixels: what ocf_pair_input writes); y: pixel-major bf16 [B,16,Ho,Wo],
 * Ho = (H - 1) / 2 + 1; w = the module's [16,3,3,3] weight (element strides, fp32 or bf16).
 * replaces: F.conv2d for that layer [the library: 55 us + a 16 us fill at 16 x 3 x 436 x 1024] (the input image needs no gradient) */
int ocf_conv3x3s2_c3(const ocf_tensor4* x, const void* w, int64_t w_so, int64_t w_si, int64_t w_sky, int64_t w_skx, int w_is_f32,
                     const ocf_tensor4* y, int B, int H, int W, int dtype, void* stream);
/* weight gradient of that layer: gw (fp32 [16][3][3][3], element strides, ADDED to) += sum over output pixels of gy[p][o] * x[2p + tap][ci];
 * x as above, gy pixel-major bf16 [B,16,Ho,Wo] with 16-byte aligned pixels; workspace of ocf_conv3x3s2_c3_wgrad_workspace_bytes() bytes.
 * replaces: the weight half of the layer's autograd [the library: 88 us + a fill and a cast launch] */
int ocf_conv3x3s2_c3_wgrad_workspace_bytes(int B, int H, int W, int64_t* bytes);
int ocf_conv3x3s2_c3_wgrad(const ocf_tensor4* x, const ocf_tensor4* gy, float* gw, int64_t g_so, int64_t g_si, int64_t g_sky, int64_t g_skx,
                           void* workspace, int B, int H, int W, int dtype, void* stream);

/* ---- the two images of a pair as one bf16 channels_last batch: x fp32 [B, 6, H, W] (any strides; channels 0-2 / 3-5 = the two images),
 * out bf16 dense channels_last [2B, 3, H, W] (8-byte aligned): out[b] = x[b, 0:3], out[B + b] = x[b, 3:6], rounded to bf16.
 * replaces: torch.cat((inputs[:, 0:3], inputs[:, 3:6]), 0) of the paired encoder pass + the autocast cast of its first convolution
 *           (reference call sites flow_net.py:100-101, flow_occ_net.py:149-150 run the encoder once per image) */
int ocf_pair_input(const ocf_tensor4* x, void* out, int B, int H, int W, void* stream);

/* ---- bilinear resize of an fp32 image batch without gradient: x [B,C,H,W] -> y [B,C,h,w], both fp32 with any element strides (x may be a
 * channel slice of the pair tensor); align_corners as torch.nn.functional.interpolate (source index = dst * (H-1)/(h-1), or
 * max((dst + 0.5) * H/h - 0.5, 0)); one thread per output element, the framework kernel's fp32 arithmetic.
 * replaces: F.interpolate(img1, scale_factor=0.25, mode="bilinear", align_corners=True) of the PWC unsupervised step (models/model.py:63-69) */
int ocf_resize_bilinear(const ocf_tensor4* x, const ocf_tensor4* y, int B, int C, int H, int W, int h, int w, int align_corners, void* stream);

/* ---- supervised head of FlowOccModel: x4 bilinear up-sampling (align_corners = False) of the quarter-resolution flow [B,2,h,w] and occlusion
 * probability [B,1,h,w] (bf16 or fp32, any strides; occ_lr / occ_gt may both be NULL), mean L1 against flow_gt fp32 [B,2,4h,4w] and mean
 * binary cross entropy (logs clamped at -100) against occ_gt fp32 [B,1,4h,4w], without the full-resolution predictions in memory.
 * _fwd: sums = 2 doubles of scratch (zeroed by the launcher), losses = 2 floats (the two means), partials = fp32 scratch of 27 * B * h * w elements kept
 *       for _bwd, or NULL when no gradient is wanted.
 * _bwd: grad_*_loss = device fp32 scalars (dLoss/d mean; NULL = 0); writes gflow_lr / gocc_lr (either may be NULL) in the tensors' dtype.
 * replaces: `self.upsample1(refined), self.upsample2(occ)` (flow_occ_net.py:56-57,93; cost_volume_flow_occ_net.py:43-44,83) +
 *           `F.l1_loss(flow_pred, flow)`, `F.binary_cross_entropy(occ_pred, occ)` (models/flow_occ_model.py:48-55) and their autograd */
int ocf_flow_occ_head_fwd(const ocf_tensor4* flow_lr, const ocf_tensor4* occ_lr, const ocf_tensor4* flow_gt, const ocf_tensor4* occ_gt,
                          double* sums, float* losses, float* partials, int B, int h, int w, int flow_dtype, int occ_dtype, void* stream);
int ocf_flow_occ_head_bwd(const float* partials, const float* grad_flow_loss, const float* grad_occ_loss, const ocf_tensor4* gflow_lr,
                          const ocf_tensor4* gocc_lr, int B, int h, int w, int flow_dtype, int occ_dtype, void* stream);

/* ---- Adam over flat buffers: one launch for a whole network.  params / grads / exp_avg / exp_avg_sq: fp32 arrays of n elements in ONE
 * element order (n % 4 == 0, 16-byte aligned); params_bf16: optional bf16 array of n elements that receives the updated parameters
 * (the compute copies of a master-weight set-up), or NULL; step: device fp32 scalar holding the step count t >= 1 of THIS update.
 *   m <- m + (1 - b1)(g - m);  v <- b2 v + (1 - b2) g^2;  p <- p - lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps);  g += wd * p first.
 * replaces: torch.optim.Adam(...).step() of models/flow_model.py:105-106 / model.py configure_optimizers [the framework's multi-tensor
 *           kernels over 229 parameter tensors + the copy of the fp32 masters into the bf16 weights] */
int ocf_adam_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* params_bf16, int64_t n, float lr, double beta1,
                  double beta2, float eps, float weight_decay, const float* step, void* stream);

/* ---- BatchNorm2d (batch statistics) + LeakyReLU of the feature-pyramid encoder
 * replaces: nn.BatchNorm2d + nn.LeakyReLU(0.1) of the reference's `conv_bn_relu` blocks (models/networks/feature_pyramid_net.py:5-95), i.e.
 * torch.batch_norm(training) + leaky_relu and their autograd.  Dense channels_last x / y / gy / gx of `dtype` (OCF_F32 / OCF_BF16), C a
 * multiple of 4 (OCF_EUNSUPPORTED otherwise: use the framework operators); gamma, beta, statistics and parameter gradients fp32.
 *   training != 0: batch statistics; save_mean / save_invstd [C] receive mean and 1/sqrt(var + eps) (biased variance) for backward,
 *                  running_mean / running_var (optional) are updated with `momentum` (unbiased variance, as torch), *num_batches_tracked
 *                  (optional, int64) is incremented.  `workspace`: ocf_bn_act_workspace_bytes(C) bytes of scratch (any contents); one
 *                  workspace serves consecutive launches on one stream.
 *   training == 0: y = lrelu((x - running_mean) / sqrt(running_var + eps) * gamma + beta); no statistics pass, no workspace.
 * backward (training-mode forward only): gx, dgamma [C], dbeta [C] (overwritten).
 * groups (1 or 2, training only): the batch is `groups` consecutive, equally sized statistics groups -- the result of calling the module
 * on each part in turn (statistics per group, save_mean / save_invstd [groups][C], running buffers updated group after group, parameter
 * gradients summed), in the same launches.  The encoder runs both images of a pair as one batch this way. */
int ocf_bn_act_workspace_bytes(int C, int groups, int64_t* bytes);
int ocf_bn_act_fwd(const ocf_tensor4* x, const ocf_tensor4* y, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd, void* workspace, int B, int C,
                   int H, int W, float eps, float momentum, float negative_slope, int training, int groups, int dtype, void* stream);
int ocf_bn_act_bwd(const ocf_tensor4* x, const ocf_tensor4* y, const ocf_tensor4* gy, const ocf_tensor4* gx, const float* gamma,
                   const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, void* workspace, int B, int C, int H, int W,
                   float negative_slope, int groups, int dtype, void* stream);

/* ocf_bn_act_bwd without the forward output y: the sign of the pre-activation is recomputed from x with the forward's own coefficients
 * (gamma * invstd, beta - mean * gamma * invstd, same fma and rounding: the same bits), a third less traffic in the statistics pass and a
 * quarter less in the dx pass.  Same results as ocf_bn_act_bwd. */
int ocf_bn_act_bwd_x(const ocf_tensor4* x, const ocf_tensor4* gy, const ocf_tensor4* gx, const float* gamma, const float* beta,
                     const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, void* workspace, int B, int C, int H, int W,
                     float negative_slope, int groups, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OCFLOW_HIP_H */
