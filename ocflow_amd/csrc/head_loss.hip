// The supervised head of FlowOccModel in two launches: the x4 bilinear up-sampling of the network's quarter-resolution flow and occlusion
// outputs (nn.Upsample(scale_factor=4, mode="bilinear"): flow_occ_net.py:56-57, 93; cost_volume_flow_occ_net.py:43-44, 83), the L1 flow
// loss and the binary cross entropy of the occlusion map against full-resolution ground truth (models/flow_occ_model.py:48-55), forward
// AND the gradient w.r.t. the quarter-resolution tensors -- the full-resolution predictions never exist in HBM.
// The framework runs this as 27 launches per step (two casts, two up-samplings, sub / abs / mean, BCE / mean, their backward: sign, scale,
// two up-sampling backwards, fills and casts): 257 us at 8 x 436 x 1024, most of it launch floor.
//   forward : thread = one quarter-resolution pixel = its 4 x 4 block of full-resolution pixels.  Every one of the 16 samples takes its four
//             taps from the 3 x 3 quarter-resolution neighbourhood (src = (dst + 0.5) / 4 - 0.5, clamped at 0; second tap clamped to the
//             last row / column: PyTorch's align_corners = False rule); the thread sums |pred - gt| and the BCE terms (fp64 totals), and
//             accumulates d(loss sums)/d(neighbour) for the 3 x 3 x 3 (two flow channels + occlusion) neighbours in registers, written
//             as 27 floats per thread ("partials", saved for the backward).
//   backward: thread = one quarter-resolution pixel: the sum of the nine partials its neighbours hold for it, times the incoming gradient
//             of each mean, in the tensors' own dtype.  No atomics: run-to-run identical.
#include "ocf_common.h"

namespace ocf {
namespace hl {
struct Args {
  T4 flow, occ, fgt, ogt;  // quarter-resolution flow [B,2,h,w] / occlusion [B,1,h,w] (bf16 or fp32), ground truth fp32 [B,2,4h,4w] / [B,1,4h,4w]
  double* sums;            // [0] = sum |flow_pred - flow_gt|, [1] = sum BCE terms
  float* part;             // [27][B h w] or null (no gradient wanted)
  int B, h, w, flow_bf16, occ_bf16;
};

__device__ __forceinline__ float ld(const T4& t, long off, int is_bf16) {
  return is_bf16 ? (float)((const bf16*)t.p)[off] : ((const float*)t.p)[off];
}

__global__ __launch_bounds__(256) void head_fwd_kernel(const Args a) {
  __shared__ double s_red[2][4];
  const int npx = a.B * a.h * a.w;
  const int i = blockIdx.x * 256 + threadIdx.x;
  double l1 = 0.0, bce = 0.0;
  if (i < npx) {
    float l1f = 0.f, bcef = 0.f;  // this thread's 32 + 16 terms in fp32, the totals in fp64
    const int bx = i % a.w, r = i / a.w, by = r % a.h, b = r / a.h;
    const bool has_occ = a.occ.p != nullptr;
    // the 3 x 3 neighbourhood (clamped coordinates: a clamped slot is only ever addressed through the clamped tap index)
    float nf[3][3][3];
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
      for (int dx = 0; dx < 3; dx++) {
        const int yy = min(max(by + dy - 1, 0), a.h - 1), xx = min(max(bx + dx - 1, 0), a.w - 1);
        const long of = (long)b * a.flow.sb + (long)yy * a.flow.sh + (long)xx * a.flow.sw;
        nf[dy][dx][0] = ld(a.flow, of, a.flow_bf16);
        nf[dy][dx][1] = ld(a.flow, of + a.flow.sc, a.flow_bf16);
        nf[dy][dx][2] = has_occ ? ld(a.occ, (long)b * a.occ.sb + (long)yy * a.occ.sh + (long)xx * a.occ.sw, a.occ_bf16) : 0.f;
      }
    float g[3][3][3];
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
      for (int dx = 0; dx < 3; dx++)
#pragma unroll
        for (int c = 0; c < 3; c++) g[dy][dx][c] = 0.f;
    // the ground truth of the whole 4 x 4 block FIRST (12 float4 loads when the rows are contiguous): issued per sample, right before
    // their use, they were sixteen dependent round trips at three waves per SIMD (40 us for 67 MB)
    float gtv[4][3][4];
    {
      const bool vec = a.fgt.sw == 1 && (!has_occ || a.ogt.sw == 1) && (((uintptr_t)a.fgt.p | (uintptr_t)a.ogt.p) & 15) == 0 &&
                       ((a.fgt.sh | a.fgt.sc | a.fgt.sb | a.ogt.sh | a.ogt.sb) & 3) == 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float* fr = (const float*)a.fgt.p + (long)b * a.fgt.sb + (long)(4 * by + j) * a.fgt.sh + (long)(4 * bx) * a.fgt.sw;
        const float* orow = has_occ ? (const float*)a.ogt.p + (long)b * a.ogt.sb + (long)(4 * by + j) * a.ogt.sh + (long)(4 * bx) * a.ogt.sw : nullptr;
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float* src = c < 2 ? fr + (long)c * a.fgt.sc : orow;
          if (c == 2 && !has_occ) {
#pragma unroll
            for (int k = 0; k < 4; k++) gtv[j][c][k] = 0.f;
          } else if (vec) {
            const f32x4 v = *(const f32x4*)src;
            gtv[j][c][0] = v[0]; gtv[j][c][1] = v[1]; gtv[j][c][2] = v[2]; gtv[j][c][3] = v[3];
          } else {
            const long sw = c < 2 ? a.fgt.sw : a.ogt.sw;
#pragma unroll
            for (int k = 0; k < 4; k++) gtv[j][c][k] = src[k * sw];
          }
        }
      }
    }
    // Sample j of a block reads rows (slot 0, slot 1) of the neighbourhood for j = 0, 1 and (slot 1, slot 2) for j = 2, 3, with the second
    // tap's weight ly = 0.625, 0.875, 0.125, 0.375 -- static register indices.  At the image border PyTorch's rule (source coordinate
    // clamped at 0; second tap clamped to the last row) puts the whole weight on the border row: the weights are folded onto slot 1 there.
#pragma unroll
    for (int j = 0; j < 4; j++) {
      constexpr float LY[4] = {0.625f, 0.875f, 0.125f, 0.375f};
      const int rA = j < 2 ? 0 : 1, rB = rA + 1;
      float hB = LY[j], hA = 1.f - LY[j];
      if (j < 2 && by == 0) { hA = 0.f; hB = 1.f; }
      if (j >= 2 && by == a.h - 1) { hA = 1.f; hB = 0.f; }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int cA = k < 2 ? 0 : 1, cB = cA + 1;
        float wB = LY[k], wA = 1.f - LY[k];
        if (k < 2 && bx == 0) { wA = 0.f; wB = 1.f; }
        if (k >= 2 && bx == a.w - 1) { wA = 1.f; wB = 0.f; }
        float dpred[3];
#pragma unroll
        for (int c = 0; c < 2; c++) {
          // (the order of the framework's kernel: h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11))
          const float p = hA * (wA * nf[rA][cA][c] + wB * nf[rA][cB][c]) + hB * (wA * nf[rB][cA][c] + wB * nf[rB][cB][c]);
          const float t = gtv[j][c][k];
          const float d = p - t;
          l1f += fabsf(d);
          dpred[c] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        }
        dpred[2] = 0.f;
        if (has_occ) {
          const float p = hA * (wA * nf[rA][cA][2] + wB * nf[rA][cB][2]) + hB * (wA * nf[rB][cA][2] + wB * nf[rB][cB][2]);
          const float t = gtv[j][2][k];
          // (F.binary_cross_entropy clamps the logs at -100; v_log_f32 / v_rcp_f32: 1 ulp, the kernel was bound by the libm sequences)
          const float lp = fmaxf(__logf(p), -100.f), lq = fmaxf(__logf(1.f - p), -100.f);
          bcef -= t * lp + (1.f - t) * lq;
          dpred[2] = (p - t) * __frcp_rn(fmaxf((1.f - p) * p, 1e-12f));
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
          g[rA][cA][c] = __builtin_fmaf(hA * wA, dpred[c], g[rA][cA][c]);
          g[rA][cB][c] = __builtin_fmaf(hA * wB, dpred[c], g[rA][cB][c]);
          g[rB][cA][c] = __builtin_fmaf(hB * wA, dpred[c], g[rB][cA][c]);
          g[rB][cB][c] = __builtin_fmaf(hB * wB, dpred[c], g[rB][cB][c]);
        }
      }
    }
    l1 = (double)l1f;
    bce = (double)bcef;
    if (a.part) {  // planar [27][pixels]: every store instruction of a wave is one contiguous 256-byte run (pixel-major rows of 27 floats were
      float* o = a.part + i;  // 27 stores of 64 scattered lines each: 55 us for this kernel)
#pragma unroll
      for (int dy = 0; dy < 3; dy++)
#pragma unroll
        for (int dx = 0; dx < 3; dx++)
#pragma unroll
          for (int c = 0; c < 3; c++) o[(long)((dy * 3 + dx) * 3 + c) * npx] = g[dy][dx][c];
    }
  }
  l1 = wave_sum_d(l1);
  bce = wave_sum_d(bce);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { s_red[0][wid] = l1; s_red[1][wid] = bce; }
  __syncthreads();
  if (threadIdx.x < 2) atomicAdd(a.sums + threadIdx.x, s_red[threadIdx.x][0] + s_red[threadIdx.x][1] + s_red[threadIdx.x][2] + s_red[threadIdx.x][3]);
}

// losses[0] = sums[0] / n_flow, losses[1] = sums[1] / n_occ  (the two means, fp32 as the framework returns them)
__global__ void head_finalize_kernel(const double* sums, float* losses, double n_flow, double n_occ) {
  if (threadIdx.x == 0) losses[0] = (float)(sums[0] / n_flow);
  if (threadIdx.x == 1) losses[1] = n_occ > 0.0 ? (float)(sums[1] / n_occ) : 0.f;
}

__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ part, const float* __restrict__ go_flow,
                                                       const float* __restrict__ go_occ, T4 gflow, T4 gocc, int B, int h, int w, float inv_nf,
                                                       float inv_no, int flow_bf16, int occ_bf16) {
  const int npx = B * h * w;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= npx) return;
  const int x = i % w, r = i / w, y = r % h, b = r / h;
  float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = -1; dy <= 1; dy++)
#pragma unroll
    for (int dx = -1; dx <= 1; dx++) {
      // the block at (y - dy, x - dx) holds this pixel in its neighbourhood slot (dy + 1, dx + 1)
      const int yy = y - dy, xx = x - dx;
      if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
      const float* p = part + (long)(((dy + 1) * 3 + (dx + 1)) * 3) * npx + ((long)(b * h + yy) * w + xx);
      s[0] += p[0]; s[1] += p[npx]; s[2] += p[2 * (long)npx];
    }
  const float gf = go_flow ? *go_flow * inv_nf : 0.f, go = go_occ ? *go_occ * inv_no : 0.f;
  if (gflow.p) {
    const long of = (long)b * gflow.sb + (long)y * gflow.sh + (long)x * gflow.sw;
    if (flow_bf16) { ((bf16*)gflow.p)[of] = (bf16)(s[0] * gf); ((bf16*)gflow.p)[of + gflow.sc] = (bf16)(s[1] * gf); }
    else { ((float*)gflow.p)[of] = s[0] * gf; ((float*)gflow.p)[of + gflow.sc] = s[1] * gf; }
  }
  if (gocc.p) {
    const long oo = (long)b * gocc.sb + (long)y * gocc.sh + (long)x * gocc.sw;
    if (occ_bf16) ((bf16*)gocc.p)[oo] = (bf16)(s[2] * go);
    else ((float*)gocc.p)[oo] = s[2] * go;
  }
}
}  // namespace hl
}  // namespace ocf

extern "C" int ocf_flow_occ_head_fwd(const ocf_tensor4* flow_lr, const ocf_tensor4* occ_lr, const ocf_tensor4* flow_gt, const ocf_tensor4* occ_gt,
                                     double* sums, float* losses, float* partials, int B, int h, int w, int flow_dtype, int occ_dtype,
                                     void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!flow_lr || !flow_gt || !flow_lr->data || !flow_gt->data || !sums || !losses || B <= 0 || h <= 0 || w <= 0) return OCF_EINVAL;
  if ((occ_lr != nullptr) != (occ_gt != nullptr) || (occ_lr && (!occ_lr->data || !occ_gt->data))) return OCF_EINVAL;
  if ((flow_dtype != OCF_F32 && flow_dtype != OCF_BF16) || (occ_lr && occ_dtype != OCF_F32 && occ_dtype != OCF_BF16)) return OCF_EINVAL;
  if ((long)B * h * w * 27 >= (1L << 31)) return OCF_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, 2 * sizeof(double), st) != hipSuccess) return OCF_ELAUNCH;
  hl::Args a;
  a.flow = mk(flow_lr); a.occ = mk(occ_lr); a.fgt = mk(flow_gt); a.ogt = mk(occ_gt);
  a.sums = sums; a.part = partials; a.B = B; a.h = h; a.w = w; a.flow_bf16 = flow_dtype == OCF_BF16; a.occ_bf16 = occ_dtype == OCF_BF16;
  const int npx = B * h * w;
  LaunchProf lp;
  lp.launch(false, hl::head_fwd_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, a);
  lp.launch(true, hl::head_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)sums, losses, (double)npx * 32.0, occ_lr ? (double)npx * 16.0 : 0.0);
  return launch_status();
}

extern "C" int ocf_flow_occ_head_bwd(const float* partials, const float* grad_flow_loss, const float* grad_occ_loss, const ocf_tensor4* gflow_lr,
                                     const ocf_tensor4* gocc_lr, int B, int h, int w, int flow_dtype, int occ_dtype, void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!partials || B <= 0 || h <= 0 || w <= 0 || (!gflow_lr && !gocc_lr)) return OCF_EINVAL;
  if ((gflow_lr && !gflow_lr->data) || (gocc_lr && !gocc_lr->data)) return OCF_EINVAL;
  const int npx = B * h * w;
  LaunchProf lp;
  lp.launch(true, hl::head_bwd_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partials, grad_flow_loss,
            grad_occ_loss, mk(gflow_lr), mk(gocc_lr), B, h, w, (float)(1.0 / ((double)npx * 32.0)), (float)(1.0 / ((double)npx * 16.0)),
            (int)(flow_dtype == OCF_BF16), (int)(occ_dtype == OCF_BF16));
  return launch_status();
}
