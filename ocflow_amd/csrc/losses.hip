// K4 photometric (Charbonnier, optionally fused with the AC-True image warp and the occlusion mask),
// K5 edge-aware smoothness (orders 1 and 2 in one pass), K6 range map, K3 moments.
// All HBM-bound streaming kernels: one thread per pixel, lanes = consecutive x (coalesced NCHW image
// planes), wave-shuffle + LDS block reduction, one fp64 atomic per block and sum.
// reference: models/model.py:27-46 (photometric), :53-66,93-114 (smoothness), :223-305,391 (range map),
//            models/networks/correlation_layer.py:59-64 (moments)
#include <math.h>
#include "ocf_common.h"

namespace ocf {

constexpr int LB = 256;
// loops over the channel count: fully unrolled in the CT = 3 instantiations, left alone for a runtime count
#define OCF_UNROLL_CT _Pragma("unroll 3")

template <int NS>
__device__ __forceinline__ void block_reduce_atomic(double (&v)[NS], double* sums) {
  __shared__ double red[NS][LB / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NS; i++) {
    const double s = wave_sum_d(v[i]);
    if (lane == 0) red[i][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < LB / 64; w++) s += red[threadIdx.x][w];
    atomicAdd(&sums[threadIdx.x], s);
  }
}

__device__ __forceinline__ float charb(float d) { return sqrtf(__builtin_fmaf(d, d, 1e-6f)); }

// ------------------------------------------------------------------------------------ photometric
struct __attribute__((packed, aligned(4))) F3 { float v[3]; };  // the three channels of a channels_last RGB pixel: one 12-byte load
struct PhotoPix {
  float w[4];
  long off[4];
  float wx1, wy1;
};
__device__ __forceinline__ PhotoPix photo_setup(const T4& img2, const T4& flow, int b, int y, int x, int H, int W,
                                                const WarpGeom& wg) {
  PhotoPix p;
  const float* fl = (const float*)flow.p;
  const long fo = (long)b * flow.sb + (long)y * flow.sh + (long)x * flow.sw;
  const Bilin s = bilin_setup(((float)x + fl[fo]) * wg.ax + wg.bx, ((float)y + fl[fo + flow.sc]) * wg.ay + wg.by, H, W);
  p.wx1 = s.wx1; p.wy1 = s.wy1;
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const int ddx = n & 1, ddy = n >> 1;
    const bool ok = (ddx ? s.vx1 : s.vx0) & (ddy ? s.vy1 : s.vy0);
    p.w[n] = ok ? (ddx ? s.wx1 : 1.f - s.wx1) * (ddy ? s.wy1 : 1.f - s.wy1) : 0.f;
    p.off[n] = ok ? (long)b * img2.sb + (long)(s.y0 + ddy) * img2.sh + (long)(s.x0 + ddx) * img2.sw : -1;
  }
  return p;
}

template <int CT>
__global__ __launch_bounds__(LB) void photo_fwd_kernel(T4 img1, T4 img2, T4 flow, T4 occ, double* sums, int B, int C, int H,
                                                       int W, WarpGeom wg, long npix) {
  // per-thread partial sums in fp32 (a thread visits at most a few dozen pixels), widened to fp64 for the block / grid reduction
  float part[2] = {0.f, 0.f};
  const float* i1 = (const float*)img1.p;
  const float* i2 = (const float*)img2.p;
  for (long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    long r = idx / W;
    const int y = r % H, b = r / H;
    // (the mask value is loaded from an always-valid address and selected afterwards: a load inside the `occ.p ?` branch is waited for at
    // the join, which made it a round trip of its own in front of the flow -> taps chain)
    const long o1 = (long)b * img1.sb + (long)y * img1.sh + (long)x * img1.sw;
    const float occ_raw = *(occ.p ? (const float*)occ.p + ((long)b * occ.sb + (long)y * occ.sh + (long)x * occ.sw) : i1 + o1);
    float e = 0.f;
    if (CT == 3 && flow.p && img1.sc == 1 && img2.sc == 1) {
      // channels_last RGB (in the model: the two halves of a six-channel image pair): a tap is ONE 12-byte load instead of three 4-byte
      // loads 24 bytes apart across the wave -- 5 instead of 15 image loads per pixel.  Taps outside the image load the pixel's own
      // position (always valid) and are zeroed by the select.
      const PhotoPix p = photo_setup(img2, flow, b, y, x, H, W, wg);
      const long own = (long)b * img2.sb + (long)y * img2.sh + (long)x * img2.sw;
      F3 nb[4];
#pragma unroll
      for (int n = 0; n < 4; n++) nb[n] = *(const F3*)(i2 + (p.off[n] >= 0 ? p.off[n] : own));
      const F3 a = *(const F3*)(i1 + o1);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        float v = 0.f;
#pragma unroll
        for (int n = 0; n < 4; n++) v = __builtin_fmaf(p.w[n], p.off[n] >= 0 ? nb[n].v[c] : 0.f, v);
        e += charb(v - a.v[c]);
      }
    } else if (flow.p) {
      const PhotoPix p = photo_setup(img2, flow, b, y, x, H, W, wg);
      OCF_UNROLL_CT for (int c = 0; c < (CT ? CT : C); c++) {
        // all four neighbour loads issued before the first use (select, not branch: a branch per tap serialises the loads)
        float nb[4];
#pragma unroll
        for (int n = 0; n < 4; n++) nb[n] = p.off[n] >= 0 ? i2[p.off[n] + (long)c * img2.sc] : 0.f;
        float v = 0.f;
#pragma unroll
        for (int n = 0; n < 4; n++) v = __builtin_fmaf(p.w[n], nb[n], v);
        e += charb(v - i1[o1 + (long)c * img1.sc]);
      }
    } else {
      const long o2 = (long)b * img2.sb + (long)y * img2.sh + (long)x * img2.sw;
      OCF_UNROLL_CT for (int c = 0; c < (CT ? CT : C); c++) e += charb(i2[o2 + (long)c * img2.sc] - i1[o1 + (long)c * img1.sc]);
    }
    const float vis = occ.p ? 1.f - occ_raw : 1.f;
    part[0] = __builtin_fmaf(e, vis, part[0]);
    part[1] += vis;
  }
  double acc[2] = {(double)part[0], (double)part[1]};
  block_reduce_atomic<2>(acc, sums);
}

template <int CT>
__global__ __launch_bounds__(LB) void photo_bwd_kernel(T4 img1, T4 img2, T4 flow, T4 occ, const float* scale, T4 gflow,
                                                       T4 gpred, int B, int C, int H, int W, WarpGeom wg, long npix) {
  const float* i1 = (const float*)img1.p;
  const float* i2 = (const float*)img2.p;
  const float sc = *scale;
  for (long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    long r = idx / W;
    const int y = r % H, b = r / H;
    // (the mask value is loaded from an always-valid address and selected afterwards: a load inside the `occ.p ?` branch is waited for at
    // the join, which made it a round trip of its own in front of the flow -> taps chain)
    const long o1 = (long)b * img1.sb + (long)y * img1.sh + (long)x * img1.sw;
    const float occ_raw = *(occ.p ? (const float*)occ.p + ((long)b * occ.sb + (long)y * occ.sh + (long)x * occ.sw) : i1 + o1);
    if (CT == 3 && flow.p && img1.sc == 1 && img2.sc == 1) {  // channels_last RGB: packed 12-byte loads (see photo_fwd_kernel)
      const PhotoPix p = photo_setup(img2, flow, b, y, x, H, W, wg);
      const long own = (long)b * img2.sb + (long)y * img2.sh + (long)x * img2.sw;
      F3 t[4];
#pragma unroll
      for (int n = 0; n < 4; n++) t[n] = *(const F3*)(i2 + (p.off[n] >= 0 ? p.off[n] : own));
      const F3 a = *(const F3*)(i1 + o1);
      const float k = sc * (occ.p ? 1.f - occ_raw : 1.f);
      float gx = 0.f, gy = 0.f;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        float nb[4];
        float v = 0.f;
#pragma unroll
        for (int n = 0; n < 4; n++) {
          nb[n] = p.off[n] >= 0 ? t[n].v[c] : 0.f;
          v = __builtin_fmaf(p.w[n], nb[n], v);
        }
        const float dlt = v - a.v[c];
        const float ge = k * dlt / charb(dlt);
        gx = __builtin_fmaf(ge, (1.f - p.wy1) * (nb[1] - nb[0]) + p.wy1 * (nb[3] - nb[2]), gx);
        gy = __builtin_fmaf(ge, (1.f - p.wx1) * (nb[2] - nb[0]) + p.wx1 * (nb[3] - nb[1]), gy);
      }
      if (gflow.p) {
        float* gf = (float*)gflow.p;
        const long o = (long)b * gflow.sb + (long)y * gflow.sh + (long)x * gflow.sw;
        gf[o] = gx * wg.ax;
        gf[o + gflow.sc] = gy * wg.ay;
      }
    } else if (flow.p) {
      const PhotoPix p = photo_setup(img2, flow, b, y, x, H, W, wg);
      const float k = sc * (occ.p ? 1.f - occ_raw : 1.f);
      float gx = 0.f, gy = 0.f;
      OCF_UNROLL_CT for (int c = 0; c < (CT ? CT : C); c++) {
        float nb[4];
        float v = 0.f;
#pragma unroll
        for (int n = 0; n < 4; n++) {
          nb[n] = p.off[n] >= 0 ? i2[p.off[n] + (long)c * img2.sc] : 0.f;
          v = __builtin_fmaf(p.w[n], nb[n], v);
        }
        const float dlt = v - i1[o1 + (long)c * img1.sc];
        const float ge = k * dlt / charb(dlt);
        gx = __builtin_fmaf(ge, (1.f - p.wy1) * (nb[1] - nb[0]) + p.wy1 * (nb[3] - nb[2]), gx);
        gy = __builtin_fmaf(ge, (1.f - p.wx1) * (nb[2] - nb[0]) + p.wx1 * (nb[3] - nb[1]), gy);
      }
      if (gflow.p) {
        float* gf = (float*)gflow.p;
        const long o = (long)b * gflow.sb + (long)y * gflow.sh + (long)x * gflow.sw;
        gf[o] = gx * wg.ax;
        gf[o + gflow.sc] = gy * wg.ay;
      }
    } else if (gpred.p) {
      const long o2 = (long)b * img2.sb + (long)y * img2.sh + (long)x * img2.sw;
      float* gp = (float*)gpred.p;
      const long og = (long)b * gpred.sb + (long)y * gpred.sh + (long)x * gpred.sw;
      const float k = sc * (occ.p ? 1.f - occ_raw : 1.f);
      OCF_UNROLL_CT for (int c = 0; c < (CT ? CT : C); c++) {
        const float dlt = i2[o2 + (long)c * img2.sc] - i1[o1 + (long)c * img1.sc];
        gp[og + (long)c * gpred.sc] = k * dlt / charb(dlt);
      }
    }
  }
}

// ------------------------------------------------------------------------------------- masked L1
// sums[0] = sum_{b,c,y,x} |pred - target| * mask[b,0,y,x],  sums[1] = sum mask   (inpainting_model.py:71: the completion loss is
// taken over occluded pixels only); backward: gpred = scale * sign(pred - target) * mask  (torch.abs: sign(0) = 0)
__global__ __launch_bounds__(LB) void masked_l1_fwd_kernel(T4 pred, T4 target, T4 mask, double* sums, int B, int C, int H, int W, long npix) {
  float part[2] = {0.f, 0.f};
  const float* pp = (const float*)pred.p;
  const float* tp = (const float*)target.p;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    long r = idx / W;
    const int y = r % H, b = r / H;
    const float m = ((const float*)mask.p)[(long)b * mask.sb + (long)y * mask.sh + (long)x * mask.sw];
    const long op = (long)b * pred.sb + (long)y * pred.sh + (long)x * pred.sw;
    const long ot = (long)b * target.sb + (long)y * target.sh + (long)x * target.sw;
    float e = 0.f;
    for (int c = 0; c < C; c++) e += fabsf(pp[op + (long)c * pred.sc] - tp[ot + (long)c * target.sc]);
    part[0] = __builtin_fmaf(e, m, part[0]);
    part[1] += m;
  }
  double acc[2] = {(double)part[0], (double)part[1]};
  block_reduce_atomic<2>(acc, sums);
}

__global__ __launch_bounds__(LB) void masked_l1_bwd_kernel(T4 pred, T4 target, T4 mask, const float* scale, T4 gpred, int B, int C, int H,
                                                           int W, long npix) {
  const float* pp = (const float*)pred.p;
  const float* tp = (const float*)target.p;
  float* gp = (float*)gpred.p;
  const float sc = *scale;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    long r = idx / W;
    const int y = r % H, b = r / H;
    const float k = sc * ((const float*)mask.p)[(long)b * mask.sb + (long)y * mask.sh + (long)x * mask.sw];
    const long op = (long)b * pred.sb + (long)y * pred.sh + (long)x * pred.sw;
    const long ot = (long)b * target.sb + (long)y * target.sh + (long)x * target.sw;
    const long og = (long)b * gpred.sb + (long)y * gpred.sh + (long)x * gpred.sw;
    for (int c = 0; c < C; c++) {
      const float d = pp[op + (long)c * pred.sc] - tp[ot + (long)c * target.sc];
      gp[og + (long)c * gpred.sc] = d > 0.f ? k : (d < 0.f ? -k : 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------- smoothness
// exp(-mean_c (alpha * (I[p+s] - I[p]))^2) for a step of `s` pixels along x (dirx) or y.
// CT: compile-time channel count (3 = RGB, the only case the reference has: all 2*CT loads are issued together), 0 = runtime C
template <int CT>
__device__ __forceinline__ float edge_w(const float* im, long o, long step, long sc, int C, float alpha) {
  const int n = CT ? CT : C;
  float m = 0.f;
#pragma unroll
  for (int c = 0; c < n; c++) {
    const float d = alpha * (im[o + step + (long)c * sc] - im[o + (long)c * sc]);
    m = __builtin_fmaf(d, d, m);
  }
  return expf(-m / (float)n);
}

template <int CT>
__global__ __launch_bounds__(LB) void smooth_fwd_kernel(T4 img, T4 flow, double* sums, int B, int C, int H, int W, float alpha,
                                                        long npix) {
  float part[4] = {0.f, 0.f, 0.f, 0.f};  // fp32 per thread, fp64 across threads (see photo_fwd_kernel)
  const float* im = (const float*)img.p;
  const float* fl = (const float*)flow.p;
  for (long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    long r = idx / W;
    const int y = r % H, b = r / H;
    const long oi = (long)b * img.sb + (long)y * img.sh + (long)x * img.sw;
    const long of = (long)b * flow.sb + (long)y * flow.sh + (long)x * flow.sw;
    if (x + 1 < W) {
      const float w = edge_w<CT>(im, oi, img.sw, img.sc, C, alpha);
      float s = 0.f;
      for (int f = 0; f < 2; f++) s += charb(fl[of + flow.sw + f * flow.sc] - fl[of + f * flow.sc]);
      part[0] = __builtin_fmaf(w, s, part[0]);
    }
    if (y + 1 < H) {
      const float w = edge_w<CT>(im, oi, img.sh, img.sc, C, alpha);
      float s = 0.f;
      for (int f = 0; f < 2; f++) s += charb(fl[of + flow.sh + f * flow.sc] - fl[of + f * flow.sc]);
      part[1] = __builtin_fmaf(w, s, part[1]);
    }
    if (x + 2 < W) {
      const float w = edge_w<CT>(im, oi, 2 * img.sw, img.sc, C, alpha);
      float s = 0.f;
      for (int f = 0; f < 2; f++) {
        const long o = of + f * flow.sc;
        s += charb((fl[o + 2 * flow.sw] - fl[o + flow.sw]) - (fl[o + flow.sw] - fl[o]));
      }
      part[2] = __builtin_fmaf(w, s, part[2]);
    }
    if (y + 2 < H) {
      const float w = edge_w<CT>(im, oi, 2 * img.sh, img.sc, C, alpha);
      float s = 0.f;
      for (int f = 0; f < 2; f++) {
        const long o = of + f * flow.sc;
        s += charb((fl[o + 2 * flow.sh] - fl[o + flow.sh]) - (fl[o + flow.sh] - fl[o]));
      }
      part[3] = __builtin_fmaf(w, s, part[3]);
    }
  }
  double acc[4] = {(double)part[0], (double)part[1], (double)part[2], (double)part[3]};
  block_reduce_atomic<4>(acc, sums);
}

// phi(p) = w1(p) * d/charb(d), d = F[p+1]-F[p]           (defined for 0 <= p < n-1)
// psi(p) = w2(p) * dd/charb(dd), dd = F[p+2]-2F[p+1]+F[p]  (defined for 0 <= p < n-2)
// dL/dF[p] = c1*(phi(p-1) - phi(p)) + c2*(psi(p-2) - 2 psi(p-1) + psi(p))
// both flow channels at once: the edge weights depend on the image only
template <int CT>
__device__ __forceinline__ void smooth_axis_grad(const float* im, const float* fl, long oi, long of, long fsc, long istep, long isc,
                                                 long fstep, int C, float alpha, int p, int n, float c1, float c2, float* g) {
  if (c1 != 0.f) {
    if (p >= 1) {
      const float w = c1 * edge_w<CT>(im, oi - istep, istep, isc, C, alpha);
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const float d = fl[of + f * fsc] - fl[of + f * fsc - fstep];
        g[f] += w * d / charb(d);
      }
    }
    if (p + 1 < n) {
      const float w = c1 * edge_w<CT>(im, oi, istep, isc, C, alpha);
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const float d = fl[of + f * fsc + fstep] - fl[of + f * fsc];
        g[f] -= w * d / charb(d);
      }
    }
  }
  if (c2 != 0.f) {
    if (p >= 2) {
      const float w = c2 * edge_w<CT>(im, oi - 2 * istep, 2 * istep, isc, C, alpha);
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const long o = of + f * fsc;
        const float d = (fl[o] - fl[o - fstep]) - (fl[o - fstep] - fl[o - 2 * fstep]);
        g[f] += w * d / charb(d);
      }
    }
    if (p >= 1 && p + 1 < n) {
      const float w = 2.f * c2 * edge_w<CT>(im, oi - istep, 2 * istep, isc, C, alpha);
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const long o = of + f * fsc;
        const float d = (fl[o + fstep] - fl[o]) - (fl[o] - fl[o - fstep]);
        g[f] -= w * d / charb(d);
      }
    }
    if (p + 2 < n) {
      const float w = c2 * edge_w<CT>(im, oi, 2 * istep, isc, C, alpha);
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const long o = of + f * fsc;
        const float d = (fl[o + 2 * fstep] - fl[o + fstep]) - (fl[o + fstep] - fl[o]);
        g[f] += w * d / charb(d);
      }
    }
  }
}

template <int CT>
__global__ __launch_bounds__(LB) void smooth_bwd_kernel(T4 img, T4 flow, const float* coef, T4 gflow, int B, int C, int H, int W,
                                                        float alpha, long npix) {
  const float* im = (const float*)img.p;
  const float* fl = (const float*)flow.p;
  float* gf = (float*)gflow.p;
  const float cx1 = coef[0], cy1 = coef[1], cx2 = coef[2], cy2 = coef[3];
  for (long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    long r = idx / W;
    const int y = r % H, b = r / H;
    const long oi = (long)b * img.sb + (long)y * img.sh + (long)x * img.sw;
    const long of = (long)b * flow.sb + (long)y * flow.sh + (long)x * flow.sw;
    const long og = (long)b * gflow.sb + (long)y * gflow.sh + (long)x * gflow.sw;
    float g[2] = {0.f, 0.f};
    smooth_axis_grad<CT>(im, fl, oi, of, flow.sc, img.sw, img.sc, flow.sw, C, alpha, x, W, cx1, cx2, g);
    smooth_axis_grad<CT>(im, fl, oi, of, flow.sc, img.sh, img.sc, flow.sh, C, alpha, y, H, cy1, cy2, g);
    gf[og] = g[0];
    gf[og + gflow.sc] = g[1];
  }
}

// ---- RGB fast path (round 3): one lane = one image COLUMN, walking down SMR rows with the rows it has already loaded kept in registers.
// The flat pixel-per-thread kernels above put the rows y, y+1, y+2 of a pixel in different workgroups -- on different XCDs, i.e. behind
// different L2s -- so every image / flow row came from HBM about three times (rocprofv3, 8 x 3 x 448 x 1024: 234 MB fetched for 73 MB
// of algorithmic bytes, 91 us forward, 86 us backward) and every term re-loaded its operands (44 / 112 loads per pixel).  Here a wave
// owns a 64-column x SMR-row strip: a row is fetched once per strip (+2 / +4 halo rows per SMR), the vertical neighbours come from the
// register window, the horizontal ones from the same cache lines the wave has just touched (15 / 25 loads per pixel).
// Same expressions, term by term, as the kernels above, with the hardware's 1-ulp square root / reciprocal square root / exp2 in place
// of the correctly-rounded library expansions (sqrtf, division, expf are ~12 / ~10 / ~15 instructions each and there are 30 of them per
// pixel in the backward pass, which was instruction-bound with them: 106 us for 102 MB): d / sqrt(d^2 + eps) = d * rsq(d^2 + eps).
// rows per strip: a row is fetched once per strip + 2 (forward) / 4 (backward) halo rows.  Forward 16 (halo 1.125 x; round 3's 8: 34.1 ->
// 26.3 us at 8 x 3 x 436 x 1024 on one box), backward 8 (with 16 its register window costs an occupancy step: 43.5 -> 48.4 us)
constexpr int SMR_F = 16, SMR_B = 8;
__device__ __forceinline__ float charb_hw(float d) { return __builtin_amdgcn_sqrtf(__builtin_fmaf(d, d, 1e-6f)); }
__device__ __forceinline__ float dcharb_hw(float d) { return d * __builtin_amdgcn_rsqf(__builtin_fmaf(d, d, 1e-6f)); }  // d / charb(d)
struct Px5 { float i[3]; float f[2]; };

// CL (image channel stride 1): a channels_last image -- in the model the first or last three of the six channels of an image pair -- is
// ONE 12-byte load per pixel instead of three 4-byte loads 24 bytes apart across the wave
template <bool CL>
__device__ __forceinline__ Px5 load_px5(const float* im, const float* fl, long oi, long of, long isc, long fsc) {
  Px5 p;
  if (CL) {
    const F3 t = *(const F3*)(im + oi);
#pragma unroll
    for (int c = 0; c < 3; c++) p.i[c] = t.v[c];
  } else {
#pragma unroll
    for (int c = 0; c < 3; c++) p.i[c] = im[oi + (long)c * isc];
  }
#pragma unroll
  for (int f = 0; f < 2; f++) p.f[f] = fl[of + (long)f * fsc];
  return p;
}
// edge_w<3> on loaded values: a = the pixel itself, b = the pixel one or two steps further
__device__ __forceinline__ float edge3(const Px5& a, const Px5& b, float alpha) {
  float m = 0.f;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float d = alpha * (b.i[c] - a.i[c]);
    m = __builtin_fmaf(d, d, m);
  }
  return __builtin_amdgcn_exp2f(m * (-1.4426950408889634f / 3.f));
}

struct StripId { int b, x, y0; };
template <int SMR>
__device__ __forceinline__ StripId strip_of(int xblocks, int ychunks) {
  int bid = blockIdx.x;
  const int xb = bid % xblocks;
  bid /= xblocks;
  const int yc = bid % ychunks;
  return StripId{bid / ychunks, xb * 64 + (int)(threadIdx.x & 63), (yc * (LB / 64) + (int)(threadIdx.x >> 6)) * SMR};
}

template <bool CL>
__global__ __launch_bounds__(LB) void smooth_fwd_rows_kernel(T4 img, T4 flow, double* sums, int H, int W, float alpha, int xblocks, int ychunks) {
  const float* im = (const float*)img.p;
  const float* fl = (const float*)flow.p;
  const StripId t = strip_of<SMR_F>(xblocks, ychunks);
  const bool xin = t.x < W;
  const int xc = xin ? t.x : W - 1, x1 = xc + 1 < W ? xc + 1 : W - 1, x2 = xc + 2 < W ? xc + 2 : W - 1;
  const long ib = (long)t.b * img.sb, fb = (long)t.b * flow.sb;
  float part[4] = {0.f, 0.f, 0.f, 0.f};
  Px5 p1 = {}, p2 = {};  // rows y - 1, y - 2 of this column
  // the loads of row r + 1 are issued before row r is evaluated: a wave walks its rows one after the other, and without the lookahead
  // every row is a full memory round trip (37.8 us at 8 x 3 x 448 x 1024 with 1.9 TB/s of algorithmic bytes)
  auto fetch = [&](int r, Px5& c, Px5& a1, Px5& a2) __attribute__((always_inline)) {
    const int y = t.y0 + r;
    const int yc = y < H ? y : H - 1;
    const long oi = ib + (long)yc * img.sh, of = fb + (long)yc * flow.sh;
    c = load_px5<CL>(im, fl, oi + (long)xc * img.sw, of + (long)xc * flow.sw, img.sc, flow.sc);
    if (r < SMR_F) {
      a1 = load_px5<CL>(im, fl, oi + (long)x1 * img.sw, of + (long)x1 * flow.sw, img.sc, flow.sc);
      a2 = load_px5<CL>(im, fl, oi + (long)x2 * img.sw, of + (long)x2 * flow.sw, img.sc, flow.sc);
    }
  };
  Px5 cn = {}, a1n = {}, a2n = {};
  fetch(0, cn, a1n, a2n);
#pragma unroll
  for (int r = 0; r < SMR_F + 2; r++) {
    const int y = t.y0 + r;
    const Px5 c = cn, a1 = a1n, a2 = a2n;
    if (r + 1 < SMR_F + 2) fetch(r + 1, cn, a1n, a2n);
    if (y < H) {  // (uniform per wave)
      if (r < SMR_F) {  // the x terms of row y (one of the strip's own rows)
        if (xin && t.x + 1 < W) {
          float s = 0.f;
#pragma unroll
          for (int f = 0; f < 2; f++) s += charb_hw(a1.f[f] - c.f[f]);
          part[0] = __builtin_fmaf(edge3(c, a1, alpha), s, part[0]);
        }
        if (xin && t.x + 2 < W) {
          float s = 0.f;
#pragma unroll
          for (int f = 0; f < 2; f++) s += charb_hw((a2.f[f] - a1.f[f]) - (a1.f[f] - c.f[f]));
          part[2] = __builtin_fmaf(edge3(c, a2, alpha), s, part[2]);
        }
      }
      if (xin && r >= 1 && r <= SMR_F) {  // first-order y term of row y - 1
        float s = 0.f;
#pragma unroll
        for (int f = 0; f < 2; f++) s += charb_hw(c.f[f] - p1.f[f]);
        part[1] = __builtin_fmaf(edge3(p1, c, alpha), s, part[1]);
      }
      if (xin && r >= 2) {  // second-order y term of row y - 2
        float s = 0.f;
#pragma unroll
        for (int f = 0; f < 2; f++) s += charb_hw((c.f[f] - p1.f[f]) - (p1.f[f] - p2.f[f]));
        part[3] = __builtin_fmaf(edge3(p2, c, alpha), s, part[3]);
      }
      p2 = p1;
      p1 = c;
    }
  }
  double acc[4] = {(double)part[0], (double)part[1], (double)part[2], (double)part[3]};
  block_reduce_atomic<4>(acc, sums);
}

// smooth_axis_grad<3> on loaded values: v[0..4] = the pixels at p - 2 .. p + 2 along the axis (entries outside 0..n-1 are never used)
__device__ __forceinline__ void smooth_axis_grad5(const Px5* v, float alpha, int p, int n, float c1, float c2, float* g) {
  if (c1 != 0.f) {
    if (p >= 1) {
      const float w = c1 * edge3(v[1], v[2], alpha);
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const float d = v[2].f[f] - v[1].f[f];
        g[f] += w * dcharb_hw(d);
      }
    }
    if (p + 1 < n) {
      const float w = c1 * edge3(v[2], v[3], alpha);
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const float d = v[3].f[f] - v[2].f[f];
        g[f] -= w * dcharb_hw(d);
      }
    }
  }
  if (c2 != 0.f) {
    if (p >= 2) {
      const float w = c2 * edge3(v[0], v[2], alpha);
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const float d = (v[2].f[f] - v[1].f[f]) - (v[1].f[f] - v[0].f[f]);
        g[f] += w * dcharb_hw(d);
      }
    }
    if (p >= 1 && p + 1 < n) {
      const float w = 2.f * c2 * edge3(v[1], v[3], alpha);
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const float d = (v[3].f[f] - v[2].f[f]) - (v[2].f[f] - v[1].f[f]);
        g[f] -= w * dcharb_hw(d);
      }
    }
    if (p + 2 < n) {
      const float w = c2 * edge3(v[2], v[4], alpha);
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const float d = (v[4].f[f] - v[3].f[f]) - (v[3].f[f] - v[2].f[f]);
        g[f] += w * dcharb_hw(d);
      }
    }
  }
}

template <bool CL>
__global__ __launch_bounds__(LB) void smooth_bwd_rows_kernel(T4 img, T4 flow, const float* coef, T4 gflow, int H, int W, float alpha, int xblocks,
                                                             int ychunks) {
  const float* im = (const float*)img.p;
  const float* fl = (const float*)flow.p;
  float* gf = (float*)gflow.p;
  const float cx1 = coef[0], cy1 = coef[1], cx2 = coef[2], cy2 = coef[3];
  const StripId t = strip_of<SMR_B>(xblocks, ychunks);
  if (t.y0 >= H) return;
  const bool xin = t.x < W;
  const int xc = xin ? t.x : W - 1;
  int xs[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const int xx = xc + k - 2;
    xs[k] = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
  }
  const long ib = (long)t.b * img.sb, fb = (long)t.b * flow.sb;
  auto row = [&](int y, int xx) __attribute__((always_inline)) {  // (row clamped into the image: such entries are never used)
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
    return load_px5<CL>(im, fl, ib + (long)yc * img.sh + (long)xx * img.sw, fb + (long)yc * flow.sh + (long)xx * flow.sw, img.sc, flow.sc);
  };
  Px5 col[5];  // rows y - 2 .. y + 2 of this column
#pragma unroll
  for (int k = 1; k < 5; k++) col[k] = row(t.y0 - 3 + k, xc);
  // (one-row lookahead as in the forward kernel: the new bottom row and the four horizontal neighbours of the NEXT row are in flight
  // while this row is evaluated)
  Px5 nbot = row(t.y0 + 2, xc), nh[5];
#pragma unroll
  for (int k = 0; k < 5; k++)
    if (k != 2) nh[k] = row(t.y0, xs[k]);
#pragma unroll
  for (int r = 0; r < SMR_B; r++) {
    const int y = t.y0 + r;
#pragma unroll
    for (int k = 0; k < 4; k++) col[k] = col[k + 1];
    col[4] = nbot;
    Px5 hor[5];
#pragma unroll
    for (int k = 0; k < 5; k++) hor[k] = k == 2 ? col[2] : nh[k];
    if (r + 1 < SMR_B) {
      nbot = row(y + 3, xc);
#pragma unroll
      for (int k = 0; k < 5; k++)
        if (k != 2) nh[k] = row(y + 1, xs[k]);
    }
    if (y < H) {  // (uniform per wave)
      float g[2] = {0.f, 0.f};
      smooth_axis_grad5(hor, alpha, t.x, W, cx1, cx2, g);
      smooth_axis_grad5(col, alpha, y, H, cy1, cy2, g);
      if (xin) {
        const long og = (long)t.b * gflow.sb + (long)y * gflow.sh + (long)t.x * gflow.sw;
        gf[og] = g[0];
        gf[og + gflow.sc] = g[1];
      }
    }
  }
}

// -------------------------------------------------------------------------------------- range map
// One workgroup = a 32x8 tile of SOURCE pixels.  Their four bilinear targets are accumulated in an LDS window (tile + RSR pixels
// on every side) with ds_add_f32 and the window is flushed with one global atomic per touched target: a smooth flow sends the
// 1024 contributions of a tile to ~300-500 distinct targets, so the global float-atomic count -- the bound of this kernel --
// drops 2-3x (199 -> ~90 us at 8 x 436 x 1024).  Targets outside the window (|flow| > RSR) go straight to global memory.
// (Round 3, measured and dropped: flushing only the bounding box of the touched cells -- 11 instead of 20 flush instructions per
// workgroup on a smooth flow -- was 12 us SLOWER, 110.5 vs 98.7 us: the wave reductions and the extra barrier cost more than the
// partially filled atomic instructions they remove.)
constexpr int RSW = 32, RSH = 8, RSR = 8, RWW = RSW + 2 * RSR + 1, RWH = RSH + 2 * RSR + 1;
// The window accumulates in 32-bit FIXED POINT (integer ds_add_u32 instead of ds_add_f32): a weight is in [0, 1], a cell receives at most
// 256 x 4 of them from one workgroup (< 2^10), so 2^-22 steps fit 32 bits; the rounding of one contribution (<= 1.2e-7 absolute: half a step) is that of an
// fp32 add into a sum of order 1.
constexpr float RFIX = 4194304.f;  // 2^22
__global__ __launch_bounds__(LB) void range_splat_kernel(T4 flow, float* range, int B, int H, int W, int tiles_x, int tiles_y) {
  __shared__ unsigned win[RWH * RWW];  // fixed point, 2^-22 (see RFIX)
  const float* fl = (const float*)flow.p;
  const int tile = blockIdx.x;
  const int tpi = tiles_x * tiles_y;
  const int b = tile / tpi, trem = tile - b * tpi;
  const int ty0 = (trem / tiles_x) * RSH, tx0 = (trem % tiles_x) * RSW;
  for (int i = threadIdx.x; i < RWH * RWW; i += LB) win[i] = 0u;
  __syncthreads();
  const int x = tx0 + (threadIdx.x & (RSW - 1)), y = ty0 + (threadIdx.x >> 5);
  float* dst = range + (long)b * H * W;
  if (x < W && y < H) {
    const long fo = (long)b * flow.sb + (long)y * flow.sh + (long)x * flow.sw;
    const float cx = (float)x + fl[fo], cy = (float)y + fl[fo + flow.sc];
    if (cx == cx && cy == cy) {
      float fx = floorf(cx), fy = floorf(cy);
      const float rx = cx - fx, ry = cy - fy;
      fx = fminf(fmaxf(fx, -2.f), (float)W + 1.f);
      fy = fminf(fmaxf(fy, -2.f), (float)H + 1.f);
      const int x0 = (int)fx, y0 = (int)fy;
#pragma unroll
      for (int n = 0; n < 4; n++) {
        const int di = n & 1, dj = n >> 1;
        const int tx = x0 + di, ty = y0 + dj;
        if (tx >= 0 && tx < W && ty >= 0 && ty < H) {
          const float w = (di ? rx : 1.f - rx) * (dj ? ry : 1.f - ry);
          const int wx = tx - (tx0 - RSR), wy = ty - (ty0 - RSR);
          if (wx >= 0 && wx < RWW && wy >= 0 && wy < RWH) atomicAdd(&win[wy * RWW + wx], (unsigned)__builtin_fmaf(w, RFIX, 0.5f));
          else atomicAdd(dst + (long)ty * W + tx, w);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RWH * RWW; i += LB) {
    const unsigned q = win[i];
    if (q != 0u) {
      const float v = (float)q * (1.f / RFIX);
      const int wy = i / RWW, wx = i - wy * RWW;
      atomicAdd(dst + (long)(ty0 - RSR + wy) * W + (tx0 - RSR + wx), v);  // in range by construction: only in-image targets were added
    }
  }
}

// Gather form of the splat (same idea as warp_bwd_gather_kernel, warp.hip): LDS float atomics retire one wave instruction per ~120
// cycles on gfx950 -- the LDS window above spends half of its 99 us in its sixteen ds_add_f32 per tile.  Here a workgroup owns a 32 x 8
// tile of RANGE-MAP pixels: phase 1 (lanes = source pixels within RG_R of the tile) pushes every tap that lands in the tile onto that
// pixel's list in LDS (fixed slot per (source, tap), one integer exchange on the list head), phase 2 (one thread per pixel) sums the
// list's weights and stores -- no zero-fill, no float atomics, bit-identical from run to run for the near sources.  Sources whose taps
// lie further than RG_R pixels away are added afterwards by range_far_kernel with global float atomics.
// MEASURED (8 x 436 x 1024): 137 us against 99 us for the LDS-window splat -- with ONE value per pixel the list building (four returning
// LDS exchanges per source, ~120 cycles per wave instruction as well, and a 4.5x redundant scan of the flow field) costs more than
// the float atomics it replaces; the same scheme pays in the warp backward, where a list entry is amortised over 32 channels.  Kept
// selectable (ocf_range_map_select(0)) because its near-source part is bit-reproducible.
constexpr int RG_R = 8, RG_WW = RSW + 2 * RG_R, RG_WH = RSH + 2 * RG_R, RG_WIN = RG_WW * RG_WH;
struct RangeTap { int x0, y0; float rx, ry; bool ok; };
__device__ __forceinline__ RangeTap range_tap(const float* fl, long fo, long fsc, int x, int y, int H, int W) {
  RangeTap t;
  const float cx = (float)x + fl[fo], cy = (float)y + fl[fo + fsc];
  t.ok = cx == cx && cy == cy;
  float fx = floorf(cx), fy = floorf(cy);
  t.rx = cx - fx; t.ry = cy - fy;
  fx = fminf(fmaxf(fx, -2.f), (float)W + 1.f);
  fy = fminf(fmaxf(fy, -2.f), (float)H + 1.f);
  t.x0 = (int)fx; t.y0 = (int)fy;
  return t;
}
__device__ __forceinline__ bool range_near(const RangeTap& t, int x, int y) {
  return t.x0 >= x - RG_R && t.x0 + 1 <= x + RG_R && t.y0 >= y - RG_R && t.y0 + 1 <= y + RG_R;
}

__global__ __launch_bounds__(LB) void range_gather_kernel(T4 flow, float* range, int B, int H, int W, int tiles_x, int tiles_y) {
  __shared__ float s_w[4 * RG_WIN];
  __shared__ unsigned s_nx[4 * RG_WIN];
  __shared__ unsigned s_head[RSW * RSH];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  const int tpi = tiles_x * tiles_y;
  const int b = tile / tpi, trem = tile - b * tpi;
  const int ty0 = (trem / tiles_x) * RSH, tx0 = (trem % tiles_x) * RSW;
  const float* fl = (const float*)flow.p + (long)b * flow.sb;
  s_head[tid] = 0xffffu;
  __syncthreads();
  for (int i = tid; i < RG_WIN; i += LB) {
    const int wy = i / RG_WW, wx = i - wy * RG_WW;
    const int sx = tx0 - RG_R + wx, sy = ty0 - RG_R + wy;
    if (sx >= 0 && sx < W && sy >= 0 && sy < H) {
      const RangeTap t = range_tap(fl, (long)sy * flow.sh + (long)sx * flow.sw, flow.sc, sx, sy, H, W);
      if (t.ok && range_near(t, sx, sy)) {
        unsigned prev[4];
        bool push[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int tx = t.x0 + (k & 1), ty = t.y0 + (k >> 1);
          const int lx = tx - tx0, ly = ty - ty0;
          push[k] = tx >= 0 && tx < W && ty >= 0 && ty < H && lx >= 0 && lx < RSW && ly >= 0 && ly < RSH;
          prev[k] = 0xffffu;
          if (push[k]) prev[k] = atomicExch(&s_head[ly * RSW + lx], (unsigned)(4 * i + k));
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (push[k]) {
            s_w[4 * i + k] = ((k & 1) ? t.rx : 1.f - t.rx) * ((k >> 1) ? t.ry : 1.f - t.ry);
            s_nx[4 * i + k] = prev[k];
          }
      }
    }
  }
  __syncthreads();
  float acc = 0.f;
  for (unsigned cur = s_head[tid]; cur != 0xffffu; cur = s_nx[cur]) acc += s_w[cur];
  const int y = ty0 + tid / RSW, x = tx0 + (tid % RSW);
  if (y < H && x < W) range[((long)b * H + y) * W + x] = acc;
}

// the far sources (taps more than RG_R pixels away from the source): global float atomics, after range_gather_kernel's plain stores
__global__ __launch_bounds__(LB) void range_far_kernel(T4 flow, float* range, int B, int H, int W, long npix) {
  const float* fl = (const float*)flow.p;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    const long r = idx / W;
    const int y = r % H, b = r / H;
    const RangeTap t = range_tap(fl + (long)b * flow.sb, (long)y * flow.sh + (long)x * flow.sw, flow.sc, x, y, H, W);
    if (!t.ok || range_near(t, x, y)) continue;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int tx = t.x0 + (k & 1), ty = t.y0 + (k >> 1);
      if (tx >= 0 && tx < W && ty >= 0 && ty < H)
        atomicAdd(range + ((long)b * H + ty) * W + tx, ((k & 1) ? t.rx : 1.f - t.rx) * ((k >> 1) ? t.ry : 1.f - t.ry));
    }
  }
}

__global__ __launch_bounds__(LB) void occ_from_range_kernel(const float* range, float* occ, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    occ[i] = 1.f - fminf(fmaxf(range[i], 0.f), 1.f);
}

// ---------------------------------------------------------------------------------------- moments
template <typename T>
__global__ __launch_bounds__(LB) void moments_kernel(T4 f, double* sums, int C, int H, int W, int blocks_per_b) {
  const int b = blockIdx.x / blocks_per_b, blk = blockIdx.x % blocks_per_b;
  const long n = (long)C * H * W;
  double acc[2] = {0.0, 0.0};
  float s1 = 0.f, s2 = 0.f;
  int cnt = 0;
  for (long idx = (long)blk * LB + threadIdx.x; idx < n; idx += (long)blocks_per_b * LB) {
    // iterate in memory order of whichever layout is dense
    int c, y, x;
    long r = idx;
    if (f.sc == 1) { c = r % C; r /= C; x = r % W; y = r / W; }
    else           { x = r % W; r /= W; y = r % H; c = r / H; }
    const float v = ldf<T>(f.p, (long)b * f.sb + (long)c * f.sc + (long)y * f.sh + (long)x * f.sw);
    s1 += v; s2 = __builtin_fmaf(v, v, s2);
    if (++cnt == 64) { acc[0] += (double)s1; acc[1] += (double)s2; s1 = s2 = 0.f; cnt = 0; }
  }
  acc[0] += (double)s1; acc[1] += (double)s2;
  block_reduce_atomic<2>(acc, sums + 2 * b);
}

// A sample that is ONE contiguous run of n elements (dense NCHW or dense channels_last: what normalize_features is handed) needs no
// index arithmetic at all: 16-byte loads over the run.  The generic kernel above pays two 64-bit divisions per 2-byte element -- 12-18 us
// per launch at every pyramid level of configs[2] (20 launches per step, profiles/r5_cfg3_step_breakdown.txt).
template <typename T>
__global__ __launch_bounds__(LB) void moments_dense_kernel(const T* __restrict__ f, long sb, double* sums, long n, int blocks_per_b) {
  constexpr int U = Unit<T>::U;
  const int b = blockIdx.x / blocks_per_b, blk = blockIdx.x % blocks_per_b;
  const T* p = f + (long)b * sb;
  const long nu = n / U;
  double acc[2] = {0.0, 0.0};
  float s1 = 0.f, s2 = 0.f;
  int cnt = 0;
  for (long u = (long)blk * LB + threadIdx.x; u < nu; u += (long)blocks_per_b * LB) {
    float v[U];
    unit_to_f<T>(*(const u32x4*)(p + u * U), v);
#pragma unroll
    for (int k = 0; k < U; k++) { s1 += v[k]; s2 = __builtin_fmaf(v[k], v[k], s2); }
    if (++cnt == 8) { acc[0] += (double)s1; acc[1] += (double)s2; s1 = s2 = 0.f; cnt = 0; }
  }
  if (blk == 0) {  // the tail that does not fill a 16-byte unit
    for (long e = nu * U + threadIdx.x; e < n; e += LB) {
      const float v = (float)p[e];
      s1 += v; s2 = __builtin_fmaf(v, v, s2);
    }
  }
  acc[0] += (double)s1; acc[1] += (double)s2;
  block_reduce_atomic<2>(acc, sums + 2 * b);
}

// ------------------------------------------------------------------------------------------- SSIM
struct SsimWin { float w[16]; };

// Separable, LDS-tiled: one workgroup = a 32x8 tile of output pixels of one (sample, channel) plane.  The (8+ws-1) x (32+ws-1)
// input patches of both images are staged once (zero padding, branch-free clamped loads), the five windowed moments are
// filtered horizontally into LDS and vertically into registers: 2*ws taps per output instead of ws^2 (ws = 11: 22 vs 121,
// 877 -> ~150 us at 8 x 3 x 436 x 1024), then the SSIM quotient; block reduce -> one fp64 atomic per block and sample.
constexpr int STW = 32, STH = 8, SMAXW = 16, SPH = STH + SMAXW - 1, SPW = STW + SMAXW - 1;
// dmaps != nullptr (backward pass 1): besides the sums, store at every output pixel the partial derivatives of the SSIM quotient
// w.r.t. the three windowed moments that depend on the FIRST image, times coef[sample]:
//   D0 = dS/d mu1,  D1 = dS/d E[a^2],  D2 = dS/d E[ab]      (planes [3][B][C][Ho][Wo] fp32)
__global__ __launch_bounds__(LB) void ssim_fwd_kernel(T4 a, T4 b, double* sums, int C, int H, int W, int Ho, int Wo, int ws, int pad,
                                                      SsimWin win, int tiles_x, int ntiles, float* dmaps, const float* coef) {
  __shared__ float sa[SPH][SPW + 1], sb[SPH][SPW + 1];
  __shared__ float hm[5][SPH][STW];
  const int bi = blockIdx.z, c = blockIdx.y;
  const float* pa = (const float*)a.p + (long)bi * a.sb + (long)c * a.sc;
  const float* pb = (const float*)b.p + (long)bi * b.sb + (long)c * b.sc;
  const int ph = STH + ws - 1, pw = STW + ws - 1;
  const int lx = threadIdx.x & (STW - 1), ly = threadIdx.x >> 5;
  double acc[1] = {0.0};
  // a workgroup walks several tiles: every block ends in one fp64 atomic on its sample's sum, and ~2300 of those per address
  // (one tile per block at 436x1024) cost more than the filtering itself
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int ty0 = (tile / tiles_x) * STH, tx0 = (tile % tiles_x) * STW;
    __syncthreads();  // the previous tile's vertical pass is done with hm / sa / sb
    for (int e = threadIdx.x; e < ph * pw; e += LB) {
      const int r = e / pw, x = e - r * pw;
      const int gy = ty0 - pad + r, gx = tx0 - pad + x;
      const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
      const float va = pa[(long)gyc * a.sh + (long)gxc * a.sw], vb = pb[(long)gyc * b.sh + (long)gxc * b.sw];
      sa[r][x] = ok ? va : 0.f;
      sb[r][x] = ok ? vb : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < ph * STW; e += LB) {
      const int r = e / STW, x = e - r * STW;
      float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
      for (int j = 0; j < ws; j++) {
        const float w = win.w[j], va = sa[r][x + j], vb = sb[r][x + j];
        m1 = __builtin_fmaf(w, va, m1);
        m2 = __builtin_fmaf(w, vb, m2);
        s11 = __builtin_fmaf(w, va * va, s11);
        s22 = __builtin_fmaf(w, vb * vb, s22);
        s12 = __builtin_fmaf(w, va * vb, s12);
      }
      hm[0][r][x] = m1; hm[1][r][x] = m2; hm[2][r][x] = s11; hm[3][r][x] = s22; hm[4][r][x] = s12;
    }
    __syncthreads();
    const int yo = ty0 + ly, xo = tx0 + lx;
    if (yo < Ho && xo < Wo) {
      float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
      for (int i = 0; i < ws; i++) {
        const float w = win.w[i];
        m1 = __builtin_fmaf(w, hm[0][ly + i][lx], m1);
        m2 = __builtin_fmaf(w, hm[1][ly + i][lx], m2);
        s11 = __builtin_fmaf(w, hm[2][ly + i][lx], s11);
        s22 = __builtin_fmaf(w, hm[3][ly + i][lx], s22);
        s12 = __builtin_fmaf(w, hm[4][ly + i][lx], s12);
      }
      const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
      const float v11 = s11 - m1 * m1, v22 = s22 - m2 * m2, v12 = s12 - m1 * m2;
      const float A1 = 2.f * m1 * m2 + c1, A2 = 2.f * v12 + c2, B1 = m1 * m1 + m2 * m2 + c1, B2 = v11 + v22 + c2;
      const float S = (A1 * A2) / (B1 * B2);
      acc[0] += (double)S;
      if (dmaps) {
        const float k = coef[bi], inv = 1.f / (B1 * B2);
        const long plane = (long)Ho * Wo, o = ((long)bi * C + c) * plane + (long)yo * Wo + xo, stride = (long)gridDim.z * C * plane;
        dmaps[o] = k * (2.f * m2 * (A2 - A1) * inv - 2.f * m1 * S / B1 + 2.f * m1 * S / B2);
        dmaps[o + stride] = -k * S / B2;
        dmaps[o + 2 * stride] = k * 2.f * A1 * inv;
      }
    }
  }
  block_reduce_atomic<1>(acc, sums + bi);
}

// backward pass 2: grad_a[p] = (G^T * D0)(p) + 2 a(p) (G^T * D1)(p) + b(p) (G^T * D2)(p), G^T = the transposed (correlation) form of
// the zero-padded Gaussian filter; separable through LDS like the forward.  One workgroup = a 32 x 8 tile of INPUT pixels.
__global__ __launch_bounds__(LB) void ssim_bwd_kernel(T4 a, T4 b, const float* dmaps, T4 ga, int B, int C, int H, int W, int Ho, int Wo, int ws,
                                                      int pad, SsimWin win, int tiles_x) {
  __shared__ float sd[3][SPH][SPW + 1];
  __shared__ float hm[3][SPH][STW];
  const int bi = blockIdx.z, c = blockIdx.y;
  const int ty0 = (blockIdx.x / tiles_x) * STH, tx0 = (blockIdx.x % tiles_x) * STW;
  const int ph = STH + ws - 1, pw = STW + ws - 1;
  const long plane = (long)Ho * Wo, stride = (long)B * C * plane;
  const float* dp = dmaps + ((long)bi * C + c) * plane;
  const int uy0 = ty0 + pad - (ws - 1), ux0 = tx0 + pad - (ws - 1);  // first output row / column that reaches this tile
  for (int e = threadIdx.x; e < ph * pw; e += LB) {
    const int r = e / pw, x = e - r * pw;
    const int uy = uy0 + r, ux = ux0 + x;
    const bool ok = uy >= 0 && uy < Ho && ux >= 0 && ux < Wo;
    const long o = (long)(uy < 0 ? 0 : (uy >= Ho ? Ho - 1 : uy)) * Wo + (ux < 0 ? 0 : (ux >= Wo ? Wo - 1 : ux));
    const float d0 = dp[o], d1 = dp[o + stride], d2 = dp[o + 2 * stride];
    sd[0][r][x] = ok ? d0 : 0.f;
    sd[1][r][x] = ok ? d1 : 0.f;
    sd[2][r][x] = ok ? d2 : 0.f;
  }
  __syncthreads();
  // out(p) = sum_t w[t] D(p + pad - t): in patch coordinates x + (ws-1) - t, i.e. tap i = ws-1-t at offset i with weight w[ws-1-i]
  for (int e = threadIdx.x; e < ph * STW; e += LB) {
    const int r = e / STW, x = e - r * STW;
    float h0 = 0.f, h1 = 0.f, h2 = 0.f;
    for (int i = 0; i < ws; i++) {
      const float w = win.w[ws - 1 - i];
      h0 = __builtin_fmaf(w, sd[0][r][x + i], h0);
      h1 = __builtin_fmaf(w, sd[1][r][x + i], h1);
      h2 = __builtin_fmaf(w, sd[2][r][x + i], h2);
    }
    hm[0][r][x] = h0; hm[1][r][x] = h1; hm[2][r][x] = h2;
  }
  __syncthreads();
  const int lx = threadIdx.x & (STW - 1), ly = threadIdx.x >> 5;
  const int y = ty0 + ly, x = tx0 + lx;
  if (y < H && x < W) {
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    for (int i = 0; i < ws; i++) {
      const float w = win.w[ws - 1 - i];
      g0 = __builtin_fmaf(w, hm[0][ly + i][lx], g0);
      g1 = __builtin_fmaf(w, hm[1][ly + i][lx], g1);
      g2 = __builtin_fmaf(w, hm[2][ly + i][lx], g2);
    }
    const float va = ((const float*)a.p)[(long)bi * a.sb + (long)c * a.sc + (long)y * a.sh + (long)x * a.sw];
    const float vb = ((const float*)b.p)[(long)bi * b.sb + (long)c * b.sc + (long)y * b.sh + (long)x * b.sw];
    ((float*)ga.p)[(long)bi * ga.sb + (long)c * ga.sc + (long)y * ga.sh + (long)x * ga.sw] = g0 + 2.f * va * g1 + vb * g2;
  }
}

// ------------------------------------------------------------------------------------------ PSNR
// sums[b] = sum over (C,H,W) of (a - b)^2 for sample b (fp64); blocks_per_b blocks per sample, memory order of whichever
// layout is dense
__global__ __launch_bounds__(LB) void sqdiff_kernel(T4 a, T4 bt, double* sums, int C, int H, int W, int blocks_per_b) {
  const int b = blockIdx.x / blocks_per_b, blk = blockIdx.x % blocks_per_b;
  const long n = (long)C * H * W;
  const float* pa = (const float*)a.p;
  const float* pb = (const float*)bt.p;
  float part = 0.f;
  double acc[1] = {0.0};
  int cnt = 0;
  for (long idx = (long)blk * LB + threadIdx.x; idx < n; idx += (long)blocks_per_b * LB) {
    int c, y, x;
    long r = idx;
    if (a.sc == 1) { c = r % C; r /= C; x = r % W; y = r / W; }
    else           { x = r % W; r /= W; y = r % H; c = r / H; }
    const float d = pa[(long)b * a.sb + (long)c * a.sc + (long)y * a.sh + (long)x * a.sw] -
                    pb[(long)b * bt.sb + (long)c * bt.sc + (long)y * bt.sh + (long)x * bt.sw];
    part = __builtin_fmaf(d, d, part);
    if (++cnt == 64) { acc[0] += (double)part; part = 0.f; cnt = 0; }
  }
  acc[0] += (double)part;
  block_reduce_atomic<1>(acc, sums + b);
}


static inline int rgrid(long n) {
  long blocks = (n + LB - 1) / LB;
  return (int)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));
}
// kernels that end in block_reduce_atomic: every block issues its fp64 atomics to the SAME few addresses, which the L2
// serialises -- 4096 blocks cost more in the tail than the streaming pass itself (photo fwd 88 us vs 46 us for its backward)
static inline int rgrid_reduce(long n) {
  long blocks = (n + LB - 1) / LB;
  return (int)(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks));
}

}  // namespace ocf

using namespace ocf;

extern "C" int ocf_photo_loss_fwd(const ocf_tensor4* img1, const ocf_tensor4* img2, const ocf_tensor4* flow,
                                  const ocf_tensor4* occ, double* sums, int B, int C, int H, int W, void* stream) {
  ocf::clear_stale_error();
  if (!img1 || !img2 || !sums || !img1->data || !img2->data) return OCF_EINVAL;
  if ((flow && !flow->data) || (occ && !occ->data)) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, 2 * sizeof(double), st) != hipSuccess) return OCF_ELAUNCH;
  const long npix = (long)B * H * W;
  if (C == 3)
    LaunchProf().launch(true, photo_fwd_kernel<3>, dim3(rgrid_reduce(npix)), dim3(LB), 0, st, mk(img1), mk(img2), mk(flow), mk(occ), sums, B, C,
                       H, W, WarpGeom(H, W, 1), npix);
  else
    LaunchProf().launch(true, photo_fwd_kernel<0>, dim3(rgrid_reduce(npix)), dim3(LB), 0, st, mk(img1), mk(img2), mk(flow), mk(occ), sums, B, C,
                       H, W, WarpGeom(H, W, 1), npix);
  return launch_status();
}

extern "C" int ocf_photo_loss_bwd(const ocf_tensor4* img1, const ocf_tensor4* img2, const ocf_tensor4* flow,
                                  const ocf_tensor4* occ, const float* scale, const ocf_tensor4* gflow,
                                  const ocf_tensor4* gpred, int B, int C, int H, int W, void* stream) {
  ocf::clear_stale_error();
  if (!img1 || !img2 || !scale || !img1->data || !img2->data) return OCF_EINVAL;
  if ((flow && !flow->data) || (occ && !occ->data) || (gflow && !gflow->data) || (gpred && !gpred->data)) return OCF_EINVAL;
  if ((flow && !gflow) || (!flow && !gpred)) return OCF_EINVAL;
  if (flow && gpred) return OCF_EUNSUPPORTED; /* grad w.r.t. the un-warped image through the fused warp */
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  const long npix = (long)B * H * W;
  if (C == 3)
    LaunchProf().launch(true, photo_bwd_kernel<3>, dim3(rgrid(npix)), dim3(LB), 0, (hipStream_t)stream, mk(img1), mk(img2), mk(flow),
                       mk(occ), scale, mk(gflow), mk(gpred), B, C, H, W, WarpGeom(H, W, 1), npix);
  else
    LaunchProf().launch(true, photo_bwd_kernel<0>, dim3(rgrid(npix)), dim3(LB), 0, (hipStream_t)stream, mk(img1), mk(img2), mk(flow),
                       mk(occ), scale, mk(gflow), mk(gpred), B, C, H, W, WarpGeom(H, W, 1), npix);
  return launch_status();
}

extern "C" int ocf_masked_l1_fwd(const ocf_tensor4* pred, const ocf_tensor4* target, const ocf_tensor4* mask, double* sums, int B, int C,
                                 int H, int W, void* stream) {
  ocf::clear_stale_error();
  if (!pred || !target || !mask || !sums || !pred->data || !target->data || !mask->data) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, 2 * sizeof(double), st) != hipSuccess) return OCF_ELAUNCH;
  const long npix = (long)B * H * W;
  LaunchProf().launch(true, masked_l1_fwd_kernel, dim3(rgrid_reduce(npix)), dim3(LB), 0, st, mk(pred), mk(target), mk(mask), sums, B, C, H, W, npix);
  return launch_status();
}

extern "C" int ocf_masked_l1_bwd(const ocf_tensor4* pred, const ocf_tensor4* target, const ocf_tensor4* mask, const float* scale,
                                 const ocf_tensor4* gpred, int B, int C, int H, int W, void* stream) {
  ocf::clear_stale_error();
  if (!pred || !target || !mask || !scale || !gpred || !pred->data || !target->data || !mask->data || !gpred->data) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  const long npix = (long)B * H * W;
  LaunchProf().launch(true, masked_l1_bwd_kernel, dim3(rgrid(npix)), dim3(LB), 0, (hipStream_t)stream, mk(pred), mk(target), mk(mask), scale, mk(gpred),
                     B, C, H, W, npix);
  return launch_status();
}

extern "C" int ocf_smooth_loss_fwd(const ocf_tensor4* img, const ocf_tensor4* flow, double* sums, int B, int C, int H, int W,
                                   float alpha, void* stream) {
  ocf::clear_stale_error();
  if (!img || !flow || !sums || !img->data || !flow->data) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, 4 * sizeof(double), st) != hipSuccess) return OCF_ELAUNCH;
  const long npix = (long)B * H * W;
  const int xblocks = (W + 63) / 64, ychunks = (H + SMR_F * (LB / 64) - 1) / (SMR_F * (LB / 64));
  const bool rows = C == 3 && (long)B * xblocks * ychunks <= 0x7fffffffL;
  if (rows && img->sc == 1)
    LaunchProf().launch(true, smooth_fwd_rows_kernel<true>, dim3((unsigned)(B * xblocks * ychunks)), dim3(LB), 0, st, mk(img), mk(flow), sums, H, W,
                       alpha, xblocks, ychunks);
  else if (rows)
    LaunchProf().launch(true, smooth_fwd_rows_kernel<false>, dim3((unsigned)(B * xblocks * ychunks)), dim3(LB), 0, st, mk(img), mk(flow), sums, H, W,
                       alpha, xblocks, ychunks);
  else if (C == 3)
    LaunchProf().launch(true, smooth_fwd_kernel<3>, dim3(rgrid_reduce(npix)), dim3(LB), 0, st, mk(img), mk(flow), sums, B, C, H, W, alpha, npix);
  else
    LaunchProf().launch(true, smooth_fwd_kernel<0>, dim3(rgrid_reduce(npix)), dim3(LB), 0, st, mk(img), mk(flow), sums, B, C, H, W, alpha, npix);
  return launch_status();
}

extern "C" int ocf_smooth_loss_bwd(const ocf_tensor4* img, const ocf_tensor4* flow, const float* coef, const ocf_tensor4* gflow,
                                   int B, int C, int H, int W, float alpha, void* stream) {
  ocf::clear_stale_error();
  if (!img || !flow || !coef || !gflow || !img->data || !flow->data || !gflow->data) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  const long npix = (long)B * H * W;
  const int xblocks = (W + 63) / 64, ychunks = (H + SMR_B * (LB / 64) - 1) / (SMR_B * (LB / 64));
  if (C == 3 && (long)B * xblocks * ychunks <= 0x7fffffffL && img->sc == 1)
    LaunchProf().launch(true, smooth_bwd_rows_kernel<true>, dim3((unsigned)(B * xblocks * ychunks)), dim3(LB), 0, (hipStream_t)stream, mk(img), mk(flow),
                       coef, mk(gflow), H, W, alpha, xblocks, ychunks);
  else if (C == 3 && (long)B * xblocks * ychunks <= 0x7fffffffL)
    LaunchProf().launch(true, smooth_bwd_rows_kernel<false>, dim3((unsigned)(B * xblocks * ychunks)), dim3(LB), 0, (hipStream_t)stream, mk(img), mk(flow),
                       coef, mk(gflow), H, W, alpha, xblocks, ychunks);
  else if (C == 3)
    LaunchProf().launch(true, smooth_bwd_kernel<3>, dim3(rgrid(npix)), dim3(LB), 0, (hipStream_t)stream, mk(img), mk(flow), coef, mk(gflow),
                       B, C, H, W, alpha, npix);
  else
    LaunchProf().launch(true, smooth_bwd_kernel<0>, dim3(rgrid(npix)), dim3(LB), 0, (hipStream_t)stream, mk(img), mk(flow), coef, mk(gflow),
                       B, C, H, W, alpha, npix);
  return launch_status();
}

extern "C" int ocf_range_map(const ocf_tensor4* flow, float* range, float* occ, int B, int H, int W, int policy, void* stream) {
  ocf::clear_stale_error();
  if (!flow || !flow->data || !range) return OCF_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if (policy < OCF_RANGE_POLICY_AUTO || policy > OCF_RANGE_POLICY_GATHER) return OCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long npix = (long)B * H * W;
  const int tiles_x = (W + RSW - 1) / RSW, tiles_y = (H + RSH - 1) / RSH;
  if ((long)B * tiles_x * tiles_y > 0x7fffffffL) return OCF_EUNSUPPORTED;
  LaunchProf lp;
  if (policy != OCF_RANGE_POLICY_GATHER) {  // automatic = the LDS-window splat (99 vs 137 us at 8 x 436 x 1024)
    if (hipMemsetAsync(range, 0, npix * sizeof(float), st) != hipSuccess) return OCF_ELAUNCH;
    lp.launch(!occ, range_splat_kernel, dim3((unsigned)(B * tiles_x * tiles_y)), dim3(LB), 0, st, mk(flow), range, B, H, W, tiles_x, tiles_y);
  } else {
    lp.launch(false, range_gather_kernel, dim3((unsigned)(B * tiles_x * tiles_y)), dim3(LB), 0, st, mk(flow), range, B, H, W, tiles_x, tiles_y);
    lp.launch(!occ, range_far_kernel, dim3(rgrid(npix)), dim3(LB), 0, st, mk(flow), range, B, H, W, npix);
  }
  if (occ) lp.launch(true, occ_from_range_kernel, dim3(rgrid(npix)), dim3(LB), 0, st, (const float*)range, occ, npix);
  return launch_status();
}

static int ssim_launch(const ocf_tensor4* img1, const ocf_tensor4* img2, double* sums, int B, int C, int H, int W, int window_size,
                       void* stream, float* dmaps, const float* coef, const ocf_tensor4* ga);

extern "C" int ocf_ssim_fwd(const ocf_tensor4* img1, const ocf_tensor4* img2, double* sums, int B, int C, int H, int W,
                            int window_size, void* stream) {
  return ssim_launch(img1, img2, sums, B, C, H, W, window_size, stream, nullptr, nullptr, nullptr);
}

extern "C" int ocf_ssim_bwd_workspace_bytes(int B, int C, int H, int W, int window_size, int64_t* bytes) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || window_size < 1 || !bytes) return OCF_EINVAL;
  const int pad = window_size / 2;
  *bytes = 3LL * B * C * (int64_t)(H + 2 * pad - window_size + 1) * (W + 2 * pad - window_size + 1) * 4 + (int64_t)B * 8;
  return OCF_OK;
}

extern "C" int ocf_ssim_bwd(const ocf_tensor4* img1, const ocf_tensor4* img2, const float* coef, const ocf_tensor4* gimg1, void* workspace,
                            int B, int C, int H, int W, int window_size, void* stream) {
  if (!coef || !gimg1 || !gimg1->data || !workspace) return OCF_EINVAL;
  // workspace = [B doubles of scratch sums][3 derivative planes]
  return ssim_launch(img1, img2, (double*)workspace, B, C, H, W, window_size, stream, (float*)((double*)workspace + B), coef, gimg1);
}

static int ssim_launch(const ocf_tensor4* img1, const ocf_tensor4* img2, double* sums, int B, int C, int H, int W, int window_size,
                       void* stream, float* dmaps, const float* coef, const ocf_tensor4* ga) {
  ocf::clear_stale_error();
  if (!img1 || !img2 || !sums || !img1->data || !img2->data) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || window_size < 1) return OCF_EINVAL;
  if (window_size > 16) return OCF_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, (size_t)B * sizeof(double), st) != hipSuccess) return OCF_ELAUNCH;
  SsimWin win;
  double tot = 0.0, g[16];
  for (int i = 0; i < window_size; i++) { const double d = i - window_size / 2; g[i] = exp(-(d * d) / (2.0 * 1.5 * 1.5)); tot += g[i]; }
  for (int i = 0; i < 16; i++) win.w[i] = i < window_size ? (float)(g[i] / tot) : 0.f;
  const int pad = window_size / 2, Ho = H + 2 * pad - window_size + 1, Wo = W + 2 * pad - window_size + 1;
  const int tiles_x = (Wo + STW - 1) / STW, tiles_y = (Ho + STH - 1) / STH;
  if ((long)tiles_x * tiles_y > 0x7fffffffL || C > 65535 || B > 65535) return OCF_EUNSUPPORTED;
  const int ntiles = tiles_x * tiles_y;
  int gx = (4096 + B * C - 1) / (B * C);  // ~4096 workgroups in total, each walking ntiles / gx tiles
  if (gx > ntiles) gx = ntiles;
  if (gx < 1) gx = 1;
  LaunchProf lp;
  lp.launch(!ga, ssim_fwd_kernel, dim3((unsigned)gx, (unsigned)C, (unsigned)B), dim3(LB), 0, st, mk(img1), mk(img2), sums, C, H, W, Ho, Wo,
            window_size, pad, win, tiles_x, ntiles, dmaps, coef);
  if (ga) {
    const int btx = (W + STW - 1) / STW, bty = (H + STH - 1) / STH;
    lp.launch(true, ssim_bwd_kernel, dim3((unsigned)(btx * bty), (unsigned)C, (unsigned)B), dim3(LB), 0, st, mk(img1), mk(img2), (const float*)dmaps,
              mk(ga), B, C, H, W, Ho, Wo, window_size, pad, win, btx);
  }
  return launch_status();
}

extern "C" int ocf_moments(const ocf_tensor4* f, double* sums, int B, int C, int H, int W, int dtype, void* stream) {
  ocf::clear_stale_error();
  if (!f || !f->data || !sums) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, 2 * (size_t)B * sizeof(double), st) != hipSuccess) return OCF_ELAUNCH;
  const long n = (long)C * H * W;
  if (dtype != OCF_F32 && dtype != OCF_BF16) return OCF_EINVAL;
  // dense sample (either layout), 16-byte aligned: the contiguous-run kernel
  {
    const bool cl = (f->sc == 1 || C == 1) && (f->sw == C || W == 1) && (f->sh == (int64_t)W * C || H == 1);
    const bool nchw = (f->sw == 1 || W == 1) && (f->sh == W || H == 1) && (f->sc == (int64_t)H * W || C == 1);
    const int es = dtype == OCF_F32 ? 4 : 2;
    if ((cl || nchw) && (B == 1 || f->sb >= n) && (uintptr_t)f->data % 16 == 0 && ((int64_t)f->sb * es) % 16 == 0) {
      const long nu = n / (16 / es);
      long bpb = (nu + LB * 4 - 1) / (LB * 4);  // >= 4 units per thread
      if (bpb < 1) bpb = 1;
      if (bpb > 128) bpb = 128;
      if (dtype == OCF_F32)
        hipLaunchKernelGGL(moments_dense_kernel<float>, dim3((unsigned)(B * bpb)), dim3(LB), 0, st, (const float*)f->data, (long)f->sb, sums, n, (int)bpb);
      else
        hipLaunchKernelGGL(moments_dense_kernel<bf16>, dim3((unsigned)(B * bpb)), dim3(LB), 0, st, (const bf16*)f->data, (long)f->sb, sums, n, (int)bpb);
      return launch_status();
    }
  }
  long bpb = (n + LB * 16 - 1) / (LB * 16);
  if (bpb < 1) bpb = 1;
  if (bpb > 512) bpb = 512;
  if (dtype == OCF_F32)
    hipLaunchKernelGGL(moments_kernel<float>, dim3((unsigned)(B * bpb)), dim3(LB), 0, st, mk(f), sums, C, H, W, (int)bpb);
  else if (dtype == OCF_BF16)
    hipLaunchKernelGGL(moments_kernel<bf16>, dim3((unsigned)(B * bpb)), dim3(LB), 0, st, mk(f), sums, C, H, W, (int)bpb);
  else
    return OCF_EINVAL;
  return launch_status();
}

namespace ocf {
ProfileSlot& profile_slot() {
  static thread_local ProfileSlot slot = {nullptr, nullptr};
  return slot;
}
}  // namespace ocf

extern "C" int ocf_stream_capture_id(void* stream, unsigned long long* id) {
  if (!id) return OCF_EINVAL;
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long cid = 0;
  clear_stale_error();
  if (hipStreamGetCaptureInfo((hipStream_t)stream, &status, &cid) != hipSuccess) { (void)hipGetLastError(); return OCF_ELAUNCH; }
  *id = status == hipStreamCaptureStatusActive ? cid : 0ull;
  return OCF_OK;
}

extern "C" int ocf_profile_next_launch(void* start_event, void* stop_event) {
  ocf::ProfileSlot& p = ocf::profile_slot();
  p.start = (hipEvent_t)start_event;
  p.stop = (hipEvent_t)stop_event;
  return OCF_OK;
}

extern "C" int ocf_abi_version(void) { return OCF_ABI_VERSION; }

extern "C" const char* ocf_status_string(int s) {
  switch (s) {
    case OCF_OK: return "ok";
    case OCF_EINVAL: return "invalid argument";
    case OCF_EUNSUPPORTED: return "unsupported configuration";
    case OCF_ELAUNCH: return "HIP launch error";
    case OCF_ENODEVICE: return "no HIP device";
    default: return "unknown status";
  }
}

extern "C" int ocf_device_info(char* name, int name_len, int* compute_units) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return OCF_ENODEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return OCF_ENODEVICE;
  if (name && name_len > 0) {
    int i = 0;
    for (; i < name_len - 1 && prop.gcnArchName[i]; i++) name[i] = prop.gcnArchName[i];
    name[i] = 0;
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  return OCF_OK;
}

extern "C" int ocf_sqdiff(const ocf_tensor4* img1, const ocf_tensor4* img2, double* sums, int B, int C, int H, int W, void* stream) {
  ocf::clear_stale_error();
  if (!img1 || !img2 || !sums || !img1->data || !img2->data) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, (size_t)B * sizeof(double), st) != hipSuccess) return OCF_ELAUNCH;
  const long n = (long)C * H * W;
  long bpb = (n + (long)LB * 8 - 1) / ((long)LB * 8);
  if (bpb < 1) bpb = 1;
  if (bpb > 512) bpb = 512;
  if ((long)B * bpb > 0x7fffffffL) return OCF_EUNSUPPORTED;
  hipLaunchKernelGGL(sqdiff_kernel, dim3((unsigned)(B * bpb)), dim3(LB), 0, st, mk(img1), mk(img2), sums, C, H, W, (int)bpb);
  return launch_status();
}
