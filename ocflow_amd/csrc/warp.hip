// K2: backward bilinear warp with zero padding, both align_corners variants.
// reference: cost_volume_flow_net.py:121-151 (AC-False), models/model.py:191-221 (AC-True).
//
// channels_last (the throughput layout): forward = one lane per (pixel, 16-byte channel unit) -- warp_fwd_cl_kernel; backward = one
// lane per (pixel, channel) -- warp_bwd_cl_kernel: grad_img is a scatter (float atomics whose wave instructions cover whole
// contiguous channel runs of a neighbour pixel), grad_flow a wave-shuffle reduction over the pixel's channel lanes followed by
// one float atomic pair per lane group (so grad_flow IS accumulated with atomics there and is zeroed by the launcher).
// NCHW / strided views: one thread per output pixel walking the channels (warp_fwd_kernel / warp_bwd_kernel; grad_flow in
// registers, written once).  Run-to-run bit-identical variant: det.hip.
#include "ocf_common.h"

namespace ocf {

template <typename T, int VEC>
__global__ __launch_bounds__(256) void warp_fwd_kernel(T4 img, T4 flow, T4 out, int B, int C, int H, int W, float fs,
                                                       WarpGeom wg, long npix) {
  constexpr int U = Unit<T>::U;
  const float* fl = (const float*)flow.p;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    long r = idx / W;
    const int y = r % H, b = r / H;
    const long fo = (long)b * flow.sb + (long)y * flow.sh + (long)x * flow.sw;
    const float uu = fl[fo] * fs, vv = fl[fo + flow.sc] * fs;
    const Bilin s = bilin_setup(((float)x + uu) * wg.ax + wg.bx, ((float)y + vv) * wg.ay + wg.by, H, W);
    float w[4];
    long off[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int ddx = n & 1, ddy = n >> 1;
      const bool ok = (ddx ? s.vx1 : s.vx0) & (ddy ? s.vy1 : s.vy0);
      w[n] = ok ? (ddx ? s.wx1 : 1.f - s.wx1) * (ddy ? s.wy1 : 1.f - s.wy1) : 0.f;
      off[n] = ok ? (long)b * img.sb + (long)(s.y0 + ddy) * img.sh + (long)(s.x0 + ddx) * img.sw : -1;
    }
    const long oo = (long)b * out.sb + (long)y * out.sh + (long)x * out.sw;
    for (int c0 = 0; c0 < C; c0 += U) {
      float a[U];
#pragma unroll
      for (int j = 0; j < U; j++) a[j] = 0.f;
#pragma unroll
      for (int n = 0; n < 4; n++) {
        if (off[n] >= 0) {
          float f[U];
          load_unit_f<T, VEC>(img, off[n], c0, C, f);
#pragma unroll
          for (int j = 0; j < U; j++) a[j] = __builtin_fmaf(w[n], f[j], a[j]);
        }
      }
      store_unit_f<T, VEC>(out, oo, c0, C, a);
    }
  }
}

// channels_last forward: one lane per (pixel, 16-byte unit).  The lanes of a pixel read whole contiguous channel runs of each
// bilinear neighbour (coalesced) and the channel loop of warp_fwd_kernel becomes lane parallelism: 49 -> ~12 us at level 2
// (8 x 32 x 109 x 256) and the small levels leave their sequential-latency regime (26 us for 8 x 196 x 7 x 16).
// VEC = 16 / 8 as in load_unit_full (8: pixels that are a multiple of 8 bytes; the last unit may be a half unit).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void warp_fwd_cl_kernel(T4 img, T4 flow, T4 out, int B, int C, int H, int W, float fs, WarpGeom wg,
                                                          long total, int upp) {
  constexpr int U = Unit<T>::U;
  const float* fl = (const float*)flow.p;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long pix = idx / upp;
    const int c0 = (int)(idx - pix * upp) * U;
    const int x = pix % W;
    const long r = pix / W;
    const int y = r % H, b = r / H;
    const long fo = (long)b * flow.sb + (long)y * flow.sh + (long)x * flow.sw;
    const float uu = fl[fo] * fs, vv = fl[fo + flow.sc] * fs;
    const Bilin s = bilin_setup(((float)x + uu) * wg.ax + wg.bx, ((float)y + vv) * wg.ay + wg.by, H, W);
    float a[U];
#pragma unroll
    for (int j = 0; j < U; j++) a[j] = 0.f;
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int ddx = n & 1, ddy = n >> 1;
      const bool ok = (ddx ? s.vx1 : s.vx0) & (ddy ? s.vy1 : s.vy0);
      if (ok) {
        const float w = (ddx ? s.wx1 : 1.f - s.wx1) * (ddy ? s.wy1 : 1.f - s.wy1);
        const long off = (long)b * img.sb + (long)(s.y0 + ddy) * img.sh + (long)(s.x0 + ddx) * img.sw;
        float f[U];
        load_unit_f<T, VEC>(img, off, c0, C, f);
#pragma unroll
        for (int j = 0; j < U; j++) a[j] = __builtin_fmaf(w, f[j], a[j]);
      }
    }
    store_unit_f<T, VEC>(out, (long)b * out.sb + (long)y * out.sh + (long)x * out.sw, c0, C, a);
  }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void warp_bwd_kernel(T4 img, T4 flow, T4 gout, T4 gimg, T4 gflow, int B, int C, int H, int W,
                                                       float fs, WarpGeom wg, long npix) {
  constexpr int U = Unit<T>::U;
  const float* fl = (const float*)flow.p;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    long r = idx / W;
    const int y = r % H, b = r / H;
    const long fo = (long)b * flow.sb + (long)y * flow.sh + (long)x * flow.sw;
    const float uu = fl[fo] * fs, vv = fl[fo + flow.sc] * fs;
    const Bilin s = bilin_setup(((float)x + uu) * wg.ax + wg.bx, ((float)y + vv) * wg.ay + wg.by, H, W);
    bool ok[4];
    float w[4];
    long off[4], goff[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int ddx = n & 1, ddy = n >> 1;
      ok[n] = (ddx ? s.vx1 : s.vx0) & (ddy ? s.vy1 : s.vy0);
      w[n] = (ddx ? s.wx1 : 1.f - s.wx1) * (ddy ? s.wy1 : 1.f - s.wy1);
      off[n] = (long)b * img.sb + (long)(s.y0 + ddy) * img.sh + (long)(s.x0 + ddx) * img.sw;
      goff[n] = (long)b * gimg.sb + (long)(s.y0 + ddy) * gimg.sh + (long)(s.x0 + ddx) * gimg.sw;
    }
    const long go = (long)b * gout.sb + (long)y * gout.sh + (long)x * gout.sw;
    float gx = 0.f, gy = 0.f;
    for (int c0 = 0; c0 < C; c0 += U) {
      float g[U];
      load_unit_f<T, VEC>(gout, go, c0, C, g);
      float v[4][U];
#pragma unroll
      for (int n = 0; n < 4; n++) {
        if (ok[n]) {
          load_unit_f<T, VEC>(img, off[n], c0, C, v[n]);
          if (gimg.p) {
#pragma unroll
            for (int j = 0; j < U; j++)
              if (c0 + j < C) atomicAdd((float*)gimg.p + goff[n] + (long)(c0 + j) * gimg.sc, w[n] * g[j]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < U; j++) v[n][j] = 0.f;
        }
      }
      // d out / d px = (1-wy)(v01 - v00) + wy (v11 - v10);  d out / d py = (1-wx)(v10 - v00) + wx (v11 - v01)
#pragma unroll
      for (int j = 0; j < U; j++) {
        const float dxv = (1.f - s.wy1) * (v[1][j] - v[0][j]) + s.wy1 * (v[3][j] - v[2][j]);
        const float dyv = (1.f - s.wx1) * (v[2][j] - v[0][j]) + s.wx1 * (v[3][j] - v[1][j]);
        gx = __builtin_fmaf(g[j], dxv, gx);
        gy = __builtin_fmaf(g[j], dyv, gy);
      }
    }
    if (gflow.p) {
      float* gf = (float*)gflow.p;
      const long o = (long)b * gflow.sb + (long)y * gflow.sh + (long)x * gflow.sw;
      gf[o] = gx * wg.ax * fs;
      gf[o + gflow.sc] = gy * wg.ay * fs;
    }
  }
}

// Backward, channels_last fast path: one LANE per (pixel, channel), channel fastest.  The 4 float-atomic instructions of a
// wave then each cover whole contiguous channel runs of a neighbour pixel (C=32: two 128-byte rows per instruction, the
// shape the memory-side atomic units run at full rate: MI355X_MICROARCH "Global float atomics"), instead of 64 lanes in 64
// different rows.  grad_flow: shuffle-reduce over the G lanes of a pixel that share a wave, then one atomic per group
// (gflow zeroed by the launcher).
template <typename T>
__global__ __launch_bounds__(256) void warp_bwd_cl_kernel(T4 img, T4 flow, T4 gout, T4 gimg, T4 gflow, int B, int C, int H, int W,
                                                          float fs, WarpGeom wg, long total, int G) {
  const float* fl = (const float*)flow.p;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = idx < total;
  const long pix = live ? idx / C : 0;
  const int c = live ? (int)(idx - pix * C) : 0;
  const int x = pix % W;
  const long r2 = pix / W;
  const int y = r2 % H, b = r2 / H;
  float gx = 0.f, gy = 0.f;
  if (live) {
    const long fo = (long)b * flow.sb + (long)y * flow.sh + (long)x * flow.sw;
    const float uu = fl[fo] * fs, vv = fl[fo + flow.sc] * fs;
    const Bilin s = bilin_setup(((float)x + uu) * wg.ax + wg.bx, ((float)y + vv) * wg.ay + wg.by, H, W);
    const float g = ldf<T>(gout.p, (long)b * gout.sb + (long)y * gout.sh + (long)x * gout.sw + c);
    float v[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int ddx = n & 1, ddy = n >> 1;
      const bool ok = (ddx ? s.vx1 : s.vx0) & (ddy ? s.vy1 : s.vy0);
      v[n] = 0.f;
      if (ok) {
        const long po = (long)(s.y0 + ddy) * img.sh + (long)(s.x0 + ddx) * img.sw;
        v[n] = ldf<T>(img.p, (long)b * img.sb + po + c);
        if (gimg.p) {
          const float w = (ddx ? s.wx1 : 1.f - s.wx1) * (ddy ? s.wy1 : 1.f - s.wy1);
          atomicAdd((float*)gimg.p + (long)b * gimg.sb + (long)(s.y0 + ddy) * gimg.sh + (long)(s.x0 + ddx) * gimg.sw + c, w * g);
        }
      }
    }
    gx = g * ((1.f - s.wy1) * (v[1] - v[0]) + s.wy1 * (v[3] - v[2]));
    gy = g * ((1.f - s.wx1) * (v[2] - v[0]) + s.wx1 * (v[3] - v[1]));
  }
  if (gflow.p) {
    float* gf = (float*)gflow.p;
    const long o = (long)b * gflow.sb + (long)y * gflow.sh + (long)x * gflow.sw;
    if (C >= 64) {
      // a wave (64 consecutive (pixel, channel) lanes) covers at most two pixels: one segmented wave sum per pixel and one
      // atomic pair per (wave, pixel).  With the power-of-two groups below, C = 196 (G = 4) issued 49 atomic pairs per pixel
      // to the same two addresses: 34 us for the 7x16 top level, most of it same-address serialisation.
      const long p0 = __shfl(pix, 0, 64);
      const bool first = live && pix == p0, second = live && pix != p0;
      const float ax = wave_sum(first ? gx : 0.f), ay = wave_sum(first ? gy : 0.f);
      const float bx = wave_sum(second ? gx : 0.f), by = wave_sum(second ? gy : 0.f);
      const int lane = threadIdx.x & 63;
      // the first lane of each segment issues that segment's atomics (lane 0, and the first lane whose pixel differs)
      const unsigned long long m2 = __ballot(second);
      if (lane == 0 && live) {
        atomicAdd(gf + o, ax * wg.ax * fs);
        atomicAdd(gf + o + gflow.sc, ay * wg.ay * fs);
      }
      if (m2 != 0ull && lane == __ffsll((long long)m2) - 1) {
        atomicAdd(gf + o, bx * wg.ax * fs);
        atomicAdd(gf + o + gflow.sc, by * wg.ay * fs);
      }
    } else {
      // G = largest power of two <= 64 dividing C: lanes [k*G, (k+1)*G) of a wave belong to one pixel
      for (int oo = G >> 1; oo > 0; oo >>= 1) {
        gx += __shfl_xor(gx, oo, 64);
        gy += __shfl_xor(gy, oo, 64);
      }
      if (live && (threadIdx.x & (G - 1)) == 0) {
        atomicAdd(gf + o, gx * wg.ax * fs);
        atomicAdd(gf + o + gflow.sc, gy * wg.ay * fs);
      }
    }
  }
}

template <typename T> static bool wvec_ok(const ocf_tensor4* t) {
  constexpr int U = Unit<T>::U;
  return t->sc == 1 && ((uintptr_t)t->data % 16 == 0) && t->sw % U == 0 && t->sh % U == 0 && t->sb % U == 0;
}

static inline int grid_for(long n) {
  long blocks = (n + 255) / 256;
  return (int)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
}

template <typename T>
static int warp_fwd_t(const ocf_tensor4* img, const ocf_tensor4* flow, const ocf_tensor4* out, int B, int C, int H, int W,
                      float fs, int ac, hipStream_t st) {
  const long npix = (long)B * H * W;
  const WarpGeom wg(H, W, ac);
  const int vec = min_vec(vec_width<T>(img), vec_width<T>(out));
  if (vec >= 8 && C >= 2 * Unit<T>::U) {  // channels_last with several units per pixel: lanes over (pixel, unit)
    const int upp = (C + Unit<T>::U - 1) / Unit<T>::U;
    const long total = npix * upp;
    const dim3 g2(grid_for(total)), b2(256);
    LaunchProf lp;
    if (vec == 16)
      lp.launch(true, warp_fwd_cl_kernel<T, 16>, g2, b2, 0, st, mk(img), mk(flow), mk(out), B, C, H, W, fs, wg, total, upp);
    else
      lp.launch(true, warp_fwd_cl_kernel<T, 8>, g2, b2, 0, st, mk(img), mk(flow), mk(out), B, C, H, W, fs, wg, total, upp);
    return launch_status();
  }
  const dim3 grid(grid_for(npix)), block(256);
  if (vec == 16)
    hipLaunchKernelGGL((warp_fwd_kernel<T, 16>), grid, block, 0, st, mk(img), mk(flow), mk(out), B, C, H, W, fs, wg, npix);
  else if (vec == 8)
    hipLaunchKernelGGL((warp_fwd_kernel<T, 8>), grid, block, 0, st, mk(img), mk(flow), mk(out), B, C, H, W, fs, wg, npix);
  else
    hipLaunchKernelGGL((warp_fwd_kernel<T, 0>), grid, block, 0, st, mk(img), mk(flow), mk(out), B, C, H, W, fs, wg, npix);
  return launch_status();
}

template <typename T>
static int warp_bwd_t(const ocf_tensor4* img, const ocf_tensor4* flow, const ocf_tensor4* gout, const ocf_tensor4* gimg,
                      const ocf_tensor4* gflow, int B, int C, int H, int W, float fs, int ac, hipStream_t st) {
  const long npix = (long)B * H * W;
  const WarpGeom wg(H, W, ac);
  // channels_last fast path: unit channel stride everywhere and a dense fp32 grad_flow the launcher can zero
  const bool cl = img->sc == 1 && gout->sc == 1 && (!gimg || gimg->sc == 1) && C >= 4;
  const bool gf_dense = !gflow || (gflow->sw == 1 && gflow->sh == W && gflow->sc == (long)H * W && gflow->sb == 2L * H * W);
  if (cl && gf_dense && npix * C < (1L << 40)) {
    if (gflow && hipMemsetAsync(gflow->data, 0, (size_t)npix * 2 * sizeof(float), st) != hipSuccess) return OCF_ELAUNCH;
    int G = 1;
    while (G < 64 && C % (G * 2) == 0) G *= 2;
    const long total = npix * C;
    const long blocks = (total + 255) / 256;
    LaunchProf lp;
    lp.launch(true, warp_bwd_cl_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, mk(img), mk(flow), mk(gout), mk(gimg), mk(gflow), B, C, H,
              W, fs, wg, total, G);
    return launch_status();
  }
  if (wvec_ok<T>(img) && wvec_ok<T>(gout))
    hipLaunchKernelGGL((warp_bwd_kernel<T, true>), dim3(grid_for(npix)), dim3(256), 0, st, mk(img), mk(flow), mk(gout),
                       mk(gimg), mk(gflow), B, C, H, W, fs, wg, npix);
  else
    hipLaunchKernelGGL((warp_bwd_kernel<T, false>), dim3(grid_for(npix)), dim3(256), 0, st, mk(img), mk(flow), mk(gout),
                       mk(gimg), mk(gflow), B, C, H, W, fs, wg, npix);
  return launch_status();
}

}  // namespace ocf

extern "C" int ocf_warp_fwd(const ocf_tensor4* img, const ocf_tensor4* flow, const ocf_tensor4* out, int B, int C, int H,
                            int W, float flow_scale, int align_corners, int dtype, void* stream) {
  ocf::clear_stale_error();
  if (!img || !flow || !out || !img->data || !flow->data || !out->data) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OCF_F32) return ocf::warp_fwd_t<float>(img, flow, out, B, C, H, W, flow_scale, align_corners, st);
  if (dtype == OCF_BF16) return ocf::warp_fwd_t<ocf::bf16>(img, flow, out, B, C, H, W, flow_scale, align_corners, st);
  return OCF_EINVAL;
}

extern "C" int ocf_warp_bwd(const ocf_tensor4* img, const ocf_tensor4* flow, const ocf_tensor4* gout, const ocf_tensor4* gimg,
                            const ocf_tensor4* gflow, int B, int C, int H, int W, float flow_scale, int align_corners,
                            int dtype, void* stream) {
  ocf::clear_stale_error();
  if (!img || !flow || !gout || !img->data || !flow->data || !gout->data) return OCF_EINVAL;
  if ((gimg && !gimg->data) || (gflow && !gflow->data) || (!gimg && !gflow)) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OCF_F32) return ocf::warp_bwd_t<float>(img, flow, gout, gimg, gflow, B, C, H, W, flow_scale, align_corners, st);
  if (dtype == OCF_BF16)
    return ocf::warp_bwd_t<ocf::bf16>(img, flow, gout, gimg, gflow, B, C, H, W, flow_scale, align_corners, st);
  return OCF_EINVAL;
}
