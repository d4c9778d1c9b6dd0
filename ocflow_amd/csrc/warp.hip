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

// FULL (host-checked: 16-byte aligned pixels, C a whole number of units): one vector load, no per-channel tail path.  load_unit_f's
// vector-or-tail branch ends in a wait, so several of them in a row are dependent round trips however the source orders them.
template <typename T, int VEC, bool FULL>
__device__ __forceinline__ void load_unit_sel(const T4& t, long pix_off, int c0, int C, float* f) {
  if (FULL) unit_to_f<T>(*(const u32x4*)(t.p + (pix_off + c0) * (long)sizeof(T)), f);
  else load_unit_f<T, VEC>(t, pix_off, c0, C, f);
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void warp_fwd_kernel(T4 img, T4 flow, T4 out, int B, int C, int H, int W, float fs,
                                                       WarpGeom wg, long npix) {
  constexpr int U = Unit<T>::U;
  const float* fl = (const float*)flow.p;
  for (long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
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
// FULL: C is a whole number of units with VEC = 16 (host-checked): the loads have no per-channel tail path -- with it every tap is a
// vector-or-tail branch that ends in a wait, i.e. the four taps are four dependent round trips however they are written
template <typename T, int VEC, bool FULL = false>
__global__ __launch_bounds__(256) void warp_fwd_cl_kernel(T4 img, T4 flow, T4 out, int B, int C, int H, int W, float fs, WarpGeom wg,
                                                          long total, int upp) {
  constexpr int U = Unit<T>::U;
  const float* fl = (const float*)flow.p;
  for (long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long pix = idx / upp;
    const int c0 = (int)(idx - pix * upp) * U;
    const int x = pix % W;
    const long r = pix / W;
    const int y = r % H, b = r / H;
    const long fo = (long)b * flow.sb + (long)y * flow.sh + (long)x * flow.sw;
    const float uu = fl[fo] * fs, vv = fl[fo + flow.sc] * fs;
    const Bilin s = bilin_setup(((float)x + uu) * wg.ax + wg.bx, ((float)y + vv) * wg.ay + wg.by, H, W);
    // all four neighbour loads are issued before the first use: a tap outside the image loads the output pixel's own position (always
    // valid) and is skipped by a SELECT on the accumulator -- a branch per tap made the four loads four dependent round trips
    const long own = (long)b * img.sb + (long)y * img.sh + (long)x * img.sw;
    float f[4][U];
    bool ok[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int ddx = n & 1, ddy = n >> 1;
      ok[n] = (ddx ? s.vx1 : s.vx0) & (ddy ? s.vy1 : s.vy0);
      const long off = ok[n] ? (long)b * img.sb + (long)(s.y0 + ddy) * img.sh + (long)(s.x0 + ddx) * img.sw : own;
      if (FULL) unit_to_f<T>(*(const u32x4*)(img.p + (off + c0) * (long)sizeof(T)), f[n]);
      else load_unit_f<T, VEC>(img, off, c0, C, f[n]);
    }
    float a[U];
#pragma unroll
    for (int j = 0; j < U; j++) a[j] = 0.f;
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int ddx = n & 1, ddy = n >> 1;
      const float w = (ddx ? s.wx1 : 1.f - s.wx1) * (ddy ? s.wy1 : 1.f - s.wy1);
#pragma unroll
      for (int j = 0; j < U; j++) a[j] = ok[n] ? __builtin_fmaf(w, f[n][j], a[j]) : a[j];
    }
    if (FULL) *(u32x4*)(out.p + ((long)b * out.sb + (long)y * out.sh + (long)x * out.sw + c0) * (long)sizeof(T)) = f_to_unit<T>(a);
    else store_unit_f<T, VEC>(out, (long)b * out.sb + (long)y * out.sh + (long)x * out.sw, c0, C, a);
  }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void warp_bwd_kernel(T4 img, T4 flow, T4 gout, T4 gimg, T4 gflow, int B, int C, int H, int W,
                                                       float fs, WarpGeom wg, long npix) {
  constexpr int U = Unit<T>::U;
  const float* fl = (const float*)flow.p;
  for (long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
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
  const long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
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


// ---------------------------------------------------------------------------------------------------------------------------------
// Backward, channels_last, without the global float-atomic bound (warp_bwd_cl_kernel runs at the chip-wide atomic rate: 4 atomics per
// (pixel, channel), 114 MB of them at the 8 x 32 x 109 x 256 level = 100 us).  Measured on the way: LDS float atomics are no way out --
// ds_add_f32 retires ~1 wave instruction per 120 cycles and ds_add_u32 per ~20-30, so a destination tile accumulated with them ran
// 180 / 40 us.  What is fast is a GATHER, and LDS integer atomics are only needed to build its index:
//   kernel A (warp_bwd_gather_kernel): one workgroup owns a 32 x 8 tile of grad_img pixels x 32 channels.  Phase 1, lanes = the source
//     pixels within NEAR_R of the tile: each bilinear tap that lands inside the tile is pushed onto that destination pixel's linked
//     list in LDS (one slot counter, one ds_wrxchg per tap: ~2000 integer atomics per tile instead of 61,000 float ones).  Phase 2,
//     lanes = (destination pixel, channel): walk the pixel's list, sum weight * grad_out[source] in a register, store -- plain
//     stores, no zero-fill of grad_img, eight pixels per half-wave in flight so that the grad_out loads (L2) overlap.
//     A source is "near" when its four taps lie within NEAR_R pixels of itself; near sources are handled here, by exactly the tiles
//     their taps fall into.
//   kernel B (warp_bwd_flow_kernel), afterwards: lanes = (source pixel, 16-byte channel unit): grad_flow = sum_c grad_out * d(sample)/
//     d(flow) by a shuffle reduction over the pixel's unit lanes, written with a plain store (no zero-fill, no atomics), and the
//     grad_img float atomics of the FAR sources (large motion) only.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int WT_W = 32, WT_H = 8, NEAR_R = 6, WWIN_W = WT_W + 2 * NEAR_R, WWIN_H = WT_H + 2 * NEAR_R, WWIN = WWIN_W * WWIN_H;
constexpr unsigned WL_END = 0xffffu;

__device__ __forceinline__ bool warp_is_near(const Bilin& s, int x, int y) {
  return s.x0 >= x - NEAR_R && s.x0 + 1 <= x + NEAR_R && s.y0 >= y - NEAR_R && s.y0 + 1 <= y + NEAR_R;
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void warp_bwd_gather_kernel(T4 flow, T4 gout, T4 gimg, int B, int C, int H, int W, float fs, WarpGeom wg,
                                                              int tiles_x, int tiles_y) {
  constexpr int ES = (int)sizeof(T), U = Unit<T>::U, NU = 32 / U;  // 16-byte units per 32-channel block
  __shared__ float s_w[4 * WWIN];        // list entry (window source i, tap k) lives in slot 4 i + k: weight
  __shared__ unsigned s_nx[4 * WWIN];    //                                                            next entry of the same pixel
  __shared__ unsigned s_head[WT_W * WT_H];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x, c0 = blockIdx.y * 32;
  const int tpi = tiles_x * tiles_y;
  const int b = tile / tpi, trem = tile - b * tpi;
  const int ty0 = (trem / tiles_x) * WT_H, tx0 = (trem % tiles_x) * WT_W;
  const float* fl = (const float*)flow.p + (long)b * flow.sb;
  s_head[tid] = WL_END;  // 256 threads == WT_W * WT_H
  __syncthreads();
  // ---- phase 1, lanes = window sources: push every in-tile, in-image tap of every near source onto its destination pixel's list.
  // The slot of an entry is fixed (4 i + k), so the only atomic is the exchange on the pixel's head; the four of a lane are
  // independent and in flight together.
  for (int i = tid; i < WWIN; i += 256) {
    const int wy = i / WWIN_W, wx = i - wy * WWIN_W;
    const int sx = tx0 - NEAR_R + wx, sy = ty0 - NEAR_R + wy;
    if (sx >= 0 && sx < W && sy >= 0 && sy < H) {
      const long fo = (long)sy * flow.sh + (long)sx * flow.sw;
      const float uu = fl[fo] * fs, vv = fl[fo + flow.sc] * fs;
      const Bilin s = bilin_setup(((float)sx + uu) * wg.ax + wg.bx, ((float)sy + vv) * wg.ay + wg.by, H, W);
      if (warp_is_near(s, sx, sy)) {
        unsigned prev[4];
        bool push[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int ddx = k & 1, ddy = k >> 1;
          const int lx = s.x0 + ddx - tx0, ly = s.y0 + ddy - ty0;
          push[k] = ((ddx ? s.vx1 : s.vx0) & (ddy ? s.vy1 : s.vy0)) && lx >= 0 && lx < WT_W && ly >= 0 && ly < WT_H;
          prev[k] = WL_END;
          if (push[k]) prev[k] = atomicExch(&s_head[ly * WT_W + lx], (unsigned)(4 * i + k));
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (push[k]) {
            s_w[4 * i + k] = ((k & 1) ? s.wx1 : 1.f - s.wx1) * ((k >> 1) ? s.wy1 : 1.f - s.wy1);
            s_nx[4 * i + k] = prev[k];
          }
      }
    }
  }
  __syncthreads();
  // ---- phase 2: ONE THREAD PER DESTINATION PIXEL: walk the pixel's list two entries at a time, 32 channel sums in registers, 16-byte
  // loads of the sources' grad_out rows (L2).  (Lanes = (pixel, channel) with a half-wave per pixel serialised 32 list walks per thread
  // and ran 100 us at the level-2 shape: every walk is a chain of dependent reads.)
  const char* gb = gout.p + (long)b * gout.sb * ES;
  float acc[32];
#pragma unroll
  for (int k = 0; k < 32; k++) acc[k] = 0.f;
  auto src_row = [&](unsigned e) __attribute__((always_inline)) {
    const int i = (int)(e >> 2);
    const int wy = i / WWIN_W, wx = i - wy * WWIN_W;
    return gb + ((long)(ty0 - NEAR_R + wy) * gout.sh + (long)(tx0 - NEAR_R + wx) * gout.sw + c0) * ES;
  };
  T4 gv = gout;  // unit loader view: pixel base pointer, channel offset relative to c0
  for (unsigned cur = s_head[tid]; cur != WL_END;) {
    const unsigned n0 = s_nx[cur];
    const bool two = n0 != WL_END;
    const unsigned e1 = two ? n0 : cur;
    const float w0 = s_w[cur], w1 = two ? s_w[e1] : 0.f;
    const char* r0 = src_row(cur);
    const char* r1 = src_row(e1);
    float f0[NU][U], f1[NU][U];
#pragma unroll
    for (int u = 0; u < NU; u++) {
      gv.p = (char*)r0; load_unit_sel<T, VEC, false>(gv, 0, u * U, C - c0, f0[u]);
      gv.p = (char*)r1; load_unit_sel<T, VEC, false>(gv, 0, u * U, C - c0, f1[u]);
    }
#pragma unroll
    for (int u = 0; u < NU; u++)
#pragma unroll
      for (int k = 0; k < U; k++) acc[u * U + k] = __builtin_fmaf(w1, f1[u][k], __builtin_fmaf(w0, f0[u][k], acc[u * U + k]));
    cur = two ? s_nx[e1] : WL_END;
  }
  const int y = ty0 + tid / WT_W, x = tx0 + (tid % WT_W);
  if (y < H && x < W) {
    float* gi = (float*)gimg.p + (long)b * gimg.sb + (long)y * gimg.sh + (long)x * gimg.sw + c0;
    if (c0 + 32 <= C && ((gimg.sw | gimg.sh | gimg.sb) & 3) == 0) {
#pragma unroll
      for (int k = 0; k < 32; k += 4) *(f32x4*)(gi + k) = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
    } else {
#pragma unroll
      for (int k = 0; k < 32; k++)
        if (c0 + k < C) gi[k] = acc[k];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// BINNED variant of kernel A (round 3): the window form above only serves sources whose taps land within NEAR_R = 6 pixels of themselves;
// what the model feeds it is not that tame -- with random-init weights 25 % of the level-2 and 44 % of the level-3 sources move further
// (scripts/flow_stats.py) and went through the float-atomic far path, which made the launch 2.5x slower inside the model (99.6 us at
// 8 x 32 x 109 x 256) than in the small-motion micro-benchmark (39 us).  Here the sources of a destination tile are not FOUND by scanning
// a window, they are LISTED: a binning pass (one thread per source pixel, 8 bytes of flow each) appends the source's index to the bin
// of every destination tile one of its taps lands in (1-4 tiles; one LDS atomic per (source, tile) pair, one global atomic per (workgroup, tile)), and the gather kernel
// walks its tile's bin instead of a window -- any displacement is "near".  A bin holds WBIN_CAP entries (3.5x the average of a 32 x 8
// tile); pairs beyond that go to an overflow list that the per-source kernel afterwards scatters with float atomics (taps of that
// tile only), so convergent flows stay correct.
// workspace: [tiles + 1 counters (the last one = overflow count) | bins: tiles x WBIN_CAP u32 | overflow: 4 * B*H*W x (source, tile) u32 pairs]
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int WBIN_CAP = 896;

// One returning device-scope atomic per (WORKGROUP, destination tile) instead of one per (source, tile), and all of a workgroup's
// atomics in flight together: the 256 sources of a 32 x 8 patch land in a handful of tiles; the workgroup counts its (source, tile) pairs
// per destination tile in an LDS hash table (LDS atomics: the pair's rank inside the workgroup falls out of the counting), then the
// occupied table slots reserve their bin ranges with ONE global atomic each -- issued by different lanes, so they overlap instead of
// forming the dependent chain the wave-aggregated form had (one returning atomic per distinct tile of a wave, one after the other, four
// times over: 21 us at 8 x 109 x 256 inside the model) -- and the pairs are written at base + rank.  (One atomic per pair measured 4x
// SLOWER than the window kernel: 270,000 returning atomics on 896 counters are ~50-140 us of memory-side atomic traffic there.)
constexpr int WBIN_HS = 2048;  // hash slots (a workgroup has at most 4 * 256 distinct pairs: load factor <= 0.5)
constexpr unsigned WBIN_EMPTY = 0xffffffffu;

// grid = source tiles (32 x 8 pixels, the destination tile shape), lanes = pixels: a wave = two rows of 32 sources
__global__ __launch_bounds__(256) void warp_bwd_bin_kernel(T4 flow, int B, int H, int W, float fs, WarpGeom wg, int tiles_x, int tiles_y,
                                                           unsigned* cnt, unsigned* bins, unsigned* ovf, long npix) {
  __shared__ unsigned s_key[WBIN_HS];  // destination tile of the slot
  __shared__ unsigned s_cnt[WBIN_HS];  // pairs of this workgroup for it; after the reservation: first bin slot of the workgroup's range
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < WBIN_HS / 256; k++) {
    s_key[tid + 256 * k] = WBIN_EMPTY;
    s_cnt[tid + 256 * k] = 0u;
  }
  const int tpi = tiles_x * tiles_y;
  const int b = blockIdx.x / tpi, trem = blockIdx.x - b * tpi;
  const int x = (trem % tiles_x) * WT_W + (tid & (WT_W - 1)), y = (trem / tiles_x) * WT_H + (tid >> 5);
  const bool live = x < W && y < H;
  const int xc = live ? x : 0, yc = live ? y : 0;
  const float* fl = (const float*)flow.p + (long)b * flow.sb + (long)yc * flow.sh + (long)xc * flow.sw;
  const Bilin s = bilin_setup(((float)xc + fl[0] * fs) * wg.ax + wg.bx, ((float)yc + fl[flow.sc] * fs) * wg.ay + wg.by, H, W);
  // destination tiles of the (in-image) taps: columns x0, x0 + 1 and rows y0, y0 + 1 each fall into one or two tiles
  const int ntiles = B * tpi;
  const int txa = s.vx0 ? s.x0 / WT_W : -1, txb = s.vx1 ? (s.x0 + 1) / WT_W : -1;
  const int tya = s.vy0 ? s.y0 / WT_H : -1, tyb = s.vy1 ? (s.y0 + 1) / WT_H : -1;
  const unsigned src = (unsigned)(yc * W + xc);
  const int tb = b * tpi;
  const bool xb = txb >= 0 && txb != txa, yb = tyb >= 0 && tyb != tya;
  const bool act[4] = {live && txa >= 0 && tya >= 0, live && xb && tya >= 0, live && yb && txa >= 0, live && yb && xb};
  const unsigned tl[4] = {(unsigned)(tb + tya * tiles_x + txa), (unsigned)(tb + tya * tiles_x + txb), (unsigned)(tb + tyb * tiles_x + txa),
                          (unsigned)(tb + tyb * tiles_x + txb)};
  __syncthreads();
  unsigned slot[4], rank[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    slot[k] = 0u;
    rank[k] = 0u;
    if (act[k]) {
      unsigned h = (tl[k] * 2654435761u) >> 21;  // 11 bits
      for (;;) {  // linear probing; terminates: the table has twice as many slots as the workgroup has pairs
        const unsigned prev = atomicCAS(&s_key[h], WBIN_EMPTY, tl[k]);
        if (prev == WBIN_EMPTY || prev == tl[k]) break;
        h = (h + 1u) & (unsigned)(WBIN_HS - 1);
      }
      slot[k] = h;
      rank[k] = atomicAdd(&s_cnt[h], 1u);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < WBIN_HS / 256; k++) {
    const int i = tid + 256 * k;
    const unsigned key = s_key[i];
    if (key != WBIN_EMPTY) s_cnt[i] = atomicAdd(&cnt[key], s_cnt[i]);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (act[k]) {
      const unsigned pos = s_cnt[slot[k]] + rank[k];
      if (pos < (unsigned)WBIN_CAP) {
        bins[(long)tl[k] * WBIN_CAP + pos] = src;
      } else {  // (a bin's counter keeps counting past its capacity: the gather kernel clamps it)
        const unsigned o = atomicAdd(&cnt[ntiles], 1u);
        ovf[2 * (long)o] = src;
        ovf[2 * (long)o + 1] = tl[k];
      }
    }
}

template <typename T, int VEC, bool FULL = false>
__global__ __launch_bounds__(256) void warp_bwd_gather_binned_kernel(T4 flow, T4 gout, T4 gimg, int B, int C, int H, int W, float fs,
                                                                     WarpGeom wg, int tiles_x, int tiles_y, const unsigned* cnt,
                                                                     const unsigned* bins) {
  constexpr int ES = (int)sizeof(T), U = Unit<T>::U, NU = 32 / U;
  __shared__ float s_w[4 * WBIN_CAP];      // list entry (bin entry e, tap k) lives in slot 4 e + k: weight
  __shared__ unsigned s_nx[4 * WBIN_CAP];  //                                                         next entry of the same pixel
  __shared__ unsigned s_src[WBIN_CAP];     // source pixel (y * W + x) of bin entry e
  __shared__ unsigned s_head[WT_W * WT_H];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x, c0 = blockIdx.y * 32;
  const int tpi = tiles_x * tiles_y;
  const int b = tile / tpi, trem = tile - b * tpi;
  const int ty0 = (trem / tiles_x) * WT_H, tx0 = (trem % tiles_x) * WT_W;
  const float* fl = (const float*)flow.p + (long)b * flow.sb;
  s_head[tid] = WL_END;
  const int n = min((int)cnt[tile], WBIN_CAP);
  __syncthreads();
  // ---- phase 1, lanes = bin entries: push every tap of the source that lands in THIS tile onto its destination pixel's list
  for (int e = tid; e < n; e += 256) {
    const unsigned src = bins[(long)tile * WBIN_CAP + e];
    const int sy = (int)(src / (unsigned)W), sx = (int)(src - (unsigned)sy * (unsigned)W);
    s_src[e] = ((unsigned)sy << 16) | (unsigned)sx;  // (H, W < 65536: host-checked) -- phase 2 decodes it without a division
    const long fo = (long)sy * flow.sh + (long)sx * flow.sw;
    const Bilin s = bilin_setup(((float)sx + fl[fo] * fs) * wg.ax + wg.bx, ((float)sy + fl[fo + flow.sc] * fs) * wg.ay + wg.by, H, W);
    unsigned prev[4];
    bool push[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int ddx = k & 1, ddy = k >> 1;
      const int lx = s.x0 + ddx - tx0, ly = s.y0 + ddy - ty0;
      push[k] = ((ddx ? s.vx1 : s.vx0) & (ddy ? s.vy1 : s.vy0)) && lx >= 0 && lx < WT_W && ly >= 0 && ly < WT_H;
      prev[k] = WL_END;
      if (push[k]) prev[k] = atomicExch(&s_head[ly * WT_W + lx], (unsigned)(4 * e + k));
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (push[k]) {
        s_w[4 * e + k] = ((k & 1) ? s.wx1 : 1.f - s.wx1) * ((k >> 1) ? s.wy1 : 1.f - s.wy1);
        s_nx[4 * e + k] = prev[k];
      }
  }
  __syncthreads();
  // ---- phase 2: one thread per destination pixel walks its list two entries at a time (as in the window kernel)
  const char* gb = gout.p + (long)b * gout.sb * ES;
  float acc[32];
#pragma unroll
  for (int k = 0; k < 32; k++) acc[k] = 0.f;
  auto src_row = [&](unsigned ent) __attribute__((always_inline)) {
    const unsigned src = s_src[ent >> 2];
    const int sy = (int)(src >> 16), sx = (int)(src & 0xffffu);
    return gb + ((long)sy * gout.sh + (long)sx * gout.sw + c0) * ES;
  };
  T4 gv = gout;
  for (unsigned cur = s_head[tid]; cur != WL_END;) {
    const unsigned n0 = s_nx[cur];
    const bool two = n0 != WL_END;
    const unsigned e1 = two ? n0 : cur;
    const float w0 = s_w[cur], w1 = two ? s_w[e1] : 0.f;
    const char* r0 = src_row(cur);
    const char* r1 = src_row(e1);
    float f0[NU][U], f1[NU][U];
#pragma unroll
    for (int u = 0; u < NU; u++) {
      gv.p = (char*)r0; load_unit_sel<T, VEC, FULL>(gv, 0, u * U, C - c0, f0[u]);
      gv.p = (char*)r1; load_unit_sel<T, VEC, FULL>(gv, 0, u * U, C - c0, f1[u]);
    }
#pragma unroll
    for (int u = 0; u < NU; u++)
#pragma unroll
      for (int k = 0; k < U; k++) acc[u * U + k] = __builtin_fmaf(w1, f1[u][k], __builtin_fmaf(w0, f0[u][k], acc[u * U + k]));
    cur = two ? s_nx[e1] : WL_END;
  }
  const int y = ty0 + tid / WT_W, x = tx0 + (tid % WT_W);
  if (y < H && x < W) {
    float* gi = (float*)gimg.p + (long)b * gimg.sb + (long)y * gimg.sh + (long)x * gimg.sw + c0;
    if (c0 + 32 <= C && ((gimg.sw | gimg.sh | gimg.sb) & 3) == 0) {
#pragma unroll
      for (int k = 0; k < 32; k += 4) *(f32x4*)(gi + k) = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
    } else {
#pragma unroll
      for (int k = 0; k < 32; k++)
        if (c0 + k < C) gi[k] = acc[k];
    }
  }
}

// kernel B: lanes = (source pixel, 16-byte channel unit).  UPP = units per pixel (a power of two <= 64 so that a pixel's lanes share a wave)
template <typename T, int VEC, bool FULL = false>
__global__ __launch_bounds__(256) void warp_bwd_flow_kernel(T4 img, T4 flow, T4 gout, T4 gimg, T4 gflow, int B, int C, int H, int W, float fs,
                                                            WarpGeom wg, long total, int upp, int upp_pad, const unsigned* ovf_cnt,
                                                            const unsigned* ovf, int tiles_x, int tiles_y) {
  constexpr int U = Unit<T>::U;
  const float* fl = (const float*)flow.p;
  const long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  const bool live = idx < total;
  const long pix = live ? idx / upp_pad : 0;
  const int un = live ? (int)(idx - pix * upp_pad) : 0;
  const int c0 = un * U;
  const bool on = live && un < upp;
  const int x = pix % W;
  const long r = pix / W;
  const int y = r % H, b = r / H;
  float gx = 0.f, gy = 0.f;
  const long fo = (long)b * flow.sb + (long)y * flow.sh + (long)x * flow.sw;
  const float uu = fl[fo] * fs, vv = fl[fo + flow.sc] * fs;
  const Bilin s = bilin_setup(((float)x + uu) * wg.ax + wg.bx, ((float)y + vv) * wg.ay + wg.by, H, W);
  if (on) {
    float g[U], v[4][U];
    bool ok[4];
    load_unit_sel<T, VEC, FULL>(gout, (long)b * gout.sb + (long)y * gout.sh + (long)x * gout.sw, c0, C, g);
#pragma unroll
    for (int n = 0; n < 4; n++) {  // (all five loads in flight before the first use: the masks are applied afterwards)
      const int ddx = n & 1, ddy = n >> 1;
      ok[n] = (ddx ? s.vx1 : s.vx0) & (ddy ? s.vy1 : s.vy0);
      const int xc = min(max(s.x0 + ddx, 0), W - 1), yc = min(max(s.y0 + ddy, 0), H - 1);
      load_unit_sel<T, VEC, FULL>(img, (long)b * img.sb + (long)yc * img.sh + (long)xc * img.sw, c0, C, v[n]);
    }
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
      for (int k = 0; k < U; k++) v[n][k] = ok[n] ? v[n][k] : 0.f;
#pragma unroll
    for (int k = 0; k < U; k++) {
      gx = __builtin_fmaf(g[k], (1.f - s.wy1) * (v[1][k] - v[0][k]) + s.wy1 * (v[3][k] - v[2][k]), gx);
      gy = __builtin_fmaf(g[k], (1.f - s.wx1) * (v[2][k] - v[0][k]) + s.wx1 * (v[3][k] - v[1][k]), gy);
    }
  }
  // ---- far sources (the gather kernel did not take them): float atomics, but COMPACTED.  Done by the lanes that hold the pixel
  // (8 channels x 4 taps per lane, any lane of the wave far => 32 sparsely populated atomic instructions for the whole wave) this ran
  // 237 us inside the benchmark model at 8x32x109x256, where 25 % of the sources are far -- 2.3x the plain scatter kernel.  Instead
  // the wave walks its far pixels one at a time (ballot; the pixel's sample geometry broadcast through SGPRs) with lanes = (2 taps x
  // 32 channels): every atomic instruction is two full 128-byte rows.
  if (gimg.p && ovf_cnt) {
    // BINNED mode: nothing is "far"; what is scattered here are the (source, tile) pairs that did not fit their tile's bin -- normally
    // none.  Wave w of the grid takes overflow entries w, w + #waves, ...; lanes = (2 taps in x) x 32 channels, taps outside that
    // tile are skipped (they were binned, or overflowed, with their own tile).
    const unsigned novf = *ovf_cnt;
    if (novf) {
      const int lane = threadIdx.x & 63, ch = lane & 31, ddx = lane >> 5;
      const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
      const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
      for (long e = wave0; e < (long)novf; e += nwaves) {
        const unsigned src = ovf[2 * e], tl = ovf[2 * e + 1];
        const int tpi = tiles_x * tiles_y;
        const int ob = (int)(tl / (unsigned)tpi), trem = (int)(tl - (unsigned)ob * (unsigned)tpi);
        const int oty0 = (trem / tiles_x) * WT_H, otx0 = (trem % tiles_x) * WT_W;
        const int sy = (int)(src / (unsigned)W), sx = (int)(src - (unsigned)sy * (unsigned)W);
        const long ofo = (long)ob * flow.sb + (long)sy * flow.sh + (long)sx * flow.sw;
        const Bilin os = bilin_setup(((float)sx + fl[ofo] * fs) * wg.ax + wg.bx, ((float)sy + fl[ofo + flow.sc] * fs) * wg.ay + wg.by, H, W);
        const int tx = os.x0 + ddx;
        const bool vx = (ddx ? os.vx1 : os.vx0) && tx >= otx0 && tx < otx0 + WT_W;
        const float wxx = ddx ? os.wx1 : 1.f - os.wx1;
        const long go = (long)ob * gout.sb + (long)sy * gout.sh + (long)sx * gout.sw;
        float* gi = (float*)gimg.p + (long)ob * gimg.sb + (long)tx * gimg.sw;
        for (int cb = 0; cb < C; cb += 32) {
          const int c = cb + ch;
          const float gv = c < C ? ldf<T>(gout.p, go + c) : 0.f;
#pragma unroll
          for (int ddy = 0; ddy < 2; ddy++) {
            const int tyy = os.y0 + ddy;
            const bool ok = vx && (ddy ? os.vy1 : os.vy0) && tyy >= oty0 && tyy < oty0 + WT_H && c < C;
            if (ok) atomicAdd(gi + (long)tyy * gimg.sh + c, wxx * (ddy ? os.wy1 : 1.f - os.wy1) * gv);
          }
        }
      }
    }
  } else if (gimg.p) {
    const bool far = live && un == 0 && !warp_is_near(s, x, y);
    unsigned long long todo = __ballot(far);
    const int lane = threadIdx.x & 63;
    const int ch = lane & 31, ddx = lane >> 5;
    while (todo) {
      const int l = __builtin_ctzll(todo);
      todo &= todo - 1;
      const int px0 = __builtin_amdgcn_readlane(s.x0, l), py0 = __builtin_amdgcn_readlane(s.y0, l);
      const float pwx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s.wx1), l));
      const float pwy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s.wy1), l));
      const int vbits = __builtin_amdgcn_readlane((int)s.vx0 | ((int)s.vx1 << 1) | ((int)s.vy0 << 2) | ((int)s.vy1 << 3), l);
      const int sx = __builtin_amdgcn_readlane(x, l), sy = __builtin_amdgcn_readlane(y, l), sb = __builtin_amdgcn_readlane(b, l);
      const bool vx = (vbits >> ddx) & 1;
      const float wxx = ddx ? pwx : 1.f - pwx;
      const long go = (long)sb * gout.sb + (long)sy * gout.sh + (long)sx * gout.sw;
      float* gi = (float*)gimg.p + (long)sb * gimg.sb + (long)px0 * gimg.sw + (long)ddx * gimg.sw;
      for (int cb = 0; cb < C; cb += 32) {
        const int c = cb + ch;
        const float gv = c < C ? ldf<T>(gout.p, go + c) : 0.f;
#pragma unroll
        for (int ddy = 0; ddy < 2; ddy++) {
          const bool ok = vx && ((vbits >> (2 + ddy)) & 1) && c < C;
          const float w = wxx * (ddy ? pwy : 1.f - pwy);
          if (ok) atomicAdd(gi + (long)(py0 + ddy) * gimg.sh + c, w * gv);
        }
      }
    }
  }
  if (gflow.p) {
    for (int o = upp_pad >> 1; o > 0; o >>= 1) {
      gx += __shfl_xor(gx, o, 64);
      gy += __shfl_xor(gy, o, 64);
    }
    if (live && un == 0) {
      float* gf = (float*)gflow.p + (long)b * gflow.sb + (long)y * gflow.sh + (long)x * gflow.sw;
      gf[0] = gx * wg.ax * fs;
      gf[gflow.sc] = gy * wg.ay * fs;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// SMALL maps (H * W <= WSMALL_PX: the three coarsest pyramid levels, 28 x 64 / 14 x 32 / 7 x 16): the whole backward in ONE launch.
// At these sizes the three-launch binned form is launch-floor bound (5.3 + 9.9 + 5.8 us at 8 x 96 x 28 x 64 for 7.9 MB) and the scatter
// kernel needs a zero-fill of grad_img and of grad_flow beside it.  Here a whole IMAGE is the gather window: a workgroup that owns a
// 32 x 8 destination tile x 32 channels lists the taps of ALL H * W sources of its image (no "near" test, no bins, no overflow, nothing
// far: no atomics on global memory and no zero-fill), then gathers as warp_bwd_gather_kernel does.  The per-source half (grad_flow) is
// done by further workgroups of the SAME grid (blockIdx.x >= nA), lanes = (source pixel, 16-byte unit) as in warp_bwd_flow_kernel.
// grad_img is written in OT: fp32 (the ABI's default) or the feature dtype (policy bit OCF_WARP_BWD_GIMG_NATIVE: saves the cast).
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int WSMALL_PX = 1792;  // 16 bytes of dynamic LDS per source (28 KB), list slots addressed by 16-bit links

template <typename T, typename OT, int VEC, bool FULL>
__global__ __launch_bounds__(256) void warp_bwd_small_kernel(T4 img, T4 flow, T4 gout, T4 gimg, T4 gflow, int B, int C, int H, int W, float fs,
                                                             WarpGeom wg, int tiles_x, int tiles_y, int ncb, int nA, long totalB, int upp,
                                                             int upp_pad) {
  constexpr int ES = (int)sizeof(T), U = Unit<T>::U, NU = 32 / U;
  extern __shared__ unsigned s_dyn[];  // [H W] (wx1, wy1) of every source (float2) | [4 H W] next entry of slot 4 i + tap (u16)
  __shared__ unsigned s_head[WT_W * WT_H];
  const int tid = threadIdx.x;
  const int HW = H * W;
  if ((int)blockIdx.x < nA) {
    if (!gimg.p) return;
    float* s_wxy = (float*)s_dyn;
    unsigned short* s_nx = (unsigned short*)(s_dyn + 2 * HW);
    const int tc = blockIdx.x, tile = tc / ncb, c0 = (tc - tile * ncb) * 32;
    const int tpi = tiles_x * tiles_y;
    const int b = tile / tpi, trem = tile - b * tpi;
    const int ty0 = (trem / tiles_x) * WT_H, tx0 = (trem % tiles_x) * WT_W;
    const float* fl = (const float*)flow.p + (long)b * flow.sb;
    s_head[tid] = WL_END;
    // this thread's <= 7 sources: ALL flow loads first (one round trip), then the list pushes
    constexpr int SPT = WSMALL_PX / 256;
    float fu[SPT], fv[SPT];
#pragma unroll
    for (int k = 0; k < SPT; k++) {
      const int i = tid + 256 * k, ic = i < HW ? i : 0;
      const int sy = ic / W, sx = ic - sy * W;
      const long fo = (long)sy * flow.sh + (long)sx * flow.sw;
      fu[k] = fl[fo];
      fv[k] = fl[fo + flow.sc];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SPT; k++) {
      const int i = tid + 256 * k;
      if (i < HW) {
        const int sy = i / W, sx = i - sy * W;
        const Bilin s = bilin_setup(((float)sx + fu[k] * fs) * wg.ax + wg.bx, ((float)sy + fv[k] * fs) * wg.ay + wg.by, H, W);
        unsigned prev[4];
        bool push[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int ddx = q & 1, ddy = q >> 1;
          const int lx = s.x0 + ddx - tx0, ly = s.y0 + ddy - ty0;
          push[q] = ((ddx ? s.vx1 : s.vx0) & (ddy ? s.vy1 : s.vy0)) && lx >= 0 && lx < WT_W && ly >= 0 && ly < WT_H;
          prev[q] = WL_END;
          if (push[q]) prev[q] = atomicExch(&s_head[ly * WT_W + lx], (unsigned)(4 * i + q));
        }
        if (push[0] | push[1] | push[2] | push[3]) *(float2*)(s_wxy + 2 * i) = float2{s.wx1, s.wy1};
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (push[q]) s_nx[4 * i + q] = (unsigned short)prev[q];
      }
    }
    __syncthreads();
    // gather: EPR list entries per round (the walk through LDS is cheap, the grad_out rows are L2 round trips: a pixel with the usual
    // ~4 contributions finishes in one round); rows are kept raw (16-byte units) until the multiply-add
    constexpr int EPR = 4;
    const char* gb = gout.p + (long)b * gout.sb * ES;
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; k++) acc[k] = 0.f;
    auto src_row = [&](unsigned e) __attribute__((always_inline)) {
      const int i = (int)(e >> 2);
      const int sy = i / W, sx = i - sy * W;
      return gb + ((long)sy * gout.sh + (long)sx * gout.sw + c0) * ES;
    };
    const bool fullcb = VEC == 16 && c0 + 32 <= C;  // (workgroup-uniform) whole units only: plain vector loads, no tail branch
    const int nu_live = (min(C - c0, 32) + U - 1) / U;  // units of this channel block that hold channels at all
    T4 gv = gout;
    for (unsigned cur = s_head[tid]; cur != WL_END;) {
      unsigned e[EPR];
      float wgt[EPR];
      bool has[EPR];
      e[0] = cur;
      has[0] = true;
      unsigned nxt = s_nx[cur];  // (16-bit links: WL_END = 0xffff ends a list)
#pragma unroll
      for (int q = 1; q < EPR; q++) {
        has[q] = nxt != WL_END;
        e[q] = has[q] ? nxt : e[q - 1];  // (a finished list re-reads its last row with weight 0)
        if (has[q]) nxt = s_nx[nxt];
      }
#pragma unroll
      for (int q = 0; q < EPR; q++) {  // the tap's weight from its source's fractional sample position
        const float2 wxy = *(const float2*)(s_wxy + 2 * (e[q] >> 2));
        const float w = ((e[q] & 1) ? wxy.x : 1.f - wxy.x) * ((e[q] & 2) ? wxy.y : 1.f - wxy.y);
        wgt[q] = has[q] ? w : 0.f;
      }
      if (fullcb) {
        u32x4 raw[EPR][NU];
#pragma unroll
        for (int q = 0; q < EPR; q++) {
          const char* r = src_row(e[q]);
#pragma unroll
          for (int u = 0; u < NU; u++) raw[q][u] = *(const u32x4*)(r + u * 16);
        }
#pragma unroll
        for (int q = 0; q < EPR; q++)
#pragma unroll
          for (int u = 0; u < NU; u++) {
            float f[U];
            unit_to_f<T>(raw[q][u], f);
#pragma unroll
            for (int k = 0; k < U; k++) acc[u * U + k] = __builtin_fmaf(wgt[q], f[k], acc[u * U + k]);
          }
      } else {
        for (int q = 0; q < EPR; q++) {  // (partial channel block / 8-byte pixels: the generic unit loader, live units only)
          if (!has[q]) break;
          gv.p = (char*)src_row(e[q]);
#pragma unroll
          for (int u = 0; u < NU; u++)
            if (u < nu_live) {
              float f[U];
              load_unit_sel<T, VEC, false>(gv, 0, u * U, C - c0, f);
#pragma unroll
              for (int k = 0; k < U; k++) acc[u * U + k] = __builtin_fmaf(wgt[q], f[k], acc[u * U + k]);
            }
        }
      }
      cur = nxt;  // the entry after the last one of this round, or the end
    }
    const int y = ty0 + tid / WT_W, x = tx0 + (tid % WT_W);
    if (y < H && x < W) {
      OT* gi = (OT*)gimg.p + (long)b * gimg.sb + (long)y * gimg.sh + (long)x * gimg.sw + c0;
      const bool vecst = c0 + 32 <= C && ((gimg.sw | gimg.sh | gimg.sb) & (16 / (int)sizeof(OT) - 1)) == 0 && ((uintptr_t)gimg.p & 15) == 0;
      if (vecst && sizeof(OT) == 4) {
#pragma unroll
        for (int k = 0; k < 32; k += 4) *(f32x4*)((float*)gi + k) = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
      } else if (vecst) {
#pragma unroll
        for (int k = 0; k < 32; k += 8) {
          bf16x8 v;
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] = (bf16)acc[k + j];
          *(bf16x8*)((bf16*)gi + k) = v;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 32; k++)
          if (c0 + k < C) gi[k] = (OT)acc[k];
      }
    }
    return;
  }
  // ---- per-source half: grad_flow (plain stores)
  if (!gflow.p) return;
  const float* fl = (const float*)flow.p;
  const long idx = ((long)blockIdx.x - nA) * 256 + tid;
  const bool live = idx < totalB;
  const long pix = live ? idx / upp_pad : 0;
  const int un = live ? (int)(idx - pix * upp_pad) : 0;
  const int c0 = un * U;
  const bool on = live && un < upp;
  const int x = pix % W;
  const long r = pix / W;
  const int y = r % H, b = r / H;
  float gx = 0.f, gy = 0.f;
  const long fo = (long)b * flow.sb + (long)y * flow.sh + (long)x * flow.sw;
  const float uu = fl[fo] * fs, vv = fl[fo + flow.sc] * fs;
  const Bilin s = bilin_setup(((float)x + uu) * wg.ax + wg.bx, ((float)y + vv) * wg.ay + wg.by, H, W);
  if (on) {
    float g[U], v[4][U];
    bool ok[4];
    load_unit_sel<T, VEC, FULL>(gout, (long)b * gout.sb + (long)y * gout.sh + (long)x * gout.sw, c0, C, g);
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int ddx = n & 1, ddy = n >> 1;
      ok[n] = (ddx ? s.vx1 : s.vx0) & (ddy ? s.vy1 : s.vy0);
      const int xc = min(max(s.x0 + ddx, 0), W - 1), yc = min(max(s.y0 + ddy, 0), H - 1);
      load_unit_sel<T, VEC, FULL>(img, (long)b * img.sb + (long)yc * img.sh + (long)xc * img.sw, c0, C, v[n]);
    }
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
      for (int k = 0; k < U; k++) v[n][k] = ok[n] ? v[n][k] : 0.f;
#pragma unroll
    for (int k = 0; k < U; k++) {
      gx = __builtin_fmaf(g[k], (1.f - s.wy1) * (v[1][k] - v[0][k]) + s.wy1 * (v[3][k] - v[2][k]), gx);
      gy = __builtin_fmaf(g[k], (1.f - s.wx1) * (v[2][k] - v[0][k]) + s.wx1 * (v[3][k] - v[1][k]), gy);
    }
  }
  for (int o = upp_pad >> 1; o > 0; o >>= 1) {
    gx += __shfl_xor(gx, o, 64);
    gy += __shfl_xor(gy, o, 64);
  }
  if (live && un == 0) {
    float* gf = (float*)gflow.p + (long)b * gflow.sb + (long)y * gflow.sh + (long)x * gflow.sw;
    gf[0] = gx * wg.ax * fs;
    gf[gflow.sc] = gy * wg.ay * fs;
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
    if (vec == 16 && C % Unit<T>::U == 0)
      lp.launch(true, warp_fwd_cl_kernel<T, 16, true>, g2, b2, 0, st, mk(img), mk(flow), mk(out), B, C, H, W, fs, wg, total, upp);
    else if (vec == 16)
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

// does the gather + flow kernel pair take this request?  (channels_last, 8/16-byte aligned pixels, enough work to pay for two launches:
// below ~1M (pixel, channel) pairs the single scatter kernel is faster -- 8.1 vs 14.7 us at 8 x 196 x 7 x 16)
template <typename T>
static bool warp_bwd_use_gather(const ocf_tensor4* img, const ocf_tensor4* gout, const ocf_tensor4* gimg, const ocf_tensor4* gflow, int B, int C,
                                int H, int W, int policy) {
  const long npix = (long)B * H * W;
  const bool cl = img->sc == 1 && gout->sc == 1 && gimg && gimg->sc == 1 && C >= 4;
  const bool gf_dense = !gflow || (gflow->sw == 1 && gflow->sh == W && gflow->sc == (long)H * W && gflow->sb == 2L * H * W);
  policy &= 0xff;
  if (!cl || !gf_dense || npix * C >= (1L << 40) || policy == OCF_WARP_BWD_POLICY_SCATTER ||
      ((policy == OCF_WARP_BWD_POLICY_AUTO || policy == OCF_WARP_BWD_POLICY_SMALL) && npix * C < (1L << 20)) || (long)H * W >= (1L << 31))
    return false;
  const int vec = min_vec(vec_width<T>(img), vec_width<T>(gout));
  const int upp = (C + Unit<T>::U - 1) / Unit<T>::U;
  if (!(vec >= 8 && upp <= 64 && C % (vec / (int)sizeof(T)) == 0)) return false;
  // the grid limits of the two launches belong to the SAME decision (ocf_warp_bwd_plan and the launcher must agree on who zeroes grad_img)
  const long tiles = (long)B * ((W + WT_W - 1) / WT_W) * ((H + WT_H - 1) / WT_H);
  int upp_pad = 1;
  while (upp_pad < upp) upp_pad *= 2;
  return tiles <= 0x7fffffffL && (npix * upp_pad + 255) / 256 <= 0x7fffffffL;
}

// the single-launch small-map kernel: same layout requirements as the gather pair, whole image within the LDS list budget
template <typename T>
static bool warp_bwd_use_small(const ocf_tensor4* img, const ocf_tensor4* gout, const ocf_tensor4* gimg, const ocf_tensor4* gflow, int B, int C,
                               int H, int W, int policy) {
  const int pol = policy & 0xff;
  if (pol != OCF_WARP_BWD_POLICY_AUTO && pol != OCF_WARP_BWD_POLICY_SMALL) return false;
  if ((long)H * W > WSMALL_PX || !gimg) return false;
  return warp_bwd_use_gather<T>(img, gout, gimg, gflow, B, C, H, W, OCF_WARP_BWD_POLICY_GATHER);
}

// workspace of the binned path: counters padded to a multiple of 64 entries (bins stay 256-byte aligned)
static inline long warp_ws_counters(long tiles) { return ((tiles + 1 + 63) / 64) * 64; }

template <typename T>
static int warp_bwd_t(const ocf_tensor4* img, const ocf_tensor4* flow, const ocf_tensor4* gout, const ocf_tensor4* gimg,
                      const ocf_tensor4* gflow, void* workspace, int B, int C, int H, int W, float fs, int ac, int policy, hipStream_t st) {
  const long npix = (long)B * H * W;
  const WarpGeom wg(H, W, ac);
  // channels_last fast path: unit channel stride everywhere and a dense fp32 grad_flow the launcher can zero
  const bool cl = img->sc == 1 && gout->sc == 1 && (!gimg || gimg->sc == 1) && C >= 4;
  const bool gf_dense = !gflow || (gflow->sw == 1 && gflow->sh == W && gflow->sc == (long)H * W && gflow->sb == 2L * H * W);
  if (cl && gf_dense && npix * C < (1L << 40)) {
    int G = 1;
    while (G < 64 && C % (G * 2) == 0) G *= 2;
    const long total = npix * C;
    const long blocks = (total + 255) / 256;
    const int vec = min_vec(vec_width<T>(img), vec_width<T>(gout));
    const int upp = (C + Unit<T>::U - 1) / Unit<T>::U;
    const bool native = (policy & OCF_WARP_BWD_GIMG_NATIVE) != 0;
    policy &= 0xff;
    if (warp_bwd_use_small<T>(img, gout, gimg, gflow, B, C, H, W, policy)) {
      const int tx = (W + WT_W - 1) / WT_W, ty = (H + WT_H - 1) / WT_H, ncb = (C + 31) / 32;
      const long nA = (long)B * tx * ty * ncb;
      int upp_pad = 1;
      while (upp_pad < upp) upp_pad *= 2;
      const long totalb = gflow ? npix * upp_pad : 0, nB = (totalb + 255) / 256;
      if (nA + nB > 0x7fffffffL) return OCF_EUNSUPPORTED;
      const dim3 grid((unsigned)(nA + nB));
      const size_t lds = (size_t)H * W * 16;  // 8 bytes of sample position + 4 x 2 bytes of list links per source: 28 KB at 28 x 64
      const bool full = vec == 16 && C % Unit<T>::U == 0;
      LaunchProf lp;
#define OCF_WSMALL(OT, V, F)                                                                                                              \
  lp.launch(true, warp_bwd_small_kernel<T, OT, V, F>, grid, dim3(256), lds, st, mk(img), mk(flow), mk(gout), mk(gimg), mk(gflow), B, C, H, W, \
            fs, wg, tx, ty, ncb, (int)nA, totalb, upp, upp_pad)
      if (native) {
        if (full) OCF_WSMALL(T, 16, true); else if (vec == 16) OCF_WSMALL(T, 16, false); else OCF_WSMALL(T, 8, false);
      } else {
        if (full) OCF_WSMALL(float, 16, true); else if (vec == 16) OCF_WSMALL(float, 16, false); else OCF_WSMALL(float, 8, false);
      }
#undef OCF_WSMALL
      return launch_status();
    }
    if (native) return OCF_EINVAL;  // (only the plan that answered "native grad_img allowed" may be given one)
    if (warp_bwd_use_gather<T>(img, gout, gimg, gflow, B, C, H, W, policy)) {
      // gather kernel for the near sources (plain stores: writes the whole of grad_img, no zero-fill needed), then the per-source
      // kernel: grad_flow (plain stores) + float atomics for the far sources
      const int tx = (W + WT_W - 1) / WT_W, ty = (H + WT_H - 1) / WT_H;
      const long tiles = (long)B * tx * ty;
      int upp_pad = 1;
      while (upp_pad < upp) upp_pad *= 2;
      const long totalb = npix * upp_pad;
      const dim3 gb((unsigned)((totalb + 255) / 256));
      const bool binned = workspace && policy != OCF_WARP_BWD_POLICY_GATHER && H < 65536 && W < 65536;
      if (binned) {
        // counters (zeroed here) | bins | overflow pairs: see the note above warp_bwd_bin_kernel
        unsigned* cnt = (unsigned*)workspace;
        unsigned* bins = cnt + warp_ws_counters(tiles);
        unsigned* ovf = bins + tiles * WBIN_CAP;
        if (hipMemsetAsync(cnt, 0, (size_t)warp_ws_counters(tiles) * 4, st) != hipSuccess) return OCF_ELAUNCH;
        LaunchProf lp;
        lp.launch(false, warp_bwd_bin_kernel, dim3((unsigned)tiles), dim3(256), 0, st, mk(flow), B, H, W, fs, wg, tx, ty, cnt, bins, ovf, npix);
        const bool full32 = vec == 16 && C % 32 == 0, full = vec == 16 && C % Unit<T>::U == 0;  // (see load_unit_sel)
        if (full32)
          lp.launch(false, warp_bwd_gather_binned_kernel<T, 16, true>, dim3((unsigned)tiles, (unsigned)(C / 32)), dim3(256), 0, st, mk(flow),
                    mk(gout), mk(gimg), B, C, H, W, fs, wg, tx, ty, (const unsigned*)cnt, (const unsigned*)bins);
        else if (vec == 16)
          lp.launch(false, warp_bwd_gather_binned_kernel<T, 16>, dim3((unsigned)tiles, (unsigned)((C + 31) / 32)), dim3(256), 0, st, mk(flow),
                    mk(gout), mk(gimg), B, C, H, W, fs, wg, tx, ty, (const unsigned*)cnt, (const unsigned*)bins);
        else
          lp.launch(false, warp_bwd_gather_binned_kernel<T, 8>, dim3((unsigned)tiles, (unsigned)((C + 31) / 32)), dim3(256), 0, st, mk(flow),
                    mk(gout), mk(gimg), B, C, H, W, fs, wg, tx, ty, (const unsigned*)cnt, (const unsigned*)bins);
        if (full)
          lp.launch(true, warp_bwd_flow_kernel<T, 16, true>, gb, dim3(256), 0, st, mk(img), mk(flow), mk(gout), mk(gimg), mk(gflow), B, C, H, W, fs,
                    wg, totalb, upp, upp_pad, (const unsigned*)(cnt + tiles), (const unsigned*)ovf, tx, ty);
        else if (vec == 16)
          lp.launch(true, warp_bwd_flow_kernel<T, 16>, gb, dim3(256), 0, st, mk(img), mk(flow), mk(gout), mk(gimg), mk(gflow), B, C, H, W, fs, wg,
                    totalb, upp, upp_pad, (const unsigned*)(cnt + tiles), (const unsigned*)ovf, tx, ty);
        else
          lp.launch(true, warp_bwd_flow_kernel<T, 8>, gb, dim3(256), 0, st, mk(img), mk(flow), mk(gout), mk(gimg), mk(gflow), B, C, H, W, fs, wg,
                    totalb, upp, upp_pad, (const unsigned*)(cnt + tiles), (const unsigned*)ovf, tx, ty);
        return launch_status();
      }
      {
        LaunchProf lp;
        if (vec == 16)
          lp.launch(false, warp_bwd_gather_kernel<T, 16>, dim3((unsigned)tiles, (unsigned)((C + 31) / 32)), dim3(256), 0, st, mk(flow), mk(gout),
                    mk(gimg), B, C, H, W, fs, wg, tx, ty);
        else
          lp.launch(false, warp_bwd_gather_kernel<T, 8>, dim3((unsigned)tiles, (unsigned)((C + 31) / 32)), dim3(256), 0, st, mk(flow), mk(gout),
                    mk(gimg), B, C, H, W, fs, wg, tx, ty);
        if (vec == 16)
          lp.launch(true, warp_bwd_flow_kernel<T, 16>, gb, dim3(256), 0, st, mk(img), mk(flow), mk(gout), mk(gimg), mk(gflow), B, C, H, W, fs, wg,
                    totalb, upp, upp_pad, (const unsigned*)nullptr, (const unsigned*)nullptr, tx, ty);
        else
          lp.launch(true, warp_bwd_flow_kernel<T, 8>, gb, dim3(256), 0, st, mk(img), mk(flow), mk(gout), mk(gimg), mk(gflow), B, C, H, W, fs, wg,
                    totalb, upp, upp_pad, (const unsigned*)nullptr, (const unsigned*)nullptr, tx, ty);
        return launch_status();
      }
    }
    if (gflow && hipMemsetAsync(gflow->data, 0, (size_t)npix * 2 * sizeof(float), st) != hipSuccess) return OCF_ELAUNCH;
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

extern "C" int ocf_warp_bwd_workspace_bytes(int B, int C, int H, int W, int64_t* bytes) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !bytes) return OCF_EINVAL;
  const long tiles = (long)B * ((W + ocf::WT_W - 1) / ocf::WT_W) * ((H + ocf::WT_H - 1) / ocf::WT_H);
  *bytes = (ocf::warp_ws_counters(tiles) + tiles * ocf::WBIN_CAP + 8L * B * H * W) * 4;
  return OCF_OK;
}

extern "C" int ocf_warp_bwd_plan(const ocf_tensor4* img, const ocf_tensor4* gout, const ocf_tensor4* gimg, const ocf_tensor4* gflow, int B, int C,
                                 int H, int W, int dtype, int policy) {
  if (!img || !gout || !img->data || !gout->data) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if ((policy & ~OCF_WARP_BWD_GIMG_NATIVE) < OCF_WARP_BWD_POLICY_AUTO || (policy & ~OCF_WARP_BWD_GIMG_NATIVE) > OCF_WARP_BWD_POLICY_SMALL) return OCF_EINVAL;
  bool gather = false, small = false;
  if (dtype == OCF_F32) {
    small = ocf::warp_bwd_use_small<float>(img, gout, gimg, gflow, B, C, H, W, policy);
    gather = small || ocf::warp_bwd_use_gather<float>(img, gout, gimg, gflow, B, C, H, W, policy);
  } else if (dtype == OCF_BF16) {
    small = ocf::warp_bwd_use_small<ocf::bf16>(img, gout, gimg, gflow, B, C, H, W, policy);
    gather = small || ocf::warp_bwd_use_gather<ocf::bf16>(img, gout, gimg, gflow, B, C, H, W, policy);
  } else return OCF_EINVAL;
  // bit 0: grad_img must be zeroed by the caller (scatter kernel: float atomics); bit 1: grad_img may be given in `dtype` instead of fp32
  // (pass OCF_WARP_BWD_GIMG_NATIVE with the policy then): the single-launch small-map kernel writes it with plain stores
  return (gather ? 0 : 1) | (small ? 2 : 0);
}

extern "C" int ocf_warp_bwd(const ocf_tensor4* img, const ocf_tensor4* flow, const ocf_tensor4* gout, const ocf_tensor4* gimg,
                            const ocf_tensor4* gflow, int B, int C, int H, int W, float flow_scale, int align_corners,
                            int dtype, int policy, void* workspace, void* stream) {
  ocf::clear_stale_error();
  if (!img || !flow || !gout || !img->data || !flow->data || !gout->data) return OCF_EINVAL;
  if ((gimg && !gimg->data) || (gflow && !gflow->data) || (!gimg && !gflow)) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if ((policy & ~OCF_WARP_BWD_GIMG_NATIVE) < OCF_WARP_BWD_POLICY_AUTO || (policy & ~OCF_WARP_BWD_GIMG_NATIVE) > OCF_WARP_BWD_POLICY_SMALL) return OCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OCF_F32) return ocf::warp_bwd_t<float>(img, flow, gout, gimg, gflow, workspace, B, C, H, W, flow_scale, align_corners, policy, st);
  if (dtype == OCF_BF16)
    return ocf::warp_bwd_t<ocf::bf16>(img, flow, gout, gimg, gflow, workspace, B, C, H, W, flow_scale, align_corners, policy, st);
  return OCF_EINVAL;
}
