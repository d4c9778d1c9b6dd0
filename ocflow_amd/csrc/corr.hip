// K1+K2: local cost volume, optionally fused with the backward warp (and occlusion scaling) of
// the second feature map.  gfx950 only.
//
// Forward kernel (corr_fwd_tiled):
//   workgroup  = one 32x8 pixel tile of one image pair, (2d+1) waves: wave w owns the displacement
//                row dy = w-d, lane l owns a strip of 4 consecutive pixels (sx = l&7, row = l>>3).
//   HBM -> LDS = per block of 4 channel "units" (a unit = 16 B = 8 bf16 / 4 fp32 channels) the f1
//                tile and the f2 halo tile (+-d) are staged ONCE; with flow != NULL the halo tile is
//                produced by the bilinear gather itself (the warped tensor never exists in HBM).
//   LDS layout = [unit][row][x&3][x>>2] in 16-B slots, row pitch 40 slots: every ds_read_b128 of a
//                wave (4-pixel strips => lane stride 4 pixels) is bank-conflict free.
//   compute    = per unit 4 + (4+2d) ds_read_b128 feed 4*(2d+1)*... v_dot2c_f32_bf16 (bf16) or
//                v_fma_f32 (fp32); each f2 slot read is reused by up to 4 pixels of the strip.
//   store      = 1/C scaling fused; 4 pixels x (2d+1) dx per lane, vector stores when NCHW.
//
// reference semantics: models/networks/correlation_layer.py:7-40 (+ warp cost_volume_flow_net.py:121-151)
#include "ocf_common.h"

namespace ocf {

constexpr int TW = 32, TH = 8, PX = 4, NU = 4, NQP = 10;

template <int D> struct Geo {
  static constexpr int N = 2 * D + 1;
  static constexpr int HH = TH + 2 * D, HW = TW + 2 * D;
  static constexpr int F2_SLOTS = NU * HH * 4 * NQP;
  static constexpr int F1_SLOTS = NU * TH * 4 * NQP;
  static constexpr int NT = 64 * N;
};

template <typename T, bool VEC, bool WARP, int D>
__global__ __launch_bounds__(64 * (2 * D + 1)) void corr_fwd_tiled(T4 f1, T4 f2, T4 flow, T4 occ, T4 out, int B, int C,
                                                                  int H, int W, float fs, WarpGeom wg, int tiles_x,
                                                                  int tiles_y, int out_vec) {
  using G = Geo<D>;
  constexpr int U = Unit<T>::U;
  __shared__ u32x4 s_f2[G::F2_SLOTS];
  __shared__ u32x4 s_f1[G::F1_SLOTS];

  const int tid = threadIdx.x;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tpi = tiles_x * tiles_y;
  const int b = tile / tpi;
  const int trem = tile - b * tpi;
  const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;

  const int wave = tid >> 6, lane = tid & 63;
  const int sx = lane & 7, ry = lane >> 3;

  float acc[PX][G::N];
#pragma unroll
  for (int i = 0; i < PX; i++)
#pragma unroll
    for (int j = 0; j < G::N; j++) acc[i][j] = 0.f;

  const int ncb = (C + NU * U - 1) / (NU * U);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  for (int cb = 0; cb < ncb; cb++) {
    const int cbase = cb * NU * U;
    // ---------------- stage f2 halo tile (plain copy, or bilinear gather when WARP) -------------
    for (int it = tid; it < G::HH * G::HW * NU; it += G::NT) {
      int u, xl, r;
      if (VEC) { u = it % NU; int pos = it / NU; xl = pos % G::HW; r = pos / G::HW; }
      else     { xl = it % G::HW; int q = it / G::HW; r = q % G::HH; u = q / G::HH; }
      const int gy = ty0 - D + r, gx = tx0 - D + xl, c0 = cbase + u * U;
      u32x4 val = zero4;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W && c0 < C) {
        if (!WARP) {
          const long po = (long)b * f2.sb + (long)gy * f2.sh + (long)gx * f2.sw;
          if (occ.p == nullptr) {
            val = load_unit<T, VEC>(f2, po, c0, C);
          } else {
            float f[U];
            load_unit_f<T, VEC>(f2, po, c0, C, f);
            const float o = ((const float*)occ.p)[(long)b * occ.sb + (long)gy * occ.sh + (long)gx * occ.sw];
#pragma unroll
            for (int j = 0; j < U; j++) f[j] *= o;
            val = f_to_unit<T>(f);
          }
        } else {
          const float* fl = (const float*)flow.p;
          const long fo = (long)b * flow.sb + (long)gy * flow.sh + (long)gx * flow.sw;
          const float uu = fl[fo] * fs, vv = fl[fo + flow.sc] * fs;
          const Bilin s = bilin_setup(((float)gx + uu) * wg.ax + wg.bx, ((float)gy + vv) * wg.ay + wg.by, H, W);
          float a[U];
#pragma unroll
          for (int j = 0; j < U; j++) a[j] = 0.f;
          const long pb = (long)b * f2.sb;
#pragma unroll
          for (int n = 0; n < 4; n++) {
            const int ddx = n & 1, ddy = n >> 1;
            const bool ok = (ddx ? s.vx1 : s.vx0) & (ddy ? s.vy1 : s.vy0);
            if (ok) {
              const float w = (ddx ? s.wx1 : 1.f - s.wx1) * (ddy ? s.wy1 : 1.f - s.wy1);
              float f[U];
              load_unit_f<T, VEC>(f2, pb + (long)(s.y0 + ddy) * f2.sh + (long)(s.x0 + ddx) * f2.sw, c0, C, f);
#pragma unroll
              for (int j = 0; j < U; j++) a[j] = __builtin_fmaf(w, f[j], a[j]);
            }
          }
          if (occ.p != nullptr) {
            const float o = ((const float*)occ.p)[(long)b * occ.sb + (long)gy * occ.sh + (long)gx * occ.sw];
#pragma unroll
            for (int j = 0; j < U; j++) a[j] *= o;
          }
          val = f_to_unit<T>(a);
        }
      }
      s_f2[((u * G::HH + r) * 4 + (xl & 3)) * NQP + (xl >> 2)] = val;
    }
    // ---------------- stage f1 tile -------------------------------------------------------------
    for (int it = tid; it < TH * TW * NU; it += G::NT) {
      int u, xl, r;
      if (VEC) { u = it % NU; int pos = it / NU; xl = pos % TW; r = pos / TW; }
      else     { xl = it % TW; int q = it / TW; r = q % TH; u = q / TH; }
      const int gy = ty0 + r, gx = tx0 + xl, c0 = cbase + u * U;
      u32x4 val = zero4;
      if (gy < H && gx < W && c0 < C)
        val = load_unit<T, VEC>(f1, (long)b * f1.sb + (long)gy * f1.sh + (long)gx * f1.sw, c0, C);
      s_f1[((u * TH + r) * 4 + (xl & 3)) * NQP + (xl >> 2)] = val;
    }
    __syncthreads();
    // ---------------- compute: wave = dy row, lane = 4-pixel strip ----------------------------
#pragma unroll
    for (int u = 0; u < NU; u++) {
      if (cbase + u * U < C) {
        u32x4 a[PX];
#pragma unroll
        for (int i = 0; i < PX; i++) a[i] = s_f1[((u * TH + ry) * 4 + i) * NQP + sx];
        const u32x4* row = &s_f2[(u * G::HH + ry + wave) * 4 * NQP + sx];
#pragma unroll
        for (int j = 0; j < PX + 2 * D; j++) {
          const u32x4 v = row[(j & 3) * NQP + (j >> 2)];
#pragma unroll
          for (int i = 0; i < PX; i++) {
            const int dxi = j - i;
            if (dxi >= 0 && dxi < G::N) acc[i][dxi] = unit_dot<T>(a[i], v, acc[i][dxi]);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---------------- epilogue: scale by 1/C and store ---------------------------------------------
  const float invC = 1.f / (float)C;
  const int y = ty0 + ry, x0 = tx0 + PX * sx;
  if (y >= H || x0 >= W) return;
  const long ob = (long)b * out.sb + (long)y * out.sh;
  if (out_vec && x0 + PX <= W) {
#pragma unroll
    for (int dxi = 0; dxi < G::N; dxi++) {
      const long o = ob + (long)(wave * G::N + dxi) * out.sc + x0;
      if (sizeof(T) == 4) {
        f32x4 v = {acc[0][dxi] * invC, acc[1][dxi] * invC, acc[2][dxi] * invC, acc[3][dxi] * invC};
        *(f32x4*)(out.p + o * 4) = v;
      } else {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 v = {(bf16)(acc[0][dxi] * invC), (bf16)(acc[1][dxi] * invC), (bf16)(acc[2][dxi] * invC), (bf16)(acc[3][dxi] * invC)};
        *(bf16x4*)(out.p + o * 2) = v;
      }
    }
  } else {
#pragma unroll
    for (int dxi = 0; dxi < G::N; dxi++) {
      const long o = ob + (long)(wave * G::N + dxi) * out.sc;
#pragma unroll
      for (int i = 0; i < PX; i++)
        if (x0 + i < W) stf<T>(out.p, o + (long)(x0 + i) * out.sw, acc[i][dxi] * invC);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Backward (v1): one thread per (pixel, channel unit); both gradients are gathers, no atomics.
// ------------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void corr_bwd_gather(T4 f1, T4 w2, T4 g, T4 g1, T4 g2, int B, int C, int H, int W, int d,
                                                       int nunits, long total) {
  constexpr int U = Unit<T>::U;
  const int n = 2 * d + 1;
  const float invC = 1.f / (float)C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int u, x, y, b;
    long r = idx;
    if (VEC) { u = r % nunits; r /= nunits; x = r % W; r /= W; y = r % H; b = r / H; }
    else     { x = r % W; r /= W; y = r % H; r /= H; u = r % nunits; b = r / nunits; }
    const int c0 = u * U;
    float a1[U], a2[U];
#pragma unroll
    for (int j = 0; j < U; j++) { a1[j] = 0.f; a2[j] = 0.f; }
    const long gb = (long)b * g.sb;
    for (int dy = -d; dy <= d; dy++) {
      for (int dx = -d; dx <= d; dx++) {
        const int k = (dy + d) * n + dx + d;
        const int y2 = y + dy, x2 = x + dx;
        if (g1.p && y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) {
          const float gk = ldf<T>(g.p, gb + (long)k * g.sc + (long)y * g.sh + (long)x * g.sw);
          float f[U];
          load_unit_f<T, VEC>(w2, (long)b * w2.sb + (long)y2 * w2.sh + (long)x2 * w2.sw, c0, C, f);
#pragma unroll
          for (int j = 0; j < U; j++) a1[j] = __builtin_fmaf(gk, f[j], a1[j]);
        }
        const int y3 = y - dy, x3 = x - dx;
        if (g2.p && y3 >= 0 && y3 < H && x3 >= 0 && x3 < W) {
          const float gk = ldf<T>(g.p, gb + (long)k * g.sc + (long)y3 * g.sh + (long)x3 * g.sw);
          float f[U];
          load_unit_f<T, VEC>(f1, (long)b * f1.sb + (long)y3 * f1.sh + (long)x3 * f1.sw, c0, C, f);
#pragma unroll
          for (int j = 0; j < U; j++) a2[j] = __builtin_fmaf(gk, f[j], a2[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < U; j++) { a1[j] *= invC; a2[j] *= invC; }
    if (g1.p) store_unit_f<T, VEC>(g1, (long)b * g1.sb + (long)y * g1.sh + (long)x * g1.sw, c0, C, a1);
    if (g2.p) store_unit_f<T, VEC>(g2, (long)b * g2.sb + (long)y * g2.sh + (long)x * g2.sw, c0, C, a2);
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <typename T> static bool vec_ok(const ocf_tensor4* t) {
  constexpr int U = Unit<T>::U;
  return t->sc == 1 && ((uintptr_t)t->data % 16 == 0) && t->sw % U == 0 && t->sh % U == 0 && t->sb % U == 0;
}

template <typename T, bool VEC, bool WARP>
static int launch_fwd_d(const T4& f1, const T4& f2, const T4& flow, const T4& occ, const T4& out, int B, int C, int H, int W,
                        int d, float fs, int ac, int out_vec, hipStream_t st) {
  const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;
  const WarpGeom wg(H, W, ac);
  const dim3 grid((unsigned)(B * tx * ty));
#define OCF_LAUNCH_D(DD)                                                                                   \
  case DD:                                                                                                 \
    hipLaunchKernelGGL((corr_fwd_tiled<T, VEC, WARP, DD>), grid, dim3(64 * (2 * DD + 1)), 0, st, f1, f2, flow, occ, out, \
                       B, C, H, W, fs, wg, tx, ty, out_vec);                                               \
    break;
  switch (d) {
    OCF_LAUNCH_D(1) OCF_LAUNCH_D(2) OCF_LAUNCH_D(3) OCF_LAUNCH_D(4)
    default: return OCF_EUNSUPPORTED;
  }
#undef OCF_LAUNCH_D
  return launch_status();
}

template <typename T>
static int corr_fwd_t(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ,
                      const ocf_tensor4* out, int B, int C, int H, int W, int d, float fs, int ac, hipStream_t st) {
  const bool vec = vec_ok<T>(f1) && vec_ok<T>(f2);
  // 4-pixel vector store per (strip, dx): needs unit x-stride and 4-element aligned rows
  const int out_vec = out->sw == 1 && ((uintptr_t)out->data % (PX * sizeof(T)) == 0) && out->sc % PX == 0 &&
                      out->sh % PX == 0 && out->sb % PX == 0;
  const T4 a = mk(f1), bq = mk(f2), fl = mk(flow), oc = mk(occ), o = mk(out);
  if (flow) {
    return vec ? launch_fwd_d<T, true, true>(a, bq, fl, oc, o, B, C, H, W, d, fs, ac, out_vec, st)
               : launch_fwd_d<T, false, true>(a, bq, fl, oc, o, B, C, H, W, d, fs, ac, out_vec, st);
  }
  return vec ? launch_fwd_d<T, true, false>(a, bq, fl, oc, o, B, C, H, W, d, fs, ac, out_vec, st)
             : launch_fwd_d<T, false, false>(a, bq, fl, oc, o, B, C, H, W, d, fs, ac, out_vec, st);
}

template <typename T>
static int corr_bwd_t(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* g, const ocf_tensor4* g1,
                      const ocf_tensor4* g2, int B, int C, int H, int W, int d, hipStream_t st) {
  constexpr int U = Unit<T>::U;
  bool vec = vec_ok<T>(f1) && vec_ok<T>(w2);
  if (g1) vec = vec && vec_ok<T>(g1);
  if (g2) vec = vec && vec_ok<T>(g2);
  const int nunits = (C + U - 1) / U;
  const long total = (long)B * H * W * nunits;
  const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  if (vec)
    hipLaunchKernelGGL((corr_bwd_gather<T, true>), dim3(blocks), dim3(256), 0, st, mk(f1), mk(w2), mk(g), mk(g1), mk(g2), B, C,
                       H, W, d, nunits, total);
  else
    hipLaunchKernelGGL((corr_bwd_gather<T, false>), dim3(blocks), dim3(256), 0, st, mk(f1), mk(w2), mk(g), mk(g1), mk(g2), B, C,
                       H, W, d, nunits, total);
  return launch_status();
}

}  // namespace ocf

extern "C" int ocf_corr_fwd(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ,
                            const ocf_tensor4* out, int B, int C, int H, int W, int d, float flow_scale, int align_corners,
                            int dtype, void* stream) {
  ocf::clear_stale_error();
  if (!f1 || !f2 || !out || !f1->data || !f2->data || !out->data) return OCF_EINVAL;
  if ((flow && !flow->data) || (occ && !occ->data)) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || d < 0) return OCF_EINVAL;
  if (d < 1 || d > OCF_MAX_DISPLACEMENT) return OCF_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OCF_F32) return ocf::corr_fwd_t<float>(f1, f2, flow, occ, out, B, C, H, W, d, flow_scale, align_corners, st);
  if (dtype == OCF_BF16) return ocf::corr_fwd_t<ocf::bf16>(f1, f2, flow, occ, out, B, C, H, W, d, flow_scale, align_corners, st);
  return OCF_EINVAL;
}

extern "C" int ocf_corr_bwd(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* gout, const ocf_tensor4* g1,
                            const ocf_tensor4* g2, int B, int C, int H, int W, int d, int dtype, void* stream) {
  ocf::clear_stale_error();
  if (!f1 || !w2 || !gout || !f1->data || !w2->data || !gout->data) return OCF_EINVAL;
  if ((g1 && !g1->data) || (g2 && !g2->data) || (!g1 && !g2)) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || d < 0) return OCF_EINVAL;
  if (d < 1 || d > OCF_MAX_DISPLACEMENT) return OCF_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OCF_F32) return ocf::corr_bwd_t<float>(f1, w2, gout, g1, g2, B, C, H, W, d, st);
  if (dtype == OCF_BF16) return ocf::corr_bwd_t<ocf::bf16>(f1, w2, gout, g1, g2, B, C, H, W, d, st);
  return OCF_EINVAL;
}
