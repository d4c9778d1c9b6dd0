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
#include <stdlib.h>
#include "ocf_common.h"
#include "corr_internal.h"

namespace ocf {

constexpr int TW = 32, TH = 8, PX = 4, NU = 4, NQP = 10;

template <int D> struct Geo {
  static constexpr int N = 2 * D + 1;
  static constexpr int NK = N * N;
  static constexpr int HH = TH + 2 * D, HW = TW + 2 * D;
  static constexpr int F2_SLOTS = NU * HH * 4 * NQP;
  static constexpr int F1_SLOTS = NU * TH * 4 * NQP;
  static constexpr int NT = 64 * N;
  static constexpr int F2_ITEMS = HH * HW * NU, F1_ITEMS = TH * TW * NU;
  static constexpr int F2_PER = (F2_ITEMS + NT - 1) / NT, F1_PER = (F1_ITEMS + NT - 1) / NT;
};

// output modes of the epilogue
enum { OUT_DIRECT = 0, OUT_DIRECT_VEC = 1, OUT_LDS = 2, OUT_LDS_VEC = 3 };

// Staging = "one wave per tile row": the row index is wave-uniform (scalar address math), the 64 lanes sweep the row's
// (pixel, unit) items.  VEC != 0 (channels_last): item = 4*pixel + unit, i.e. consecutive lanes read consecutive 16-byte
// units (a 32-channel bf16 pixel is 64 contiguous bytes, pixels are contiguous along x => 1 KiB per wave-instruction);
// VEC == 0 (NCHW / arbitrary strides): item = unit*ROWLEN + pixel, lanes sweep x within a channel plane.
template <int VEC, int ROWLEN>
__device__ __forceinline__ void row_item(int it, int& u, int& xl) {
  if (VEC != 0) { u = it & (NU - 1); xl = it >> 2; }
  else          { u = it / ROWLEN; xl = it - u * ROWLEN; }
}

// MODE 0: plain copy staging; 1: f2 scaled by occ while staged; 2: f2 gathered through the warp (occ optional)
template <typename T, int VEC, int MODE, int D>
__global__ __launch_bounds__(64 * (2 * D + 1)) void corr_fwd_tiled(T4 f1, T4 f2, T4 flow, T4 occ, T4 out, int B, int C,
                                                                  int H, int W, float fs, WarpGeom wg, int tiles_x,
                                                                  int tiles_y, int out_mode, float lrelu) {
  using G = Geo<D>;
  constexpr int U = Unit<T>::U;
  constexpr int ES = (int)sizeof(T);
  // ONE shared array (staging tiles, later reused as the output transposition buffer)
  __shared__ u32x4 smem[G::F2_SLOTS + G::F1_SLOTS];
  u32x4* s_f2 = smem;
  u32x4* s_f1 = smem + G::F2_SLOTS;

#ifdef OCF_TUNING_BUILD  // ablation switches exist only in the library tools/kbench builds for itself (-DOCF_TUNING_BUILD)
  const int ablate = out_mode >> 8;  // bit0 no compute, bit1 no global loads, bit2 no output
  out_mode &= 0xff;
#else
  constexpr int ablate = 0;
#endif
  const int tid = threadIdx.x;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tpi = tiles_x * tiles_y;
  const int b = tile / tpi;
  const int trem = tile - b * tpi;
  const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int sx = lane & 7, ry = lane >> 3;

  float acc[PX][G::N];
#pragma unroll
  for (int i = 0; i < PX; i++)
#pragma unroll
    for (int j = 0; j < G::N; j++) acc[i][j] = 0.f;

  const int ncb = (C + NU * U - 1) / (NU * U);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // per-image bases (64-bit once); everything inside an image is 32-bit element arithmetic (host checks the range)
  const char* f1b = f1.p + (long)b * f1.sb * ES;
  const char* f2b = f2.p + (long)b * f2.sb * ES;
  const float* flb = MODE == 2 ? (const float*)flow.p + (long)b * flow.sb : nullptr;
  const float* ocb = (MODE >= 1 && occ.p) ? (const float*)occ.p + (long)b * occ.sb : nullptr;
  const int f1sh = (int)f1.sh, f1sw = (int)f1.sw, f1sc = (int)f1.sc;
  const int f2sh = (int)f2.sh, f2sw = (int)f2.sw, f2sc = (int)f2.sc;
  T4 f1i = f1, f2i = f2;  // image-local views for the unit loaders
  f1i.p = (char*)f1b; f2i.p = (char*)f2b;

  for (int cb = 0; cb < ncb; cb++) {
    const int cbase = cb * NU * U;
    // ---------------- stage the f2 halo tile: wave w takes rows w, w+N, ... -------------------------
    for (int r = wave; r < G::HH; r += G::N) {
      const int gy = ty0 - D + r;
      const bool rowok = gy >= 0 && gy < H;
#pragma unroll
      for (int k = 0; k < (G::HW * NU + 63) / 64; k++) {
        const int it = lane + 64 * k;
        if (it < G::HW * NU) {
          int u, xl;
          row_item<VEC, G::HW>(it, u, xl);
          const int gx = tx0 - D + xl, c0 = cbase + u * U;
          u32x4 val = zero4;
          if (rowok && gx >= 0 && gx < W && c0 < C && !(ablate & 2)) {
            if (MODE == 0) {
              val = load_unit<T, VEC>(f2i, gy * f2sh + gx * f2sw, c0, C);
            } else {
              float a[U];
              if (MODE == 1) {
                load_unit_f<T, VEC>(f2i, gy * f2sh + gx * f2sw, c0, C, a);
              } else {
                const int fo = gy * (int)flow.sh + gx * (int)flow.sw;
                const float uu = flb[fo] * fs, vv = flb[fo + flow.sc] * fs;
                const Bilin sm = bilin_setup(((float)gx + uu) * wg.ax + wg.bx, ((float)gy + vv) * wg.ay + wg.by, H, W);
#pragma unroll
                for (int j = 0; j < U; j++) a[j] = 0.f;
#pragma unroll
                for (int n = 0; n < 4; n++) {
                  const int ddx = n & 1, ddy = n >> 1;
                  const bool ok = (ddx ? sm.vx1 : sm.vx0) & (ddy ? sm.vy1 : sm.vy0);
                  if (ok) {
                    const float w = (ddx ? sm.wx1 : 1.f - sm.wx1) * (ddy ? sm.wy1 : 1.f - sm.wy1);
                    float f[U];
                    load_unit_f<T, VEC>(f2i, (sm.y0 + ddy) * f2sh + (sm.x0 + ddx) * f2sw, c0, C, f);
#pragma unroll
                    for (int j = 0; j < U; j++) a[j] = __builtin_fmaf(w, f[j], a[j]);
                  }
                }
              }
              if (ocb) {
                const float o = ocb[gy * (int)occ.sh + gx * (int)occ.sw];
#pragma unroll
                for (int j = 0; j < U; j++) a[j] *= o;
              }
              val = f_to_unit<T>(a);
            }
          }
          s_f2[((u * G::HH + r) * 4 + (xl & 3)) * NQP + (xl >> 2)] = val;
        }
      }
    }
    // ---------------- stage the f1 tile -----------------------------------------------------------
    for (int r = wave; r < TH; r += G::N) {
      const int gy = ty0 + r;
#pragma unroll
      for (int k = 0; k < (TW * NU) / 64; k++) {
        const int it = lane + 64 * k;
        int u, xl;
        row_item<VEC, TW>(it, u, xl);
        const int gx = tx0 + xl, c0 = cbase + u * U;
        u32x4 val = zero4;
        if (gy < H && gx < W && c0 < C && !(ablate & 2)) val = load_unit<T, VEC>(f1i, gy * f1sh + gx * f1sw, c0, C);
        s_f1[((u * TH + r) * 4 + (xl & 3)) * NQP + (xl >> 2)] = val;
      }
    }
    __syncthreads();
    // ---------------- compute: wave = dy row, lane = 4-pixel strip ----------------------------
#pragma unroll
    for (int u = 0; u < NU; u++) {
      if (cbase + u * U < C && !(ablate & 1)) {
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

  // ---------------- epilogue: scale by 1/C (+ optional LeakyReLU) and store ---------------------------------------------
  const float invC = 1.f / (float)C;
#pragma unroll
  for (int i = 0; i < PX; i++)
#pragma unroll
    for (int dxi = 0; dxi < G::N; dxi++) {
      const float t = acc[i][dxi] * invC;
      acc[i][dxi] = t > 0.f ? t : t * lrelu;  // (lrelu = 1: the plain mean over channels, bit for bit)
    }
  const int y = ty0 + ry, x0 = tx0 + PX * sx;
  if ((ablate & 4) && acc[0][0] != 123.f) return;
  if (out_mode == OUT_LDS || out_mode == OUT_LDS_VEC) {
    // channels_last output: transpose through LDS so that global stores are whole contiguous pixel runs
    // (NK contiguous outputs per pixel, TW*NK per tile row when the tensor is dense).
    constexpr int ROWS = (sizeof(T) == 2) ? TH : TH / 2;  // rows per pass that fit the staging buffer
    static_assert(ROWS * TW * G::NK * sizeof(T) <= (G::F2_SLOTS + G::F1_SLOTS) * 16, "output tile does not fit LDS");
    T* s_out = (T*)smem;
    const int vrows = min(TH, H - ty0), vcols = min(TW, W - tx0);
#pragma unroll
    for (int pass = 0; pass < TH / ROWS; pass++) {
      const int r0 = pass * ROWS;
      if (ry >= r0 && ry < r0 + ROWS) {
        T* dst = s_out + ((ry - r0) * TW + PX * sx) * G::NK + wave * G::N;
#pragma unroll
        for (int i = 0; i < PX; i++)
#pragma unroll
          for (int dxi = 0; dxi < G::N; dxi++) dst[i * G::NK + dxi] = (T)(acc[i][dxi]);
      }
      __syncthreads();
      const int nrows = min(ROWS, vrows - r0);
      if (out_mode == OUT_LDS_VEC) {
        // dense channels_last: each tile row is ONE contiguous run of vcols*NK elements, 16-byte aligned at both ends
        const int row_elems = vcols * G::NK;
        const int nvec = (row_elems * (int)sizeof(T)) / 16;
        const int tail0 = nvec * (16 / (int)sizeof(T));
        for (int r = 0; r < nrows; r++) {
          char* gp = out.p + ((long)b * out.sb + (long)(ty0 + r0 + r) * out.sh + (long)tx0 * out.sw) * (long)sizeof(T);
          const u32x4* lp = (const u32x4*)(s_out + r * TW * G::NK);
          for (int v = tid; v < nvec; v += G::NT) ((u32x4*)gp)[v] = lp[v];
          for (int e = tail0 + tid; e < row_elems; e += G::NT) ((T*)gp)[e] = s_out[r * TW * G::NK + e];
        }
      } else {
        const int per_row = vcols * G::NK;
        for (int e = tid; e < nrows * per_row; e += G::NT) {
          const int r = e / per_row, q = e - r * per_row;
          const int xx = q / G::NK, k = q - xx * G::NK;
          ((T*)out.p)[(long)b * out.sb + (long)k * out.sc + (long)(ty0 + r0 + r) * out.sh + (long)(tx0 + xx) * out.sw] =
              s_out[(r * TW + xx) * G::NK + k];
        }
      }
      if (pass + 1 < TH / ROWS) __syncthreads();
    }
    return;
  }
  if (y >= H || x0 >= W) return;
  const long ob = (long)b * out.sb + (long)y * out.sh;
  if (out_mode == OUT_DIRECT_VEC && x0 + PX <= W) {
#pragma unroll
    for (int dxi = 0; dxi < G::N; dxi++) {
      const long o = ob + (long)(wave * G::N + dxi) * out.sc + x0;
      if (sizeof(T) == 4) {
        f32x4 v = {acc[0][dxi], acc[1][dxi], acc[2][dxi], acc[3][dxi]};
        *(f32x4*)(out.p + o * 4) = v;
      } else {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 v = {(bf16)(acc[0][dxi]), (bf16)(acc[1][dxi]), (bf16)(acc[2][dxi]), (bf16)(acc[3][dxi])};
        *(bf16x4*)(out.p + o * 2) = v;
      }
    }
  } else {
#pragma unroll
    for (int dxi = 0; dxi < G::N; dxi++) {
      const long o = ob + (long)(wave * G::N + dxi) * out.sc;
#pragma unroll
      for (int i = 0; i < PX; i++)
        if (x0 + i < W) stf<T>(out.p, o + (long)(x0 + i) * out.sw, acc[i][dxi]);
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
// Backward (v2, channels_last 16-byte units): LDS-tiled gather.  One workgroup = 16x4 pixel tile x 4 channel units
// (32 bf16 / 16 fp32 channels per pass), 256 threads = (pixel, unit).  Per pass the +-d halo tile of the OTHER operand
// (w2 for g1, f1 for g2) and the needed slice of grad_out are staged once in LDS:
//   g1[p][c] = 1/C sum_k gout[p][k]        * w2[p + delta_k][c]     coefficients: the thread's own pixel (tile of gout)
//   g2[p][c] = 1/C sum_k gout[q][K-1-k']   * f1[q][c], q = p + delta_k'   coefficients: the SOURCE pixel, mirrored tap (halo of gout)
// so each (2d+1)^2-tap sum reads LDS only: one broadcast 2/4-byte coefficient + one 16-byte unit per tap, 8 FMAs.
// v1 (corr_bwd_gather below) issued all 4*(2d+1)^2 loads per thread against L1/L2 and fetched 14x the algorithmic bytes from
// HBM at the level-2 shape (rocprofv3 FETCH_SIZE 924 MB); it remains the fallback for strided layouts.
// ------------------------------------------------------------------------------------------------
constexpr int BTW = 16, BTH = 4, BNU = 4;

// VEC = 16: whole 16-byte units (C % U == 0, 16-byte aligned pixels); VEC = 8: 8-byte granularity (pixel size a multiple of
// 8 bytes, e.g. C = 196 bf16 = 392-byte pixels: two 8-byte accesses per unit, the last unit may be a half unit).
// Occlusion (forward was corr(f1, w2 * occ), occ [B,1,H,W] fp32; occ.p == nullptr: none):
//   g1 pass (G2 = false): `other` is the UNMASKED w2; each staged halo unit is multiplied by occ of its pixel (the product
//                         tensor w2*occ is never materialised),
//   g2 pass (G2 = true):  acc is the gradient w.r.t. (w2*occ); the kernel stores acc * occ[q] (gradient w.r.t. w2) and, when
//                         gocc.p != nullptr, adds sum_c acc[c] * w2[q][c] to gocc[q] (4 unit lanes of a pixel reduced by shuffle,
//                         one fp32 atomic per pixel and channel block).  `w2own` = the unmasked w2 (read at the own pixel).
template <typename T, bool G2, int D, int VEC>
__global__ __launch_bounds__(256) void corr_bwd_tiled(T4 other, T4 g, T4 gout_t, T4 occ, T4 w2own, T4 gocc, int B, int C, int H, int W,
                                                      int tiles_x, int tiles_y, int ablate) {
#ifndef OCF_TUNING_BUILD
  ablate = 0;  // the ablation switches only exist in the tuning build of tools/kbench
#endif
  constexpr int U = Unit<T>::U, ES = (int)sizeof(T);
  constexpr int N = 2 * D + 1, NK = N * N;
  constexpr int HH = BTH + 2 * D, HW = BTW + 2 * D;
  constexpr int GROWS = G2 ? HH : BTH, GCOLS = G2 ? HW : BTW;   // region of grad_out staged
  // g1: LDS row image == global row image (all NK taps of the tile's own pixels).  g2: a halo pixel in halo row py is only ever
  // read with the dy rows r in [py - (BTH-1), py] -- at most BTH*N CONTIGUOUS taps -- so only that window is staged (36 of 81
  // at d = 4): 2.25x less staging traffic and LDS (46.6 -> 20.7 KB bf16), i.e. four workgroups per CU instead of two.
  constexpr int WN = BTH * N;
  constexpr int GPITCH = G2 ? WN : ((NK * ES + 7) / 8) * (8 / ES);  // g1: NK taps, pitch rounded up to 8 bytes (84 elements bf16)
  auto kstart = [](int py) __attribute__((always_inline)) {  // first tap of the window kept for halo row py
    const int rhi = py < 2 * D ? py : 2 * D;
    const int ks = NK - (rhi + 1) * N;
    return ks > NK - WN ? NK - WN : ks;
  };
  __shared__ u32x4 s_op[BNU * HH * HW];                           // [row][x][unit]
  __shared__ __attribute__((aligned(16))) T s_g[GROWS * GCOLS * GPITCH + 8];

  const int tid = threadIdx.x;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tpi = tiles_x * tiles_y;
  const int b = tile / tpi;
  const int trem = tile - b * tpi;
  const int ty0 = (trem / tiles_x) * BTH, tx0 = (trem % tiles_x) * BTW;
  const int u = tid & (BNU - 1), px = tid >> 2;
  const int lx = px & (BTW - 1), ly = px >> 4;
  const int y = ty0 + ly, x = tx0 + lx;
  const float invC = 1.f / (float)C;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ---- stage the grad_out slice once (all channel passes reuse it).  Interior tiles of a channel-contiguous grad_out copy
  // per-pixel tap windows with 8-byte vectors; edge tiles (and other layouts) go element by element.
  {
    const int gy0 = G2 ? ty0 - D : ty0, gx0 = G2 ? tx0 - D : tx0;
    if (ablate & 1) {  // kernel-tuning aid (tools/kbench, OCF_ABLATE_BWD): no grad_out staging
    } else if (G2 && g.sc == 1 && (WN * ES) % 8 == 0) {
      // per halo pixel one window of WN taps = CH eight-byte chunks at a 2-byte aligned address (unaligned global loads), any
      // pixel pitch; all of a thread's chunks are in flight before the first LDS store
      constexpr int CH = WN * ES / 8, NV = GROWS * GCOLS * CH, SB = (NV + 255) / 256;
      u32x2 val[SB];
#pragma unroll
      for (int j = 0; j < SB; j++) {
        const int v = tid + j * 256;
        // branch-free: a load inside a divergent branch is followed by s_waitcnt vmcnt(0) at the join, which serialised all of
        // a thread's loads behind each other's latency (ISA check); invalid lanes read the tile's first valid window and discard it
        const int vv = v < NV ? v : 0;
        const int pi = vv / CH, c = vv - pi * CH;
        const int r = pi / GCOLS, xx = pi - r * GCOLS;
        const int gy = gy0 + r;
        const int gx = gx0 + xx;
        const bool ok = v < NV && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
        const u32x2 ld = *(const u32x2*)(g.p + ((long)b * g.sb + (long)gyc * g.sh + (long)gxc * g.sw + kstart(r)) * ES + c * 8);
        val[j] = ok ? ld : u32x2{0u, 0u};
      }
#pragma unroll
      for (int j = 0; j < SB; j++) {
        const int v = tid + j * 256;
        if (v < NV) *(u32x2*)((char*)s_g + (long)v * 8) = val[j];   // [pixel][chunk] == [pixel][WN taps]
      }
    } else if (!G2 && g.sc == 1) {
      // g1: the NK taps of each tile pixel as CH eight-byte chunks from a 2-byte aligned address (any pixel pitch: the
      // gradient may be a channel slice of the decoder-input gradient); the last chunk holds the tail of the pixel and is read
      // element-wise so that nothing past the pixel (and the tensor) is touched
      constexpr int CH = GPITCH * ES / 8, NV = GROWS * GCOLS * CH, SB = (NV + 255) / 256, TAIL = NK - (CH - 1) * (8 / ES);
      u32x2 val[SB];
#pragma unroll
      for (int j = 0; j < SB; j++) {
        const int v = tid + j * 256;
        // branch-free (see the g2 path): the last chunk of a pixel re-reads its final full 8 bytes and shifts the tail into place
        const int vv = v < NV ? v : 0;
        const int pi = vv / CH, c = vv - pi * CH;
        const int r = pi / GCOLS, xx = pi - r * GCOLS;
        const int gy = gy0 + r;
        const int gx = gx0 + xx;
        const bool ok = v < NV && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
        const bool last = c == CH - 1;
        // the last chunk starts (8/ES - TAIL) elements early so that it ends exactly at the pixel's last tap
        const long eoff = ((long)b * g.sb + (long)gyc * g.sh + (long)gxc * g.sw) + (last ? NK - 8 / ES : c * (8 / ES));
        u32x2 ld = *(const u32x2*)(g.p + eoff * ES);
        if (last) {  // keep the TAIL newest elements, in the low positions
          constexpr int SH = (8 / ES - TAIL) * ES * 8;  // bits to drop
          const unsigned long long w = __builtin_bit_cast(unsigned long long, ld) >> SH;
          ld = __builtin_bit_cast(u32x2, w);
        }
        val[j] = ok ? ld : u32x2{0u, 0u};
      }
#pragma unroll
      for (int j = 0; j < SB; j++) {
        const int v = tid + j * 256;
        if (v < NV) *(u32x2*)((char*)s_g + (long)v * 8) = val[j];   // [pixel][chunk]
      }
    } else {
      for (int e = tid; e < GROWS * GCOLS * GPITCH; e += 256) {
        const int pi = e / GPITCH, j = e - pi * GPITCH;
        const int r = pi / GCOLS, xx = pi - r * GCOLS;
        const int k = G2 ? kstart(r) + j : j;
        const int gy = gy0 + r, gx = gx0 + xx;
        T v = (T)0.f;
        if (k < NK && gy >= 0 && gy < H && gx >= 0 && gx < W) v = ((const T*)g.p)[(long)b * g.sb + (long)gy * g.sh + (long)gx * g.sw + (long)k * g.sc];
        s_g[e] = v;
      }
    }
  }
  // gridDim.y > 1: the channel blocks of a tile are spread over workgroups (small pyramid levels: too few tiles to fill the
  // chip and a sequential channel loop per tile); each re-stages the grad_out slice, which is cheap at those sizes
  const int ncb = (C + BNU * U - 1) / (BNU * U);
  const int cb_per = (ncb + (int)gridDim.y - 1) / (int)gridDim.y;
  const int cb_lo = (int)blockIdx.y * cb_per, cb_hi = min(ncb, cb_lo + cb_per);
  for (int cb = cb_lo; cb < cb_hi; cb++) {
    const int cbase = cb * BNU * U;
    if (cb > cb_lo) __syncthreads();
    // ---- stage the halo tile of the other operand for this channel block (all of a thread's loads in flight together)
    constexpr int NIT = (HH * HW * BNU + 255) / 256;
    u32x4 sv[NIT];
    float so[NIT];
#pragma unroll
    for (int j = 0; j < NIT; j++) {
      const int it = tid + j * 256;
      const int uu = it & (BNU - 1), pos = it >> 2;
      const int xx = pos % HW, r = pos / HW;
      const int gy = ty0 - D + r, gx = tx0 - D + xx, c0 = cbase + uu * U;
      // branch-free loads (clamped address, result discarded when out of range): see the grad_out staging above
      const bool ok = !(ablate & 2) && it < HH * HW * BNU && gy >= 0 && gy < H && gx >= 0 && gx < W && c0 < C;
      const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
      const int c0c = c0 < C ? c0 : 0;
      const char* src = other.p + ((long)b * other.sb + (long)gyc * other.sh + (long)gxc * other.sw + c0c) * ES;
      u32x4 v;
      if (VEC == 16) {
        v = *(const u32x4*)src;
      } else {
        const u32x2 lo = *(const u32x2*)src;
        const bool full = c0c + U / 2 < C;
        const u32x2 hi = *(const u32x2*)(src + (full ? 8 : 0));
        v = u32x4{lo[0], lo[1], full ? hi[0] : 0u, full ? hi[1] : 0u};
      }
      so[j] = 1.f;
      if (!G2 && occ.p) so[j] = ((const float*)occ.p)[(long)b * occ.sb + (long)gyc * occ.sh + (long)gxc * occ.sw];
      sv[j] = ok ? v : zero4;
    }
#pragma unroll
    for (int j = 0; j < NIT; j++) {
      const int it = tid + j * 256;
      if (it < HH * HW * BNU) {
        u32x4 v = sv[j];
        if (!G2 && occ.p) {
          float f[U];
          unit_to_f<T>(v, f);
#pragma unroll
          for (int k = 0; k < U; k++) f[k] = (float)(T)(f[k] * so[j]);  // rounded like the materialised product w2 * occ
          v = f_to_unit<T>(f);
        }
        const int uu = it & (BNU - 1), pos = it >> 2;
        const int xx = pos % HW, r = pos / HW;
        s_op[(r * HW + xx) * BNU + uu] = v;
      }
    }
    __syncthreads();
    float acc[U];
#pragma unroll
    for (int j = 0; j < U; j++) acc[j] = 0.f;
#pragma unroll 1
    for (int r = 0; r < ((ablate & 4) ? 0 : N); r++) {
      const u32x4* orow = &s_op[((ly + r) * HW + lx) * BNU + u];
      // g1: own pixel, taps r*N .. r*N+N-1;  g2: source pixel (ly+r, lx+s) of the halo, mirrored tap NK-1-(r*N+s)
      const T* grow = G2 ? &s_g[((ly + r) * GCOLS + lx) * GPITCH + (NK - 1 - r * N) - kstart(ly + r)] : &s_g[(ly * GCOLS + lx) * GPITCH + r * N];
#pragma unroll
      for (int sft = 0; sft < N; sft++) {
        const float gk = (float)(G2 ? grow[sft * GPITCH - sft] : grow[sft]);
        float f[U];
        unit_to_f<T>(orow[sft * BNU], f);
#pragma unroll
        for (int j = 0; j < U; j++) acc[j] = __builtin_fmaf(gk, f[j], acc[j]);
      }
    }
    const int c0 = cbase + u * U;
#pragma unroll
    for (int j = 0; j < U; j++) acc[j] *= invC;
    if (G2 && occ.p) {
      // all 4 unit lanes of a pixel take part in the shuffles; out-of-image / out-of-range lanes contribute 0
      const bool live = y < H && x < W && c0 < C;
      float o = 0.f, part = 0.f;
      if (live) {
        o = ((const float*)occ.p)[(long)b * occ.sb + (long)y * occ.sh + (long)x * occ.sw];
        if (gocc.p) {
          float f[U];
          const char* src = w2own.p + ((long)b * w2own.sb + (long)y * w2own.sh + (long)x * w2own.sw + c0) * ES;
          if (VEC == 16) {
            unit_to_f<T>(*(const u32x4*)src, f);
          } else {
            const u32x2 lo = *(const u32x2*)src;
            u32x2 hi = {0u, 0u};
            if (c0 + U / 2 < C) hi = *(const u32x2*)(src + 8);
            unit_to_f<T>(u32x4{lo[0], lo[1], hi[0], hi[1]}, f);
          }
          // the reference chain rounds the gradient w.r.t. (w2*occ) to the feature dtype before this product
#pragma unroll
          for (int j = 0; j < U; j++) part = __builtin_fmaf((float)(T)acc[j], f[j], part);
        }
      }
      if (gocc.p) {
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        if (live && u == 0) atomicAdd((float*)gocc.p + ((long)b * gocc.sb + (long)y * gocc.sh + (long)x * gocc.sw), part);
      }
#pragma unroll
      for (int j = 0; j < U; j++) acc[j] = (float)(T)acc[j] * o;
    }
    if (y < H && x < W && c0 < C) {
      char* dstp = gout_t.p + ((long)b * gout_t.sb + (long)y * gout_t.sh + (long)x * gout_t.sw + c0) * ES;
      const u32x4 o = f_to_unit<T>(acc);
      if (VEC == 16) {
        *(u32x4*)dstp = o;
      } else {
        *(u32x2*)dstp = u32x2{o[0], o[1]};
        if (c0 + U / 2 < C) *(u32x2*)(dstp + 8) = u32x2{o[2], o[3]};
      }
    }
  }
}

template <typename T, bool G2, int VEC>
static void launch_bwd_tiled(LaunchProf& lp, bool last, const ocf_tensor4* other, const ocf_tensor4* g, const ocf_tensor4* dst, int B, int C,
                             int H, int W, int d, hipStream_t st, const ocf_tensor4* occ = nullptr, const ocf_tensor4* w2own = nullptr,
                             const ocf_tensor4* gocc = nullptr) {
  const int tx = (W + BTW - 1) / BTW, ty = (H + BTH - 1) / BTH;
  const int ncb = (C + BNU * Unit<T>::U - 1) / (BNU * Unit<T>::U);
  const int split = (B * tx * ty <= 128) ? ncb : 1;  // under half a tile per CU: one workgroup per (tile, channel block); measured
                                                     // 81 -> 28 us and 60 -> 34 us (g1+g2) at the two top levels, slower from 224 tiles up
  const dim3 grid((unsigned)(B * tx * ty), (unsigned)split), block(256);
#ifdef OCF_TUNING_BUILD
  static const int ablate = getenv("OCF_ABLATE_BWD") ? atoi(getenv("OCF_ABLATE_BWD")) : 0;
#else
  constexpr int ablate = 0;
#endif
  switch (d) {
    case 1: lp.launch(last, corr_bwd_tiled<T, G2, 1, VEC>, grid, block, 0, st, mk(other), mk(g), mk(dst), mk(occ), mk(w2own), mk(gocc), B, C, H, W, tx, ty, ablate); break;
    case 2: lp.launch(last, corr_bwd_tiled<T, G2, 2, VEC>, grid, block, 0, st, mk(other), mk(g), mk(dst), mk(occ), mk(w2own), mk(gocc), B, C, H, W, tx, ty, ablate); break;
    case 3: lp.launch(last, corr_bwd_tiled<T, G2, 3, VEC>, grid, block, 0, st, mk(other), mk(g), mk(dst), mk(occ), mk(w2own), mk(gocc), B, C, H, W, tx, ty, ablate); break;
    default: lp.launch(last, corr_bwd_tiled<T, G2, 4, VEC>, grid, block, 0, st, mk(other), mk(g), mk(dst), mk(occ), mk(w2own), mk(gocc), B, C, H, W, tx, ty, ablate); break;
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <typename T> static bool vec_ok(const ocf_tensor4* t) { return vec_width<T>(t) == 16; }

template <typename T, int VEC, int MODE>
static int launch_fwd_d(const T4& f1, const T4& f2, const T4& flow, const T4& occ, const T4& out, int B, int C, int H, int W,
                        int d, float fs, int ac, int out_mode, hipStream_t st) {
  const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;
  const WarpGeom wg(H, W, ac);
  const dim3 grid((unsigned)(B * tx * ty));
#define OCF_LAUNCH_D(DD)                                                                                   \
  case DD:                                                                                                 \
    hipLaunchKernelGGL((corr_fwd_tiled<T, VEC, MODE, DD>), grid, dim3(64 * (2 * DD + 1)), 0, st, f1, f2, flow, occ, out, \
                       B, C, H, W, fs, wg, tx, ty, out_mode, corr_fwd_lrelu());                            \
    break;
  switch (d) {
    OCF_LAUNCH_D(1) OCF_LAUNCH_D(2) OCF_LAUNCH_D(3) OCF_LAUNCH_D(4)
    default: return OCF_EUNSUPPORTED;
  }
#undef OCF_LAUNCH_D
  return launch_status();
}

template <typename T, bool ELEMENT_LOADS_ONLY = false>
static int corr_fwd_t(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ,
                      const ocf_tensor4* out, int B, int C, int H, int W, int d, float fs, int ac, hipStream_t st) {
  // in-image offsets are 32-bit in the kernels
  auto span = [&](const ocf_tensor4* t, long c) { return (labs(t->sh) * H + labs(t->sw) * W + labs(t->sc) * c) * (long)sizeof(T); };
  if (span(f1, C) >= (1L << 31) || span(f2, C) >= (1L << 31)) return OCF_EUNSUPPORTED;
  const int vec = min_vec(vec_width<T>(f1), vec_width<T>(f2));
  const int nk = (2 * d + 1) * (2 * d + 1);
  const long es = sizeof(T);
  int out_mode;
  if (out->sc == 1 && out->sw != 1) {
    // channels_last-like output: LDS-transposed epilogue; vector copy when the tensor is dense and rows are 16-B aligned
    const bool dense = out->sw == nk && ((uintptr_t)out->data % 16 == 0) && (out->sh * es) % 16 == 0 && (out->sb * es) % 16 == 0 &&
                       ((long)TW * nk * es) % 16 == 0;
    out_mode = dense ? OUT_LDS_VEC : OUT_LDS;
  } else {
    const bool v = out->sw == 1 && ((uintptr_t)out->data % (PX * sizeof(T)) == 0) && out->sc % PX == 0 && out->sh % PX == 0 &&
                   out->sb % PX == 0;
    out_mode = v ? OUT_DIRECT_VEC : OUT_DIRECT;
  }
#ifdef OCF_TUNING_BUILD
  if (const char* ab = getenv("OCF_ABLATE")) out_mode |= atoi(ab) << 8;
#endif
  const T4 a = mk(f1), bq = mk(f2), fl = mk(flow), oc = mk(occ), o = mk(out);
#define OCF_DISPATCH(V)                                                                                              \
  (flow ? launch_fwd_d<T, V, 2>(a, bq, fl, oc, o, B, C, H, W, d, fs, ac, out_mode, st)                               \
        : (occ ? launch_fwd_d<T, V, 1>(a, bq, fl, oc, o, B, C, H, W, d, fs, ac, out_mode, st)                        \
               : launch_fwd_d<T, V, 0>(a, bq, fl, oc, o, B, C, H, W, d, fs, ac, out_mode, st)))
  if constexpr (!ELEMENT_LOADS_ONLY) {
    if (vec == 16) return OCF_DISPATCH(16);
    if (vec == 8) return OCF_DISPATCH(8);
  }
  return OCF_DISPATCH(0);
#undef OCF_DISPATCH
}

int corr_fwd_bf16_vector(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ,
                         const ocf_tensor4* out, int B, int C, int H, int W, int d, float fs, int ac, hipStream_t st) {
  ScopedProfile prof(st);
  return corr_fwd_t<bf16, true>(f1, f2, flow, occ, out, B, C, H, W, d, fs, ac, st);
}

// which kernel a backward request gets (enum ocf_corr_bwd_variant); shared by the launcher and by ocf_corr_bwd_variant
template <typename T>
static int corr_bwd_variant_t(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* g, const ocf_tensor4* g1, const ocf_tensor4* g2,
                              int B, int C, int H, int W, int d, int policy) {
  constexpr int U = Unit<T>::U;
  if (sizeof(T) == 2 && corr_bwd_mfma_eligible(f1, w2, g, g1, g2, B, C, H, W, d, policy)) return OCF_CORR_BWD_MFMA;
  bool vec = vec_ok<T>(f1) && vec_ok<T>(w2);
  if (g1) vec = vec && vec_ok<T>(g1);
  if (g2) vec = vec && vec_ok<T>(g2);
  if (vec && C % U == 0 && g->sc == 1) return OCF_CORR_BWD_TILED16;
  int v8 = min_vec(vec_width<T>(f1), vec_width<T>(w2));
  if (g1) v8 = min_vec(v8, vec_width<T>(g1));
  if (g2) v8 = min_vec(v8, vec_width<T>(g2));
  if (v8 >= 8 && (C * (long)sizeof(T)) % 8 == 0 && g->sc == 1) return OCF_CORR_BWD_TILED8;
  return OCF_CORR_BWD_GATHER;
}

template <typename T>
static int corr_bwd_t(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* g, const ocf_tensor4* g1,
                      const ocf_tensor4* g2, int B, int C, int H, int W, int d, int policy, hipStream_t st, const ocf_tensor4* occ = nullptr,
                      const ocf_tensor4* gocc = nullptr) {
  constexpr int U = Unit<T>::U;
  if (sizeof(T) == 2 && corr_bwd_mfma_eligible(f1, w2, g, g1, g2, B, C, H, W, d, policy))
    return corr_bwd_bf16_mfma(f1, w2, g, g1, g2, B, C, H, W, st, occ, gocc);  // (zeroes d/docc itself where its kernel accumulates into it)
  if (gocc && hipMemsetAsync(gocc->data, 0, sizeof(float) * (size_t)B * H * W, st) != hipSuccess) return OCF_ELAUNCH;
  bool vec = vec_ok<T>(f1) && vec_ok<T>(w2);
  if (g1) vec = vec && vec_ok<T>(g1);
  if (g2) vec = vec && vec_ok<T>(g2);
  // tiled path: 16-byte units everywhere, whole units only, grad_out with unit channel stride
  LaunchProf lp;
  if (vec && C % U == 0 && g->sc == 1) {
    if (g1) launch_bwd_tiled<T, false, 16>(lp, !g2, w2, g, g1, B, C, H, W, d, st, occ);
    if (g2) launch_bwd_tiled<T, true, 16>(lp, true, f1, g, g2, B, C, H, W, d, st, occ, w2, gocc);
    return launch_status();
  }
  // same kernel at 8-byte granularity: pixels that are a multiple of 8 bytes (C = 196 bf16 at the top pyramid level)
  int v8 = min_vec(vec_width<T>(f1), vec_width<T>(w2));
  if (g1) v8 = min_vec(v8, vec_width<T>(g1));
  if (g2) v8 = min_vec(v8, vec_width<T>(g2));
  if (v8 >= 8 && (C * (long)sizeof(T)) % 8 == 0 && g->sc == 1) {
    if (g1) launch_bwd_tiled<T, false, 8>(lp, !g2, w2, g, g1, B, C, H, W, d, st, occ);
    if (g2) launch_bwd_tiled<T, true, 8>(lp, true, f1, g, g2, B, C, H, W, d, st, occ, w2, gocc);
    return launch_status();
  }
  if (occ) return OCF_EUNSUPPORTED;  // the gather fallback has no occlusion path: the caller composes it from plain launches
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
                            float leaky_slope, int dtype, int policy, void* stream) {
  ocf::clear_stale_error();
  if (!(leaky_slope >= 0.f) || leaky_slope > 1.f) return OCF_EINVAL;
  // 0 = no activation: the kernels take the slope itself, 1 = identity
  ocf::ScopedCorrLrelu epi(leaky_slope == 0.f ? 1.f : leaky_slope);
  if (!f1 || !f2 || !out || !f1->data || !f2->data || !out->data) return OCF_EINVAL;
  if ((flow && !flow->data) || (occ && !occ->data)) return OCF_EINVAL;
  if (policy < OCF_CORR_POLICY_AUTO || policy > OCF_CORR_POLICY_WAVE) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || d < 0) return OCF_EINVAL;
  if (d < 1 || d > OCF_MAX_DISPLACEMENT) return OCF_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  // bf16: ONE kernel; a profiling request (ocf_profile_next_launch) is handed to the launch itself (hipExtLaunchKernelGGL takes
  // the events and stamps them with the dispatch's own begin / end times, like the profiler's kernel trace) instead of
  // bracketing it with event records, which add ~3 us of queue processing to a 10-30 us kernel
  if (dtype == OCF_BF16) return ocf::corr_fwd_bf16_mfma(f1, f2, flow, occ, out, B, C, H, W, d, flow_scale, align_corners, policy, st);
  ocf::ScopedProfile prof(st);
  if (dtype == OCF_F32) return ocf::corr_fwd_t<float>(f1, f2, flow, occ, out, B, C, H, W, d, flow_scale, align_corners, st);
  return OCF_EINVAL;
}

extern "C" int ocf_corr_fwd_variant(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ,
                                    const ocf_tensor4* out, int B, int C, int H, int W, int d, int dtype, int policy) {
  if (!f1 || !f2 || !out || !f1->data || !f2->data || !out->data) return OCF_EINVAL;
  if (policy < OCF_CORR_POLICY_AUTO || policy > OCF_CORR_POLICY_WAVE) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || d < 0) return OCF_EINVAL;
  if (d < 1 || d > OCF_MAX_DISPLACEMENT) return OCF_EUNSUPPORTED;
  if (dtype == OCF_F32) return OCF_CORR_VECTOR_F32;
  if (dtype != OCF_BF16) return OCF_EINVAL;
  return ocf::corr_fwd_bf16_variant(f1, f2, flow, occ, out, B, C, H, W, d, policy);
}

extern "C" int ocf_corr_bwd(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* gout, const ocf_tensor4* g1,
                            const ocf_tensor4* g2, int B, int C, int H, int W, int d, int dtype, int policy, void* stream) {
  ocf::clear_stale_error();
  if (!f1 || !w2 || !gout || !f1->data || !w2->data || !gout->data) return OCF_EINVAL;
  if ((g1 && !g1->data) || (g2 && !g2->data) || (!g1 && !g2)) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || d < 0) return OCF_EINVAL;
  if (policy < OCF_CORR_BWD_POLICY_AUTO || policy > OCF_CORR_BWD_POLICY_MFMA) return OCF_EINVAL;
  if (d < 1 || d > OCF_MAX_DISPLACEMENT) return OCF_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OCF_F32) return ocf::corr_bwd_t<float>(f1, w2, gout, g1, g2, B, C, H, W, d, policy, st);
  if (dtype == OCF_BF16) return ocf::corr_bwd_t<ocf::bf16>(f1, w2, gout, g1, g2, B, C, H, W, d, policy, st);
  return OCF_EINVAL;
}

extern "C" int ocf_corr_bwd_variant(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* gout, const ocf_tensor4* g1,
                                    const ocf_tensor4* g2, int B, int C, int H, int W, int d, int dtype, int policy) {
  if (!f1 || !w2 || !gout || !f1->data || !w2->data || !gout->data || (!g1 && !g2)) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || d < 0) return OCF_EINVAL;
  if (policy < OCF_CORR_BWD_POLICY_AUTO || policy > OCF_CORR_BWD_POLICY_MFMA) return OCF_EINVAL;
  if (d < 1 || d > OCF_MAX_DISPLACEMENT) return OCF_EUNSUPPORTED;
  if (dtype == OCF_F32) return ocf::corr_bwd_variant_t<float>(f1, w2, gout, g1, g2, B, C, H, W, d, policy);
  if (dtype == OCF_BF16) return ocf::corr_bwd_variant_t<ocf::bf16>(f1, w2, gout, g1, g2, B, C, H, W, d, policy);
  return OCF_EINVAL;
}

/* occlusion-aware backward: see include/ocflow_hip.h */
extern "C" int ocf_corr_occ_bwd(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* occ, const ocf_tensor4* gout,
                                const ocf_tensor4* g1, const ocf_tensor4* g2, const ocf_tensor4* gocc, int B, int C, int H, int W, int d,
                                int dtype, int policy, void* stream) {
  ocf::clear_stale_error();
  if (!f1 || !w2 || !occ || !gout || !f1->data || !w2->data || !occ->data || !gout->data) return OCF_EINVAL;
  if (policy < OCF_CORR_BWD_POLICY_AUTO || policy > OCF_CORR_BWD_POLICY_MFMA) return OCF_EINVAL;
  if ((g1 && !g1->data) || (g2 && !g2->data) || (gocc && !gocc->data) || (!g1 && !g2)) return OCF_EINVAL;
  if (gocc && !g2) return OCF_EINVAL;  // the occlusion gradient is a by-product of the g2 pass
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || d < 0) return OCF_EINVAL;
  if (d < 1 || d > OCF_MAX_DISPLACEMENT) return OCF_EUNSUPPORTED;
  if (gocc && !(gocc->sw == 1 && gocc->sh == W && gocc->sb == (long)H * W)) return OCF_EUNSUPPORTED;  // dense fp32 [B,1,H,W]
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OCF_F32) return ocf::corr_bwd_t<float>(f1, w2, gout, g1, g2, B, C, H, W, d, policy, st, occ, gocc);
  if (dtype == OCF_BF16) return ocf::corr_bwd_t<ocf::bf16>(f1, w2, gout, g1, g2, B, C, H, W, d, policy, st, occ, gocc);
  return OCF_EINVAL;
}
