// K1+K2 forward for bf16 features on the CDNA4 matrix cores.
//
// Why MFMA: with v_dot2c_f32_bf16 the vector-ALU formulation of this kernel measured 14 us of pure dot-product issue at
// the level-2 shape (B=8, C=32, 109x256: 1.16 GFLOP) -- the measured v_dot2c rate (~8 cycles per wave64 instruction) caps it
// at ~1.6 TB/s of algorithmic traffic, BELOW the HBM roofline.  Arithmetic intensity is 17-34 FLOP/B, right at the fp32
// vector ridge (hard parts), so the contraction is moved to the matrix pipe to make the kernel
// memory-bound again.
//
// Band-GEMM formulation: for one image row y, one displacement row dy and a strip of 16 pixels x0..x0+15,
//     P[m][n] = sum_c f1[y, x0+m, c] * f2[y+dy, xh+n, c]          (16 x 16, K = channels)
// is one v_mfma_f32_16x16x32_bf16 per 32 channels; the 2d+1 needed outputs of pixel m are the band n-m in [0, 2d] of two
// adjacent 16-column tiles (28 % of the MACs are useful at d=4 -- still ~9x the useful v_dot2c rate).
//
// Work decomposition (gfx950, wave64):
//   workgroup  = 32 x 4 pixel tile, 2d+1 waves; wave w owns displacement row dy = w - d (4 rows x 4 MFMA per 32 channels).
//   LDS image  = pixel-major [row][x][4 units of 16 B], unit order XOR-permuted by ((x>>2)&3) so that BOTH the staging
//                ds_write_b128 (8-lane groups = 2 whole pixels = 128 contiguous bytes) and the MFMA fragment
//                ds_read_b128 (lane = (pixel l&15, k-group l>>4)) are bank-conflict free.
//   staging    = one wave per tile row (wave-uniform row math), lanes sweep 16-byte units; channels_last reads are whole
//                64-byte pixels, 1 KiB per wave instruction; MODE 1 scales by occ, MODE 2 gathers through the warp.
//   multi-block (C > 32): accumulate in registers across 32-channel blocks; the next block's global loads are issued
//                before the MFMAs of the current one (register prefetch).
//   epilogue   = band elements (x 1/C) -> LDS [row][x][85-element pitch] (pitch chosen so the 4 k-groups of a wave hit
//                disjoint banks), then whole contiguous pixel runs -> HBM with 16-byte stores (channels_last, dense).
// reference semantics: models/networks/correlation_layer.py:7-40 (+ warp cost_volume_flow_net.py:121-151)
#include <stdlib.h>
#include <hip/hip_ext.h>
#include "ocf_common.h"
#include "corr_internal.h"

namespace ocf {

namespace mf {
constexpr int TW = 32, NUM = 4;             // tile width, units (of 8 bf16) per 32-channel block; tile height TH is a template parameter
constexpr int ROWPX = 40;                   // halo row pitch in pixels (32 + 2*4); fragment reads of the 3rd tile may run
constexpr int SLACK_PX = 8;                 // 8 px past the last row: harmless, those MFMA columns are never used

template <int D, int TH> struct G {
  static constexpr int N = 2 * D + 1, NK = N * N;
  static constexpr int HH = TH + 2 * D;
  static constexpr int F2_SLOTS = (HH * ROWPX + SLACK_PX) * NUM;
  static constexpr int F1_SLOTS = TH * TW * NUM;
  // output staging image [pixel][dy][NG groups of 4 columns]: pixel m keeps the 4-column groups floor(m/4)..+NG-1 of its
  // D row, its band element dx sits at (m&3)+dx inside them -- every lane writes ONE aligned 8-byte vector per MFMA pair
  static constexpr int NG = (N + 6) / 4;
  static constexpr int PDY = 4 * NG, S = N * PDY;  // element pitches: per dy, per pixel (108 at d=4)
  static constexpr int OUT_ELEMS = TH * TW * S;
  static constexpr int NT = 64 * N;
  static constexpr int F2_ROWS_PER_WAVE = (HH + N - 1) / N;
  static constexpr int F2_IT = (ROWPX * NUM + 63) / 64;  // 160 items per row -> 3 lane sweeps
  static constexpr int F1_IT = (TW * NUM) / 64;           // 128 items per row -> 2 lane sweeps
  static constexpr int F1_ROWS_PER_WAVE = (TH + N - 1) / N;  // 1 at d >= 2 (N >= TH waves), 2 at d = 1
};

__device__ __forceinline__ int slot(int rowpx, int row, int x, int u) { return (row * rowpx + x) * NUM + (((x >> 2) & 3) ^ ((3 * u) & 3)); }

template <int VEC, int ROWLEN>
__device__ __forceinline__ void row_item(int it, int& u, int& x) {
  if (VEC != 0) { u = it & (NUM - 1); x = it >> 2; }
  else          { u = it / ROWLEN; x = it - u * ROWLEN; }
}
}  // namespace mf

enum { MOUT_GENERIC = 0, MOUT_CL_VEC = 1 };

// MODE 0: plain; 1: occ given (applied to the band outputs: corr(f1, f2*occ)[dy,dx](x) = occ(x+d) * corr(f1,f2)[dy,dx](x));
// MODE 2: f2 gathered through the warp while staged (occ optional, applied the same way).
// MULTI: C > 32, accumulate over 32-channel blocks in registers.
// The kernel is PERSISTENT: gridDim.x <= 2 workgroups per CU, each walks tiles of one XCD's contiguous tile range; the
// global loads of the next tile (or channel block) are issued before the MFMAs of the current one, and the 16-byte output
// stores of a tile drain while the next one is staged.
template <int VEC, int MODE, int D, bool MULTI, int TH>
__global__ __launch_bounds__(64 * (2 * D + 1), (D == 1 || (MODE == 2 && D == 3)) ? 2 : ((MULTI || TH > 4) ? 3 : 5)) void corr_fwd_mfma(T4 f1, T4 f2, T4 flow, T4 occ, T4 out, int B, int C, int H,
                                                                 int W, float fs, WarpGeom wg, int tiles_x, int tiles_y,
                                                                 int out_mode, float lrelu) {
  using namespace mf;
  using GG = G<D, TH>;
  typedef bf16 T;
  constexpr int U = 8, ES = 2;
  typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
  constexpr int OCC_W = 48;  // halo occlusion tile pitch (floats)
  extern __shared__ u32x4 smem[];  // lds_bytes<D, TH>(MODE) dynamic bytes
  u32x4* s_f2 = smem;
  u32x4* s_f1 = smem + GG::F2_SLOTS;
  unsigned short* s_out = (unsigned short*)(smem + GG::F2_SLOTS + GG::F1_SLOTS);
  float* s_occ = (float*)(smem + GG::F2_SLOTS + GG::F1_SLOTS + (GG::OUT_ELEMS * ES + 15) / 16);

#ifdef OCF_TUNING_BUILD  // ablation switches exist only in the library tools/kbench builds for itself (-DOCF_TUNING_BUILD)
  const int ablate = out_mode >> 8;  // bit0 no MFMA, bit1 no global loads, bit2 no output
  out_mode &= 0xff;
#else
  constexpr int ablate = 0;
#endif
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int ln = lane & 15, lg = lane >> 4;
  const int tpi = tiles_x * tiles_y;
  const int ntiles = B * tpi;
  // XCD-affine persistent schedule: blocks b and b+8 share an XCD; XCD x owns tiles [x*per_xcd, (x+1)*per_xcd)
  const int per_xcd = (ntiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3, slots = gridDim.x >> 3;

  // register prefetch of the next TILE in the single-block variant (1 workgroup per CU, 137-160 VGPRs, no spills); the multi-block
  // variant would spill with 32 more live registers beside its 64 accumulators (measured: 25 -> 15 us at level 3 without it)
  constexpr bool PF = !MULTI;
  const int ncb = MULTI ? (C + 31) / 32 : 1;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const int f1sh = (int)f1.sh, f1sw = (int)f1.sw, f2sh = (int)f2.sh, f2sw = (int)f2.sw;
  const float invC = 1.f / (float)C;
  const bool has_occ = MODE >= 1 && occ.p != nullptr;

  u32x4 pf2[GG::F2_ROWS_PER_WAVE][GG::F2_IT], pf1[GG::F1_ROWS_PER_WAVE][GG::F1_IT];

  // ---- fetch: global -> registers for (tile at b, ty0, tx0; channel block cbase) -----------------------------------
  auto fetch = [&](int b, int ty0, int tx0, int cbase) __attribute__((always_inline)) {
    T4 f1i = f1, f2i = f2;
    f1i.p = f1.p + (long)b * f1.sb * ES;
    f2i.p = f2.p + (long)b * f2.sb * ES;
#pragma unroll
    for (int rr = 0; rr < GG::F2_ROWS_PER_WAVE; rr++) {
      const int r = wave + rr * GG::N;
      const int gy = ty0 - D + r;
      const bool rowok = r < GG::HH && gy >= 0 && gy < H;
#pragma unroll
      for (int k = 0; k < GG::F2_IT; k++) {
        const int it = lane + 64 * k;
        int u, x;
        row_item<VEC, ROWPX>(it, u, x);
        const int gx = tx0 - D + x, c0 = cbase + u * U;
        u32x4 val = zero4;
        if (rowok && it < ROWPX * NUM && x < TW + 2 * D && gx >= 0 && gx < W && c0 < C && !(ablate & 2)) {
          if (MODE != 2) {
            val = load_unit_full<T, VEC>(f2i, gy * f2sh + gx * f2sw, c0, C);
          } else {
            const float* flb = (const float*)flow.p + (long)b * flow.sb;
            const int fo = gy * (int)flow.sh + gx * (int)flow.sw;
            const float uu = flb[fo] * fs, vv = flb[fo + flow.sc] * fs;
            const Bilin sm = bilin_setup(((float)gx + uu) * wg.ax + wg.bx, ((float)gy + vv) * wg.ay + wg.by, H, W);
            float a[U];
#pragma unroll
            for (int j = 0; j < U; j++) a[j] = 0.f;
#pragma unroll
            for (int n = 0; n < 4; n++) {
              const int ddx = n & 1, ddy = n >> 1;
              const bool ok = (ddx ? sm.vx1 : sm.vx0) & (ddy ? sm.vy1 : sm.vy0);
              if (ok) {
                const float w = (ddx ? sm.wx1 : 1.f - sm.wx1) * (ddy ? sm.wy1 : 1.f - sm.wy1);
                float f[U];
                load_unit_full_f<T, VEC>(f2i, (sm.y0 + ddy) * f2sh + (sm.x0 + ddx) * f2sw, c0, C, f);
#pragma unroll
                for (int j = 0; j < U; j++) a[j] = __builtin_fmaf(w, f[j], a[j]);
              }
            }
            val = f_to_unit<T>(a);
          }
        }
        pf2[rr][k] = val;
      }
    }
#pragma unroll
    for (int rr = 0; rr < GG::F1_ROWS_PER_WAVE; rr++) {
      const int r = wave + rr * GG::N, gy = ty0 + r;
#pragma unroll
      for (int k = 0; k < GG::F1_IT; k++) {
        const int it = lane + 64 * k;
        int u, x;
        row_item<VEC, TW>(it, u, x);
        const int gx = tx0 + x, c0 = cbase + u * U;
        u32x4 val = zero4;
        if (r < TH && gy < H && gx < W && c0 < C && !(ablate & 2)) val = load_unit_full<T, VEC>(f1i, gy * f1sh + gx * f1sw, c0, C);
        pf1[rr][k] = val;
      }
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int rr = 0; rr < GG::F2_ROWS_PER_WAVE; rr++) {
      const int r = wave + rr * GG::N;
      if (r < GG::HH) {
#pragma unroll
        for (int k = 0; k < GG::F2_IT; k++) {
          const int it = lane + 64 * k;
          int u, x;
          row_item<VEC, ROWPX>(it, u, x);
          if (it < ROWPX * NUM) s_f2[slot(ROWPX, r, x, u)] = pf2[rr][k];
        }
      }
    }
#pragma unroll
    for (int rr = 0; rr < GG::F1_ROWS_PER_WAVE; rr++) {
      const int r = wave + rr * GG::N;
      if (r < TH) {
#pragma unroll
        for (int k = 0; k < GG::F1_IT; k++) {
          const int it = lane + 64 * k;
          int u, x;
          row_item<VEC, TW>(it, u, x);
          s_f1[slot(TW, r, x, u)] = pf1[rr][k];
        }
      }
    }
  };
  // occlusion halo tile straight to LDS (tiny: HH x 48 floats), 0 outside the image
  auto stage_occ = [&](int b, int ty0, int tx0) __attribute__((always_inline)) {
    if (MODE >= 1) {
      const float* ocb = has_occ ? (const float*)occ.p + (long)b * occ.sb : nullptr;
      for (int e = tid; e < GG::HH * OCC_W; e += GG::NT) {
        const int r = e / OCC_W, x = e - r * OCC_W;
        const int gy = ty0 - D + r, gx = tx0 - D + x;
        float o = has_occ ? 0.f : 1.f;
        if (has_occ && gy >= 0 && gy < H && gx >= 0 && gx < W) o = ocb[gy * (int)occ.sh + gx * (int)occ.sw];
        s_occ[e] = o;
      }
    }
  };

  // ---- band write.  The MFMAs compute the TRANSPOSED product D[n][m] = sum_c f2[n][c] * f1[m][c] (f2 fragment as A, f1 as B),
  // so lane (m = l&15, g = l>>4) holds, for ITS pixel m, the four consecutive columns n = 4g..4g+3 in its four registers:
  // four consecutive dx of one (pixel, dy).  Column group g of tile s (resp. s+1) is group g - m/4 (resp. g + 4 - m/4) of the
  // pixel's NG kept groups; the two tiles' valid lanes are disjoint, so ONE masked, 8-byte aligned ds_write_b64 per MFMA pair
  // stores the whole band (the 16 lanes of a k-group hit 16 distinct bank pairs: pixel pitch 54 dwords).
  const int gq = lg - (ln >> 2);                      // group index for tile s; tile s+1: gq + 4
  const bool bvalA = gq >= 0 && gq < GG::NG, bvalB = gq + 4 >= 0 && gq + 4 < GG::NG;
  const int boff = (ln * GG::S + wave * GG::PDY + (bvalA ? gq : gq + 4) * 4) * ES;
  auto band_write = [&](const f32x4& dA, const f32x4& dB, int ry, int s) __attribute__((always_inline)) {
    if (bvalA || bvalB) {
      f32x4 v = bvalA ? dA : dB;
      if (MODE >= 1) {
        // occ of the f2 pixel = D row n = 4g + r of column tile s (A) or s+1 (B): four consecutive floats
        const f32x4 o = *(const f32x4*)(s_occ + (ry + wave) * OCC_W + 16 * (bvalA ? s : s + 1) + 4 * lg);
        v = v * o;
      }
      v = v * invC;
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = v[r] > 0.f ? v[r] : v[r] * lrelu;
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      const bf16x4 h = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
      *(bf16x4*)((char*)s_out + (ry * TW + 16 * s) * GG::S * ES + boff) = h;
    }
  };

  // LDS element offset of output (pixel x of tile row r, k = dy*N + dx)
  auto oidx = [&](int r, int x, int k) __attribute__((always_inline)) {
    const int dy = k / GG::N, dx = k - dy * GG::N;
    return (r * TW + x) * GG::S + dy * GG::PDY + (x & 3) + dx;
  };
  f32x4 acc[MULTI ? TH : 1][4];

  // first tile of this workgroup
  int ti = slot_in_xcd;
  int tile = xcd * per_xcd + ti;
  bool have = ti < per_xcd && tile < ntiles;
  int b = 0, ty0 = 0, tx0 = 0;
  if (have) {
    b = tile / tpi;
    const int trem = tile - b * tpi;
    ty0 = (trem / tiles_x) * TH; tx0 = (trem % tiles_x) * TW;
    if (PF) fetch(b, ty0, tx0, 0);
  }
  while (have) {
    // next tile of this workgroup (for the prefetch at the end of the channel loop)
    const int nti = ti + slots;
    const int ntile = xcd * per_xcd + nti;
    const bool nhave = nti < per_xcd && ntile < ntiles;
    int nb = 0, nty0 = 0, ntx0 = 0;
    if (nhave) {
      nb = ntile / tpi;
      const int trem = ntile - nb * tpi;
      nty0 = (trem / tiles_x) * TH; ntx0 = (trem % tiles_x) * TW;
    }
    if (MULTI) {
#pragma unroll
      for (int i = 0; i < TH; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int cb = 0; cb < ncb; cb++) {
      if (!PF) fetch(b, ty0, tx0, cb * 32);
      commit();
      if (cb == 0) stage_occ(b, ty0, tx0);
      __syncthreads();
      // loads of the next channel block / next tile fly under the MFMAs, band writes and output stores
      if (PF) {
        if (cb + 1 < ncb) fetch(b, ty0, tx0, (cb + 1) * 32);
        else if (nhave) fetch(nb, nty0, ntx0, 0);
      }
      if (!(ablate & 1)) {
#pragma unroll
        for (int ry = 0; ry < TH; ry++) {
          const bf16x8v a0 = __builtin_bit_cast(bf16x8v, s_f1[slot(TW, ry, ln, lg)]);
          const bf16x8v a1 = __builtin_bit_cast(bf16x8v, s_f1[slot(TW, ry, 16 + ln, lg)]);
          const int hr = ry + wave;
          const bf16x8v b0 = __builtin_bit_cast(bf16x8v, s_f2[slot(ROWPX, hr, ln, lg)]);
          const bf16x8v b1 = __builtin_bit_cast(bf16x8v, s_f2[slot(ROWPX, hr, 16 + ln, lg)]);
          const bf16x8v b2 = __builtin_bit_cast(bf16x8v, s_f2[slot(ROWPX, hr, 32 + ln, lg)]);
          if (MULTI) {
            acc[ry][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, a0, acc[ry][0], 0, 0, 0);
            acc[ry][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, a0, acc[ry][1], 0, 0, 0);
            acc[ry][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, a1, acc[ry][2], 0, 0, 0);
            acc[ry][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, a1, acc[ry][3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);  // keep one row's fragments live at a time (register pressure)
          } else {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, a0, z, 0, 0, 0);
            const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, a0, z, 0, 0, 0);
            const f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, a1, z, 0, 0, 0);
            const f32x4 d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, a1, z, 0, 0, 0);
            if (!(ablate & 4)) {
              band_write(d0, d1, ry, 0);
              band_write(d2, d3, ry, 1);
            }
          }
        }
      }
      if (MULTI && cb + 1 < ncb) __syncthreads();  // staging tiles are rewritten by the next commit
    }
    if (MULTI && !(ablate & 4)) {
#pragma unroll
      for (int ry = 0; ry < TH; ry++) {
        band_write(acc[ry][0], acc[ry][1], ry, 0);
        band_write(acc[ry][2], acc[ry][3], ry, 1);
      }
    }
    __syncthreads();

    // ---- copy-out: LDS [row][x][S] -> out[b, k, y, x]; the stores drain while the next tile is staged -------------
    if (!(ablate & 4) && !(ablate & 8)) {
      const int vrows = min(TH, H - ty0), vcols = min(TW, W - tx0);
      if (out_mode == MOUT_CL_VEC && GG::N >= 7) {  // (a chunk of 8 crosses at most one dy boundary only when N >= 7)
        // dense channels_last: a tile row is one contiguous run of vcols*NK bf16, 16-byte aligned at the start
        const int row_elems = vcols * GG::NK;
        const int nvec = row_elems >> 3;
        for (int r = 0; r < vrows; r++) {
          char* gp = out.p + ((long)b * out.sb + (long)(ty0 + r) * out.sh + (long)tx0 * out.sw) * ES;
          for (int v = tid; v < nvec; v += GG::NT) {
            // 8 consecutive outputs starting at (pixel px, k).  In the grouped LDS image consecutive k are consecutive elements
            // except for ONE possible jump inside the chunk, at the next dy boundary (element index w): +3 elements (group
            // padding), or, when that boundary is also the pixel boundary, +4 + ((px+1)&3) - (px&3).
            const int e0 = v * 8;
            const int px = e0 / GG::NK, k = e0 - px * GG::NK;
            const int dy = k / GG::N, dx = k - dy * GG::N;
            const int w = GG::N - dx;
            const int jump = (dy == GG::N - 1) ? (GG::S - GG::N * GG::PDY) + (GG::PDY - GG::N) + 1 + (((px + 1) & 3) - (px & 3)) - 1
                                               : GG::PDY - GG::N;
            const char* lpA = (const char*)s_out + (((r * TW + px) * GG::S) + dy * GG::PDY + (px & 3) + dx) * ES;
            const char* lpB = lpA + jump * ES;
            unsigned short h[8];
#pragma unroll
            for (int j = 0; j < 8; j++) h[j] = *(const unsigned short*)((j >= w ? lpB : lpA) + j * ES);
            u32x4 pk = {(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16),
                        (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16)};
            ((u32x4*)gp)[v] = pk;
          }
          for (int e = nvec * 8 + tid; e < row_elems; e += GG::NT) {
            const int px = e / GG::NK, k = e - px * GG::NK;
            ((unsigned short*)gp)[e] = s_out[oidx(r, px, k)];
          }
        }
      } else if (out.sw == 1) {
        // NCHW-like: lanes sweep x inside a (k, row) line
        const int per_k = vrows * vcols;
        for (int e = tid; e < GG::NK * per_k; e += GG::NT) {
          const int k = e / per_k, q = e - k * per_k;
          const int r = q / vcols, x = q - r * vcols;
          ((unsigned short*)out.p)[(long)b * out.sb + (long)k * out.sc + (long)(ty0 + r) * out.sh + (long)(tx0 + x)] =
              s_out[oidx(r, x, k)];
        }
      } else {
        const int per_row = vcols * GG::NK;
        for (int e = tid; e < vrows * per_row; e += GG::NT) {
          const int r = e / per_row, q = e - r * per_row;
          const int x = q / GG::NK, k = q - x * GG::NK;
          ((unsigned short*)out.p)[(long)b * out.sb + (long)k * out.sc + (long)(ty0 + r) * out.sh + (long)(tx0 + x) * out.sw] =
              s_out[oidx(r, x, k)];
        }
      }
    }
    // The next commit rewrites s_f1/s_f2 (every MFMA fragment read finished before the barrier above) and the next band
    // writes to s_out happen after the next tile's staging barrier, i.e. after every thread's copy-out reads: no extra barrier.
    ti = nti; tile = ntile; have = nhave; b = nb; ty0 = nty0; tx0 = ntx0;
  }
}

template <int D, int TH> static size_t lds_bytes(int mode) {
  using GG = mf::G<D, TH>;
  return (size_t)(GG::F2_SLOTS + GG::F1_SLOTS) * 16 + ((GG::OUT_ELEMS * 2 + 15) / 16) * 16 + (mode >= 1 ? GG::HH * 48 * 4 : 0);
}

static int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
            ? p.multiProcessorCount : 256;
    n = (n / 8) * 8;
    if (n < 8) n = 8;
  }
  return n;
}

// tile height: 8 rows for the single-block (C <= 32) variant -- half the tiles, 2/3 of the halo rows per pixel, one workgroup
// per CU (116 KB LDS); 4 rows for the multi-block variant (64 accumulator registers per lane instead of 128), 2 rows when a
// level has fewer 4-row tiles than CUs: those launches are a sequential channel-block loop on a fraction of the chip, and twice
// the workgroups with half the MFMA / epilogue work each measured 14.0 -> 10.6, 10.8 -> 8.5, 10.0 -> 8.3 us at the three top
// levels (B = 8), while the 448-tile level got slower (15.9 -> 20.5 us: more halo rows per pixel) and keeps 4 rows.
// One-row tiles do the same again below a quarter of that (10.6 -> 8.6 and 8.5 -> 6.6 us at the two top levels; 8.2 -> 9.7 at
// the third, which stays at 2 rows).
constexpr int TH_SINGLE = 8, TH_MULTI = 4, TH_MULTI_SMALL = 2, TH_MULTI_TINY = 1;

template <int VEC, int MODE, int D, bool MULTI, int TH>
static int launch_mfma_th(const T4& f1, const T4& f2, const T4& flow, const T4& occ, const T4& out, int B, int C, int H, int W,
                          float fs, int ac, int out_mode, hipStream_t st) {
  const int tx = (W + mf::TW - 1) / mf::TW, ty = (H + TH - 1) / TH;
  const WarpGeom wg(H, W, ac);
  // persistent grid: a multiple of 8 (XCD-affine schedule), as many workgroups as are co-resident (queried once per variant)
  const int ntiles = B * tx * ty;
  const dim3 block(64 * (2 * D + 1));
  const size_t lds = lds_bytes<D, TH>(MODE);
  static int occ_blocks = 0;
  if (occ_blocks == 0) {
    int nb = 0;
    (void)hipFuncSetAttribute((const void*)corr_fwd_mfma<VEC, MODE, D, MULTI, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, corr_fwd_mfma<VEC, MODE, D, MULTI, TH>, (int)block.x, lds);
    occ_blocks = (e == hipSuccess && nb >= 1) ? (nb > 2 ? 2 : nb) : 1;
  }
  int g = ((ntiles + 7) / 8) * 8;
  const int cap = occ_blocks * num_cus();
  if (g > cap) g = cap;
  ProfileSlot& ps = profile_slot();
  if (ps.start && ps.stop) {  // one-shot: the events receive this dispatch's begin / end timestamps
    hipEvent_t e0 = ps.start, e1 = ps.stop;
    ps.start = nullptr; ps.stop = nullptr;
    hipExtLaunchKernelGGL((corr_fwd_mfma<VEC, MODE, D, MULTI, TH>), dim3((unsigned)g), block, lds, st, e0, e1, 0, f1, f2, flow, occ, out, B, C,
                          H, W, fs, wg, tx, ty, out_mode, corr_fwd_lrelu());
  } else {
    hipLaunchKernelGGL((corr_fwd_mfma<VEC, MODE, D, MULTI, TH>), dim3((unsigned)g), block, lds, st, f1, f2, flow, occ, out, B, C, H, W, fs,
                       wg, tx, ty, out_mode, corr_fwd_lrelu());
  }
  return launch_status();
}

// widest channel run both operands give the staging loads: 16 bytes, 8 bytes, or 0 = element loads only (not a band-kernel request)
static int band_vec(const ocf_tensor4* f1, const ocf_tensor4* f2, int C) {
  int vec = min_vec(vec_width<bf16>(f1), vec_width<bf16>(f2));
  // the vector staging paths have no per-channel tail: 16-byte units need C % 8 == 0, 8-byte halves C % 4 == 0
  if (vec == 16 && C % 8 != 0) vec = 8;
  if (vec == 8 && C % 4 != 0) vec = 0;
  return vec;
}

// which band-GEMM instantiation a shape gets (also reported by ocf_corr_fwd_variant)
// `warp`: with the bilinear gather of the second operand staged in-kernel the 4-row tile does not fit the register budget of a
// nine-wave workgroup (168 VGPRs: it spilled 41-61 dwords): such requests get 2-row tiles at every size, 1-row tiles at d = 4 (the
// 2-row tile still spilled 2-3 dwords there)
static int band_variant(int B, int C, int H, int W, bool warp = false, int d = 0) {
  if (C <= 32) return OCF_CORR_BAND_TH8;
  const long tiles4 = (long)B * ((W + mf::TW - 1) / mf::TW) * ((H + TH_MULTI - 1) / TH_MULTI);
  if (tiles4 * 4 < num_cus() || (warp && d == 4)) return OCF_CORR_BAND_MULTI_TH1;
  if (tiles4 < num_cus() || warp) return OCF_CORR_BAND_MULTI_TH2;
  return OCF_CORR_BAND_MULTI_TH4;
}

template <int VEC, int MODE, int D>
static int launch_mfma(const T4& f1, const T4& f2, const T4& flow, const T4& occ, const T4& out, int B, int C, int H, int W,
                       float fs, int ac, int out_mode, hipStream_t st) {
  switch (band_variant(B, C, H, W, MODE == 2, D)) {
    case OCF_CORR_BAND_TH8: return launch_mfma_th<VEC, MODE, D, false, TH_SINGLE>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, out_mode, st);
    case OCF_CORR_BAND_MULTI_TH1: return launch_mfma_th<VEC, MODE, D, true, TH_MULTI_TINY>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, out_mode, st);
    case OCF_CORR_BAND_MULTI_TH2:
      if constexpr (!(MODE == 2 && D == 4)) return launch_mfma_th<VEC, MODE, D, true, TH_MULTI_SMALL>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, out_mode, st);
      return OCF_EUNSUPPORTED;  // (not reached: band_variant)
    default:
      if constexpr (MODE != 2) return launch_mfma_th<VEC, MODE, D, true, TH_MULTI>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, out_mode, st);
      return OCF_EUNSUPPORTED;  // (not reached: band_variant)
  }
}

template <int VEC, int MODE>
static int launch_mfma_d(const T4& f1, const T4& f2, const T4& flow, const T4& occ, const T4& out, int B, int C, int H, int W,
                         int d, float fs, int ac, int out_mode, hipStream_t st) {
  switch (d) {
    case 1: return launch_mfma<VEC, MODE, 1>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, out_mode, st);
    case 2: return launch_mfma<VEC, MODE, 2>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, out_mode, st);
    case 3: return launch_mfma<VEC, MODE, 3>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, out_mode, st);
    case 4: return launch_mfma<VEC, MODE, 4>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, out_mode, st);
    default: return OCF_EUNSUPPORTED;
  }
}

int corr_fwd_bf16_variant(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ,
                          const ocf_tensor4* out, int B, int C, int H, int W, int d, int policy) {
  if (policy == OCF_CORR_POLICY_BAND_ONLY) return band_vec(f1, f2, C) == 0 ? OCF_CORR_VECTOR_BF16 : band_variant(B, C, H, W, flow != nullptr, d);
  const bool wave_ok = policy != OCF_CORR_POLICY_NO_WAVE && corr_wave_eligible(f1, f2, flow, occ, out, B, C, H, W, d);
  if (policy == OCF_CORR_POLICY_WAVE && wave_ok) return OCF_CORR_WAVE;
  const int v = corr_rows_variant(f1, f2, flow, out, C, H, W, d);  // (0 when the output is not a dense cost volume; with a flow: C <= 64 only)
  if (v == OCF_CORR_DIRECT && wave_ok) return OCF_CORR_WAVE;       // small levels: independent waves instead of nine-wave workgroups
  if (v) return v;
  // no row kernel for this request: the wave kernel takes the small levels with the warp gathered in registers, and cost volumes
  // written into a wider buffer (any pixel pitch) that the band kernel would write element by element
  if (wave_ok && (C > 64 || !flow)) return OCF_CORR_WAVE;
  if (band_vec(f1, f2, C) == 0) return OCF_CORR_VECTOR_BF16;
  return band_variant(B, C, H, W, flow != nullptr, d);
}

int corr_fwd_bf16_mfma(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ,
                       const ocf_tensor4* out, int B, int C, int H, int W, int d, float fs, int ac, int policy, hipStream_t st) {
  const int variant = corr_fwd_bf16_variant(f1, f2, flow, occ, out, B, C, H, W, d, policy);
  if (variant == OCF_CORR_WAVE) return corr_fwd_bf16_wave(f1, f2, flow, occ, out, B, C, H, W, fs, ac, st);
  if (variant <= OCF_CORR_DIRECT) return corr_fwd_bf16_rows(f1, f2, occ, out, B, C, H, W, st, flow, fs, ac);
  auto span = [&](const ocf_tensor4* t, long c) { return (labs(t->sh) * H + labs(t->sw) * W + labs(t->sc) * c) * 2L; };
  if (span(f1, C) >= (1L << 31) || span(f2, C) >= (1L << 31)) return OCF_EUNSUPPORTED;
  const int vec = band_vec(f1, f2, C);
  // no 8-/16-byte channel runs (NCHW, odd strides, C % 4 != 0): the vector-ALU kernel with element loads.  The band kernel's
  // element-load staging spilled up to 377 VGPR dwords at the register budget of its nine-wave workgroups and is gone.
  if (vec == 0) return corr_fwd_bf16_vector(f1, f2, flow, occ, out, B, C, H, W, d, fs, ac, st);
  const int nk = (2 * d + 1) * (2 * d + 1);
  const bool dense = out->sc == 1 && out->sw == nk && ((uintptr_t)out->data % 16 == 0) && (out->sh * 2) % 16 == 0 &&
                     (out->sb * 2) % 16 == 0 && ((long)mf::TW * nk * 2) % 16 == 0;
  int out_mode = dense ? MOUT_CL_VEC : MOUT_GENERIC;
#ifdef OCF_TUNING_BUILD
  if (const char* ab = getenv("OCF_ABLATE")) out_mode |= atoi(ab) << 8;
#endif
  const T4 a = mk(f1), bq = mk(f2), fl = mk(flow), oc = mk(occ), o = mk(out);
#define OCF_DISPATCH(V)                                                                            \
  (flow ? launch_mfma_d<V, 2>(a, bq, fl, oc, o, B, C, H, W, d, fs, ac, out_mode, st)               \
        : (occ ? launch_mfma_d<V, 1>(a, bq, fl, oc, o, B, C, H, W, d, fs, ac, out_mode, st)        \
               : launch_mfma_d<V, 0>(a, bq, fl, oc, o, B, C, H, W, d, fs, ac, out_mode, st)))
  if (vec == 16) return OCF_DISPATCH(16);
  return OCF_DISPATCH(8);
#undef OCF_DISPATCH
}

}  // namespace ocf
