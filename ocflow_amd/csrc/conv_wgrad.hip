// Weight gradient of the 3x3 / stride-1 "same" convolutions of the estimator and context networks (the flow-decoder convs
// as hand-written MFMA kernels; this is the weight-gradient third of it), bf16 channels_last, gfx950.
//
//   dW[co][ci][ky][kx] = sum_{b,y,x} gy[b,y,x,co] * x[b, y + (ky-1) dil, x + (kx-1) dil, ci]        (zero outside the image)
//
// Why this one first: in the round-1 profile MIOpen's bf16 weight-gradient kernels were 3.2 ms of the 16.7 ms step PLUS 1.8 ms of
// helper launches -- every layer zero-fills an fp32 split-K workspace and casts it to bf16 afterwards (341 launches of ~5 us).  Here
// the split-K partial sums go straight into ONE pre-zeroed fp32 gradient buffer with float atomics: no workspace, no cast.
//
// GEMM view: M = Co, N = 9 * Ci, K = pixels.  Both operands are K-major in memory the WRONG way round for MFMA fragments (a pixel's
// channels are contiguous, the fragment wants 8 consecutive pixels of one channel), so the LDS tiles keep the memory layout
// ([32-channel block][pixel][32 ch], 64-byte pixel rows: conflict-free) and the fragments are read with ds_read_b64_tr_b16, the
// transposing LDS read of gfx950 (two per v_mfma_f32_32x32x16_bf16 operand).
//
// Decomposition: workgroup = (chunk of image rows, 32-channel block of Ci, group of <= 128 Co); wave = one 32-row block of Co with all
// nine taps (9 accumulator tiles = 144 registers), so a gy fragment is read once per k-step and reused nine times.  Per 64-pixel
// k-tile the gy tile and the three shifted x rows are staged through registers (next tile's loads fly under the MFMAs).
// Epilogue: one float atomic per accumulator element into dW (lanes = consecutive ci: whole 128-byte segments).
#include "ocf_common.h"

namespace ocf {
namespace wg {
constexpr int KT = 64;        // pixels per k-tile
constexpr int MAXD = 16;      // largest dilation (context_net.py:15)
constexpr int XR = KT + 2 * MAXD;  // x row tile, pixels
}  // namespace wg

typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// transposing fragment read: LDS tile [pixel][32 ch] (64-byte rows).  Lane l: channel l & 31, pixels kb + 8 (l >> 5) + 0..7.
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int kb, int lane) {
  const int i = lane & 15, grp = (lane >> 4) & 1, h = lane >> 5;
  const int q = i >> 2, p = i & 3;
  const char* a0 = tile + (kb + 8 * h + q) * 64 + (16 * grp + 4 * p) * 2;
  typedef s4v __attribute__((address_space(3))) * lds_s4;
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)a0);
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(a0 + 4 * 64));
  typedef short s8v __attribute__((ext_vector_type(8)));
  const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(const char* __restrict__ x, const char* __restrict__ gy, float* __restrict__ gw,
                                                               long so, long si, long sky, long skx, int B, int Ci, int Co, int H, int W,
                                                               int dil, int rows_per_chunk) {
  using namespace wg;
  __shared__ __attribute__((aligned(16))) char smem[4 * KT * 64 + 3 * XR * 64];
  char* s_gy = smem;                 // [m-block][KT px][32 ch]
  char* s_x = smem + 4 * KT * 64;    // [ky][XR px][32 ch]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> (row chunk, ci block, co group).  The ci blocks of one row chunk read the same gy rows: they get block ids 8 apart
  // (same XCD under round-robin dispatch, adjacent in time) so that all but the first read them from that XCD's L2 (speed only)
  const int ncib = (Ci + 31) >> 5;
  const int bid = blockIdx.x;
  const int chunk = ((bid >> 3) / ncib) * 8 + (bid & 7), cib = (bid >> 3) % ncib, cog = blockIdx.y;
  const int co0 = cog * 128, ci0 = cib * 32;
  const int co_n = min(128, Co - co0);
  const int mb = (co_n + 31) >> 5;
  const int nrows = B * H;
  const int r0 = min(nrows, chunk * rows_per_chunk), r1 = min(nrows, r0 + rows_per_chunk);
  const int xw = KT + 2 * dil;  // pixels of an x row tile in use
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  f32x16 acc[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[a][c][e] = 0.f;

  // staging registers of one k-tile: gy = 64 px x (mb*4) units -> <= 4 per thread; x = 3 rows x xw px x 4 units -> <= 5 per thread
  u32x4 rg[4], rx[5];
  const int ntx = (W + KT - 1) / KT;
  const int ntiles = (r1 - r0) * ntx;
  auto fetch = [&](int t) __attribute__((always_inline)) {
    const int row = r0 + t / ntx, x0 = (t % ntx) * KT;
    const int b = row / H, y = row - b * H;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int u = tid + 256 * k;            // unit index: pixel-major, then 8-channel group
      const int px = u / (mb * 4), c8 = u - px * (mb * 4);
      const int gx = x0 + px, ch = co0 + c8 * 8;
      const bool ok = k < mb && gx < W && ch < Co;
      const long off = (((long)row * W + (ok ? gx : 0)) * Co + (ok ? ch : 0)) * 2;
      const u32x4 v = *(const u32x4*)(gy + off);
      rg[k] = ok ? v : zero4;
    }
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const int u = tid + 256 * k;            // (ky, pixel, unit)
      const int per_row = xw * 4;
      const int ky = u / per_row, rem = u - ky * per_row;
      const int j = rem >> 2, c8 = rem & 3;
      const int gyy = y + (ky - 1) * dil, gx = x0 - dil + j, ch = ci0 + c8 * 8;
      const bool ok = ky < 3 && gyy >= 0 && gyy < H && gx >= 0 && gx < W && ch < Ci;
      const long off = ((((long)b * H + (ok ? gyy : 0)) * W + (ok ? gx : 0)) * Ci + (ok ? ch : 0)) * 2;
      const u32x4 v = *(const u32x4*)(x + off);
      rx[k] = ok ? v : zero4;
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int u = tid + 256 * k;
      const int px = u / (mb * 4), c8 = u - px * (mb * 4);
      if (k < mb) *(u32x4*)(s_gy + (c8 >> 2) * (KT * 64) + px * 64 + (c8 & 3) * 16) = rg[k];
    }
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const int u = tid + 256 * k;
      const int per_row = xw * 4;
      const int ky = u / per_row, rem = u - ky * per_row;
      if (ky < 3) *(u32x4*)(s_x + ky * (XR * 64) + rem * 16) = rx[k];
    }
  };

  if (ntiles > 0) fetch(0);
  for (int t = 0; t < ntiles; t++) {
    commit();
    __syncthreads();
    if (t + 1 < ntiles) fetch(t + 1);
    if (wave < mb) {
      const char* ga = s_gy + wave * (KT * 64);
#pragma unroll 1
      for (int ks = 0; ks < KT / 16; ks++) {
        const bf16x8 a = tr_frag(ga, 16 * ks, lane);
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
          for (int kx = 0; kx < 3; kx++) {
            const bf16x8 bq = tr_frag(s_x + ky * (XR * 64), 16 * ks + kx * dil, lane);
            acc[ky][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc[ky][kx], 0, 0, 0);
          }
      }
    }
    __syncthreads();
  }
  // ---- epilogue: D[row = co][col = ci]: lane holds column l & 31, rows (e & 3) + 8 (e >> 2) + 4 (l >> 5)
  if (wave < mb && ntiles > 0) {
    const int ci = ci0 + (lane & 31);
    if (ci < Ci) {
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++)
#pragma unroll
          for (int e = 0; e < 16; e++) {
            const int co = co0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (co < Co) atomicAdd(gw + (long)co * so + (long)ci * si + ky * sky + kx * skx, acc[ky][kx][e]);
          }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Version 2 for the large layers: the first kernel is latency-bound -- one k-tile of prefetch (28.7 KB per workgroup, 0.5 us of MFMA
// work) cannot cover a 2-3 us memory latency, it ran at 10 % of the matrix-core rate.  Here the operands go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers) into a THREE-stage ring, so two k-tiles (~80 KB) are in flight under the MFMAs of
// the third, with ONE barrier per k-tile (counted vmcnt: the DMA of later tiles stays in flight across it).  Workgroup = 8 waves =
// (4 Co blocks) x (2 Ci blocks) x 9 taps: twice the MACs per staged byte of version 1.
// ---------------------------------------------------------------------------------------------------------------------------------
namespace wg2 {
constexpr int KT = 64, GY_BYTES = 4 * KT * 64, NPW = 7;  // DMA pieces (1 KiB wave instructions) every wave issues per k-tile
}
__device__ u32x4 g_wzero[64];  // 1 KiB of zeros: the source of out-of-image / out-of-range DMA lanes

__device__ __forceinline__ void dma16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_v2_kernel(const char* __restrict__ x, const char* __restrict__ gy, float* __restrict__ gw,
                                                                  long so, long si, long sky, long skx, int B, int Ci, int Co, int H, int W,
                                                                  int dil, int rows_per_chunk, int XP, int tune) {
  using namespace wg2;
#ifndef OCF_TUNING_BUILD
  tune = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) char sm2[];
  const int stage_bytes = GY_BYTES + 2 * 3 * XP * 64;
  char* s_dummy = sm2 + 3 * stage_bytes;  // 1 KiB: where surplus DMA pieces land
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncp = (Ci + 63) >> 6;
  const int bid = blockIdx.x;
  const int chunk = ((bid >> 3) / ncp) * 8 + (bid & 7), cp = (bid >> 3) % ncp, cog = blockIdx.y;
  const int co0 = cog * 128, ci0 = cp * 64;
  const int mb = (min(128, Co - co0) + 31) >> 5, nc = (min(64, Ci - ci0) + 31) >> 5;
  const int nrows = B * H;
  const int r0 = min(nrows, chunk * rows_per_chunk), r1 = min(nrows, r0 + rows_per_chunk);
  const int ntx = (W + KT - 1) / KT;
  const int ntiles = (r1 - r0) * ntx;
  const int xw = KT + 2 * dil, xpg = XP >> 4;
  const int npieces = 4 * mb + nc * 3 * xpg;
  const char* zp = (const char*)g_wzero;

  // ---- issue the DMA of k-tile t into ring stage (t % 3): every wave exactly NPW pieces (surplus ones copy zeros to a dummy KiB)
  auto issue = [&](int t) __attribute__((always_inline)) {
    const int row = r0 + t / ntx, x0 = (t % ntx) * KT;
    const int b = row / H, y = row - b * H;
    char* st = sm2 + (t % 3) * stage_bytes;
#pragma unroll
    for (int k = 0; k < NPW; k++) {
      const int q = wave + 8 * k;  // wave-uniform piece id
      const char* src = zp + lane * 16;
      char* dst = s_dummy;
      if (q < 4 * mb) {
        const int blk = q >> 2, pg = q & 3;
        const int gx = x0 + pg * 16 + (lane >> 2), ch = co0 + blk * 32 + (lane & 3) * 8;
        if (gx < W && ch < Co) src = gy + (((long)row * W + gx) * Co + ch) * 2;
        dst = st + blk * (KT * 64) + pg * 1024;
      } else if (q < npieces) {
        const int q2 = q - 4 * mb;
        const int cb = q2 / (3 * xpg), rem = q2 - cb * (3 * xpg);
        const int ky = rem / xpg, pg = rem - ky * xpg;
        const int j = pg * 16 + (lane >> 2);
        const int gyy = y + (ky - 1) * dil, gx = x0 - dil + j, ch = ci0 + cb * 32 + (lane & 3) * 8;
        if (j < xw && gyy >= 0 && gyy < H && gx >= 0 && gx < W && ch < Ci) src = x + ((((long)b * H + gyy) * W + gx) * Ci + ch) * 2;
        dst = st + GY_BYTES + ((cb * 3 + ky) * XP + pg * 16) * 64;
      }
      dma16(src, dst);
    }
  };

  f32x16 acc[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[a][c][e] = 0.f;
  const int mi = wave & 3, cj = wave >> 2;
  const bool active = mi < mb && cj < nc;

  if (ntiles > 0) issue(0);
  if (ntiles > 1) issue(1);
  for (int t = 0; t < ntiles; t++) {
    // tile t has landed (this wave's share; the barrier makes it everybody's), tile t+1 may still be in flight
    if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // also: every wave is done with the MFMAs of tile t-1, whose stage tile t+2 overwrites
    if (t + 2 < ntiles) issue(t + 2);
    if (active && !(tune & 2)) {
      const char* st = sm2 + (t % 3) * stage_bytes;
      const char* ga = st + mi * (KT * 64);
      const char* xb = st + GY_BYTES + (cj * 3) * XP * 64;
#pragma unroll 1
      for (int ks = 0; ks < KT / 16; ks++) {
        const bf16x8 a = tr_frag(ga, 16 * ks, lane);
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
          for (int kx = 0; kx < 3; kx++) {
            const bf16x8 bq = tr_frag(xb + ky * (XP * 64), 16 * ks + kx * dil, lane);
            acc[ky][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc[ky][kx], 0, 0, 0);
          }
      }
    }
  }
  if (active && ntiles > 0 && !(tune & 1)) {
    const int ci = ci0 + cj * 32 + (lane & 31);
    if (ci < Ci) {
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++)
#pragma unroll
          for (int e = 0; e < 16; e++) {
            const int co = co0 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (co < Co) atomicAdd(gw + (long)co * so + (long)ci * si + ky * sky + kx * skx, acc[ky][kx][e]);
          }
    }
  }
}

}  // namespace ocf

extern "C" int ocf_conv3x3_wgrad(const ocf_tensor4* x, const ocf_tensor4* gy, float* gw, int64_t gw_so, int64_t gw_si, int64_t gw_sky,
                                 int64_t gw_skx, int B, int Ci, int Co, int H, int W, int dilation, int dtype, int policy, void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !gy || !gw || !x->data || !gy->data) return OCF_EINVAL;
  if (policy < 0 || policy > 2) return OCF_EINVAL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || dilation < 1) return OCF_EINVAL;
  if (dtype != OCF_BF16 || dilation > wg::MAXD || Ci % 8 != 0 || Co % 8 != 0) return OCF_EUNSUPPORTED;
  // dense channels_last operands with 16-byte aligned pixels
  auto dense_cl = [&](const ocf_tensor4* t, int C) {
    return t->sc == 1 && t->sw == C && t->sh == (int64_t)W * C && t->sb == (int64_t)H * W * C && ((uintptr_t)t->data % 16 == 0);
  };
  if (!dense_cl(x, Ci) || !dense_cl(gy, Co)) return OCF_EUNSUPPORTED;
  if ((long)B * H * W * (long)(Ci > Co ? Ci : Co) * 2 >= (1L << 40)) return OCF_EUNSUPPORTED;
  // automatic = version 1: measured at the benchmark's layer shapes (profiles/r2_d_wgrad_v2.txt) the ring kernel is DMA-throughput and
  // LDS-read bound where version 1 is latency bound, and ends up within +-10 % of it; it stays selectable for tuning
  const bool v2 = policy == 2;
  const int cib = v2 ? (Ci + 63) / 64 : (Ci + 31) / 32, cog = (Co + 127) / 128;
  const int nrows = B * H;
  // split-K factor: every row chunk adds a full copy of dW with float atomics (~1.3 TB/s chip-wide), while fewer than ~2 workgroups
  // per CU leave matrix cores idle -- pick the chunk count that minimises the sum of the two estimates
  const double flops = 2.0 * nrows * (double)W * Ci * Co * 9, dw_bytes = 4.0 * Ci * Co * 9;
  int best = 1;
  double best_t = 1e30;
  for (int c = 1; c <= nrows; c = c < 8 ? c + 1 : c + c / 4) {
    const double wgs = (double)c * cib * cog;
    const double full = v2 ? 256.0 : 512.0;  // version 2 runs one (eight-wave) workgroup per CU
    const double t = flops / 4.0e14 * (wgs >= full ? 1.0 : full / wgs) + c * dw_bytes / 1.3e12;
    if (t < best_t) { best_t = t; best = c; }
  }
  int chunks = best, tune = 0;
#ifdef OCF_TUNING_BUILD
  if (const char* e = getenv("OCF_WGRAD_CHUNKS")) chunks = atoi(e) > 0 ? atoi(e) : chunks;
  if (const char* e = getenv("OCF_WGRAD_TUNE")) tune = atoi(e);
#endif
  const int rpc = (nrows + chunks - 1) / chunks;
  chunks = ((nrows + rpc - 1) / rpc + 7) / 8 * 8;  // a multiple of 8 for the block -> chunk map (surplus chunks have no rows)
  LaunchProf lp;
  if (v2) {
    const int XP = dilation <= 8 ? 80 : 96;
    const size_t lds = 3 * (size_t)(wg2::GY_BYTES + 2 * 3 * XP * 64) + 1024;
    static size_t lds_set = 0;
    if (lds > lds_set) {
      (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_v2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      lds_set = lds;
    }
    lp.launch(true, conv3x3_wgrad_v2_kernel, dim3((unsigned)(chunks * cib), (unsigned)cog), dim3(512), lds, (hipStream_t)stream,
              (const char*)x->data, (const char*)gy->data, gw, (long)gw_so, (long)gw_si, (long)gw_sky, (long)gw_skx, B, Ci, Co, H, W, dilation, rpc,
              XP, tune);
    return launch_status();
  }
  lp.launch(true, conv3x3_wgrad_kernel, dim3((unsigned)(chunks * cib), (unsigned)cog), dim3(256), 0, (hipStream_t)stream, (const char*)x->data,
            (const char*)gy->data, gw, (long)gw_so, (long)gw_si, (long)gw_sky, (long)gw_skx, B, Ci, Co, H, W, dilation, rpc);
  return launch_status();
}
