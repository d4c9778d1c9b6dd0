// 3x3 / stride-1 / padding == dilation convolutions of the estimator and context networks as ONE implicit-GEMM kernel family on the
// matrix cores, bf16 channels_last, gfx950 ("flow-decoder dense block as hand MFMA implicit-GEMM", reference layers
// flow_net.py:11-37, flow_occ_net.py:11-67, context_net.py:11-17).
//
//   y[b,y,x,m] = epi( sum_{ky,kx,k} Wp[m][ky*3+kx][k] * x[b, y + (ky-1) d, x + (kx-1) d, k] )          (zero outside the image)
//   epi(v)     = lrelu(v + bias[m], slope) * (act[b,y,x,m] > 0 ? 1 : mask_slope)
//   with a residual (ocf_conv3x3_igemm_ex: the DenseNet blocks of cost_volume_flow_net.py:176-180, whose layers all read -- and whose input
//   gradients all add into -- channel ranges of ONE buffer):  mask m only on channels < act_channels, and
//       y = res_premask ? (epi0(v) + res) * m : epi0(v) * m + res            (res may be y itself: every element is read, then written, by one lane)
//
// The SAME kernel is the forward pass (Wp = the weight, epilogue = bias + LeakyReLU: the framework's conv + `add_` + `leaky_relu` in one
// launch) and the input gradient (x := grad_out, Wp = the weight with taps flipped and channels transposed, packed by
// ocf_conv3x3_pack_weights; epilogue = the LeakyReLU mask of the PREVIOUS layer taken from its output `act`: conv-backward-data +
// leaky_relu_backward in one launch).
//
// GEMM view per workgroup: M = MT*16 output channels, N = 512 output pixels (a TH x TW patch of one image), K = 9 taps x Cin.
// A operand = weights, B operand = pixels, v_mfma_f32_16x16x32_bf16: a lane's four accumulator registers are four CONSECUTIVE output
// channels of one pixel, i.e. one 8-byte store into the channels_last output.
//
// Staging: global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers), two buffers, one barrier per stage.  A stage is
// (32-channel block c, tap row ky): the TH x (TW + 2d) pixel rows the three taps of that row touch (64 bytes per pixel) and the three
// [M][32] weight slices.  Every wave instruction of the DMA writes 1 KiB lane-linearly; out-of-image pixels and channel tails are
// sourced from a zero page, so the LDS image needs no fix-up.  Both LDS images are rows of 64 bytes = 4 slots of 16 bytes (one
// 8-channel k-group each), slot g of row r stored at slot (g + 2 (r >> 2)) & 3: the ds_read_b128 of a 16x16x32 operand (lane = row
// l & 15, k-group l >> 4) is then bank-conflict free for every row offset (the hardware serves the 64 lanes in four fixed groups of 16:
// lanes {0-3, 12-15, 20-27} ...); the permutation is applied on the DMA's per-lane SOURCE address.
#include "ocf_common.h"
#include <mutex>
#include <set>
#include <utility>

namespace ocf {
namespace ig {
constexpr int NTHR = 512, NWAVE = 8, PIX_ITERS = 6;  // <= 48 KiB of pixel rows per stage
__device__ u32x4 g_zero[64];                          // 1 KiB of zeros (not const: a const source turns the selected load into flat_load)

struct Args {
  const char* x;      // [B][H][W][xp] bf16, channels [0, Cin) used
  const char* w;      // packed [rows][9][wK] bf16, rows a multiple of MT*16 (zero padded), wK a multiple of 8 (zero padded past Cin)
  char* y;            // [B][H][W][yp] bf16, channels [0, Cout) written
  const float* bias;  // [Cout] or null
  const char* act;    // [B][H][W][ap] bf16 or null
  const char* res;    // [B][H][W][rp] bf16 or null: residual added in the epilogue (row-store epilogue only)
  long xp, yp, ap, rp;  // pixel pitches, elements
  int act_ch;         // the mask applies to output channels < act_ch
  int res_premask;    // 1: (v + res) * mask, 0: v * mask + res
  int B, H, W, Cin, Cout, wK, d, tiles_x, tiles_y;
  int rows_epi;  // 1: row-store epilogue through LDS (Cout % 8 == 0, 16-byte aligned output pixels)
  int tune;  // ablation (policy bits 1024 / 2048 / 4096): no fragment reads + MFMAs / no DMA / no epilogue stores
  float slope, mslope;
};

__device__ __forceinline__ void dma16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float bf_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }

// FULL: Cout % 4 == 0, every lane's four channels exist together (8-byte loads / stores); otherwise per-channel accesses.
// Loads are issued for all MT channel tiles of a pixel before the first use, from addresses that are valid for every lane
// (out-of-range lanes read the tensor's first elements and store nothing): no load sits behind a lane-dependent branch.
template <int MT, int TW, bool FULL, bool ACT>
__device__ __forceinline__ void epilogue(const Args& a, const f32x4 (&acc)[MT][4], int wave, int l15, int cb0, int b, int x0, int y0) {
  constexpr int FC = TW / 16;
  const float slope = a.slope, mslope = a.mslope;
  f32x4 bv[MT];
#pragma unroll
  for (int m = 0; m < MT; m++) {
    bv[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
      const int cb = cb0 + m * 16;
      if (FULL) {
        bv[m] = *(const f32x4*)(a.bias + (cb < a.Cout ? cb : 0));
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) bv[m][e] = a.bias[cb + e < a.Cout ? cb + e : 0];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int f = wave * 4 + j, fr = f / FC, fc = f - fr * FC;
    const int yy = y0 + fr, xx = x0 + fc * 16 + l15;
    const bool pix_ok = yy < a.H && xx < a.W;
    const long pix = pix_ok ? ((long)b * a.H + yy) * a.W + xx : 0;
    u32x2 mv[MT];
    if (ACT && FULL) {
#pragma unroll
      for (int m = 0; m < MT; m++) {
        const int cb = cb0 + m * 16;
        mv[m] = *(const u32x2*)(a.act + (pix * a.ap + (cb < a.Cout ? cb : 0)) * 2);
      }
    }
#pragma unroll
    for (int m = 0; m < MT; m++) {
      const int cb = cb0 + m * 16;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        v[e] = acc[m][j][e] + bv[m][e];
        v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
      }
      if (FULL) {
        if (ACT) {
          v[0] *= bf_lo(mv[m][0]) > 0.f ? 1.f : mslope; v[1] *= bf_hi(mv[m][0]) > 0.f ? 1.f : mslope;
          v[2] *= bf_lo(mv[m][1]) > 0.f ? 1.f : mslope; v[3] *= bf_hi(mv[m][1]) > 0.f ? 1.f : mslope;
        }
        const bf16x2 lo = {(bf16)v[0], (bf16)v[1]}, hi = {(bf16)v[2], (bf16)v[3]};
        const u32x2 o = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
        if (pix_ok && cb < a.Cout) *(u32x2*)(a.y + (pix * a.yp + cb) * 2) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const bool ok = pix_ok && cb + e < a.Cout;
          float o = v[e];
          if (ACT) o *= (float)(((const bf16*)a.act)[pix * a.ap + (ok ? cb + e : 0)]) > 0.f ? 1.f : mslope;
          if (ok) ((bf16*)a.y)[pix * a.yp + cb + e] = (bf16)o;
        }
      }
    }
  }
}

// Row-store epilogue (Cout % 8 == 0, 16-byte aligned output pixels): the accumulators leave the wave as WHOLE pixel rows.  Each lane
// holds 4 consecutive channels of one pixel per accumulator register -- stored directly that is 32 instructions of 8 bytes per lane
// at a 2*Cout-byte stride (measured: 19 of the 85 us of 8 x 128 -> 128 x 109 x 256, a store tail bound by instruction count, not by
// bytes).  Here bias + LeakyReLU are applied in registers, the wave's 64 x (16 MT) tile goes through its own LDS slab
// ([pixel][channel], pitch 32 MT + 16 bytes: the ds_write_b64 of 16 pixels and the ds_read_b128 of a row are both conflict-free) and
// is stored 16 bytes per lane, 16 MT / 8 lanes per pixel row; the activation-derivative mask is read the same way.
template <int MT, int TW, bool ACT, bool RES = false>
__device__ __forceinline__ void epilogue_rows(const Args& a, const f32x4 (&acc)[MT][4], char* slab, int wave, int lane, int mrow0, int b, int x0,
                                              int y0) {
  constexpr int FC = TW / 16, PITCH = MT * 32 + 16, CPR = MT * 2;  // 16-byte chunks per pixel row
  const int g = lane >> 4, l15 = lane & 15;
  const float slope = a.slope, mslope = a.mslope;
#pragma unroll
  for (int m = 0; m < MT; m++) {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    const int cb = mrow0 + m * 16 + 4 * g;
    if (a.bias) bv = *(const f32x4*)(a.bias + (cb < a.Cout ? cb : 0));
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        v[e] = acc[m][j][e] + bv[e];
        v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
      }
      const bf16x2 lo = {(bf16)v[0], (bf16)v[1]}, hi = {(bf16)v[2], (bf16)v[3]};
      const u32x2 o = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
      *(u32x2*)(slab + (j * 16 + l15) * PITCH + (m * 16 + 4 * g) * 2) = o;
    }
  }
  // (the slab is private to the wave: its own LDS writes are ordered before its reads, no barrier)
#pragma unroll
  for (int i = 0; i < (64 * CPR + 63) / 64; i++) {
    const int idx = i * 64 + lane;
    const int row = idx / CPR, chunk = idx - row * CPR;  // row = j * 16 + pixel-in-fragment
    const int j = row >> 4, f = wave * 4 + j, fr = f / FC, fc = f - fr * FC;
    const int yy = y0 + fr, xx = x0 + fc * 16 + (row & 15);
    const int cb = mrow0 + chunk * 8;
    const bool ok = idx < 64 * CPR && yy < a.H && xx < a.W && cb < a.Cout;
    const long pix = ok ? ((long)b * a.H + yy) * a.W + xx : 0;
    u32x4 v = *(const u32x4*)(slab + (ok ? row : 0) * PITCH + (ok ? chunk : 0) * 16);
    if (RES) {
      // (both loads from always-valid addresses, issued together, before any use)
      u32x4 mv = {0u, 0u, 0u, 0u};
      if (ACT) mv = *(const u32x4*)(a.act + (pix * a.ap + (ok ? cb : 0)) * 2);
      const u32x4 rv = *(const u32x4*)(a.res + (pix * a.rp + (ok ? cb : 0)) * 2);
      const bool pre = a.res_premask != 0;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float lo = bf_lo(v[e]), hi = bf_hi(v[e]);
        const float rlo = bf_lo(rv[e]), rhi = bf_hi(rv[e]);
        float mlo = 1.f, mhi = 1.f;
        if (ACT) {
          mlo = (cb + 2 * e < a.act_ch && !(bf_lo(mv[e]) > 0.f)) ? mslope : 1.f;
          mhi = (cb + 2 * e + 1 < a.act_ch && !(bf_hi(mv[e]) > 0.f)) ? mslope : 1.f;
        }
        lo = pre ? (lo + rlo) * mlo : lo * mlo + rlo;
        hi = pre ? (hi + rhi) * mhi : hi * mhi + rhi;
        const bf16x2 p2 = {(bf16)lo, (bf16)hi};
        v[e] = __builtin_bit_cast(unsigned, p2);
      }
    } else if (ACT) {
      const u32x4 mv = *(const u32x4*)(a.act + (pix * a.ap + (ok ? cb : 0)) * 2);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float lo = bf_lo(v[e]), hi = bf_hi(v[e]);
        lo *= bf_lo(mv[e]) > 0.f ? 1.f : mslope;
        hi *= bf_hi(mv[e]) > 0.f ? 1.f : mslope;
        const bf16x2 p2 = {(bf16)lo, (bf16)hi};
        v[e] = __builtin_bit_cast(unsigned, p2);
      }
    }
    if (ok) *(u32x4*)(a.y + (pix * a.yp + cb) * 2) = v;
  }
}

template <int MT, int TW>
__global__ __launch_bounds__(NTHR, 2) void conv3x3_igemm_kernel(const Args a) {
  constexpr int TH = 512 / TW, FC = TW / 16, SPLIT_KX = 0;
  constexpr int W_BYTES = 3 * MT * 1024, W_ITERS = (3 * MT + NWAVE - 1) / NWAVE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = a.d, TWP = TW + 2 * d, H = a.H, W = a.W;
  const int npi = (TH * TWP * 64 + 1023) >> 10;  // wave instructions (1 KiB) per stage of pixel rows
  const int pix_bytes = npi << 10, buf_bytes = pix_bytes + W_BYTES;

  const int ntile = a.tiles_x * a.tiles_y * a.B;
  const int t = xcd_remap(blockIdx.x, ntile);
  const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, b = t / (a.tiles_x * a.tiles_y);
  const int x0 = tx * TW, y0 = ty * TH;
  const int mrow0 = blockIdx.y * (MT * 16);

  const int g = lane >> 4, l15 = lane & 15;
  // logical k-group this lane SOURCES for the DMA: it writes physical slot lane & 3 of row (16 q + (lane >> 2)), and (row >> 2) & 3 = g
  const int slot_src = ((lane & 3) - 2 * g) & 3;
  const char* zsrc = (const char*)g_zero + lane * 16;

  // ---- stage-invariant DMA source offsets (elements, relative to the stage's origin pixel) -------------------------------------
  int poff[PIX_ITERS], prow[PIX_ITERS];
#pragma unroll
  for (int k = 0; k < PIX_ITERS; k++) {
    const int q = k * NWAVE + wave;
    const int pidx = q * 16 + (lane >> 2);
    const int r = pidx / TWP, c = pidx - r * TWP;
    const int gx = x0 - d + c;
    const bool ok = q < npi && r < TH && gx >= 0 && gx < W;
    poff[k] = (int)((r * W + c) * a.xp) + slot_src * 8;
    prow[k] = ok ? r : -(1 << 20);
  }
  int woff[W_ITERS];
#pragma unroll
  for (int k = 0; k < W_ITERS; k++) {
    const int q = k * NWAVE + wave;  // = kx * MT + m
    const int kx = q / MT, m = q - kx * MT;
    woff[k] = ((m * 16 + (lane >> 2)) * 9 + kx) * a.wK + slot_src * 8;
  }
  // ---- fragment read offsets -------------------------------------------------------------------------------------------------------
  const int aoff = l15 * 64 + ((g + 2 * (l15 >> 2)) & 3) * 16;
  int boff[4][3];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int f = wave * 4 + j, fr = f / FC, fc = f - fr * FC;
#pragma unroll
    for (int kx = 0; kx < 3; kx++) {
      const int pidx = fr * TWP + fc * 16 + l15 + kx * d;
      boff[j][kx] = pidx * 64 + ((g + 2 * ((pidx >> 2) & 3)) & 3) * 16;
    }
  }

  const int nchunk = (a.Cin + 31) >> 5;
  const int nstage = nchunk * 3;
  auto issue = [&](int s) __attribute__((always_inline)) {
    const int c = s / 3, ky = s - 3 * c;
    char* buf = smem + (s & 1) * buf_bytes;
    const int ch = c * 32 + slot_src * 8;
    const bool x_ok = ch < a.Cin, w_ok = ch < a.wK;
    const int dy = (ky - 1) * d;
    const char* xb = a.x + ((((long)b * H + y0 + dy) * W + (x0 - d)) * a.xp + c * 32) * 2;
#pragma unroll
    for (int k = 0; k < PIX_ITERS; k++) {
      const int q = k * NWAVE + wave;
      if (q < npi) {
        const int yy = y0 + prow[k] + dy;
        const bool ok = x_ok && yy >= 0 && yy < H;
        dma16(ok ? xb + (long)poff[k] * 2 : zsrc, buf + q * 1024);
      }
    }
    const char* wb = a.w + (((long)mrow0 * 9 + ky * 3) * a.wK + c * 32) * 2;
#pragma unroll
    for (int k = 0; k < W_ITERS; k++) {
      const int q = k * NWAVE + wave;
      if (q < 3 * MT) dma16(w_ok ? wb + (long)woff[k] * 2 : zsrc, buf + pix_bytes + q * 1024);
    }
  };

  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (!(a.tune & 2)) issue(0);
  for (int s = 0; s < nstage; s++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of stage s has landed
    __builtin_amdgcn_s_barrier();                      // everybody's has; and every wave is done reading the other buffer
    // The two waves of a SIMD (w and w + 4) issue the next stage's DMA at DIFFERENT points of the stage: an LDS-DMA instruction holds its
    // wave at issue while the CU's load path (~60 GB/s) drains, and with all eight waves issuing first and computing afterwards the
    // matrix cores idled through the issue phase.  Waves 0-3 issue before their first tap, waves 4-7 after it.
    const bool early = wave < 4;
    if (s + 1 < nstage && early && !(a.tune & 2)) issue(s + 1);
    const char* buf = smem + (s & 1) * buf_bytes;
    const char* wbuf = buf + pix_bytes + aoff;
    // Fragment reads are software-pipelined by hand: the weight fragment of step (kx, m + 1) -- and, at the end of a tap, the four pixel
    // fragments of tap kx + 1 -- are requested BEFORE the four MFMAs of step (kx, m) issue.  Left to the compiler, every step was
    // `ds_read_b128 aq; s_waitcnt lgkmcnt(0); 4 x v_mfma` into ONE register quad: 24 exposed LDS round trips per stage and wave
    // (~100 cycles each beside 64 cycles of matrix work), which is why the fragment-read and matrix phases of this kernel ADDED in the
    // round-4 ablation instead of overlapping.
    if (!(a.tune & 1)) {
      bf16x8 bq[2][4], aq[2];
#pragma unroll
      for (int j = 0; j < 4; j++) bq[0][j] = *(const bf16x8*)(buf + boff[j][0]);
      aq[0] = *(const bf16x8*)(wbuf);
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
#pragma unroll
        for (int m = 0; m < MT; m++) {
          const int st = kx * MT + m;  // (all indices below are compile-time after unrolling)
          if (m + 1 < MT) {
            aq[(st + 1) & 1] = *(const bf16x8*)(wbuf + (st + 1) * 1024);
          } else if (kx + 1 < 3) {
#pragma unroll
            for (int j = 0; j < 4; j++) bq[(kx + 1) & 1][j] = *(const bf16x8*)(buf + boff[j][kx + 1]);
            aq[(st + 1) & 1] = *(const bf16x8*)(wbuf + (st + 1) * 1024);
          }
          __builtin_amdgcn_sched_barrier(0);  // the requests above stay above the matrix work below
#pragma unroll
          for (int j = 0; j < 4; j++) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[st & 1], bq[kx & 1][j], acc[m][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kx == SPLIT_KX && s + 1 < nstage && !early && !(a.tune & 2)) issue(s + 1);
      }
    } else if (s + 1 < nstage && !early && !(a.tune & 2)) {
      issue(s + 1);
    }
  }

  // ---- epilogue: lane = pixel l15 of fragment j, registers = channels 16 m + 4 g + 0..3 ----------------------------------------------
  const int cb0 = mrow0 + 4 * g;
  if (a.tune & 4) return;
  if (a.rows_epi) {
    __syncthreads();  // every wave is done reading the last stage's buffer: the slabs overlay the stage buffers
    char* slab = smem + wave * (64 * (MT * 32 + 16));
    if (a.res) {
      if (a.act) epilogue_rows<MT, TW, true, true>(a, acc, slab, wave, lane, mrow0, b, x0, y0);
      else epilogue_rows<MT, TW, false, true>(a, acc, slab, wave, lane, mrow0, b, x0, y0);
    } else if (a.act) epilogue_rows<MT, TW, true>(a, acc, slab, wave, lane, mrow0, b, x0, y0);
    else epilogue_rows<MT, TW, false>(a, acc, slab, wave, lane, mrow0, b, x0, y0);
    return;
  }
  if ((a.Cout & 3) == 0) {
    if (a.act) epilogue<MT, TW, true, true>(a, acc, wave, l15, cb0, b, x0, y0);
    else epilogue<MT, TW, true, false>(a, acc, wave, l15, cb0, b, x0, y0);
  } else {
    if (a.act) epilogue<MT, TW, false, true>(a, acc, wave, l15, cb0, b, x0, y0);
    else epilogue<MT, TW, false, false>(a, acc, wave, l15, cb0, b, x0, y0);
  }
}

// ---- weight packing: any-strided [Co][Ci][3][3] (fp32 or bf16) -> packed bf16 [rows_pad][9][k_pad], zero padded -----------------------
//   transpose == 0: rows = Co, k = Ci, tap t = ky*3 + kx                         (forward operand)
//   transpose == 1: rows = Ci, k = Co, tap t = (2-ky)*3 + (2-kx)                 (input-gradient operand: flipped taps, swapped channels)
struct PackDesc {
  const void* src;
  void* dst;
  long so, si, sky, skx;  // element strides of src
  int Co, Ci, rows_pad, k_pad, transpose, src_f32;
  long first;  // index of this tensor's first packed element in the launch's flat index space
};

__global__ __launch_bounds__(256) void conv3x3_pack_kernel(const PackDesc* __restrict__ descs, int n, long total) {
  // blockIdx.y = tensor; the blocks of a tensor walk its (tap, 32-row, 32-k) tiles of the VALID region only (the zero padding of the
  // packed copy is written once, when it is allocated).  A tile goes through LDS so that both the read (contiguous along the source's
  // fastest dimension) and the write (contiguous along k) are whole runs; the first versions looked up the descriptor of every
  // element (59 us per step) / wrote the padding every step with two 64-bit divisions per element (100 us).
  __shared__ float tile[32][33];
  const PackDesc p = descs[blockIdx.y];
  const int R = p.transpose ? p.Ci : p.Co, K = p.transpose ? p.Co : p.Ci;
  const int tr = (R + 31) >> 5, tk = (K + 31) >> 5, ntile = 9 * tr * tk;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  // source strides along the packed copy's row / k dimension, and which of them is the source's contiguous one
  const long s_row = p.transpose ? p.si : p.so, s_k = p.transpose ? p.so : p.si;
  const bool k_fast = s_k <= s_row;  // read with tx along k (no transpose needed) or along rows
  for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int tap = t / (tr * tk), rem = t - tap * (tr * tk);
    const int r0 = (rem / tk) << 5, k0 = (rem % tk) << 5;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const long toff = p.transpose ? (2 - ky) * p.sky + (2 - kx) * p.skx : ky * p.sky + kx * p.skx;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int a8 = ty + 8 * i;
      const int row = r0 + (k_fast ? a8 : tx), k = k0 + (k_fast ? tx : a8);
      float v = 0.f;
      if (row < R && k < K) {
        const long off = (long)row * s_row + (long)k * s_k + toff;
        v = p.src_f32 ? ((const float*)p.src)[off] : (float)(((const bf16*)p.src)[off]);
      }
      if (k_fast) tile[a8][tx] = v; else tile[tx][a8] = v;  // tile[row - r0][k - k0]
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int row = r0 + ty + 8 * i, k = k0 + tx;
      if (row < R && k < K) ((bf16*)p.dst)[((long)row * 9 + tap) * p.k_pad + k] = (bf16)tile[ty + 8 * i][tx];
    }
    __syncthreads();
  }
}

typedef void (*kern_t)(const Args);
template <int TW>
static kern_t pick_mt(int mt) {
  switch (mt) {
    case 1: return conv3x3_igemm_kernel<1, TW>;
    case 2: return conv3x3_igemm_kernel<2, TW>;
    case 3: return conv3x3_igemm_kernel<3, TW>;
    case 4: return conv3x3_igemm_kernel<4, TW>;
    case 5: return conv3x3_igemm_kernel<5, TW>;
    case 6: return conv3x3_igemm_kernel<6, TW>;
    // (no 7-tile instantiation: it spilled 16 VGPR dwords; ocf_conv3x3_plan never asks for it -- 14 tiles run as 3 x 5)
    default: return conv3x3_igemm_kernel<8, TW>;
  }
}

// dynamic LDS above 64 KiB needs the attribute once per kernel (and device): remembered per kernel pointer
static bool lds_attr_done(kern_t k) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair((const void*)k, dev);
  if (done.count(key)) return true;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
  done.insert(key);
  return true;
}

}  // namespace ig
}  // namespace ocf

extern "C" int ocf_conv3x3_plan(int Cout, int B, int H, int W, int dilation, int* mt, int* tw, int* rows_needed) {
  if (!mt || !tw || !rows_needed || Cout <= 0 || B <= 0 || H <= 0 || W <= 0 || dilation < 1) return OCF_EINVAL;
  if (dilation > 16) return OCF_EUNSUPPORTED;
  // patch width: the one of 64 / 32 / 16 that covers the image with the fewest patch pixels (ties: the widest -- wide patches keep the
  // 2 * dilation halo columns per row small, and 64 is the only width whose rows fit the LDS budget at dilation 16)
  int best = 0;
  long best_px = -1;
  for (int w : {64, 32, 16}) {
    const int th = 512 / w;
    if ((th * (w + 2 * dilation) * 64 + 1023) / 1024 > ocf::ig::PIX_ITERS * ocf::ig::NWAVE) continue;
    const long px = (long)((W + w - 1) / w) * w * (long)((H + th - 1) / th) * th;
    if (best_px < 0 || px < best_px) { best_px = px; best = w; }
  }
  if (best_px < 0) return OCF_EUNSUPPORTED;
  *tw = best;
  // output-channel blocks: as few equal blocks of <= 8 tiles of 16 channels as cover Cout; more of them while the launch would leave
  // most of the chip idle (a pyramid level of 8 x 55 x 128 pixels is 112 patches for 256 compute units)
  const long patches = best_px / 512 * B;
  const int tiles = (Cout + 15) / 16;
  int nblk = (tiles + 7) / 8, m = (tiles + nblk - 1) / nblk;
  while (patches * nblk < 192 && m > 2) {
    nblk++;
    m = (tiles + nblk - 1) / nblk;
  }
  if (m == 7) {  // (the 7-tile instantiation is the one that spills registers: 14 tiles run as 3 x 5)
    nblk++;
    m = (tiles + nblk - 1) / nblk;
  }
  nblk = (tiles + m - 1) / m;
  *mt = m;
  *rows_needed = nblk * m * 16;
  return OCF_OK;
}

int ocf_conv3x3_direct_launch(const void* x, long xp, const void* w_packed, int w_rows, int w_k, void* y, long yp, const float* bias, const void* act,
                              long ap, const void* res, long rp, int res_premask, int act_channels, int B, int Cin, int Cout, int H, int W,
                              int dilation, float negative_slope, float mask_slope, void* stream);

extern "C" int ocf_conv3x3_igemm(const ocf_tensor4* x, const void* w_packed, int w_rows, int w_k, const ocf_tensor4* y, const float* bias,
                                 const ocf_tensor4* act, int B, int Cin, int Cout, int H, int W, int dilation, float negative_slope,
                                 float mask_slope, int dtype, int policy, void* stream) {
  return ocf_conv3x3_igemm_ex(x, w_packed, w_rows, w_k, y, bias, act, nullptr, 0, Cout, B, Cin, Cout, H, W, dilation, negative_slope, mask_slope,
                              dtype, policy, stream);
}

extern "C" int ocf_conv3x3_igemm_ex(const ocf_tensor4* x, const void* w_packed, int w_rows, int w_k, const ocf_tensor4* y, const float* bias,
                                    const ocf_tensor4* act, const ocf_tensor4* res, int res_premask, int act_channels, int B, int Cin, int Cout,
                                    int H, int W, int dilation, float negative_slope, float mask_slope, int dtype, int policy, void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !y || !w_packed || !x->data || !y->data) return OCF_EINVAL;
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || dilation < 1 || policy < 0) return OCF_EINVAL;
  if ((res && !res->data) || act_channels < 0 || (res_premask != 0 && res_premask != 1)) return OCF_EINVAL;
  if (act_channels > Cout) act_channels = Cout;
  if (dtype != OCF_BF16 || Cin % 8 != 0 || w_k % 8 != 0 || w_k < Cin) return OCF_EUNSUPPORTED;
  int mt, tw, rows_pad;
  const int st = ocf_conv3x3_plan(Cout, B, H, W, dilation, &mt, &tw, &rows_pad);
  if (st != OCF_OK) return st;
  const int pw = policy & 255;
  const int tune = (policy >> 10) & 7;  // (bit 8192: keep the direct 8-byte epilogue stores, A/B)
  if (pw == 16 || pw == 32 || pw == 64) tw = pw;  // A/B: force the patch width
  else if (pw != 0) return OCF_EINVAL;
  // channels_last views: channel stride 1, rows and images dense in pixels at one pixel pitch (size-1 dimensions carry arbitrary strides)
  auto pitch = [&](const ocf_tensor4* t, int C) { return W > 1 ? t->sw : (H > 1 ? t->sh : (B > 1 ? t->sb : (int64_t)C)); };
  auto cl = [&](const ocf_tensor4* t, int C, int align) {
    const int64_t p = pitch(t, C);
    return (C == 1 || t->sc == 1) && p >= C && (H == 1 || t->sh == p * W) && (B == 1 || t->sb == p * W * H) && (uintptr_t)t->data % align == 0 &&
           (p * 2) % align == 0;
  };
  const int oal = Cout % 4 == 0 ? 8 : 2;
  if (!cl(x, Cin, 16) || !cl(y, Cout, oal) || (act && (!act->data || !cl(act, Cout, oal)))) return OCF_EUNSUPPORTED;
  if ((uintptr_t)w_packed % 16 != 0) return OCF_EUNSUPPORTED;
  // the residual / partial-mask epilogue exists in the 16-byte row form only: whole aligned 8-channel chunks of output, mask and residual
  if (act && act_channels < Cout && !res) return OCF_EUNSUPPORTED;  // (a partial mask exists in the residual epilogue only)
  const bool ex = res != nullptr;
  if (ex && (Cout % 8 != 0 || !cl(y, Cout, 16) || (act && !cl(act, Cout, 16)) || (res && !cl(res, Cout, 16)) || (policy & 8192))) return OCF_EUNSUPPORTED;
  // small maps (fewer than ~8 staged patches): the barrier-free kernel that loads its fragments straight from L2 (conv_direct.hip).
  // Measured under graph replay (profiles/r4_h_bench_conv_small_maps.txt), 8 pairs: ahead of the staged kernel and of the library at
  // 14 x 32 and 7 x 16, behind the staged kernel at 28 x 64 (28.5 vs 16.5 us for 128 -> 128)
  const bool direct = (policy & 256) || (!(policy & 512) && (long)B * H * W <= 4096);
  if (direct)
    return ocf_conv3x3_direct_launch(x->data, pitch(x, Cin), w_packed, w_rows, w_k, y->data, pitch(y, Cout), bias, act ? act->data : nullptr,
                                     act ? pitch(act, Cout) : 0, res ? res->data : nullptr, res ? pitch(res, Cout) : 0, res_premask,
                                     act_channels, B, Cin, Cout, H, W, dilation, negative_slope, mask_slope, stream);
  if (w_rows < rows_pad) return OCF_EINVAL;
  if ((long)B * H * W * (pitch(x, Cin) > pitch(y, Cout) ? pitch(x, Cin) : pitch(y, Cout)) >= (1L << 30)) return OCF_EUNSUPPORTED;  // 32-bit in-image offsets
  const int th = 512 / tw;
  ig::Args a;
  a.x = (const char*)x->data; a.w = (const char*)w_packed; a.y = (char*)y->data; a.bias = bias;
  a.act = act ? (const char*)act->data : nullptr;
  a.res = res ? (const char*)res->data : nullptr;
  a.xp = pitch(x, Cin); a.yp = pitch(y, Cout); a.ap = act ? pitch(act, Cout) : 0; a.rp = res ? pitch(res, Cout) : 0;
  a.act_ch = act_channels; a.res_premask = res_premask;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.wK = w_k; a.d = dilation;
  a.tiles_x = (W + tw - 1) / tw; a.tiles_y = (H + th - 1) / th;
  a.slope = negative_slope; a.mslope = mask_slope; a.tune = tune;
  const int npi = (th * (tw + 2 * dilation) * 64 + 1023) / 1024;
  if (npi > ig::PIX_ITERS * ig::NWAVE) return OCF_EUNSUPPORTED;
  size_t lds = 2 * ((size_t)npi * 1024 + 3 * (size_t)mt * 1024);
  auto al16 = [&](const ocf_tensor4* t, int C) { return (uintptr_t)t->data % 16 == 0 && (pitch(t, C) * 2) % 16 == 0; };
  a.rows_epi = (Cout % 8 == 0 && al16(y, Cout) && (!act || al16(act, Cout)) && (!bias || (uintptr_t)bias % 16 == 0) && !(policy & 8192)) ? 1 : 0;
  if (ex && !a.rows_epi) return OCF_EUNSUPPORTED;  // (an unaligned bias pointer)
  if (a.rows_epi) {
    const size_t slabs = (size_t)ig::NWAVE * 64 * ((size_t)mt * 32 + 16);
    if (slabs > lds) lds = slabs;
  }
  ig::kern_t k = tw == 64 ? ig::pick_mt<64>(mt) : (tw == 32 ? ig::pick_mt<32>(mt) : ig::pick_mt<16>(mt));
  if (!ig::lds_attr_done(k)) return OCF_ELAUNCH;
  LaunchProf lp;
  lp.launch(true, k, dim3((unsigned)(a.tiles_x * a.tiles_y * B), (unsigned)(rows_pad / (mt * 16))), dim3(ig::NTHR), lds, (hipStream_t)stream, a);
  return launch_status();
}

extern "C" int ocf_conv3x3_pack_bytes(int n, int64_t* desc_bytes) {
  if (!desc_bytes || n <= 0) return OCF_EINVAL;
  *desc_bytes = (int64_t)n * (int64_t)sizeof(ocf::ig::PackDesc);
  return OCF_OK;
}

// Host helper: fill descriptor `index` of a HOST array of ocf_conv3x3_pack_bytes(n) bytes.  The caller copies the array to the device
// once (the pointers are stable) and launches ocf_conv3x3_pack_weights every step.
extern "C" int ocf_conv3x3_pack_describe(void* host_descs, int index, const void* src, int64_t so, int64_t si, int64_t sky, int64_t skx, int Co,
                                         int Ci, int src_is_f32, void* dst, int rows_pad, int k_pad, int transpose, int64_t first,
                                         int64_t* next_first) {
  using namespace ocf;
  if (!host_descs || index < 0 || !src || !dst || !next_first || Co <= 0 || Ci <= 0) return OCF_EINVAL;
  const int R = transpose ? Ci : Co, K = transpose ? Co : Ci;
  if (rows_pad < R || k_pad < K) return OCF_EINVAL;
  ig::PackDesc& p = ((ig::PackDesc*)host_descs)[index];
  p.src = src; p.dst = dst; p.so = so; p.si = si; p.sky = sky; p.skx = skx;
  p.Co = Co; p.Ci = Ci; p.rows_pad = rows_pad; p.k_pad = k_pad; p.transpose = transpose ? 1 : 0; p.src_f32 = src_is_f32 ? 1 : 0;
  p.first = first;
  *next_first = first + (int64_t)rows_pad * 9 * k_pad;
  return OCF_OK;
}

extern "C" int ocf_conv3x3_pack_weights(const void* device_descs, int n, int64_t total_elements, void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!device_descs || n <= 0 || total_elements <= 0) return OCF_EINVAL;
  if (n > 65535) return OCF_EUNSUPPORTED;
  LaunchProf lp;
  // (64 blocks per tensor: a tile is a dependent load -> LDS -> store chain of ~2 us, the largest tensors have ~600 of them)
  lp.launch(true, ig::conv3x3_pack_kernel, dim3(64, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
            (const ig::PackDesc*)device_descs, n, (long)total_elements);
  return launch_status();
}
