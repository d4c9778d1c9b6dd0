// 3x3 / stride-1 / padding-1 convolution with 16 input and 16 output channels on full-resolution maps: `layer1.double_conv.3` of the
// feature pyramid (feature_pyramid_net.py:21-33: Conv2d(16, 16, 3, 1, 1, bias=False) at 218 x 512, both images of 8 pairs = 1.8 M pixels),
// forward and input gradient.  57 MB in, 57 MB out, 8 GFLOP: memory-bound by two orders of magnitude, and too narrow for the staged
// implicit-GEMM kernel (half of every 64-byte LDS pixel row would be padding).  The library's kernels for it take 54 us forward and
// 57 + 17 us (zero-fill of the output for its split accumulation) for the input gradient.
//
// One matrix-core instruction = 16 output pixels x 16 output channels x 32 K-values, K = (tap, channel) in memory order: a K-chunk is TWO
// taps x 16 channels, and the 8 consecutive K-values a lane feeds (lane = (pixel n = l & 15, k-group kg = l >> 4)) are HALF A PIXEL of one
// tap: 16 contiguous bytes of the channels_last input -- the B fragments come straight from global memory (L1 / L2 serve the 9-fold reuse
// of every input pixel), no LDS, no barrier.  Five chunks cover the nine taps (the tenth half-chunk is zero).  A fragments (weights, 5 x 16
// bytes per lane) are loaded once per wave.  The accumulator leaves the lane as 4 consecutive output channels of its pixel: one 8-byte store.
// Input gradient = the same kernel on grad_out with the weight read transposed and tap-flipped.
#include "ocf_common.h"

namespace ocf {
namespace c16 {
struct Args {
  const char* x;   // [B][H][W][xp] bf16
  char* y;         // [B][H][W][yp] bf16
  const void* w;   // [16][16][3][3] (out, in, ky, kx), element strides so / si / sky / skx, bf16 or fp32
  long so, si, sky, skx, xp, yp;
  int B, H, W, w_f32, transpose;
  long ngroups;    // groups of 16 consecutive pixels of a row
  int gpr;         // groups per row = ceil(W / 16)
  int xsegs, gps;  // column segments per row, groups per segment (even)
};

__device__ __forceinline__ float wval(const Args& a, int o, int i, int ky, int kx) {
  const long off = o * a.so + i * a.si + ky * a.sky + kx * a.skx;
  const float v = a.w_f32 ? ((const float*)a.w)[off] : (float)((const bf16*)a.w)[off];
  return (float)(bf16)v;
}

__device__ u32x4 g_zero_c16[1];

__global__ __launch_bounds__(256) void conv3x3_c16_kernel(const Args a) {
  const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
  // ---- weights: fragment (chunk c, lane l) = row m = l & 15 (output channel of this pass), tap 2c + (l >> 5), channels 8 ((l >> 4) & 1)
  // .. + 7, built once per block in LDS by 320 (chunk, lane) items and read back as one ds_read_b128 per chunk
  __shared__ bf16x8 s_w[5][64];
  for (int t = threadIdx.x; t < 320; t += 256) {
    const int c = t >> 6, l = t & 63, m = l & 15, g4 = l >> 4;
    const int tap = 2 * c + (g4 >> 1), k0 = (g4 & 1) * 8;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float f = 0.f;
      if (tap < 9) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        // forward: y[o] += w[o][i][ky][kx] x[i](+ky-1, +kx-1); input gradient: gx[i] += w[o][i][2-ky][2-kx] gy[o](+ky-1, +kx-1)
        f = a.transpose ? wval(a, k0 + j, m, 2 - ky, 2 - kx) : wval(a, m, k0 + j, ky, kx);
      }
      v[j] = (bf16)f;
    }
    s_w[c][l] = v;
  }
  __syncthreads();
  bf16x8 wa[5];
#pragma unroll
  for (int c = 0; c < 5; c++) wa[c] = s_w[c][lane];
  const u32x4* zp = g_zero_c16;
  // A block owns FOUR consecutive rows of one image (wave w = row 4 t + w) and sweeps them left to right in lockstep: the rows a wave
  // reads above and below its own are the neighbouring waves' own rows, so the 3-fold vertical reuse is served by the CU's L1 instead of
  // L2 (with the groups dealt out round-robin over the whole grid every tap load went to L2: 515 MB, 52 us for this layer).
  const int tiles_y = (a.H + 3) >> 2;
  const int seg = blockIdx.x % a.xsegs, bt = blockIdx.x / a.xsegs;  // column segment (a few per row: enough waves to cover the load latency)
  const int b = bt / tiles_y, y = (bt - b * tiles_y) * 4 + (threadIdx.x >> 6);
  if (y >= a.H) return;
  const char* xb = a.x + (long)b * a.H * a.W * a.xp * 2;
  const int xg_end = min(a.gpr, (seg + 1) * a.gps);
  for (int xg = seg * a.gps; xg < xg_end; xg += 2) {
    // two groups per iteration: ten 16-byte loads in flight per lane before the first matrix instruction
    u32x4 bq[2][5];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int px = (xg + u < xg_end ? xg + u : xg) * 16 + n;
#pragma unroll
      for (int c = 0; c < 5; c++) {
        const int tap = 2 * c + (kg >> 1);
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int sy = y + ky - 1, sx = px + kx - 1;
        const bool ok = tap < 9 && sy >= 0 && sy < a.H && sx >= 0 && sx < a.W;
        const u32x4* src = ok ? (const u32x4*)(xb + (((long)sy * a.W + sx) * a.xp + (kg & 1) * 8) * 2) : zp;
        bq[u][c] = *src;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int px = (xg + u) * 16 + n;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 5; c++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c], __builtin_bit_cast(bf16x8, bq[u][c]), acc, 0, 0, 0);
      if (px < a.W && xg + u < xg_end) {
        const bf16x2 lo = {(bf16)acc[0], (bf16)acc[1]}, hi = {(bf16)acc[2], (bf16)acc[3]};
        *(u32x2*)(a.y + ((((long)b * a.H + y) * a.W + px) * a.yp + 4 * kg) * 2) = u32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
      }
    }
  }
}
}  // namespace c16
}  // namespace ocf

extern "C" int ocf_conv3x3_c16(const ocf_tensor4* x, const void* w, int64_t w_so, int64_t w_si, int64_t w_sky, int64_t w_skx, int w_is_f32,
                               const ocf_tensor4* y, int B, int H, int W, int transpose, int dtype, void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !y || !w || !x->data || !y->data || B <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if (dtype != OCF_BF16) return OCF_EUNSUPPORTED;
  auto pitch = [&](const ocf_tensor4* t) { return W > 1 ? t->sw : (H > 1 ? t->sh : (B > 1 ? t->sb : (int64_t)16)); };
  auto ok = [&](const ocf_tensor4* t) {
    const int64_t p = pitch(t);
    return t->sc == 1 && p >= 16 && p % 8 == 0 && (H == 1 || t->sh == p * W) && (B == 1 || t->sb == p * W * H) && (uintptr_t)t->data % 16 == 0;
  };
  if (!ok(x) || !ok(y)) return OCF_EUNSUPPORTED;
  c16::Args a;
  a.x = (const char*)x->data; a.y = (char*)y->data; a.w = w; a.so = w_so; a.si = w_si; a.sky = w_sky; a.skx = w_skx;
  a.xp = pitch(x); a.yp = pitch(y); a.B = B; a.H = H; a.W = W; a.w_f32 = w_is_f32; a.transpose = transpose;
  a.gpr = (W + 15) / 16;
  a.ngroups = (long)B * H * a.gpr;
  a.xsegs = a.gpr >= 32 ? 2 : 1;  // >= 16 groups per segment
  a.gps = ((a.gpr + a.xsegs - 1) / a.xsegs + 1) & ~1;
  a.xsegs = (a.gpr + a.gps - 1) / a.gps;
  const long blocks = (long)B * ((H + 3) / 4) * a.xsegs;
  if (blocks > 0x7fffffffL) return OCF_EUNSUPPORTED;
  LaunchProf lp;
  lp.launch(true, c16::conv3x3_c16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}
