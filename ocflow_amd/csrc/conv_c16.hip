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

namespace ocf {
namespace c16 {
// ---- weight gradient of the same layer: dW[o][i][ky][kx] = sum_px gy[px][o] * x[px + (ky-1, kx-1)][i].  GEMM view: M = o, N = i, K = pixels,
// once per tap -- and a matrix-core fragment wants 8 consecutive PIXELS of one channel while both tensors are pixel-major.  The transpose
// is done by the LDS WRITES: a block owns a strip of 128 columns of one image and walks down its rows; each step stages ONE new row of x and
// one row of gy, every thread scattering the channels of two neighbouring pixels into channel-major LDS rows (32-bit writes).  The three most recent x rows stay in a ring
// (each row is read from memory once).  Wave w contracts pixels 32 w .. 32 w + 31 of the strip: per row one A fragment (gy), nine B
// fragments, nine v_mfma_f32_16x16x32_bf16 into nine accumulators that live for the whole walk; every wave ends by writing its sums to a
// workspace row, which a second launch totals into the fp32 gradient.  The next row's global loads are issued before the
// current row is computed.  (The library: 106 us + a fill and a cast launch at 16 x 218 x 512.)
constexpr int SW = 128, PITCH = SW + 8;  // strip width; gy LDS row pitch in elements (272 bytes: the 16 channel rows of a fragment read tile all 64 banks)
constexpr int XH = 8, XPITCH = SW + 2 * XH + 8;  // x rows: 8 halo pixels on either side (fragment bases stay 16-byte aligned), pitch 304 bytes

struct WArgs {
  const char* x;   // [B][H][W][xp] bf16
  const char* gy;  // [B][H][W][gp] bf16
  float* ws;       // [blocks][4 waves][2304] wave sums
  long xp, gp;
  int B, H, W, strips, ysegs, rows_per_seg;
};

// ONE channel-major copy of every x row (ring of three rows), written as 32-bit pairs of neighbouring pixels; the fragments of the taps
// kx = 0 / 2 (pixel shift -1 / +1) are assembled in registers from the aligned 16-byte read of the centre fragment and the 32-bit words
// before / after it (v_alignbyte).  (Three shifted copies written with ds_write_b16 -- 32 LDS write instructions per thread and row --
// made the kernel LDS-write bound: 80 us.)
__global__ __launch_bounds__(256) void conv3x3_c16_wgrad_kernel(const WArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned short s_gy[16 * PITCH];
  __shared__ __attribute__((aligned(16))) unsigned short s_x[3][16 * XPITCH];  // [ring slot][channel][XH + strip pixel]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kg = lane >> 4;
  int bid = blockIdx.x;
  const int strip = bid % a.strips; bid /= a.strips;
  const int seg = bid % a.ysegs, b = bid / a.ysegs;
  const int px0 = strip * SW, y_begin = seg * a.rows_per_seg, y_end = min(a.H, y_begin + a.rows_per_seg);
  const char* xb = a.x + (long)b * a.H * a.W * a.xp * 2;
  const char* gb = a.gy + (long)b * a.H * a.W * a.gp * 2;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // staging items = (pixel pair, channel half).  x: pairs q = 0..65 <-> pixels px0 - 2 + 2q, +1 (threads 0..131); gy: pairs 0..63 <->
  // pixels px0 + 2q, +1 (threads 128..255)
  const bool does_x = tid < 132, does_g = tid >= 128;
  const int xq = tid >> 1, xhalf = tid & 1, gq = (tid - 128) >> 1, ghalf = tid & 1;
  auto load_px = [&](const char* base, long pitch, int row, int gx, int half) -> u32x4 {
    const bool ok = row >= 0 && row < a.H && gx >= 0 && gx < a.W;
    return ok ? *(const u32x4*)(base + (((long)row * a.W + gx) * pitch + half * 8) * 2) : zero4;
  };
  auto put_pair = [&](unsigned short* dst_row0, int pitch, int q_elem, const u32x4& v0, const u32x4& v1) {
    // channel j of both pixels -> one 32-bit word at [half * 8 + j][q_elem, q_elem + 1]
    const unsigned w0[4] = {v0[0], v0[1], v0[2], v0[3]}, w1[4] = {v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const unsigned lo = (w0[j >> 1] >> ((j & 1) * 16)) & 0xffffu, hi = (w1[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
      *(unsigned*)(dst_row0 + j * pitch + q_elem) = lo | (hi << 16);
    }
  };
  auto stage_x = [&](int slot, const u32x4& v0, const u32x4& v1) {
    if (does_x) put_pair(&s_x[slot][(xhalf * 8) * XPITCH], XPITCH, XH - 2 + 2 * xq, v0, v1);
  };

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: x rows y_begin - 1 and y_begin into the ring
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int row = y_begin - 1 + r, slot = (row + 3) % 3;
    if (does_x) stage_x(slot, load_px(xb, a.xp, row, px0 - 2 + 2 * xq, xhalf), load_px(xb, a.xp, row, px0 - 1 + 2 * xq, xhalf));
  }
  u32x4 nx0 = zero4, nx1 = zero4, ng0 = zero4, ng1 = zero4;
  if (does_x) { nx0 = load_px(xb, a.xp, y_begin + 1, px0 - 2 + 2 * xq, xhalf); nx1 = load_px(xb, a.xp, y_begin + 1, px0 - 1 + 2 * xq, xhalf); }
  if (does_g) { ng0 = load_px(gb, a.gp, y_begin, px0 + 2 * gq, ghalf); ng1 = load_px(gb, a.gp, y_begin, px0 + 2 * gq + 1, ghalf); }
  // (raw barriers: __syncthreads() would also wait for the prefetched global loads)
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  for (int y = y_begin; y < y_end; y++) {
    lds_barrier();  // the previous row's fragment reads are done (gy buffer and the ring slot about to be overwritten)
    stage_x((y + 1) % 3, nx0, nx1);
    if (does_g) put_pair(&s_gy[(ghalf * 8) * PITCH], PITCH, 2 * gq, ng0, ng1);
    if (y + 1 < y_end) {  // next step's loads in flight during this step's matrix work
      if (does_x) { nx0 = load_px(xb, a.xp, y + 2, px0 - 2 + 2 * xq, xhalf); nx1 = load_px(xb, a.xp, y + 2, px0 - 1 + 2 * xq, xhalf); }
      if (does_g) { ng0 = load_px(gb, a.gp, y + 1, px0 + 2 * gq, ghalf); ng1 = load_px(gb, a.gp, y + 1, px0 + 2 * gq + 1, ghalf); }
    }
    lds_barrier();
    const bf16x8 af = *(const bf16x8*)(s_gy + n * PITCH + 32 * wave + 8 * kg);  // A[m = o][k = pixel]
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
      const unsigned short* xr = s_x[(y + ky - 1 + 3) % 3] + n * XPITCH + XH + 32 * wave + 8 * kg;  // B[k = pixel][n = i]
      const u32x4 cur = *(const u32x4*)xr;
      const unsigned prv = *(const unsigned*)(xr - 2), nxt = *(const unsigned*)(xr + 8);
      const u32x4 left = {__builtin_amdgcn_alignbyte(cur[0], prv, 2), __builtin_amdgcn_alignbyte(cur[1], cur[0], 2),
                          __builtin_amdgcn_alignbyte(cur[2], cur[1], 2), __builtin_amdgcn_alignbyte(cur[3], cur[2], 2)};   // x[pixel - 1]
      const u32x4 right = {__builtin_amdgcn_alignbyte(cur[1], cur[0], 2), __builtin_amdgcn_alignbyte(cur[2], cur[1], 2),
                           __builtin_amdgcn_alignbyte(cur[3], cur[2], 2), __builtin_amdgcn_alignbyte(nxt, cur[3], 2)};     // x[pixel + 1]
      acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, left), acc[ky * 3 + 0], 0, 0, 0);
      acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, cur), acc[ky * 3 + 1], 0, 0, 0);
      acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, right), acc[ky * 3 + 2], 0, 0, 0);
    }
  }
  // D[m = o = 4 kg + r][n = i] of this wave -> its workspace row (e = tap * 256 + o * 16 + i); a second, tiny launch totals the rows.
  // (One atomic per (block, weight) measured 134 us for the whole kernel: the atomics of one cache line serialise.)
  float* wrow = a.ws + ((long)blockIdx.x * 4 + wave) * 2304;
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) wrow[t * 256 + (4 * kg + r) * 16 + n] = acc[t][r];
}

// totals of the workspace rows: blockIdx.x = 256 of the 2304 weights (e = tap * 256 + o * 16 + i), blockIdx.y = a slice of the rows; one
// float atomic per (slice, weight).  (One thread per weight walking all 2048 rows was 4 loads in flight x 512 dependent rounds: ~150 us.)
__global__ __launch_bounds__(256) void conv3x3_c16_wgrad_total_kernel(const float* __restrict__ ws, int nrows, float* gw, long so, long si,
                                                                      long sky, long skx) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int per = (nrows + gridDim.y - 1) / gridDim.y, k0 = blockIdx.y * per, k1 = min(nrows, k0 + per);
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = k0;
  for (; k + 7 < k1; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] += ws[(long)(k + u) * 2304 + e];
  }
  for (; k < k1; k++) v[0] += ws[(long)k * 2304 + e];
  const int t = e >> 8, oi = e & 255, o = oi >> 4, i = oi & 15;
  if (k1 > k0) atomicAdd(gw + o * so + i * si + (t / 3) * sky + (t % 3) * skx, ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
}
}  // namespace c16
}  // namespace ocf

static void c16_wgrad_geometry(int B, int H, int W, int* strips, int* ysegs, int* rows_per_seg) {
  *strips = (W + ocf::c16::SW - 1) / ocf::c16::SW;
  // enough blocks for two per CU, at least 16 rows each
  int ys = (int)((512 + (long)B * *strips - 1) / ((long)B * *strips));
  if (ys > (H + 15) / 16) ys = (H + 15) / 16;
  if (ys < 1) ys = 1;
  *rows_per_seg = (H + ys - 1) / ys;
  *ysegs = (H + *rows_per_seg - 1) / *rows_per_seg;
}

extern "C" int ocf_conv3x3_c16_wgrad_workspace_bytes(int B, int H, int W, int64_t* bytes) {
  if (B <= 0 || H <= 0 || W <= 0 || !bytes) return OCF_EINVAL;
  int strips, ysegs, rps;
  c16_wgrad_geometry(B, H, W, &strips, &ysegs, &rps);
  *bytes = (int64_t)B * strips * ysegs * 4 * 2304 * sizeof(float);
  return OCF_OK;
}

extern "C" int ocf_conv3x3_c16_wgrad(const ocf_tensor4* x, const ocf_tensor4* gy, float* gw, int64_t g_so, int64_t g_si, int64_t g_sky,
                                     int64_t g_skx, void* workspace, int B, int H, int W, int dtype, void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !gy || !gw || !workspace || !x->data || !gy->data || B <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if (dtype != OCF_BF16) return OCF_EUNSUPPORTED;
  auto pitch = [&](const ocf_tensor4* t) { return W > 1 ? t->sw : (H > 1 ? t->sh : (B > 1 ? t->sb : (int64_t)16)); };
  auto ok = [&](const ocf_tensor4* t) {
    const int64_t p = pitch(t);
    return t->sc == 1 && p >= 16 && p % 8 == 0 && (H == 1 || t->sh == p * W) && (B == 1 || t->sb == p * W * H) && (uintptr_t)t->data % 16 == 0;
  };
  if (!ok(x) || !ok(gy)) return OCF_EUNSUPPORTED;
  c16::WArgs a;
  a.x = (const char*)x->data; a.gy = (const char*)gy->data;
  a.xp = pitch(x); a.gp = pitch(gy); a.B = B; a.H = H; a.W = W;
  c16_wgrad_geometry(B, H, W, &a.strips, &a.ysegs, &a.rows_per_seg);
  a.ws = (float*)workspace;
  const long blocks = (long)B * a.strips * a.ysegs;
  if (blocks > 0x7fffffffL) return OCF_EUNSUPPORTED;
  LaunchProf lp;
  lp.launch(false, c16::conv3x3_c16_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  lp.launch(true, c16::conv3x3_c16_wgrad_total_kernel, dim3(9, 32), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (int)blocks * 4, gw,
            (long)g_so, (long)g_si, (long)g_sky, (long)g_skx);
  return launch_status();
}

namespace ocf {
namespace c3 {
// ---- the encoder's FIRST layer: Conv2d(3, 16, 3, stride 2, padding 1, bias=False) on the bf16 channels_last image pair batch (6-byte
// pixels).  A tap ROW of an output pixel is 3 neighbouring input pixels x 3 channels = 9 CONTIGUOUS bf16; K is laid out as (row ky, 16
// slots: 9 values + 7 zeros), i.e. 48 -> two 32-wide chunks, so that a lane's 8 K-values are one (unaligned) 16-byte load of a row
// (slots 0-7) or its ninth value alone (slot 8): two loads and two v_mfma_f32_16x16x32_bf16 per 16 output pixels, no LDS for the pixels.
// (The library: igemm_fwd 55 us + a 16 us fill of the output at 16 x 3 x 436 x 1024.)
struct __attribute__((packed, aligned(2))) U16B { u32x4 v; };
struct Args {
  const char* x;  // [B][H][W][3] bf16 dense
  char* y;        // [B][Ho][Wo][yp] bf16
  const void* w;  // [16][3][3][3], element strides so / si / sky / skx, bf16 or fp32
  long so, si, sky, skx, yp;
  int B, H, W, Ho, Wo, w_f32, gpr;
};

__device__ __forceinline__ float wval(const Args& a, int o, int i, int ky, int kx) {
  const long off = o * a.so + i * a.si + ky * a.sky + kx * a.skx;
  const float v = a.w_f32 ? ((const float*)a.w)[off] : (float)((const bf16*)a.w)[off];
  return (float)(bf16)v;
}

__global__ __launch_bounds__(256) void conv3x3s2_c3_kernel(const Args a) {
  __shared__ bf16x8 s_w[2][64];
  const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
  for (int t = threadIdx.x; t < 128; t += 256) {
    const int c = t >> 6, l = t & 63, m = l & 15, g4 = l >> 4;
    const int ky = 2 * c + (g4 >> 1), e0 = (g4 & 1) * 8;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int e = e0 + j;  // slot of row ky: e = kx * 3 + ci for e < 9
      v[j] = (bf16)((ky < 3 && e < 9) ? wval(a, m, e % 3, ky, e / 3) : 0.f);
    }
    s_w[c][l] = v;
  }
  __syncthreads();
  const bf16x8 wa0 = s_w[0][lane], wa1 = s_w[1][lane];
  // a block owns four consecutive output rows of one image (wave = row) and sweeps them left to right: the input rows two neighbouring
  // output rows share come from L1
  const int tiles_y = (a.Ho + 3) >> 2;
  const int b = blockIdx.x / tiles_y, oy = (blockIdx.x - b * tiles_y) * 4 + (threadIdx.x >> 6);
  if (oy >= a.Ho) return;
  const char* xb = a.x + (long)b * a.H * a.W * 6;
  const long row_bytes = (long)a.W * 6;
  for (int xg = 0; xg < a.gpr; xg += 2) {
    u32x4 bq[2][2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int ox = (xg + u < a.gpr ? xg + u : xg) * 16 + n;
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int ky = 2 * c + (kg >> 1);
        const int iy = 2 * oy + ky - 1;
        const long e0 = (long)(2 * ox - 1) * 3 + (kg & 1) * 8;  // first element of this lane's 8 slots within the input row
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ky < 3 && iy >= 0 && iy < a.H && ox < a.Wo) {
          const char* src = xb + (long)iy * row_bytes + e0 * 2;
          if ((kg & 1) == 0) {
            // slots 0..7 = elements (2ox-1)*3 .. +7: whole row interior -> one unaligned 16-byte load; at the borders element by element
            if (ox > 0 && (2 * ox + 1) * 3 + 2 < (long)a.W * 3 + 0 && e0 * 2 + 16 <= row_bytes) {
              v = ((const U16B*)src)->v;
            } else {
              unsigned short ev[8];
#pragma unroll
              for (int j = 0; j < 8; j++) {
                const long e = e0 + j;
                ev[j] = (e >= 0 && e < (long)a.W * 3) ? *(const unsigned short*)(xb + (long)iy * row_bytes + e * 2) : (unsigned short)0;
              }
              v = u32x4{(unsigned)ev[0] | ((unsigned)ev[1] << 16), (unsigned)ev[2] | ((unsigned)ev[3] << 16), (unsigned)ev[4] | ((unsigned)ev[5] << 16),
                        (unsigned)ev[6] | ((unsigned)ev[7] << 16)};
            }
          } else {
            // slot 8 = channel 2 of pixel 2ox+1 (if inside the image); slots 9..15 are padding
            const long e = e0;  // = (2ox-1)*3 + 8
            if (e < (long)a.W * 3) v[0] = *(const unsigned short*)src;
          }
        }
        bq[u][c] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int ox = (xg + u) * 16 + n;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa0, __builtin_bit_cast(bf16x8, bq[u][0]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa1, __builtin_bit_cast(bf16x8, bq[u][1]), acc, 0, 0, 0);
      if (xg + u < a.gpr && ox < a.Wo) {
        const bf16x2 lo = {(bf16)acc[0], (bf16)acc[1]}, hi = {(bf16)acc[2], (bf16)acc[3]};
        *(u32x2*)(a.y + ((((long)b * a.Ho + oy) * a.Wo + ox) * a.yp + 4 * kg) * 2) = u32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
      }
    }
  }
}
}  // namespace c3
}  // namespace ocf

namespace ocf {
namespace c3 {
// ---- weight gradient of the first layer: dW[o][ci][ky][kx] = sum over output pixels of gy[p][o] * x[2p + (ky - 1, kx - 1)][ci].  M = o (16),
// N = (ky, kx, ci) = 27 -> two 16-column tiles, K = output pixels.  As in the 16 -> 16 kernel a block walks down a strip of 128 output
// columns: gy rows are scattered channel-major into LDS (aligned A fragments); the input rows stay PIXEL-major in LDS (6-byte pixels, three
// rows alive, two new ones per step) and a B fragment -- 8 consecutive output pixels of one (ky, kx, ci) column -- is eight ds_read_u16 at
// a 12-byte stride.  Wave sums to a workspace, a second launch totals them.  (The library: 88 us + a fill and a cast launch.)
constexpr int SW = 128, GPITCH = SW + 8, XROW = (2 * SW + 1) * 3, XCH = (XROW * 2 + 15) / 16, XPITCH = XCH * 8;  // 97 chunks of 16 bytes per row

struct WArgs {
  const char* x;   // [B][H][W][3] bf16 dense
  const char* gy;  // [B][Ho][Wo][gp] bf16
  float* ws;       // [blocks][4 waves][512]
  long gp;
  int B, H, W, Ho, Wo, strips, ysegs, rows_per_seg;
};

__global__ __launch_bounds__(256) void conv3x3s2_c3_wgrad_kernel(const WArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned short s_gy[16 * GPITCH];
  __shared__ __attribute__((aligned(16))) unsigned short s_x[3][XPITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kg = lane >> 4;
  int bid = blockIdx.x;
  const int strip = bid % a.strips; bid /= a.strips;
  const int seg = bid % a.ysegs, b = bid / a.ysegs;
  const int px0 = strip * SW, y_begin = seg * a.rows_per_seg, y_end = min(a.Ho, y_begin + a.rows_per_seg);
  const char* xb = a.x + (long)b * a.H * a.W * 6;
  const char* gb = a.gy + (long)b * a.Ho * a.Wo * a.gp * 2;
  const long row_elems = (long)a.W * 3, e_origin = (long)(2 * px0 - 1) * 3;  // LDS element 0 of an input row = this element of the image row
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // input-row chunk `ch` (8 elements) of image row iy: whole chunks inside the row with one unaligned 16-byte load, edge chunks by element
  auto load_xchunk = [&](int iy, int ch) -> u32x4 {
    if (iy < 0 || iy >= a.H) return zero4;
    const long e0 = e_origin + 8L * ch;
    const char* rowp = xb + (long)iy * row_elems * 2;
    if (e0 >= 0 && e0 + 8 <= row_elems) return ((const U16B*)(rowp + e0 * 2))->v;
    unsigned short ev[8];
#pragma unroll
    for (int j = 0; j < 8; j++) ev[j] = (e0 + j >= 0 && e0 + j < row_elems) ? *(const unsigned short*)(rowp + (e0 + j) * 2) : (unsigned short)0;
    return u32x4{(unsigned)ev[0] | ((unsigned)ev[1] << 16), (unsigned)ev[2] | ((unsigned)ev[3] << 16), (unsigned)ev[4] | ((unsigned)ev[5] << 16),
                 (unsigned)ev[6] | ((unsigned)ev[7] << 16)};
  };
  auto load_g = [&](int oy, int ox, int half) -> u32x4 {
    const bool ok = oy < a.Ho && ox < a.Wo;
    return ok ? *(const u32x4*)(gb + (((long)oy * a.Wo + ox) * a.gp + half * 8) * 2) : zero4;
  };
  // staging roles: x = 2 rows x 97 chunks (threads 0..193: row tid / 97, chunk tid % 97); gy = 64 pixel pairs x 2 halves (threads 128..255)
  const bool does_x = tid < 2 * XCH, does_g = tid >= 128;
  const int xr_i = tid / XCH, xch = tid - xr_i * XCH, gq = (tid - 128) >> 1, ghalf = tid & 1;

  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  // this lane's two B columns: nn = 16 T + n -> (ky, e = kx * 3 + ci)
  int bky[2], be[2];
#pragma unroll
  for (int T = 0; T < 2; T++) {
    const int nn = 16 * T + n;
    bky[T] = nn < 27 ? nn / 9 : -1;
    be[T] = nn < 27 ? nn % 9 : 0;
  }
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // prologue: input row 2 y_begin - 1 (the previous step's second row)
  if (tid < XCH) *(u32x4*)(s_x[(2 * y_begin - 1 + 3) % 3] + 8 * tid) = load_xchunk(2 * y_begin - 1, tid);
  u32x4 nx = zero4, ng0 = zero4, ng1 = zero4;
  if (does_x) nx = load_xchunk(2 * y_begin + xr_i, xch);
  if (does_g) { ng0 = load_g(y_begin, px0 + 2 * gq, ghalf); ng1 = load_g(y_begin, px0 + 2 * gq + 1, ghalf); }
  for (int oy = y_begin; oy < y_end; oy++) {
    lds_barrier();
    if (does_x) *(u32x4*)(s_x[(2 * oy + xr_i) % 3] + 8 * xch) = nx;
    if (does_g) {
      const unsigned w0[4] = {ng0[0], ng0[1], ng0[2], ng0[3]}, w1[4] = {ng1[0], ng1[1], ng1[2], ng1[3]};
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const unsigned lo = (w0[j >> 1] >> ((j & 1) * 16)) & 0xffffu, hi = (w1[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
        *(unsigned*)(s_gy + (ghalf * 8 + j) * GPITCH + 2 * gq) = lo | (hi << 16);
      }
    }
    if (oy + 1 < y_end) {
      if (does_x) nx = load_xchunk(2 * (oy + 1) + xr_i, xch);
      if (does_g) { ng0 = load_g(oy + 1, px0 + 2 * gq, ghalf); ng1 = load_g(oy + 1, px0 + 2 * gq + 1, ghalf); }
    }
    lds_barrier();
    const bf16x8 af = *(const bf16x8*)(s_gy + n * GPITCH + 32 * wave + 8 * kg);  // A[m = o][k = output pixel]
#pragma unroll
    for (int T = 0; T < 2; T++) {
      u32x4 bv = zero4;
      if (bky[T] >= 0) {
        const unsigned short* xr = s_x[(2 * oy + bky[T] - 1 + 3) % 3] + 6 * (32 * wave + 8 * kg) + be[T];  // B[k][nn]: 6 elements per output pixel
        unsigned short ev[8];
#pragma unroll
        for (int j = 0; j < 8; j++) ev[j] = xr[6 * j];
        bv = u32x4{(unsigned)ev[0] | ((unsigned)ev[1] << 16), (unsigned)ev[2] | ((unsigned)ev[3] << 16), (unsigned)ev[4] | ((unsigned)ev[5] << 16),
                   (unsigned)ev[6] | ((unsigned)ev[7] << 16)};
      }
      acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, bv), acc[T], 0, 0, 0);
    }
  }
  float* wrow = a.ws + ((long)blockIdx.x * 4 + wave) * 512;
#pragma unroll
  for (int T = 0; T < 2; T++)
#pragma unroll
    for (int r = 0; r < 4; r++) wrow[T * 256 + (4 * kg + r) * 16 + n] = acc[T][r];
}

// totals: thread e = T * 256 + o * 16 + n of 512, blockIdx.y = a slice of the rows
__global__ __launch_bounds__(256) void conv3x3s2_c3_wgrad_total_kernel(const float* __restrict__ ws, int nrows, float* gw, long so, long si,
                                                                       long sky, long skx) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int per = (nrows + gridDim.y - 1) / gridDim.y, k0 = blockIdx.y * per, k1 = min(nrows, k0 + per);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  int k = k0;
  for (; k + 3 < k1; k += 4) {
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] += ws[(long)(k + u) * 512 + e];
  }
  for (; k < k1; k++) v[0] += ws[(long)k * 512 + e];
  const int T = e >> 8, o = (e & 255) >> 4, nn = 16 * T + (e & 15);
  if (nn < 27 && k1 > k0) {
    const int ky = nn / 9, ee = nn % 9, kx = ee / 3, ci = ee % 3;
    atomicAdd(gw + o * so + ci * si + ky * sky + kx * skx, (v[0] + v[1]) + (v[2] + v[3]));
  }
}
}  // namespace c3
}  // namespace ocf

static void c3_wgrad_geometry(int B, int Ho, int Wo, int* strips, int* ysegs, int* rows_per_seg) {
  *strips = (Wo + ocf::c3::SW - 1) / ocf::c3::SW;
  int ys = (int)((512 + (long)B * *strips - 1) / ((long)B * *strips));
  if (ys > (Ho + 15) / 16) ys = (Ho + 15) / 16;
  if (ys < 1) ys = 1;
  *rows_per_seg = (Ho + ys - 1) / ys;
  *ysegs = (Ho + *rows_per_seg - 1) / *rows_per_seg;
}

extern "C" int ocf_conv3x3s2_c3_wgrad_workspace_bytes(int B, int H, int W, int64_t* bytes) {
  if (B <= 0 || H <= 0 || W <= 0 || !bytes) return OCF_EINVAL;
  int strips, ysegs, rps;
  c3_wgrad_geometry(B, (H - 1) / 2 + 1, (W - 1) / 2 + 1, &strips, &ysegs, &rps);
  *bytes = (int64_t)B * strips * ysegs * 4 * 512 * sizeof(float);
  return OCF_OK;
}

extern "C" int ocf_conv3x3s2_c3_wgrad(const ocf_tensor4* x, const ocf_tensor4* gy, float* gw, int64_t g_so, int64_t g_si, int64_t g_sky,
                                      int64_t g_skx, void* workspace, int B, int H, int W, int dtype, void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !gy || !gw || !workspace || !x->data || !gy->data || B <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if (dtype != OCF_BF16) return OCF_EUNSUPPORTED;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if (!(x->sc == 1 && (W == 1 || x->sw == 3) && (H == 1 || x->sh == 3L * W) && (B == 1 || x->sb == 3L * W * H))) return OCF_EUNSUPPORTED;
  const int64_t gp = Wo > 1 ? gy->sw : (Ho > 1 ? gy->sh : (B > 1 ? gy->sb : (int64_t)16));
  if (!(gy->sc == 1 && gp >= 16 && gp % 8 == 0 && (Ho == 1 || gy->sh == gp * Wo) && (B == 1 || gy->sb == gp * Wo * Ho) && (uintptr_t)gy->data % 16 == 0))
    return OCF_EUNSUPPORTED;
  c3::WArgs a;
  a.x = (const char*)x->data; a.gy = (const char*)gy->data; a.ws = (float*)workspace; a.gp = gp;
  a.B = B; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  c3_wgrad_geometry(B, Ho, Wo, &a.strips, &a.ysegs, &a.rows_per_seg);
  const long blocks = (long)B * a.strips * a.ysegs;
  if (blocks > 0x7fffffffL) return OCF_EUNSUPPORTED;
  LaunchProf lp;
  lp.launch(false, c3::conv3x3s2_c3_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  lp.launch(true, c3::conv3x3s2_c3_wgrad_total_kernel, dim3(2, 32), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (int)blocks * 4, gw,
            (long)g_so, (long)g_si, (long)g_sky, (long)g_skx);
  return launch_status();
}

extern "C" int ocf_conv3x3s2_c3(const ocf_tensor4* x, const void* w, int64_t w_so, int64_t w_si, int64_t w_sky, int64_t w_skx, int w_is_f32,
                                const ocf_tensor4* y, int B, int H, int W, int dtype, void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !y || !w || !x->data || !y->data || B <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if (dtype != OCF_BF16) return OCF_EUNSUPPORTED;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;  // floor((H + 2 - 3) / 2) + 1
  // x: dense channels_last [B,3,H,W]; y: pixel-major [B,16,Ho,Wo] with 8-byte aligned pixels
  if (!(x->sc == 1 && (W == 1 || x->sw == 3) && (H == 1 || x->sh == 3L * W) && (B == 1 || x->sb == 3L * W * H))) return OCF_EUNSUPPORTED;
  const int64_t yp = Wo > 1 ? y->sw : (Ho > 1 ? y->sh : (B > 1 ? y->sb : (int64_t)16));
  if (!(y->sc == 1 && yp >= 16 && yp % 4 == 0 && (Ho == 1 || y->sh == yp * Wo) && (B == 1 || y->sb == yp * Wo * Ho) && (uintptr_t)y->data % 8 == 0))
    return OCF_EUNSUPPORTED;
  c3::Args a;
  a.x = (const char*)x->data; a.y = (char*)y->data; a.w = w; a.so = w_so; a.si = w_si; a.sky = w_sky; a.skx = w_skx; a.yp = yp;
  a.B = B; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.w_f32 = w_is_f32; a.gpr = (Wo + 15) / 16;
  const long blocks = (long)B * ((Ho + 3) / 4);
  if (blocks > 0x7fffffffL || (long)H * W * 6 >= (1L << 40)) return OCF_EUNSUPPORTED;
  LaunchProf lp;
  lp.launch(true, c3::conv3x3s2_c3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

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
