// The same 3x3 / stride-1 / padding == dilation convolution as conv_igemm.hip for SMALL feature maps (the coarse pyramid levels of the
// estimators: 8 x 28 x 64, 8 x 14 x 32 and 8 x 7 x 16 pixels at the benchmark size; flow_net.py:11-37, flow_occ_net.py:11-67), where the
// 512-pixel patches of the staged kernel leave most of the chip idle (2 .. 28 patches) and every one of its 12 .. 39 stages is a DMA
// round trip + barrier.  Here nothing is staged: a workgroup of four waves owns (16 MT) output channels x (16 NF) pixels, its waves take
// every fourth of the 9 x Cin/32 (tap, channel block) steps, each loading its matrix-core fragments straight from global memory -- the
// maps and weights of these levels sit in L2 -- one 16-byte load per lane and fragment, four steps in flight in registers, MT x NF
// v_mfma_f32_16x16x32_bf16 per step, no barrier until the four partial sums meet in LDS.  (First version: one wave per tile, one step
// in flight: 36 .. 117 dependent L2 round trips per wave, 18.6 us against the library's 13.2 at 8 x 128 -> 128 x 14 x 32.)
// Same packed weights, same epilogue (bias + LeakyReLU, optional activation-derivative mask), same forward / input-gradient duality.
#include "ocf_common.h"

namespace ocf {
namespace dc {
__device__ u32x4 g_zero[4];

struct Args {
  const char* x;
  const char* w;
  char* y;
  const float* bias;
  const char* act;
  const char* res;  // residual (see conv_igemm.hip): Cout % 4 == 0 form only
  long xp, yp, ap, rp;
  int act_ch, res_premask;
  int B, H, W, Cin, Cout, wK, d, nfx, nfrag, ngroup;
  float slope, mslope;
};

__device__ __forceinline__ float bf_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }

template <int MT, int NF>
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(const Args a) {
  constexpr int PF = 4;  // steps in flight per wave
  __shared__ __attribute__((aligned(16))) char red[3 * MT * NF * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = blockIdx.x;  // pixel-fragment group of this workgroup
  const int mrow0 = blockIdx.y * (MT * 16);
  const int g = lane >> 4, l15 = lane & 15;
  const int H = a.H, W = a.W, d = a.d;
  const char* zsrc = (const char*)g_zero;

  // ---- the NF fragments of this workgroup: 16 consecutive pixels of one image row each ------------------------------------------------
  int py[NF], px[NF], pl[NF];  // this lane's pixel (row, column, linear index) per fragment; row = -2^20 for fragments / columns that do not exist
  long pb[NF];                 // element offset of that pixel in x
#pragma unroll
  for (int j = 0; j < NF; j++) {
    const int f = grp * NF + j;
    const int xb = f % a.nfx, row = f / a.nfx;
    const int yy = row % H, b = row / H;
    const int xx = xb * 16 + l15;
    const bool ok = f < a.nfrag && xx < W;
    py[j] = ok ? yy : -(1 << 20);
    px[j] = xx;
    pl[j] = ok ? (b * H + yy) * W + xx : 0;
    pb[j] = (long)pl[j] * a.xp;
  }
  long wrow[MT];  // element offset of this lane's weight row per channel tile
#pragma unroll
  for (int m = 0; m < MT; m++) wrow[m] = (long)(mrow0 + m * 16 + l15) * 9 * a.wK;

  const int nchunk = (a.Cin + 31) >> 5;
  const int nstep = nchunk * 9;
  const int mine = (nstep - wave + 3) >> 2;  // steps wave, wave + 4, ... of this wave
  f32x4 acc[MT][NF];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int j = 0; j < NF; j++) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 fa[PF][MT], fb[PF][NF];
  auto load = [&](int i, int set) __attribute__((always_inline)) {  // this wave's i-th step
    const int s = wave + 4 * i;
    const int c = s / 9, tap = s - 9 * c, ky = tap / 3, kx = tap - 3 * ky;
    const int ch = c * 32 + g * 8;
    const bool live = i < mine;
    const bool x_ok = live && ch < a.Cin, w_ok = live && ch < a.wK;
    const int dy = (ky - 1) * d, dx = (kx - 1) * d;
    const long toff = ((long)dy * W + dx) * a.xp + ch;
#pragma unroll
    for (int m = 0; m < MT; m++) {
      const char* p = w_ok ? a.w + (wrow[m] + (long)tap * a.wK + ch) * 2 : zsrc;
      fa[set][m] = *(const u32x4*)p;
    }
#pragma unroll
    for (int j = 0; j < NF; j++) {
      const int yy = py[j] + dy, xx = px[j] + dx;
      const bool ok = x_ok && yy >= 0 && yy < H && xx >= 0 && xx < W;
      const char* p = ok ? a.x + (pb[j] + toff) * 2 : zsrc;
      fb[set][j] = *(const u32x4*)p;
    }
  };
#pragma unroll
  for (int i = 0; i < PF - 1; i++) load(i, i);
  for (int i0 = 0; i0 < mine; i0 += PF) {
    // PF steps per trip so that the register sets are addressed statically; a step past the wave's last one loads zeros (no branch
    // around a load: a load behind a condition is waited for at the join)
#pragma unroll
    for (int u = 0; u < PF; u++) {
      load(i0 + u + PF - 1, (u + PF - 1) % PF);
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int j = 0; j < NF; j++)
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[u][m]), __builtin_bit_cast(bf16x8, fb[u][j]), acc[m][j], 0, 0, 0);
    }
  }

  // ---- the four partial sums meet in LDS; wave 0 finishes ---------------------------------------------------------------------------------
  if (wave > 0) {
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int j = 0; j < NF; j++) *(f32x4*)(red + (((wave - 1) * MT + m) * NF + j) * 1024 + lane * 16) = acc[m][j];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; w++)
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int j = 0; j < NF; j++) acc[m][j] += *(const f32x4*)(red + ((w * MT + m) * NF + j) * 1024 + lane * 16);

  // ---- epilogue: lane = pixel l15 of fragment j, registers = channels 16 m + 4 g + 0..3 ----------------------------------------------
  const float slope = a.slope, mslope = a.mslope;
  const bool full = (a.Cout & 3) == 0;
  const int cb0 = mrow0 + 4 * g;
#pragma unroll
  for (int m = 0; m < MT; m++) {
    const int cb = cb0 + m * 16;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
      for (int e = 0; e < 4; e++) bv[e] = a.bias[cb + e < a.Cout ? cb + e : 0];
    }
#pragma unroll
    for (int j = 0; j < NF; j++) {
      const bool pix_ok = py[j] >= 0;
      const long pix = pl[j];
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        v[e] = acc[m][j][e] + bv[e];
        v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
      }
      if (full) {
        if (a.res) {
          const int cs = cb < a.Cout ? cb : 0;
          u32x2 mv = {0u, 0u};
          if (a.act) mv = *(const u32x2*)(a.act + (pix * a.ap + cs) * 2);
          const u32x2 rv = *(const u32x2*)(a.res + (pix * a.rp + cs) * 2);
          const float r[4] = {bf_lo(rv[0]), bf_hi(rv[0]), bf_lo(rv[1]), bf_hi(rv[1])};
          const float mm[4] = {bf_lo(mv[0]), bf_hi(mv[0]), bf_lo(mv[1]), bf_hi(mv[1])};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float m = (a.act && cb + e < a.act_ch && !(mm[e] > 0.f)) ? mslope : 1.f;
            v[e] = a.res_premask ? (v[e] + r[e]) * m : v[e] * m + r[e];
          }
        } else if (a.act) {
          const u32x2 mv = *(const u32x2*)(a.act + (pix * a.ap + (cb < a.Cout ? cb : 0)) * 2);
          v[0] *= bf_lo(mv[0]) > 0.f ? 1.f : mslope; v[1] *= bf_hi(mv[0]) > 0.f ? 1.f : mslope;
          v[2] *= bf_lo(mv[1]) > 0.f ? 1.f : mslope; v[3] *= bf_hi(mv[1]) > 0.f ? 1.f : mslope;
        }
        const bf16x2 lo = {(bf16)v[0], (bf16)v[1]}, hi = {(bf16)v[2], (bf16)v[3]};
        const u32x2 o = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
        if (pix_ok && cb < a.Cout) *(u32x2*)(a.y + (pix * a.yp + cb) * 2) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const bool ok = pix_ok && cb + e < a.Cout;
          float o = v[e];
          if (a.act) o *= (float)(((const bf16*)a.act)[pix * a.ap + (ok ? cb + e : 0)]) > 0.f ? 1.f : mslope;
          if (ok) ((bf16*)a.y)[pix * a.yp + cb + e] = (bf16)o;
        }
      }
    }
  }
}

typedef void (*kern_t)(const Args);
template <int NF>
static kern_t pick(int mt) {
  switch (mt) {
    case 1: return conv3x3_direct_kernel<1, NF>;
    case 2: return conv3x3_direct_kernel<2, NF>;
    default: return conv3x3_direct_kernel<4, NF>;
  }
}
}  // namespace dc
}  // namespace ocf

// Launcher used by ocf_conv3x3_igemm for small maps (policy 256 forces it, 512 forbids it); same arguments, same results.
int ocf_conv3x3_direct_launch(const void* x, long xp, const void* w_packed, int w_rows, int w_k, void* y, long yp, const float* bias, const void* act,
                              long ap, const void* res, long rp, int res_premask, int act_channels, int B, int Cin, int Cout, int H, int W,
                              int dilation, float negative_slope, float mask_slope, void* stream) {
  using namespace ocf;
  const long npix = (long)B * H * W;
  dc::Args a;
  a.x = (const char*)x; a.w = (const char*)w_packed; a.y = (char*)y; a.bias = bias; a.act = (const char*)act; a.res = (const char*)res;
  a.xp = xp; a.yp = yp; a.ap = ap; a.rp = rp; a.act_ch = act_channels; a.res_premask = res_premask;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.wK = w_k; a.d = dilation;
  a.nfx = (W + 15) / 16;
  a.nfrag = B * H * a.nfx;
  a.slope = negative_slope; a.mslope = mask_slope;
  // tile per workgroup: as many output channels (<= 64) and pixels (<= 64) as still give the chip ~400 workgroups
  const int tiles = (Cout + 15) / 16;
  int mt = tiles >= 4 ? 4 : (tiles >= 2 ? 2 : 1), nf = 4;
  auto waves = [&](int m, int n) { return (long)((a.nfrag + n - 1) / n) * ((tiles + m - 1) / m); };  // workgroups
  while (waves(mt, nf) < 400 && nf > 1) nf >>= 1;
  while (waves(mt, nf) < 400 && mt > 1) mt >>= 1;
  const int nblk = (tiles + mt - 1) / mt;
  if (w_rows < nblk * mt * 16) return OCF_EINVAL;
  a.ngroup = (a.nfrag + nf - 1) / nf;
  dc::kern_t k = nf == 4 ? dc::pick<4>(mt) : (nf == 2 ? dc::pick<2>(mt) : dc::pick<1>(mt));
  (void)npix;
  LaunchProf lp;
  lp.launch(true, k, dim3((unsigned)a.ngroup, (unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}
