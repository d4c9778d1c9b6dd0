// Forward of the PWC-style decoders' feature up-sampling layers: nn.ConvTranspose2d(Ci, Co <= 2, kernel 4, stride 2, padding 1) with hundreds of
// input channels -- `upfeat6..3` of cost_volume_flow_net.py:63-65,182-183,197-198 (Ci = 529 .. 661 -> 2) and the pwoc twin's -- on a channel
// RANGE of the level's dense-block buffer (bf16 pixel-major, any pixel pitch).
//
//   y[b,co,oy,ox] = bias[co] + sum_{ci,ky,kx : oy = 2 iy - 1 + ky, ox = 2 ix - 1 + kx} x[b,ci,iy,ix] * w[ci,co,ky,kx]
//
// The library serves these through its generic backward-data implicit GEMM behind a strided copy of the 529 .. 661 channel slice and a fill /
// cast pair: 14 / 23 / 54 / 176 us at the four levels of configs[2] (profiles/r5_cfg3_step_breakdown.txt), twice per step.  Three launches here:
//
//   pack    the weight in B-fragment order, bf16 (fp32 masters rounded as autocast would), once per call
//   gemm    contrib[p][n] = sum_ci x[p][ci] * w[ci][n],  n = co * 16 + ky * 4 + kx  (a [pixels x Ci] x [Ci x 16 Co] product on the matrix
//           cores: v_mfma_f32_32x32x16_bf16, one workgroup = 32 input pixels x all 16 Co columns, the k-steps split over its four waves;
//           A fragments straight from global memory -- a lane's 8 k-values are 16 contiguous bytes of its pixel --, B fragments from the pack);
//           the features are read ONCE (26 MB at the largest level)
//   gather  y[b, co, oy, ox] = bias + the (up to) four contrib entries whose tap lands on (oy, ox): col2im without atomics.
#include "ocf_common.h"

namespace ocf {
namespace d44 {
constexpr int NT = 256, PXW = 32;  // 4 waves per workgroup, 32 pixels per wave
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Args {
  const bf16* x;   // [B][H][W][xp], channels [0, Ci) used; xp >= kpad
  const void* w;   // [Ci][Co][4][4], element strides wsi / wso / wsy / wsx, fp32 or bf16
  float* contrib;  // [B H W][16 Co]
  long xp, wsi, wso, wsy, wsx;
  long npix;
  int Ci, Co, kpad, w_f32;
};

// the weight in MFMA B-fragment order, rounded to bf16 (fp32 masters: the rounding autocast applies), zero beyond Ci and beyond 16 Co columns:
//   packed[((k * 2 + kg) * 32 + n) * 8 + e] = w[ci = k * 16 + kg * 8 + e][n]      (one k-step = 1 KB, a wave's fragment load is contiguous)
// one thread per element, once per call, into the workspace
__global__ __launch_bounds__(256) void deconv4x4_pack_kernel(const Args a, bf16* packed) {
  const int N = 16 * a.Co;
  const int i = blockIdx.x * 256 + threadIdx.x;  // = ci * 32 + n: consecutive threads read consecutive elements of a dense weight
  if (i >= 32 * a.kpad) return;
  const int ci = i >> 5, n = i & 31;
  float v = 0.f;
  if (ci < a.Ci && n < N) {
    const long off = ci * a.wsi + (n >> 4) * a.wso + ((n >> 2) & 3) * a.wsy + (n & 3) * a.wsx;
    v = a.w_f32 ? ((const float*)a.w)[off] : (float)((const bf16*)a.w)[off];
  }
  packed[(((ci >> 4) * 2 + ((ci >> 3) & 1)) * 32 + n) * 8 + (ci & 7)] = (bf16)v;
}

// workgroup = 32 pixels, its four waves split the k-steps: every wave issues ALL its loads (<= KQ A fragments from the pixels' rows + <= KQ B
// fragments from the packed weight, both straight from global memory) before the first MFMA -- one memory latency per workgroup instead of
// one per prefetch group (the layer's levels have 12 .. 768 such tiles: few waves, so latency, not bandwidth, is what a k-loop pays) --,
// then the four partial tiles meet in LDS and every wave stores a quarter of the rows
template <int KQ>
__global__ __launch_bounds__(NT) void deconv4x4_gemm_kernel(const Args a, const bf16* __restrict__ packed) {
  __shared__ float part[4][16][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = 16 * a.Co;
  const long p0 = (long)blockIdx.x * PXW;
  const int m = lane & 31, kg = lane >> 5;
  const long p = p0 + m;
  const bf16* xr = a.x + (p < a.npix ? p : 0) * a.xp + kg * 8;
  const bf16* wr = packed + lane * 8;
  const int ksteps = a.kpad >> 4, kq = (ksteps + 3) >> 2;  // (kq <= KQ: the launcher picks KQ)
  const int kb = wave * kq, cnt = min(kq, ksteps - kb);    // this wave's k-steps [kb, kb + cnt); cnt may be <= 0 for the last wave
  const int last = ksteps - 1;
  bf16x8 fa[KQ], fb[KQ];
#pragma unroll
  for (int j = 0; j < KQ; j++) {  // (indices clamped into the row: a redundant load instead of a branch)
    const int k = min(kb + j, last);
    fa[j] = *(const bf16x8*)(xr + k * 16);
    fb[j] = *(const bf16x8*)(wr + k * 512);
  }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
  for (int j = 0; j < KQ; j++)
    if (j < cnt) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[j], fb[j], acc, 0, 0, 0);  // (wave-uniform)
#pragma unroll
  for (int e = 0; e < 16; e++) part[wave][e][lane] = acc[e];
  __syncthreads();
  // D[row][col]: register e of lane l = row (e & 3) + 8 (e >> 2) + 4 (l >> 5), col l & 31; wave w sums and stores registers 4 w .. 4 w + 3
  // (32 lanes write one pixel's N consecutive floats)
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int e = wave * 4 + q;
    const float v = part[0][e][lane] + part[1][e][lane] + part[2][e][lane] + part[3][e][lane];
    const int row = q + 8 * wave + 4 * kg;
    if (m < N && p0 + row < a.npix) a.contrib[(p0 + row) * N + m] = v;
  }
}

struct GArgs {
  const float* contrib;
  const float* bias;
  bf16* y;  // [B][2H][2W][yp]
  long yp;
  int B, H, W, Co;
};

// thread = one output pixel: oy = 2 iy - 1 + ky  <=>  even oy = 2 m: (iy, ky) in {(m, 1), (m - 1, 3)}; odd oy = 2 m + 1: {(m + 1, 0), (m, 2)}
template <int CO>
__global__ __launch_bounds__(256) void deconv4x4_gather_kernel(const GArgs g) {
  const int Ho = 2 * g.H, Wo = 2 * g.W;
  const long nout = (long)g.B * Ho * Wo;
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= nout) return;
  const int ox = (int)(o % Wo);
  const long r = o / Wo;
  const int oy = (int)(r % Ho), b = (int)(r / Ho);
  const int my = oy >> 1, mx = ox >> 1;
  const int iy0 = (oy & 1) ? my + 1 : my, ky0 = (oy & 1) ? 0 : 1, iy1 = (oy & 1) ? my : my - 1, ky1 = (oy & 1) ? 2 : 3;
  const int ix0 = (ox & 1) ? mx + 1 : mx, kx0 = (ox & 1) ? 0 : 1, ix1 = (ox & 1) ? mx : mx - 1, kx1 = (ox & 1) ? 2 : 3;
  float acc[CO];
#pragma unroll
  for (int co = 0; co < CO; co++) acc[co] = g.bias ? g.bias[co] : 0.f;
  const int iys[2] = {iy0, iy1}, kys[2] = {ky0, ky1}, ixs[2] = {ix0, ix1}, kxs[2] = {kx0, kx1};
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const bool ok = iys[a] >= 0 && iys[a] < g.H && ixs[c] >= 0 && ixs[c] < g.W;
      const long p = ((long)b * g.H + (ok ? iys[a] : 0)) * g.W + (ok ? ixs[c] : 0);
#pragma unroll
      for (int co = 0; co < CO; co++) {
        const float v = g.contrib[p * (16 * CO) + co * 16 + kys[a] * 4 + kxs[c]];
        acc[co] += ok ? v : 0.f;
      }
    }
  bf16* yr = g.y + o * g.yp;
#pragma unroll
  for (int co = 0; co < CO; co++) yr[co] = (bf16)acc[co];
}
}  // namespace d44
}  // namespace ocf

static bool d44_pixel_major(const ocf_tensor4* t, int B, int C, int H, int W, long* pitch) {
  const int64_t p = W > 1 ? t->sw : (H > 1 ? t->sh : (B > 1 ? t->sb : (int64_t)C));
  *pitch = p;
  return (C == 1 || t->sc == 1) && p >= C && (H == 1 || t->sh == p * W) && (B == 1 || t->sb == p * W * H);
}

static int64_t d44_contrib_bytes(int B, int Co, int H, int W) { return ((int64_t)B * H * W * 16 * Co * 4 + 255) / 256 * 256; }

extern "C" int ocf_deconv4x4s2_fwd_workspace_bytes(int B, int Ci, int Co, int H, int W, int64_t* bytes) {
  if (!bytes || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if (Co > 2) return OCF_EUNSUPPORTED;
  *bytes = d44_contrib_bytes(B, Co, H, W) + (int64_t)32 * ((Ci + 15) / 16 * 16) * 2;  // contributions + the packed weight
  return OCF_OK;
}

extern "C" int ocf_deconv4x4s2_fwd(const ocf_tensor4* x, const void* w, int64_t w_si, int64_t w_so, int64_t w_sy, int64_t w_sx, int w_is_f32,
                                   const float* bias, const ocf_tensor4* y, void* workspace, int B, int Ci, int Co, int H, int W, int dtype,
                                   void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !y || !w || !workspace || !x->data || !y->data) return OCF_EINVAL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if (dtype != OCF_BF16 || Co > 2) return OCF_EUNSUPPORTED;
  long xp, yp;
  if (!d44_pixel_major(x, B, Ci, H, W, &xp) || !d44_pixel_major(y, B, Co, 2 * H, 2 * W, &yp)) return OCF_EUNSUPPORTED;
  const int kpad = (Ci + 15) / 16 * 16;
  // the A fragments are 16-byte loads of whole 8-channel groups up to kpad: they must stay inside the pixel's own pitch (channels beyond Ci
  // meet zero weights, but they have to be THERE -- the alignment channels of the buffer the slice comes from)
  if (xp < kpad || (xp * 2) % 16 != 0 || (uintptr_t)x->data % 16 != 0 || (uintptr_t)workspace % 16 != 0) return OCF_EUNSUPPORTED;
  const long npix = (long)B * H * W;
  if (npix * xp >= (1L << 31) || npix * 4 >= (1L << 31)) return OCF_EUNSUPPORTED;
  d44::Args a;
  a.x = (const bf16*)x->data; a.w = w; a.contrib = (float*)workspace;
  a.xp = xp; a.wsi = w_si; a.wso = w_so; a.wsy = w_sy; a.wsx = w_sx;
  a.npix = npix; a.Ci = Ci; a.Co = Co; a.kpad = kpad; a.w_f32 = w_is_f32;
  const int ksteps = kpad / 16, kq = (ksteps + 3) / 4;
  if (kq > 16) return OCF_EUNSUPPORTED;  // (Ci <= 1024)
  const long tiles = (npix + d44::PXW - 1) / d44::PXW;
  bf16* packed = (bf16*)((char*)workspace + d44_contrib_bytes(B, Co, H, W));
  LaunchProf lp;
  lp.launch(false, d44::deconv4x4_pack_kernel, dim3((unsigned)((32 * kpad + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, packed);
  if (kq <= 8) lp.launch(false, d44::deconv4x4_gemm_kernel<8>, dim3((unsigned)tiles), dim3(d44::NT), 0, (hipStream_t)stream, a, (const bf16*)packed);
  else if (kq <= 12) lp.launch(false, d44::deconv4x4_gemm_kernel<12>, dim3((unsigned)tiles), dim3(d44::NT), 0, (hipStream_t)stream, a, (const bf16*)packed);
  else lp.launch(false, d44::deconv4x4_gemm_kernel<16>, dim3((unsigned)tiles), dim3(d44::NT), 0, (hipStream_t)stream, a, (const bf16*)packed);
  d44::GArgs g;
  g.contrib = a.contrib; g.bias = bias; g.y = (bf16*)y->data; g.yp = yp; g.B = B; g.H = H; g.W = W; g.Co = Co;
  const long nout = npix * 4, gblocks = (nout + 255) / 256;
  if (Co == 1) lp.launch(true, d44::deconv4x4_gather_kernel<1>, dim3((unsigned)gblocks), dim3(256), 0, (hipStream_t)stream, g);
  else lp.launch(true, d44::deconv4x4_gather_kernel<2>, dim3((unsigned)gblocks), dim3(256), 0, (hipStream_t)stream, g);
  return launch_status();
}
