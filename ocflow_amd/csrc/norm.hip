// K3: normalize_features as kernels (reference models/networks/correlation_layer.py:59-82).
//   mu  = mean over {maps s, samples b} of the per-sample mean over (C,H,W),   var = mean over {s, b} of the per-sample biased variance,
//   y_s = (x_s - mu) / sqrt(var + 1e-16)                                         (ONE scalar pair for both maps and the whole batch, F6)
// and its exact backward through mu and var.  Round 2 did this with `ocf_moments` + ~10 elementwise / reduction passes of the framework
// per direction; here: forward = moments (per map) + ONE pass per map, backward = ONE reduction pass + ONE elementwise pass per map.  The
// scalar algebra (mu, var, the three backward coefficients) is redone by every block from the S*B*2 moment sums -- a few hundred flops
// instead of a chain of tiny framework launches.
//   backward:  G = sum g,  D = sum g * (x - mu)  over all maps;   dmu = -inv_std * G / (S B n),  dvar = -0.5 * D * inv_std^3,
//              gx_s = g * inv_std + dmu + 2 dvar / (S B n) * (x_s - m1[s][b])          (m1 = per-sample mean, n = C*H*W)
#include "ocf_common.h"

namespace ocf {
namespace nm {
constexpr int NB = 256;

struct Scal { float mu, inv_std, dmu, c1; };

// every block: mu / inv_std (and the backward coefficients when bsums != nullptr) from the moment sums; result broadcast through LDS
__device__ __forceinline__ Scal scalars(const double* sums, const double* bsums, int S, int B, double n, float* s_m1, int which) {
  __shared__ double s_red[4];
  const int tid = threadIdx.x;
  double a = 0.0, v = 0.0;
  for (int i = tid; i < S * B; i += NB) {
    const double m1 = sums[2 * i] / n;
    a += m1;
    v += sums[2 * i + 1] / n - m1 * m1;
    if (i / B == which) s_m1[i - which * B] = (float)m1;
  }
  a = wave_sum_d(a);
  v = wave_sum_d(v);
  __shared__ double s_part[2][NB / 64];
  if ((tid & 63) == 0) { s_part[0][tid >> 6] = a; s_part[1][tid >> 6] = v; }
  __syncthreads();
  if (tid == 0) {
    double sa = 0.0, sv = 0.0;
    for (int w = 0; w < NB / 64; w++) { sa += s_part[0][w]; sv += s_part[1][w]; }
    const double mu = sa / (S * B), var = sv / (S * B);
    const double inv = 1.0 / sqrt(var + 1e-16);
    s_red[0] = mu; s_red[1] = inv;
    if (bsums) {
      const float invf = (float)inv;
      const float dmu = -invf * (float)bsums[0] / (float)((double)S * B * n);
      const float dvar = -0.5f * (float)bsums[1] * invf * invf * invf;
      s_red[2] = dmu; s_red[3] = dvar * 2.0f / (float)((double)S * B * n);
    }
  }
  __syncthreads();
  Scal r;
  r.mu = (float)s_red[0]; r.inv_std = (float)s_red[1];
  r.dmu = bsums ? (float)s_red[2] : 0.f; r.c1 = bsums ? (float)s_red[3] : 0.f;
  return r;
}

// memory-order element walk of a tensor that is dense in NCHW or channels_last order (host-checked): flat index -> sample
template <typename T, int MODE>  // MODE 0: forward, 1: backward sums, 2: backward elementwise
__global__ __launch_bounds__(NB) void norm_kernel(const T* x, const T* g, T* out, const double* sums, const double* bsums_in, double* bsums_out,
                                                  int S, int which, int B, long per_sample) {
  extern __shared__ float s_m1[];  // [B]
  const Scal sc = scalars(sums, MODE == 2 ? bsums_in : nullptr, S, B, (double)per_sample, s_m1, which);
  constexpr int V = 16 / (int)sizeof(T);
  const long nvec = (long)B * per_sample / V;  // (host: per_sample % V == 0)
  double acc[2] = {0.0, 0.0};
  float p0 = 0.f, p1 = 0.f;
  int cnt = 0;
  for (long i = (long)blockIdx.x * NB + threadIdx.x; i < nvec; i += (long)gridDim.x * NB) {
    const u32x4 xv = ((const u32x4*)x)[i];
    float xf[V], gf[V], of[V];
    unit_to_f<T>(xv, xf);
    if (MODE != 0) unit_to_f<T>(((const u32x4*)g)[i], gf);
    const float m1 = MODE == 2 ? s_m1[(i * V) / per_sample] : 0.f;
#pragma unroll
    for (int k = 0; k < V; k++) {
      if (MODE == 0) of[k] = (xf[k] - sc.mu) * sc.inv_std;
      if (MODE == 1) { p0 += gf[k]; p1 = __builtin_fmaf(gf[k], xf[k] - sc.mu, p1); }
      if (MODE == 2) of[k] = gf[k] * sc.inv_std + sc.dmu + sc.c1 * (xf[k] - m1);
    }
    if (MODE == 1 && ++cnt == 16) { acc[0] += (double)p0; acc[1] += (double)p1; p0 = p1 = 0.f; cnt = 0; }
    if (MODE != 1) ((u32x4*)out)[i] = f_to_unit<T>(of);
  }
  if (MODE == 1) {
    acc[0] += (double)p0; acc[1] += (double)p1;
    acc[0] = wave_sum_d(acc[0]); acc[1] = wave_sum_d(acc[1]);
    __shared__ double s_acc[2][NB / 64];
    if ((threadIdx.x & 63) == 0) { s_acc[0][threadIdx.x >> 6] = acc[0]; s_acc[1][threadIdx.x >> 6] = acc[1]; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double a0 = 0.0, a1 = 0.0;
      for (int w = 0; w < NB / 64; w++) { a0 += s_acc[0][w]; a1 += s_acc[1][w]; }
      atomicAdd(bsums_out, a0);
      atomicAdd(bsums_out + 1, a1);
    }
  }
}

// dense in memory with the same strides (NCHW-contiguous or channels_last), 16-byte aligned, whole 16-byte units per sample
static bool dense_pair(const ocf_tensor4* a, const ocf_tensor4* b, int B, int C, int H, int W, int es) {
  auto dense = [&](const ocf_tensor4* t) {
    const bool nchw = t->sw == 1 && t->sh == W && t->sc == (int64_t)H * W && t->sb == (int64_t)C * H * W;
    const bool cl = (t->sc == 1 || C == 1) && t->sw == C && t->sh == (int64_t)W * C && t->sb == (int64_t)C * H * W;
    return (nchw || cl) && ((uintptr_t)t->data % 16 == 0);
  };
  if (!dense(a) || (b && !dense(b))) return false;
  if (b && !(a->sb == b->sb && a->sc == b->sc && a->sh == b->sh && a->sw == b->sw)) return false;
  return ((long)C * H * W * es) % 16 == 0;
}

template <int MODE>
static int launch(const ocf_tensor4* x, const ocf_tensor4* g, const ocf_tensor4* out, const double* sums, const double* bsums_in, double* bsums_out,
                  int S, int which, int B, int C, int H, int W, int dtype, hipStream_t st) {
  const long per = (long)C * H * W, total = per * B;
  const int es = dtype == OCF_BF16 ? 2 : 4;
  long blocks = (total * es / 16 + NB * 4 - 1) / (NB * 4);
  if (blocks < 1) blocks = 1;
  if (blocks > (MODE == 1 ? 1024 : 8192)) blocks = MODE == 1 ? 1024 : 8192;
  const size_t lds = (size_t)B * sizeof(float);
  LaunchProf lp;
  if (dtype == OCF_F32)
    lp.launch(true, norm_kernel<float, MODE>, dim3((unsigned)blocks), dim3(NB), lds, st, (const float*)x->data, g ? (const float*)g->data : nullptr,
              out ? (float*)out->data : nullptr, sums, bsums_in, bsums_out, S, which, B, per);
  else
    lp.launch(true, norm_kernel<bf16, MODE>, dim3((unsigned)blocks), dim3(NB), lds, st, (const bf16*)x->data, g ? (const bf16*)g->data : nullptr,
              out ? (bf16*)out->data : nullptr, sums, bsums_in, bsums_out, S, which, B, per);
  return launch_status();
}

}  // namespace nm
}  // namespace ocf

extern "C" int ocf_norm_fwd(const ocf_tensor4* x, const ocf_tensor4* y, const double* sums, int S, int which, int B, int C, int H, int W, int dtype,
                            void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !y || !x->data || !y->data || !sums) return OCF_EINVAL;
  if (S <= 0 || which < 0 || which >= S || B <= 0 || C <= 0 || H <= 0 || W <= 0 || (dtype != OCF_F32 && dtype != OCF_BF16)) return OCF_EINVAL;
  if (B > 4096 || !nm::dense_pair(x, y, B, C, H, W, dtype == OCF_BF16 ? 2 : 4)) return OCF_EUNSUPPORTED;
  return nm::launch<0>(x, nullptr, y, sums, nullptr, nullptr, S, which, B, C, H, W, dtype, (hipStream_t)stream);
}

extern "C" int ocf_norm_bwd_sums(const ocf_tensor4* x, const ocf_tensor4* gy, const double* sums, double* bsums, int zero_first, int S, int B, int C,
                                 int H, int W, int dtype, void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !gy || !x->data || !gy->data || !sums || !bsums) return OCF_EINVAL;
  if (S <= 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || (dtype != OCF_F32 && dtype != OCF_BF16)) return OCF_EINVAL;
  if (B > 4096 || !nm::dense_pair(x, gy, B, C, H, W, dtype == OCF_BF16 ? 2 : 4)) return OCF_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (zero_first && hipMemsetAsync(bsums, 0, 2 * sizeof(double), st) != hipSuccess) return OCF_ELAUNCH;
  return nm::launch<1>(x, gy, nullptr, sums, nullptr, bsums, S, 0, B, C, H, W, dtype, st);
}

extern "C" int ocf_norm_bwd(const ocf_tensor4* x, const ocf_tensor4* gy, const ocf_tensor4* gx, const double* sums, const double* bsums, int S,
                            int which, int B, int C, int H, int W, int dtype, void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !gy || !gx || !x->data || !gy->data || !gx->data || !sums || !bsums) return OCF_EINVAL;
  if (S <= 0 || which < 0 || which >= S || B <= 0 || C <= 0 || H <= 0 || W <= 0 || (dtype != OCF_F32 && dtype != OCF_BF16)) return OCF_EINVAL;
  const int es = dtype == OCF_BF16 ? 2 : 4;
  if (B > 4096 || !nm::dense_pair(x, gy, B, C, H, W, es) || !nm::dense_pair(x, gx, B, C, H, W, es)) return OCF_EUNSUPPORTED;
  return nm::launch<2>(x, gy, gx, sums, bsums, nullptr, S, which, B, C, H, W, dtype, (hipStream_t)stream);
}
