// BatchNorm2d (batch statistics) + LeakyReLU of the feature-pyramid encoder in two passes per direction.
//
// The reference's encoder is  nn.Sequential(nn.Conv2d(bias=False), nn.BatchNorm2d, nn.LeakyReLU(0.1))  x 17 per image
// (models/networks/feature_pyramid_net.py:5-95; both images of a pair go through it separately, so the statistics are per call).
// Through the framework that is 9 launches per site and step -- three batch-norm kernels and the activation forward, three batch-norm
// kernels and the activation backward, and an int64 `num_batches_tracked += 1` kernel -- most of them on small tensors at the launch
// floor: 34 sites x ~54 us = 1.8 ms of a 14.9 ms step (profiles/r2_g_rocprof_kernel_stats.csv).  Here:
//   forward   pass 1  per-channel sum / sum of squares as per-block partials (<= 64 blocks, plain stores)
//             pass 2  every block first adds the partials up in fp64 for all channels (a few thousand L2 reads) -> mean, 1/sqrt(var+eps)
//                     in LDS, then y = lrelu((x - mean) * invstd * gamma + beta); block 0 also writes the saved statistics, the running
//                     statistics and the batch counter
//   backward  pass 1  dz = lrelu'(y) * dy;  per-channel partials of sum dz, sum dz * xhat
//             pass 2  totals as above (block 0 writes dbeta, dgamma), dx = gamma * invstd * (dz - dbeta / N - xhat * dgamma / N)
// No same-address atomics (their tail grows with the block count, see epilogue.hip), no zero-fill launches, and no in-kernel
// "last block reduces" hand-off: that needs a device-scope fence per block, which on this chip writes back the XCD's whole L2 -- the
// first version of this file did that and made the training step 3 ms SLOWER (14.9 -> 17.9 ms).  The kernel boundary orders the
// partials for free.  Arithmetic: fp32 per element, fp64 for the channel totals; rounding points as in the framework
// chain (batch-norm output rounded to the activation type, then the activation).  Dense channels_last tensors only (what the bf16
// convolutions produce); anything else reports OCF_EUNSUPPORTED and the caller uses the framework's operators.
#include "ocf_common.h"

namespace ocf {
namespace bn {
constexpr int MAX_PARTS = 256;
constexpr long COL_MAX_ELEMS = 256 * 1024;  // at most this many elements: the one-launch column kernels

template <typename T> __device__ __forceinline__ float rnd(float v) { return (float)(T)v; }

// V channels per thread: bf16 8 (16 bytes) or 4 (8 bytes); fp32 4 (16 bytes)
template <typename T, int V> __device__ __forceinline__ void ldv(const T* p, float* f);
template <> __device__ __forceinline__ void ldv<bf16, 8>(const bf16* p, float* f) { unit_to_f<bf16>(*(const u32x4*)p, f); }
template <> __device__ __forceinline__ void ldv<bf16, 4>(const bf16* p, float* f) {
  const u32x2 v = *(const u32x2*)p;
  f[0] = __builtin_bit_cast(float, v[0] << 16); f[1] = __builtin_bit_cast(float, v[0] & 0xffff0000u);
  f[2] = __builtin_bit_cast(float, v[1] << 16); f[3] = __builtin_bit_cast(float, v[1] & 0xffff0000u);
}
template <> __device__ __forceinline__ void ldv<float, 4>(const float* p, float* f) {
  const f32x4 v = *(const f32x4*)p;
  f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
}
template <typename T, int V> __device__ __forceinline__ void stv(T* p, const float* f);
template <> __device__ __forceinline__ void stv<bf16, 8>(bf16* p, const float* f) { *(u32x4*)p = f_to_unit<bf16>(f); }
template <> __device__ __forceinline__ void stv<bf16, 4>(bf16* p, const float* f) {
  const bf16x2 a = {(bf16)f[0], (bf16)f[1]}, b = {(bf16)f[2], (bf16)f[3]};
  *(u32x2*)p = u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
}
template <> __device__ __forceinline__ void stv<float, 4>(float* p, const float* f) { *(f32x4*)p = f32x4{f[0], f[1], f[2], f[3]}; }

constexpr int MAX_C = 1024;  // channels of the LDS coefficient tables

// the tensor as an [R][C] matrix; thread (r = tid / G, cg = tid % G) owns V channels and walks rows r, r + step, ...
// Block totals of two per-channel quantities -> part[(block * 2 + q) * C + c]
template <int V>
__device__ __forceinline__ void store_partials(float (*red)[2 * V + 1], const float* a0, const float* a1, float* part, int C, int G) {
  const int rpb = 256 / G;
#pragma unroll
  for (int j = 0; j < V; j++) { red[threadIdx.x][j] = a0[j]; red[threadIdx.x][V + j] = a1[j]; }
  __syncthreads();
  for (int t = threadIdx.x; t < C; t += 256) {
    const int g = t / V, j = t - g * V;
    float s0 = 0.f, s1 = 0.f;
    for (int rr = 0; rr < rpb; rr++) { s0 += red[rr * G + g][j]; s1 += red[rr * G + g][V + j]; }
    part[((long)blockIdx.x * 2 + 0) * C + t] = s0;
    part[((long)blockIdx.x * 2 + 1) * C + t] = s1;
  }
}

template <typename T, int V>
__global__ void __launch_bounds__(256) bn_stats_kernel(const T* __restrict__ x, float* __restrict__ part, long R, int C, int G) {
  // blockIdx.y = statistics group (R rows each; a group is a contiguous range of the batch)
  x += (long)blockIdx.y * R * C;
  part += (long)blockIdx.y * gridDim.x * 2 * C;
  __shared__ float red[256][2 * V + 1];
  const int rpb = 256 / G;
  const int r = threadIdx.x / G, cg = threadIdx.x - r * G;
  float s[V], q[V];
#pragma unroll
  for (int j = 0; j < V; j++) s[j] = q[j] = 0.f;
  if (r < rpb) {
    const long step = (long)gridDim.x * rpb;
    for (long row0 = (long)blockIdx.x * rpb + r; row0 < R; row0 += step * 4) {
      float f[4][V];
#pragma unroll
      for (int k = 0; k < 4; k++) {  // branch-free: rows past the end re-read the last row and are ignored
        const long row = row0 + k * step < R ? row0 + k * step : R - 1;
        ldv<T, V>(x + row * C + cg * V, f[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (row0 + k * step < R) {
#pragma unroll
          for (int j = 0; j < V; j++) { s[j] += f[k][j]; q[j] = __builtin_fmaf(f[k][j], f[k][j], q[j]); }
        }
    }
  }
  store_partials<V>(red, s, q, part, C, G);
}

// totals of the two per-channel quantities over `nparts` block partials, fp64, for every channel -> tot[0][c], tot[1][c] (LDS).
// All 256 threads share the work: thread = (channel, slice of the partials) in channel blocks of <= 256, and a thread's loads are issued
// EIGHT partials at a time before the first add.  (The loop used to be one load pair per iteration with the fp64 add behind it: up to 32
// dependent L2 round trips per thread -- 64 where C does not divide 256, one thread per channel -- in the prologue of EVERY block of the
// apply / dx passes: those kernels ran 19-25 us at pyramid levels whose tensors are 5-30 MB.)
__device__ __forceinline__ void reduce_partials(const float* __restrict__ part, int nparts, int C, double (*tot)[MAX_C]) {
  __shared__ double s_r[2][256];
  for (int cb = 0; cb < C; cb += 256) {
    const int cw = C - cb < 256 ? C - cb : 256, nslots = 256 / cw;
    const int c = threadIdx.x % cw, slot = threadIdx.x / cw;
    double ds = 0.0, dq = 0.0;
    if (slot < nslots) {
      const float* pc = part + cb + c;
      for (int b0 = slot; b0 < nparts; b0 += 8 * nslots) {
        float a0[8], a1[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int b = b0 + k * nslots, bc = b < nparts ? b : b0;
          a0[k] = pc[((long)bc * 2) * C];
          a1[k] = pc[((long)bc * 2 + 1) * C];
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (b0 + k * nslots < nparts) { ds += (double)a0[k]; dq += (double)a1[k]; }
      }
    }
    s_r[0][threadIdx.x] = ds; s_r[1][threadIdx.x] = dq;
    __syncthreads();
    if (threadIdx.x < cw) {
      ds = 0.0; dq = 0.0;
      for (int k = 0; k < nslots; k++) { ds += s_r[0][k * cw + threadIdx.x]; dq += s_r[1][k * cw + threadIdx.x]; }
      tot[0][cb + threadIdx.x] = ds; tot[1][cb + threadIdx.x] = dq;
    }
    __syncthreads();
  }
}

// y = lrelu(T(x * a + b)),  a = gamma * invstd, b = beta - mean * a.  nparts > 0: batch statistics from the partials of bn_stats_kernel
// (block 0 publishes them and updates the running buffers); nparts == 0: the running statistics (evaluation mode: mean, var given)
template <typename T, int V>
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ part, int nparts,
                                                       float* mean, float* invstd, float* run_mean, float* run_var, long long* nbt,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, long R, int C, int G,
                                                       float slope, float eps, float momentum) {
  __shared__ float s_a[MAX_C], s_b[MAX_C];
  __shared__ double s_tot[2][MAX_C];
  // blockIdx.y = statistics group (1 or 2 groups): its own rows, partials and saved statistics.  The running buffers see the groups one
  // after the other, exactly as two calls of the module would: block (0, 0) -- the keeper -- also totals group 1 for that.
  __shared__ float s_m1[MAX_C], s_v1[MAX_C];
  const int grp = blockIdx.y, ngrp = gridDim.y;
  x += (long)grp * R * C;
  y += (long)grp * R * C;
  const bool keeper = nparts > 0 && blockIdx.x == 0 && grp == 0;
  if (keeper && ngrp == 2) {
    reduce_partials(part + (long)nparts * 2 * C, nparts, C, s_tot);
    for (int c = threadIdx.x; c < C; c += 256) {
      const double n = (double)R, dm = s_tot[0][c] / n;
      double var = s_tot[1][c] / n - dm * dm;
      s_m1[c] = (float)dm;
      s_v1[c] = (float)(var > 0.0 ? var : 0.0);
    }
    __syncthreads();
  }
  if (nparts > 0) reduce_partials(part + (long)grp * nparts * 2 * C, nparts, C, s_tot);
  for (int c = threadIdx.x; c < C; c += 256) {
    float m, is;
    if (nparts > 0) {
      const double n = (double)R, dm = s_tot[0][c] / n;
      double var = s_tot[1][c] / n - dm * dm;
      var = var > 0.0 ? var : 0.0;
      m = (float)dm;
      is = (float)(1.0 / sqrt(var + (double)eps));
      if (blockIdx.x == 0) {
        mean[(long)grp * C + c] = m;
        invstd[(long)grp * C + c] = is;
      }
      if (keeper) {
        const double unb = R > 1 ? n / (n - 1.0) : 1.0, mo = (double)momentum;
        double rmv = run_mean ? (double)run_mean[c] : 0.0, rvv = run_var ? (double)run_var[c] : 0.0;
        rmv = (1.0 - mo) * rmv + mo * dm;
        rvv = (1.0 - mo) * rvv + mo * var * unb;
        if (ngrp == 2) {
          rmv = (float)rmv;  // (a separate call would have stored fp32 in between)
          rvv = (float)rvv;
          rmv = (1.0 - mo) * rmv + mo * (double)s_m1[c];
          rvv = (1.0 - mo) * rvv + mo * (double)s_v1[c] * unb;
        }
        if (run_mean) run_mean[c] = (float)rmv;
        if (run_var) run_var[c] = (float)rvv;
      }
    } else {
      m = run_mean[c];
      is = 1.f / sqrtf(run_var[c] + eps);
    }
    const float a = gamma[c] * is;
    s_a[c] = a;
    s_b[c] = beta[c] - m * a;
  }
  if (keeper && threadIdx.x == 0 && nbt) *nbt += ngrp;
  __syncthreads();
  const long n_units = R * G;
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
  if (i0 >= n_units) return;
  const long stride = (long)gridDim.x * 256;  // a multiple of G (host): the thread keeps its channel group
  const int cg = (int)(i0 % G);
  float a[V], b[V];
#pragma unroll
  for (int j = 0; j < V; j++) { a[j] = s_a[cg * V + j]; b[j] = s_b[cg * V + j]; }
  for (long i = i0; i < n_units; i += 4 * stride) {
    float f[4][V];
#pragma unroll
    for (int k = 0; k < 4; k++) {  // four units in flight per thread (branch-free: a unit past the end re-reads the thread's first)
      const long ii = i + k * stride < n_units ? i + k * stride : i0;
      ldv<T, V>(x + ii * V, f[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (i + k * stride < n_units) {
#pragma unroll
        for (int j = 0; j < V; j++) {
          const float t = rnd<T>(__builtin_fmaf(f[k][j], a[j], b[j]));
          f[k][j] = t > 0.f ? t : rnd<T>(t * slope);
        }
        stv<T, V>(y + (i + k * stride) * V, f[k]);
      }
  }
}

// NOY: the forward output is not read -- the sign of the pre-activation t = T(x * a + b) is recomputed from x with the forward's own
// coefficients (a = gamma * invstd, b = beta - mean * a from the SAVED fp32 statistics, the same fma, the same rounding: the same bits),
// which takes a third of the traffic out of this pass and a quarter out of the dx pass
template <typename T, int V, bool NOY>
__global__ void __launch_bounds__(256) bn_bwd_stats_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ gy,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ part, long R, int C, int G, float slope) {
  {  // blockIdx.y = statistics group
    const long go = (long)blockIdx.y * R * C;
    x += go; gy += go;
    if (!NOY) y += go;
    mean += (long)blockIdx.y * C; invstd += (long)blockIdx.y * C;
    part += (long)blockIdx.y * gridDim.x * 2 * C;
  }
  __shared__ float red[256][2 * V + 1];
  const int rpb = 256 / G;
  const int r = threadIdx.x / G, cg = threadIdx.x - r * G;
  float s[V], q[V], mu[V], is[V], ca[V], cb[V];
#pragma unroll
  for (int j = 0; j < V; j++) {
    const int c = (cg * V + j) < C ? cg * V + j : 0;
    s[j] = q[j] = 0.f;
    mu[j] = mean[c];
    is[j] = invstd[c];
    ca[j] = NOY ? gamma[c] * is[j] : 0.f;
    cb[j] = NOY ? beta[c] - mu[j] * ca[j] : 0.f;
  }
  if (r < rpb) {
    const long step = (long)gridDim.x * rpb;
    for (long row0 = (long)blockIdx.x * rpb + r; row0 < R; row0 += step * 4) {
      float fx[4][V], fy[4][V], fg[4][V];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const long row = row0 + k * step < R ? row0 + k * step : R - 1;
        const long o = row * C + cg * V;
        ldv<T, V>(x + o, fx[k]); ldv<T, V>(gy + o, fg[k]);
        if (!NOY) ldv<T, V>(y + o, fy[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (row0 + k * step < R) {
#pragma unroll
          for (int j = 0; j < V; j++) {
            const float yv = NOY ? rnd<T>(__builtin_fmaf(fx[k][j], ca[j], cb[j])) : fy[k][j];
            const float dz = yv > 0.f ? fg[k][j] : rnd<T>(fg[k][j] * slope);
            s[j] += dz;
            q[j] = __builtin_fmaf(dz, (fx[k][j] - mu[j]) * is[j], q[j]);
          }
        }
    }
  }
  store_partials<V>(red, s, q, part, C, G);
}

// dx = gamma * invstd * (dz - dbeta / N - xhat * dgamma / N); the totals dbeta = sum dz, dgamma = sum dz * xhat from the partials
template <typename T, int V, bool NOY>
__global__ void __launch_bounds__(256) bn_bwd_dx_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ gy,
                                                        T* __restrict__ gx, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ part, int nparts, float* dgamma, float* dbeta, long R,
                                                        int C, int G, float slope) {
  __shared__ double s_tot[2][MAX_C];
  __shared__ float s_o[2][MAX_C];
  // blockIdx.y = statistics group; the parameter gradients are the sums over the groups: block (0, 0) also totals group 1
  const int grp = blockIdx.y, ngrp = gridDim.y;
  {
    const long go = (long)grp * R * C;
    x += go; gy += go; gx += go;
    if (!NOY) y += go;
    mean += (long)grp * C; invstd += (long)grp * C;
  }
  const bool keeper = blockIdx.x == 0 && grp == 0;
  if (keeper && ngrp == 2) {
    reduce_partials(part + (long)nparts * 2 * C, nparts, C, s_tot);
    for (int c = threadIdx.x; c < C; c += 256) { s_o[0][c] = (float)s_tot[0][c]; s_o[1][c] = (float)s_tot[1][c]; }
    __syncthreads();
  }
  reduce_partials(part + (long)grp * nparts * 2 * C, nparts, C, s_tot);
  const float inv_n = 1.f / (float)R;
  if (keeper)
    for (int c = threadIdx.x; c < C; c += 256) {
      dbeta[c] = (float)s_tot[0][c] + (ngrp == 2 ? s_o[0][c] : 0.f);
      dgamma[c] = (float)s_tot[1][c] + (ngrp == 2 ? s_o[1][c] : 0.f);
    }
  const long n_units = R * G;
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
  if (i0 >= n_units) return;
  const long stride = (long)gridDim.x * 256;
  const int cg = (int)(i0 % G);
  float mu[V], is[V], k0[V], k1[V], k2[V], cb[V];
#pragma unroll
  for (int j = 0; j < V; j++) {
    const int c = cg * V + j;
    mu[j] = mean[c]; is[j] = invstd[c];
    k0[j] = gamma[c] * is[j];
    cb[j] = NOY ? beta[c] - mu[j] * k0[j] : 0.f;
    k1[j] = (float)s_tot[0][c] * inv_n;
    k2[j] = (float)s_tot[1][c] * inv_n;
  }
  for (long i = i0; i < n_units; i += 2 * stride) {
    float fx[2][V], fy[2][V], fg[2][V];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const long ii = i + k * stride < n_units ? i + k * stride : i0;
      ldv<T, V>(x + ii * V, fx[k]); ldv<T, V>(gy + ii * V, fg[k]);
      if (!NOY) ldv<T, V>(y + ii * V, fy[k]);
    }
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (i + k * stride < n_units) {
#pragma unroll
        for (int j = 0; j < V; j++) {
          const float yv = NOY ? rnd<T>(__builtin_fmaf(fx[k][j], k0[j], cb[j])) : fy[k][j];
          const float dz = yv > 0.f ? fg[k][j] : rnd<T>(fg[k][j] * slope);
          fx[k][j] = k0[j] * (dz - k1[j] - (fx[k][j] - mu[j]) * is[j] * k2[j]);
        }
        stv<T, V>(gx + (i + k * stride) * V, fx[k]);
      }
  }
}

constexpr int CB = 256;  // threads per block of the column kernels (1024 measured 3-5x slower: 64 scattered lines per wave load)
// ---- small tensors (<= 256K elements: the coarsest pyramid level): ONE launch per direction.  Block = one group of V channels, all rows: its statistics need no other block.  The tensor is read twice by the same block (second time from L2); a wave's rows are
// C * sizeof(T) bytes apart, which at these sizes costs less than a second launch.
template <int V>
__device__ __forceinline__ void block_totals(float* s, float* q, float (*red)[2 * V]) {
  // wave shuffle, then one LDS round over the 16 waves; result in every thread
#pragma unroll
  for (int j = 0; j < V; j++) { s[j] = wave_sum(s[j]); q[j] = wave_sum(q[j]); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < V; j++) { red[w][j] = s[j]; red[w][V + j] = q[j]; }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < V; j++) {
    float ts = 0.f, tq = 0.f;
#pragma unroll
    for (int k = 0; k < CB / 64; k++) { ts += red[k][j]; tq += red[k][V + j]; }
    s[j] = ts; q[j] = tq;
  }
}

template <typename T, int V>
__global__ void __launch_bounds__(CB) bn_fwd_col_kernel(const T* __restrict__ x, T* __restrict__ y, float* mean, float* invstd, float* run_mean,
                                                         float* run_var, long long* nbt, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, long R, int C, float slope, float eps, float momentum, int ngrp) {
  __shared__ float red[CB / 64][2 * V];
  const int c0 = blockIdx.x * V;
  const T* x0 = x;
  T* y0 = y;
  for (int grp = 0; grp < ngrp; grp++) {  // statistics groups one after the other (the running buffers see them in order)
  x = x0 + (long)grp * R * C;
  y = y0 + (long)grp * R * C;
  float s[V], q[V];
#pragma unroll
  for (int j = 0; j < V; j++) s[j] = q[j] = 0.f;
  for (long r0 = threadIdx.x; r0 < R; r0 += 4 * CB) {
    float f[4][V];
#pragma unroll
    for (int k = 0; k < 4; k++) ldv<T, V>(x + (r0 + CB * k < R ? r0 + CB * k : r0) * C + c0, f[k]);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (r0 + CB * k < R) {
#pragma unroll
        for (int j = 0; j < V; j++) { s[j] += f[k][j]; q[j] = __builtin_fmaf(f[k][j], f[k][j], q[j]); }
      }
  }
  block_totals<V>(s, q, red);
  float a[V], b[V];
#pragma unroll
  for (int j = 0; j < V; j++) {
    const double n = (double)R, dm = (double)s[j] / n;
    double var = (double)q[j] / n - dm * dm;
    var = var > 0.0 ? var : 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const int c = c0 + j;
    if (threadIdx.x == 0) {
      mean[(long)grp * C + c] = (float)dm;
      invstd[(long)grp * C + c] = is;
      if (run_mean) run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * dm);
      if (run_var) run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * var * (R > 1 ? n / (n - 1.0) : 1.0));
    }
    a[j] = gamma[c] * is;
    b[j] = beta[c] - (float)dm * a[j];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt && grp == 0) *nbt += ngrp;
  for (long r0 = threadIdx.x; r0 < R; r0 += 4 * CB) {
    float f[4][V];
#pragma unroll
    for (int k = 0; k < 4; k++) ldv<T, V>(x + (r0 + CB * k < R ? r0 + CB * k : r0) * C + c0, f[k]);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (r0 + CB * k < R) {
#pragma unroll
        for (int j = 0; j < V; j++) {
          const float t = rnd<T>(__builtin_fmaf(f[k][j], a[j], b[j]));
          f[k][j] = t > 0.f ? t : rnd<T>(t * slope);
        }
        stv<T, V>(y + (r0 + CB * k) * C + c0, f[k]);
      }
  }
  __syncthreads();  // (red is reused by the next group)
  }
}

template <typename T, int V, bool NOY>
__global__ void __launch_bounds__(CB) bn_bwd_col_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ gy,
                                                         T* __restrict__ gx, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float* dgamma,
                                                         float* dbeta, long R, int C, float slope, int ngrp) {
  __shared__ float red[CB / 64][2 * V];
  const int c0 = blockIdx.x * V;
  const T *x0 = x, *y0 = y, *g0 = gy;
  T* gx0 = gx;
  float sb[V], sg[V];  // parameter gradients: sums over the statistics groups
#pragma unroll
  for (int j = 0; j < V; j++) sb[j] = sg[j] = 0.f;
  for (int grp = 0; grp < ngrp; grp++) {
  x = x0 + (long)grp * R * C; if (!NOY) y = y0 + (long)grp * R * C; gy = g0 + (long)grp * R * C; gx = gx0 + (long)grp * R * C;
  float s[V], q[V], mu[V], is[V], ca[V], cb[V];
#pragma unroll
  for (int j = 0; j < V; j++) {
    s[j] = q[j] = 0.f; mu[j] = mean[(long)grp * C + c0 + j]; is[j] = invstd[(long)grp * C + c0 + j];
    ca[j] = gamma[c0 + j] * is[j];
    cb[j] = NOY ? beta[c0 + j] - mu[j] * ca[j] : 0.f;
  }
  for (long r0 = threadIdx.x; r0 < R; r0 += 2 * CB) {
    float fx[2][V], fy[2][V], fg[2][V];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const long o = (r0 + CB * k < R ? r0 + CB * k : r0) * C + c0;
      ldv<T, V>(x + o, fx[k]); ldv<T, V>(gy + o, fg[k]);
      if (!NOY) ldv<T, V>(y + o, fy[k]);
    }
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (r0 + CB * k < R) {
#pragma unroll
        for (int j = 0; j < V; j++) {
          const float yv = NOY ? rnd<T>(__builtin_fmaf(fx[k][j], ca[j], cb[j])) : fy[k][j];
          const float dz = yv > 0.f ? fg[k][j] : rnd<T>(fg[k][j] * slope);
          s[j] += dz;
          q[j] = __builtin_fmaf(dz, (fx[k][j] - mu[j]) * is[j], q[j]);
        }
      }
  }
  block_totals<V>(s, q, red);
  const float inv_n = 1.f / (float)R;
  float k0[V], k1[V], k2[V];
#pragma unroll
  for (int j = 0; j < V; j++) {
    sb[j] += s[j]; sg[j] += q[j];
    k0[j] = gamma[c0 + j] * is[j];
    k1[j] = s[j] * inv_n;
    k2[j] = q[j] * inv_n;
  }
  for (long r0 = threadIdx.x; r0 < R; r0 += 2 * CB) {
    float fx[2][V], fy[2][V], fg[2][V];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const long o = (r0 + CB * k < R ? r0 + CB * k : r0) * C + c0;
      ldv<T, V>(x + o, fx[k]); ldv<T, V>(gy + o, fg[k]);
      if (!NOY) ldv<T, V>(y + o, fy[k]);
    }
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (r0 + CB * k < R) {
#pragma unroll
        for (int j = 0; j < V; j++) {
          const float yv = NOY ? rnd<T>(__builtin_fmaf(fx[k][j], ca[j], cb[j])) : fy[k][j];
          const float dz = yv > 0.f ? fg[k][j] : rnd<T>(fg[k][j] * slope);
          fx[k][j] = k0[j] * (dz - k1[j] - (fx[k][j] - mu[j]) * is[j] * k2[j]);
        }
        stv<T, V>(gx + (r0 + CB * k) * C + c0, fx[k]);
      }
  }
  __syncthreads();  // (red is reused by the next group)
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < V; j++) { dbeta[c0 + j] = sb[j]; dgamma[c0 + j] = sg[j]; }
  }
}

static bool dense_cl(const ocf_tensor4* t, int B, int C, int H, int W) {
  return t && t->data && t->sc == 1 && t->sw == C && t->sh == (int64_t)W * C && t->sb == (int64_t)H * W * C;
}

// channels per thread for this (dtype, C, alignment): 8 / 4, or 0 when the kernels do not take the tensor
template <typename T> static int pick_v(int C, std::initializer_list<const ocf_tensor4*> ts) {
  auto aligned = [&](int bytes) {
    for (const ocf_tensor4* t : ts)
      if ((uintptr_t)t->data % bytes) return false;
    return true;
  };
  if (C > MAX_C) return 0;
  if (sizeof(T) == 2 && C % 8 == 0 && aligned(16)) return 8;
  if (C % 4 == 0 && aligned(sizeof(T) * 4)) return 4;
  return 0;
}

// blocks of the statistics passes: >= 8K elements each (one or two rounds of loads per thread at the coarse levels: a block's loop is a
// chain of load round trips), at most MAX_PARTS (every block of the second pass totals all partials: reduce_partials)
static int stat_blocks(long R, int C) {
  long want = (R * C + 8191) / 8192;
  return (int)(want < 1 ? 1 : (want < MAX_PARTS ? want : MAX_PARTS));
}

static int grid_for_units(long n_units, int G) {
  // a multiple of G threads in total so that a thread's channel group is fixed; <= 512 blocks (each re-reads the partials)
  long blocks = (n_units + 1023) / 1024;
  if (blocks < 1) blocks = 1;
  if (blocks > 512) blocks = 512;
  while ((blocks * 256) % G != 0) blocks++;
  return (int)blocks;
}

template <typename T, int V>
static int fwd_v(const ocf_tensor4* x, const ocf_tensor4* y, const float* gamma, const float* beta, float* rm, float* rv, long long* nbt,
                 float* save_mean, float* save_invstd, void* workspace, long R, int C, float eps, float momentum, float slope, int training,
                 int groups, hipStream_t st) {
  // R = rows of ONE statistics group
  const int G = C / V;
  const long n_units = R * G;
  LaunchProf lp;
  if (training && R * C <= COL_MAX_ELEMS) {
    lp.launch(true, bn_fwd_col_kernel<T, V>, dim3(G), dim3(CB), 0, st, (const T*)x->data, (T*)y->data, save_mean, save_invstd, rm, rv, nbt, gamma,
              beta, R, C, slope, eps, momentum, groups);
    return launch_status();
  }
  if (training) {
    const int nb = stat_blocks(R, C);
    lp.launch(false, bn_stats_kernel<T, V>, dim3(nb, groups), dim3(256), 0, st, (const T*)x->data, (float*)workspace, R, C, G);
    lp.launch(true, bn_apply_kernel<T, V>, dim3(grid_for_units(n_units, G), groups), dim3(256), 0, st, (const T*)x->data, (T*)y->data,
              (const float*)workspace, nb, save_mean, save_invstd, rm, rv, nbt, gamma, beta, R, C, G, slope, eps, momentum);
  } else {
    lp.launch(true, bn_apply_kernel<T, V>, dim3(grid_for_units(n_units, G), groups), dim3(256), 0, st, (const T*)x->data, (T*)y->data,
              (const float*)nullptr, 0, (float*)nullptr, (float*)nullptr, rm, rv, (long long*)nullptr, gamma, beta, R, C, G, slope, eps, 0.f);
  }
  return launch_status();
}

// y == NULL (with beta): the y-free kernels (NOY)
template <typename T, int V>
static int bwd_v(const ocf_tensor4* x, const ocf_tensor4* y, const ocf_tensor4* gy, const ocf_tensor4* gx, const float* gamma,
                 const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, void* workspace, long R, int C, float slope,
                 int groups, hipStream_t st, const float* beta = nullptr) {
  const int G = C / V;
  const long n_units = R * G;
  const int nb = stat_blocks(R, C);
  LaunchProf lp;
  const T* yd = y ? (const T*)y->data : nullptr;
  if (R * C <= COL_MAX_ELEMS) {
    if (yd)
      lp.launch(true, bn_bwd_col_kernel<T, V, false>, dim3(G), dim3(CB), 0, st, (const T*)x->data, yd, (const T*)gy->data, (T*)gx->data, save_mean,
                save_invstd, gamma, beta, dgamma, dbeta, R, C, slope, groups);
    else
      lp.launch(true, bn_bwd_col_kernel<T, V, true>, dim3(G), dim3(CB), 0, st, (const T*)x->data, yd, (const T*)gy->data, (T*)gx->data, save_mean,
                save_invstd, gamma, beta, dgamma, dbeta, R, C, slope, groups);
    return launch_status();
  }
  if (yd) {
    lp.launch(false, bn_bwd_stats_kernel<T, V, false>, dim3(nb, groups), dim3(256), 0, st, (const T*)x->data, yd, (const T*)gy->data, save_mean,
              save_invstd, gamma, beta, (float*)workspace, R, C, G, slope);
    lp.launch(true, bn_bwd_dx_kernel<T, V, false>, dim3(grid_for_units(n_units, G), groups), dim3(256), 0, st, (const T*)x->data, yd,
              (const T*)gy->data, (T*)gx->data, save_mean, save_invstd, gamma, beta, (const float*)workspace, nb, dgamma, dbeta, R, C, G, slope);
  } else {
    lp.launch(false, bn_bwd_stats_kernel<T, V, true>, dim3(nb, groups), dim3(256), 0, st, (const T*)x->data, yd, (const T*)gy->data, save_mean,
              save_invstd, gamma, beta, (float*)workspace, R, C, G, slope);
    lp.launch(true, bn_bwd_dx_kernel<T, V, true>, dim3(grid_for_units(n_units, G), groups), dim3(256), 0, st, (const T*)x->data, yd,
              (const T*)gy->data, (T*)gx->data, save_mean, save_invstd, gamma, beta, (const float*)workspace, nb, dgamma, dbeta, R, C, G, slope);
  }
  return launch_status();
}

}  // namespace bn
}  // namespace ocf

using namespace ocf;

extern "C" int ocf_bn_act_workspace_bytes(int C, int groups, int64_t* bytes) {
  if (C <= 0 || groups < 1 || groups > 2 || !bytes) return OCF_EINVAL;
  *bytes = (int64_t)groups * bn::MAX_PARTS * 2 * C * sizeof(float);
  return OCF_OK;
}

extern "C" int ocf_bn_act_fwd(const ocf_tensor4* x, const ocf_tensor4* y, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd, void* workspace, int B,
                              int C, int H, int W, float eps, float momentum, float negative_slope, int training, int groups, int dtype, void* stream) {
  clear_stale_error();
  if (groups < 1 || groups > 2 || (B % (groups > 0 ? groups : 1)) != 0 || (!training && groups != 1)) return OCF_EINVAL;
  if (!x || !y || !x->data || !y->data || !gamma || !beta || B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if (dtype != OCF_F32 && dtype != OCF_BF16) return OCF_EINVAL;
  if (training && (!save_mean || !save_invstd || !workspace)) return OCF_EINVAL;
  if (!training && (!running_mean || !running_var)) return OCF_EINVAL;
  if (!bn::dense_cl(x, B, C, H, W) || !bn::dense_cl(y, B, C, H, W)) return OCF_EUNSUPPORTED;
  const long R = (long)(B / groups) * H * W;  // rows of one statistics group
  hipStream_t st = (hipStream_t)stream;
  long long* nbt = (long long*)num_batches_tracked;
  if (dtype == OCF_BF16) {
    const int v = bn::pick_v<bf16>(C, {x, y});
    if (v == 8) return bn::fwd_v<bf16, 8>(x, y, gamma, beta, running_mean, running_var, nbt, save_mean, save_invstd, workspace, R, C, eps, momentum, negative_slope, training, groups, st);
    if (v == 4) return bn::fwd_v<bf16, 4>(x, y, gamma, beta, running_mean, running_var, nbt, save_mean, save_invstd, workspace, R, C, eps, momentum, negative_slope, training, groups, st);
    return OCF_EUNSUPPORTED;
  }
  if (bn::pick_v<float>(C, {x, y}) == 4)
    return bn::fwd_v<float, 4>(x, y, gamma, beta, running_mean, running_var, nbt, save_mean, save_invstd, workspace, R, C, eps, momentum, negative_slope, training, groups, st);
  return OCF_EUNSUPPORTED;
}

extern "C" int ocf_bn_act_bwd(const ocf_tensor4* x, const ocf_tensor4* y, const ocf_tensor4* gy, const ocf_tensor4* gx, const float* gamma,
                              const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, void* workspace, int B, int C,
                              int H, int W, float negative_slope, int groups, int dtype, void* stream) {
  clear_stale_error();
  if (groups < 1 || groups > 2 || B % (groups > 0 ? groups : 1) != 0) return OCF_EINVAL;
  if (!x || !y || !gy || !gx || !gamma || !save_mean || !save_invstd || !dgamma || !dbeta || !workspace) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (dtype != OCF_F32 && dtype != OCF_BF16)) return OCF_EINVAL;
  if (!bn::dense_cl(x, B, C, H, W) || !bn::dense_cl(y, B, C, H, W) || !bn::dense_cl(gy, B, C, H, W) || !bn::dense_cl(gx, B, C, H, W))
    return OCF_EUNSUPPORTED;
  const long R = (long)(B / groups) * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OCF_BF16) {
    const int v = bn::pick_v<bf16>(C, {x, y, gy, gx});
    if (v == 8) return bn::bwd_v<bf16, 8>(x, y, gy, gx, gamma, save_mean, save_invstd, dgamma, dbeta, workspace, R, C, negative_slope, groups, st);
    if (v == 4) return bn::bwd_v<bf16, 4>(x, y, gy, gx, gamma, save_mean, save_invstd, dgamma, dbeta, workspace, R, C, negative_slope, groups, st);
    return OCF_EUNSUPPORTED;
  }
  if (bn::pick_v<float>(C, {x, y, gy, gx}) == 4)
    return bn::bwd_v<float, 4>(x, y, gy, gx, gamma, save_mean, save_invstd, dgamma, dbeta, workspace, R, C, negative_slope, groups, st);
  return OCF_EUNSUPPORTED;
}

/* the same backward WITHOUT the forward output: the LeakyReLU mask is recomputed from x (see bn_bwd_stats_kernel) */
extern "C" int ocf_bn_act_bwd_x(const ocf_tensor4* x, const ocf_tensor4* gy, const ocf_tensor4* gx, const float* gamma, const float* beta,
                                const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, void* workspace, int B, int C,
                                int H, int W, float negative_slope, int groups, int dtype, void* stream) {
  clear_stale_error();
  if (groups < 1 || groups > 2 || B % (groups > 0 ? groups : 1) != 0) return OCF_EINVAL;
  if (!x || !gy || !gx || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta || !workspace) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (dtype != OCF_F32 && dtype != OCF_BF16)) return OCF_EINVAL;
  if (!bn::dense_cl(x, B, C, H, W) || !bn::dense_cl(gy, B, C, H, W) || !bn::dense_cl(gx, B, C, H, W)) return OCF_EUNSUPPORTED;
  const long R = (long)(B / groups) * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OCF_BF16) {
    const int v = bn::pick_v<bf16>(C, {x, gy, gx});
    if (v == 8) return bn::bwd_v<bf16, 8>(x, nullptr, gy, gx, gamma, save_mean, save_invstd, dgamma, dbeta, workspace, R, C, negative_slope, groups, st, beta);
    if (v == 4) return bn::bwd_v<bf16, 4>(x, nullptr, gy, gx, gamma, save_mean, save_invstd, dgamma, dbeta, workspace, R, C, negative_slope, groups, st, beta);
    return OCF_EUNSUPPORTED;
  }
  if (bn::pick_v<float>(C, {x, gy, gx}) == 4)
    return bn::bwd_v<float, 4>(x, nullptr, gy, gx, gamma, save_mean, save_invstd, dgamma, dbeta, workspace, R, C, negative_slope, groups, st, beta);
  return OCF_EUNSUPPORTED;
}
