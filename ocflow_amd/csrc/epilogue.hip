// Convolution epilogue of the decoder / context / refinement nets: bias add + LeakyReLU in one pass, and its gradient
// (LeakyReLU backward + per-channel bias-gradient reduction) in one pass.
//
// The reference writes  nn.Sequential(nn.Conv2d(..., bias=True), nn.LeakyReLU(0.1))  (cost_volume_flow_net.py:19-21,
// flow_occ_net.py conv(), simple_flow_net.py:55-57); in PyTorch-ROCm that is conv -> add_(bias) -> leaky_relu forward (three
// passes over the activation) and leaky_relu_backward -> sum over (N,H,W) -> two dtype casts backward.  ~60 such sites per
// FlowOccNet step: 0.8 + 0.5 ms forward, 0.7 + 1.2 + 0.7 ms backward of a 19 ms step (profiles/r1_j_torch_op_profile.txt).
//
// Arithmetic matches the unfused chain bit for bit in the forward direction:
//   t = T(x + T(b))         (the bias is rounded to the activation type first, as autocast's cast does; fp32: identity)
//   y = t > 0 ? t : T(t * slope)
// backward: gx = y > 0 ? gy : T(gy * slope)   (sign(y) == sign(t) for slope > 0),  gbias[c] = sum_{n,h,w} gx   (fp32 accumulate,
// returned in fp32: one rounding less than the reference's bf16 sum followed by a cast).
#include "ocf_common.h"

namespace ocf {

template <typename T> __device__ __forceinline__ float rnd(float v) { return (float)(T)v; }

// U == Unit<T>::U: one 16-byte unit per access; U == 1: one element (channel counts that are not a multiple of the unit)
template <typename T, int U> __device__ __forceinline__ void ld_u(const T* base, long i, float* f) {
  if (U == 1) f[0] = (float)base[i];
  else unit_to_f<T>(((const u32x4*)base)[i], f);
}
template <typename T, int U> __device__ __forceinline__ void st_u(T* base, long i, const float* f) {
  if (U == 1) base[i] = (T)f[0];
  else ((u32x4*)base)[i] = f_to_unit<T>(f);
}

// ---- channels_last dense: the tensor is a [R][C] matrix, C % U == 0.  Thread = one 16-byte unit; consecutive threads walk a row.
template <typename T, int U>
__global__ void __launch_bounds__(256) bias_act_fwd_cl(T* __restrict__ x, const float* __restrict__ bias, long n_units, int G, float slope) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_units; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % G);
    float f[U];
    ld_u<T, U>(x, i, f);
#pragma unroll
    for (int j = 0; j < U; j++) {
      const float t = rnd<T>(f[j] + rnd<T>(bias[cg * U + j]));
      f[j] = t > 0.f ? t : rnd<T>(t * slope);
    }
    st_u<T, U>(x, i, f);
  }
}

// rpb = 256 / G rows of G units each per pass; thread (r = tid / G, cg = tid % G) keeps U fp32 partial sums of its channel
// group.  UNR passes are loaded before any is used (the loop is latency-bound otherwise: 2 loads + 1 store per pass); the
// partial sums are reduced over r through LDS and leave the block as one atomicAdd per channel (consecutive threads ->
// consecutive channels, i.e. whole-cache-line atomics), so the grid is kept at <= 256 blocks.
// PASS: negative_slope == 1 (plain bias epilogue): gx == gy, nothing to read from y or write to gx, only the reduction runs.
// gy_pitch: elements between consecutive pixels of gy (== C for a dense tensor; larger when gy is a channel slice of the gradient
// of a concatenation -- the dense blocks of the decoders -- whose pixels start at any 2-byte aligned address: its units are then
// read with unaligned 16-byte loads instead of being copied to a dense tensor first).
template <typename T, int U, bool PASS>
__global__ void __launch_bounds__(256) bias_act_bwd_cl(const T* __restrict__ y, const T* __restrict__ gy, T* __restrict__ gx,
                                                       float* __restrict__ gbias, long R, int G, float slope, long gy_pitch) {
  constexpr int UNR = (U == 1) ? 8 : 4;
  __shared__ float red[256][U + 1];
  const int rpb = 256 / G;               // rows per pass
  const int r = threadIdx.x / G, cg = threadIdx.x - r * G;
  float acc[U];
#pragma unroll
  for (int j = 0; j < U; j++) acc[j] = 0.f;
  if (r < rpb) {
    const long step = (long)gridDim.x * rpb;
    for (long row0 = (long)blockIdx.x * rpb + r; row0 < R; row0 += step * UNR) {
      float fy[UNR][U], fg[UNR][U];
#pragma unroll
      for (int k = 0; k < UNR; k++) {
        // branch-free loads (rows past the end re-read the last row and are ignored below): a load inside a divergent branch is
        // followed by s_waitcnt vmcnt(0) at the join, which serialises the UNR loads behind each other
        const long row = row0 + k * step < R ? row0 + k * step : R - 1;
        if (!PASS) ld_u<T, U>(y, row * G + cg, fy[k]);
        if (U == 1) fg[k][0] = (float)gy[row * gy_pitch + cg];
        else unit_to_f<T>(*(const u32x4*)(gy + row * gy_pitch + (long)cg * U), fg[k]);
      }
#pragma unroll
      for (int k = 0; k < UNR; k++) {
        const long row = row0 + k * step;
        if (row < R) {
#pragma unroll
          for (int j = 0; j < U; j++) {
            if (!PASS) fg[k][j] = fy[k][j] > 0.f ? fg[k][j] : rnd<T>(fg[k][j] * slope);
            acc[j] += fg[k][j];
          }
          if (!PASS) st_u<T, U>(gx, row * G + cg, fg[k]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < U; j++) red[threadIdx.x][j] = acc[j];
  __syncthreads();
  // thread t < G*U sums channel t (group t / U, lane t % U) over the rpb row slots
  for (int t = threadIdx.x; t < G * U; t += 256) {
    const int g = t / U, j = t - g * U;
    float s = 0.f;
    for (int rr = 0; rr < rpb; rr++) s += red[rr * G + g][j];
    atomicAdd(gbias + t, s);
  }
}

// ---- NCHW (stride_w == 1, dense planes): block = a chunk of one (b, c) plane
template <typename T>
__global__ void __launch_bounds__(256) bias_act_fwd_nchw(T* __restrict__ x, const float* __restrict__ bias, int C, long HW, int chunks,
                                                         float slope) {
  const long plane = blockIdx.x / chunks;
  const int chunk = blockIdx.x - (int)(plane * chunks);
  const int c = (int)(plane % C);
  const float b = rnd<T>(bias[c]);
  const long per = (HW + chunks - 1) / chunks;
  const long lo = chunk * per, hi = lo + per < HW ? lo + per : HW;
  T* p = x + plane * HW;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const float t = rnd<T>((float)p[i] + b);
    p[i] = (T)(t > 0.f ? t : rnd<T>(t * slope));
  }
}

template <typename T, bool PASS>
__global__ void __launch_bounds__(256) bias_act_bwd_nchw(const T* __restrict__ y, const T* __restrict__ gy, T* __restrict__ gx,
                                                         float* __restrict__ gbias, int C, long HW, int chunks, float slope) {
  const long plane = blockIdx.x / chunks;
  const int chunk = blockIdx.x - (int)(plane * chunks);
  const int c = (int)(plane % C);
  const long per = (HW + chunks - 1) / chunks;
  const long lo = chunk * per, hi = lo + per < HW ? lo + per : HW;
  const long base = plane * HW;
  float acc = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const float g = (float)gy[base + i];
    float v = g;
    if (!PASS) {
      v = (float)y[base + i] > 0.f ? g : rnd<T>(g * slope);
      gx[base + i] = (T)v;
    }
    acc += v;
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(gbias + c, part[0] + part[1] + part[2] + part[3]);
}

enum { LAYOUT_NONE = 0, LAYOUT_CL = 1, LAYOUT_CL1 = 3, LAYOUT_NCHW = 2 };
// dense channels_last ([B][H][W][C]; LAYOUT_CL: C a multiple of the 16-byte unit, LAYOUT_CL1: any C <= 256, element
// accesses) or dense NCHW; anything else is refused: the operand is the fresh output of a convolution, never a view.
template <typename T>
static int dense_layout(const T4& t, int B, int C, int H, int W) {
  constexpr int U = Unit<T>::U;
  const bool cl = (t.sc == 1 || C == 1) && (t.sw == C || W == 1) && (t.sh == (long)W * C || H == 1) && (t.sb == (long)H * W * C || B == 1);
  if (cl && C % U == 0 && C / U <= 256 && ((uintptr_t)t.p & 15) == 0) return LAYOUT_CL;
  if (cl && C <= 256 && C > 1) return LAYOUT_CL1;
  const bool nchw = (t.sw == 1 || W == 1) && (t.sh == W || H == 1) && (t.sc == (long)H * W || C == 1) && (t.sb == (long)C * H * W || B == 1);
  if (nchw) return LAYOUT_NCHW;
  return LAYOUT_NONE;
}

template <typename T>
static int bias_act_fwd_t(const T4& x, const float* bias, int B, int C, int H, int W, float slope, hipStream_t st) {
  constexpr int U = Unit<T>::U;
  const int lay = dense_layout<T>(x, B, C, H, W);
  if (lay == LAYOUT_CL) {
    const long n_units = (long)B * H * W * (C / U);
    const long want = (n_units + 255) / 256;
    const int grid = (int)(want < 8192 ? want : 8192);
    hipLaunchKernelGGL((bias_act_fwd_cl<T, U>), dim3(grid), dim3(256), 0, st, (T*)x.p, bias, n_units, C / U, slope);
  } else if (lay == LAYOUT_CL1) {
    const long n = (long)B * H * W * C;
    const long want = (n + 255) / 256;
    const int grid = (int)(want < 16384 ? want : 16384);
    hipLaunchKernelGGL((bias_act_fwd_cl<T, 1>), dim3(grid), dim3(256), 0, st, (T*)x.p, bias, n, C, slope);
  } else if (lay == LAYOUT_NCHW) {
    const long HW = (long)H * W;
    int chunks = (int)((HW + 2047) / 2048);
    const long blocks = (long)B * C * chunks;
    if (blocks > 0x7fffffffL) return OCF_EUNSUPPORTED;
    hipLaunchKernelGGL(bias_act_fwd_nchw<T>, dim3((unsigned)blocks), dim3(256), 0, st, (T*)x.p, bias, C, HW, chunks, slope);
  } else {
    return OCF_EUNSUPPORTED;
  }
  return launch_status();
}

// pixel pitch of a channel-contiguous tensor whose pixels are laid out in dense (b, y, x) order, 0 if it is not one
static long uniform_pixel_pitch(const T4& t, int B, int C, int H, int W) {
  if (!(t.sc == 1 || C == 1)) return 0;
  // strides of size-1 dimensions carry no information: take the pitch from the innermost dimension that has extent
  const long p = W > 1 ? t.sw : (H > 1 ? t.sh : (B > 1 ? t.sb : (long)C));
  if (p < C) return 0;
  if (W > 1 && H > 1 && t.sh != (long)W * p) return 0;
  if ((W > 1 || H > 1) && B > 1 && t.sb != (long)H * W * p) return 0;
  return p;
}

template <typename T, bool PASS>
static int bias_act_bwd_t(const T4& y, const T4& gy, const T4& gx, float* gbias, int B, int C, int H, int W, float slope, bool accumulate,
                          hipStream_t st) {
  constexpr int U = Unit<T>::U;
  // the layout is decided by the tensors this launch owns (y / gx; gy alone in the pass-through form); gy may additionally be a
  // channel slice with a larger pixel pitch when the others are channels_last
  int lay = PASS ? dense_layout<T>(gy, B, C, H, W) : dense_layout<T>(y, B, C, H, W);
  long gy_pitch = C;
  if (!PASS && dense_layout<T>(gx, B, C, H, W) != lay) return OCF_EUNSUPPORTED;
  if (lay == LAYOUT_CL || lay == LAYOUT_CL1 || (PASS && lay == LAYOUT_NONE)) {
    gy_pitch = uniform_pixel_pitch(gy, B, C, H, W);
    if (gy_pitch == 0) return OCF_EUNSUPPORTED;
    if (PASS && lay == LAYOUT_NONE) lay = (C % U == 0 && C / U <= 256) ? LAYOUT_CL : (C <= 256 && C > 1 ? LAYOUT_CL1 : LAYOUT_NONE);
  } else if (lay == LAYOUT_NCHW) {
    if (dense_layout<T>(gy, B, C, H, W) != LAYOUT_NCHW) return OCF_EUNSUPPORTED;
  }
  if (lay == LAYOUT_NONE) return OCF_EUNSUPPORTED;
  if (!accumulate && hipMemsetAsync(gbias, 0, sizeof(float) * C, st) != hipSuccess) return OCF_ELAUNCH;
  if (lay == LAYOUT_CL || lay == LAYOUT_CL1) {
    const int G = lay == LAYOUT_CL ? C / U : C;
    const int rpb = 256 / G;
    const long R = (long)B * H * W;
    // one unrolled iteration (4-8 passes) per thread at least, at most 256 blocks (1 per CU): every block ends in per-channel
    // atomics on the SAME C addresses, and that tail grows with the block count -- measured at 8x128x109x256: 43.0 / 31.3 / 36.0 /
    // 46.9 us for 128 / 256 / 512 / 1024 blocks
    long want = (R + (long)rpb * 8 - 1) / ((long)rpb * 8);
    const int grid = (int)(want < 1 ? 1 : (want < 256 ? want : 256));
    if (lay == LAYOUT_CL)
      hipLaunchKernelGGL((bias_act_bwd_cl<T, U, PASS>), dim3(grid), dim3(256), 0, st, (const T*)y.p, (const T*)gy.p, (T*)gx.p, gbias, R, G, slope, gy_pitch);
    else
      hipLaunchKernelGGL((bias_act_bwd_cl<T, 1, PASS>), dim3(grid), dim3(256), 0, st, (const T*)y.p, (const T*)gy.p, (T*)gx.p, gbias, R, G, slope, gy_pitch);
  } else {
    const long HW = (long)H * W;
    int chunks = (int)((HW + 2047) / 2048);  // 8 elements per thread: these are 1- and 2-channel maps, parallelism comes from chunks
    const long blocks = (long)B * C * chunks;
    if (blocks > 0x7fffffffL) return OCF_EUNSUPPORTED;
    hipLaunchKernelGGL((bias_act_bwd_nchw<T, PASS>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)y.p, (const T*)gy.p, (T*)gx.p, gbias,
                       C, HW, chunks, slope);
  }
  return launch_status();
}

}  // namespace ocf

using namespace ocf;

extern "C" int ocf_bias_act_fwd(const ocf_tensor4* x, const float* bias, int B, int C, int H, int W, float negative_slope, int dtype,
                                void* stream) {
  if (!x || !x->data || !bias || B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if (dtype != OCF_F32 && dtype != OCF_BF16) return OCF_EINVAL;
  clear_stale_error();
  ScopedProfile prof((hipStream_t)stream);
  const T4 tx = mk(x);
  return dtype == OCF_F32 ? bias_act_fwd_t<float>(tx, bias, B, C, H, W, negative_slope, (hipStream_t)stream)
                          : bias_act_fwd_t<bf16>(tx, bias, B, C, H, W, negative_slope, (hipStream_t)stream);
}

extern "C" int ocf_bias_act_bwd(const ocf_tensor4* y, const ocf_tensor4* gy, const ocf_tensor4* gx, float* gbias, int B, int C, int H,
                                int W, float negative_slope, int accumulate, int dtype, void* stream) {
  if (!gy || !gy->data || !gbias || B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if (dtype != OCF_F32 && dtype != OCF_BF16) return OCF_EINVAL;
  // the input gradient is only a separate tensor when the activation has a slope: with negative_slope == 1 it IS gy
  const bool pass = (gx == nullptr || gx->data == nullptr);
  if (pass && negative_slope != 1.0f) return OCF_EINVAL;
  if (!pass && (!y || !y->data)) return OCF_EINVAL;
  clear_stale_error();
  ScopedProfile prof((hipStream_t)stream);
  const T4 ty = mk(y), tg = mk(gy), tx = mk(gx);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == OCF_F32)
    return pass ? bias_act_bwd_t<float, true>(ty, tg, tx, gbias, B, C, H, W, negative_slope, accumulate != 0, st)
                : bias_act_bwd_t<float, false>(ty, tg, tx, gbias, B, C, H, W, negative_slope, accumulate != 0, st);
  return pass ? bias_act_bwd_t<bf16, true>(ty, tg, tx, gbias, B, C, H, W, negative_slope, accumulate != 0, st)
              : bias_act_bwd_t<bf16, false>(ty, tg, tx, gbias, B, C, H, W, negative_slope, accumulate != 0, st);
}
