// Shared device helpers for the OCFlow gfx950 kernels (wave = 64 lanes, 16-byte channel units).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "ocflow_hip.h"

namespace ocf {

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Device-side view of ocf_tensor4 (passed by value in the kernel argument segment).
struct T4 {
  char* p;
  long sb, sc, sh, sw;  // element strides
};
static inline T4 mk(const ocf_tensor4* t) {
  T4 r;
  if (t) { r.p = (char*)t->data; r.sb = t->sb; r.sc = t->sc; r.sh = t->sh; r.sw = t->sw; }
  else   { r.p = nullptr; r.sb = r.sc = r.sh = r.sw = 0; }
  return r;
}

// One 16-byte "unit" = U consecutive channels of one pixel: 4 x fp32 or 8 x bf16.
template <typename T> struct Unit;
template <> struct Unit<float> { static constexpr int U = 4; };
template <> struct Unit<bf16>  { static constexpr int U = 8; };

template <typename T> __device__ __forceinline__ float ldf(const char* base, long eoff) {
  return (float)(((const T*)base)[eoff]);
}
template <typename T> __device__ __forceinline__ void stf(char* base, long eoff, float v) {
  ((T*)base)[eoff] = (T)v;
}

// unit (u32x4 bits) <-> U floats
template <typename T> __device__ __forceinline__ void unit_to_f(u32x4 v, float* f);
template <> __device__ __forceinline__ void unit_to_f<float>(u32x4 v, float* f) {
  f32x4 x = __builtin_bit_cast(f32x4, v);
  f[0] = x[0]; f[1] = x[1]; f[2] = x[2]; f[3] = x[3];
}
template <> __device__ __forceinline__ void unit_to_f<bf16>(u32x4 v, float* f) {
  // bf16 -> f32 is a 16-bit shift
  f[0] = __builtin_bit_cast(float, v[0] << 16); f[1] = __builtin_bit_cast(float, v[0] & 0xffff0000u);
  f[2] = __builtin_bit_cast(float, v[1] << 16); f[3] = __builtin_bit_cast(float, v[1] & 0xffff0000u);
  f[4] = __builtin_bit_cast(float, v[2] << 16); f[5] = __builtin_bit_cast(float, v[2] & 0xffff0000u);
  f[6] = __builtin_bit_cast(float, v[3] << 16); f[7] = __builtin_bit_cast(float, v[3] & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ u32x4 f_to_unit(const float* f);
template <> __device__ __forceinline__ u32x4 f_to_unit<float>(const float* f) {
  f32x4 x = {f[0], f[1], f[2], f[3]};
  return __builtin_bit_cast(u32x4, x);
}
template <> __device__ __forceinline__ u32x4 f_to_unit<bf16>(const float* f) {
  bf16x8 x = {(bf16)f[0], (bf16)f[1], (bf16)f[2], (bf16)f[3], (bf16)f[4], (bf16)f[5], (bf16)f[6], (bf16)f[7]};
  return __builtin_bit_cast(u32x4, x);
}

// dot product of two units, fp32 accumulate.  bf16: 4 x v_dot2c_f32_bf16.
template <typename T> __device__ __forceinline__ float unit_dot(u32x4 a, u32x4 b, float acc);
template <> __device__ __forceinline__ float unit_dot<float>(u32x4 a, u32x4 b, float acc) {
  f32x4 x = __builtin_bit_cast(f32x4, a), y = __builtin_bit_cast(f32x4, b);
  acc = __builtin_fmaf(x[0], y[0], acc); acc = __builtin_fmaf(x[1], y[1], acc);
  acc = __builtin_fmaf(x[2], y[2], acc); acc = __builtin_fmaf(x[3], y[3], acc);
  return acc;
}
template <> __device__ __forceinline__ float unit_dot<bf16>(u32x4 a, u32x4 b, float acc) {
  bf16x8 x = __builtin_bit_cast(bf16x8, a), y = __builtin_bit_cast(bf16x8, b);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(x, x, 0, 1), __builtin_shufflevector(y, y, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(x, x, 2, 3), __builtin_shufflevector(y, y, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(x, x, 4, 5), __builtin_shufflevector(y, y, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(x, x, 6, 7), __builtin_shufflevector(y, y, 6, 7), acc, false);
  return acc;
}

// Load the unit holding channels [c0, c0+U) of pixel (b,y,x); channels >= C read as 0.
// VEC = 16: stride_c == 1 and pixels are 16-byte aligned; VEC = 8: 8-byte aligned (e.g. C = 196 bf16 = 392-byte pixels);
// VEC = 0: arbitrary strides, one scalar load per channel.  (bool true/false still mean 16/0.)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <typename T, int VEC>
__device__ __forceinline__ u32x4 load_unit(const T4& t, long pix_off, int c0, int C) {
  constexpr int U = Unit<T>::U;
  if (VEC == 16 || VEC == 1) {
    if (c0 + U <= C) return *(const u32x4*)(t.p + (pix_off + c0) * (long)sizeof(T));
  } else if (VEC == 8) {
    if (c0 + U <= C) {
      const u32x2* q = (const u32x2*)(t.p + (pix_off + c0) * (long)sizeof(T));
      const u32x2 lo = q[0], hi = q[1];
      u32x4 r = {lo[0], lo[1], hi[0], hi[1]};
      return r;
    }
  }
  float f[U];
#pragma unroll
  for (int j = 0; j < U; j++) f[j] = (c0 + j < C) ? ldf<T>(t.p, pix_off + (long)(c0 + j) * t.sc) : 0.f;
  return f_to_unit<T>(f);
}
template <typename T, int VEC>
__device__ __forceinline__ void load_unit_f(const T4& t, long pix_off, int c0, int C, float* f) {
  constexpr int U = Unit<T>::U;
  if (VEC != 0) {
    if (c0 + U <= C) { unit_to_f<T>(load_unit<T, VEC>(t, pix_off, c0, C), f); return; }
  }
#pragma unroll
  for (int j = 0; j < U; j++) f[j] = (c0 + j < C) ? ldf<T>(t.p, pix_off + (long)(c0 + j) * t.sc) : 0.f;
}
template <typename T, int VEC>
__device__ __forceinline__ void store_unit_f(const T4& t, long pix_off, int c0, int C, const float* f) {
  constexpr int U = Unit<T>::U;
  if (VEC == 16 || VEC == 1) {
    if (c0 + U <= C) { *(u32x4*)(t.p + (pix_off + c0) * (long)sizeof(T)) = f_to_unit<T>(f); return; }
  } else if (VEC == 8) {
    if (c0 + U <= C) {
      const u32x4 v = f_to_unit<T>(f);
      u32x2* q = (u32x2*)(t.p + (pix_off + c0) * (long)sizeof(T));
      u32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
      q[0] = lo; q[1] = hi;
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < U; j++) if (c0 + j < C) stf<T>(t.p, pix_off + (long)(c0 + j) * t.sc, f[j]);
}

// vector-only unit load for kernels whose host side guarantees C % U == 0 whenever VEC != 0 (no per-channel tail path:
// the tail code costs ~100 VGPRs of live scalar loads in the staging loops)
template <typename T, int VEC>
__device__ __forceinline__ u32x4 load_unit_full(const T4& t, int pix_off, int c0, int C) {
  if (VEC == 16) return *(const u32x4*)(t.p + (long)(pix_off + c0) * (long)sizeof(T));
  if (VEC == 8) {
    // 8-byte granularity: the host guarantees (C * sizeof(T)) % 8 == 0, so a partial last unit is exactly its first half
    const u32x2* q = (const u32x2*)(t.p + (long)(pix_off + c0) * (long)sizeof(T));
    const u32x2 lo = q[0];
    u32x2 hi = {0u, 0u};
    if ((c0 + Unit<T>::U / 2) < C) hi = q[1];
    u32x4 r = {lo[0], lo[1], hi[0], hi[1]};
    return r;
  }
  return load_unit<T, 0>(t, pix_off, c0, C);
}
template <typename T, int VEC>
__device__ __forceinline__ void load_unit_full_f(const T4& t, int pix_off, int c0, int C, float* f) {
  unit_to_f<T>(load_unit_full<T, VEC>(t, pix_off, c0, C), f);
}

// widest vector width (bytes) usable for the channel units of tensor t: 16, 8 or 0
template <typename T> static inline int vec_width(const ocf_tensor4* t) {
  if (!t || t->sc != 1) return 0;
  const long es = sizeof(T);
  auto ok = [&](long a) { return ((uintptr_t)t->data % a == 0) && (t->sw * es) % a == 0 && (t->sh * es) % a == 0 && (t->sb * es) % a == 0; };
  return ok(16) ? 16 : (ok(8) ? 8 : 0);
}
static inline int min_vec(int a, int b) { return a < b ? a : b; }

// Sampling geometry of the backward warp: source coordinate of output pixel (x,y).
struct WarpGeom {
  float ax, bx, ay, by;  // px = (x+u)*ax + bx
  __host__ __device__ WarpGeom(int H, int W, int align_corners) {
    float wd = (float)(W > 1 ? W - 1 : 1), hd = (float)(H > 1 ? H - 1 : 1);
    if (align_corners) { ax = (float)(W - 1) / wd; bx = 0.f; ay = (float)(H - 1) / hd; by = 0.f; }
    else               { ax = (float)W / wd;       bx = -0.5f; ay = (float)H / hd;     by = -0.5f; }
  }
};

struct Bilin {
  int x0, y0;
  float wx1, wy1;   // fractional parts; weights are (1-wx1, wx1) x (1-wy1, wy1)
  bool vx0, vx1, vy0, vy1;
};
__device__ __forceinline__ Bilin bilin_setup(float px, float py, int H, int W) {
  Bilin s;
  float fx = floorf(px), fy = floorf(py);
  s.wx1 = px - fx; s.wy1 = py - fy;
  // clamp before the int conversion so huge/NaN coordinates cannot overflow
  fx = fminf(fmaxf(fx, -2.f), (float)W + 1.f);
  fy = fminf(fmaxf(fy, -2.f), (float)H + 1.f);
  s.x0 = (int)fx; s.y0 = (int)fy;
  s.vx0 = (s.x0 >= 0) & (s.x0 < W); s.vx1 = (s.x0 + 1 >= 0) & (s.x0 + 1 < W);
  s.vy0 = (s.y0 >= 0) & (s.y0 < H); s.vy1 = (s.y0 + 1 >= 0) & (s.y0 + 1 < H);
  if (!(px == px) || !(py == py)) { s.vx0 = s.vx1 = s.vy0 = s.vy1 = false; s.wx1 = s.wy1 = 0.f; }
  return s;
}

// wave-level sum over 64 lanes (result valid in every lane)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Bijective XCD-aware remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each
// XCD a contiguous range of logical tile ids (neighbouring tiles share halos in that XCD's L2).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// hipGetLastError() is sticky per thread: a failed query inside the caller's framework (e.g. hipPointerGetAttributes
// on pageable host memory during a torch H2D copy) would otherwise be reported as OUR launch failing.
static inline void clear_stale_error() { (void)hipGetLastError(); }
// one-shot event bracketing requested through ocf_profile_next_launch() (defined in losses.hip)
struct ProfileSlot { hipEvent_t start, stop; };
ProfileSlot& profile_slot();
struct ScopedProfile {
  hipStream_t st;
  hipEvent_t stop;
  explicit ScopedProfile(hipStream_t s) : st(s), stop(nullptr) {
    ProfileSlot& p = profile_slot();
    if (p.start) { (void)hipEventRecord(p.start, st); }
    stop = p.stop;
    p.start = nullptr; p.stop = nullptr;
  }
  ~ScopedProfile() { if (stop) (void)hipEventRecord(stop, st); }
};
// One-shot dispatch-stamped timing for launchers with one or several kernels: the request of ocf_profile_next_launch() is taken
// once; the FIRST kernel launched through it gets the start event, a kernel launched with last = true the stop event
// (hipExtLaunchKernelGGL stamps them with the dispatch's own begin / end: what rocprofv3's kernel trace reports).
struct LaunchProf {
  hipEvent_t start, stop;
  LaunchProf() {
    ProfileSlot& p = profile_slot();
    start = p.start; stop = p.stop;
    p.start = nullptr; p.stop = nullptr;
  }
  template <typename... KArgs, typename... Args>
  void launch(bool last, void (*kern)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    hipEvent_t e0 = start, e1 = last ? stop : nullptr;
    start = nullptr;
    if (e0 || e1) hipExtLaunchKernelGGL(kern, grid, block, (unsigned)lds, st, e0, e1, 0, args...);
    else hipLaunchKernelGGL(kern, grid, block, (unsigned)lds, st, args...);
  }
};
static inline int launch_status() { return hipGetLastError() == hipSuccess ? OCF_OK : OCF_ELAUNCH; }

}  // namespace ocf
