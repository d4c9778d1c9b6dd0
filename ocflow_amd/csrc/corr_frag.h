// MFMA operand fragments of the bf16 cost-volume kernels loaded straight from global memory (corr_rows.hip, corr_wave.hip):
// raw load at a clamped (always valid) address, and the mask applied to it AFTER a whole burst of loads has been issued -- a load whose
// only use sits behind a condition is sunk into a branch by hipcc and followed by a wait for EVERY outstanding load.
#pragma once
#include "ocf_common.h"

namespace ocf {

// VEC = 16: 16-byte aligned pixels (C % 8 == 0); VEC = 8: 8-byte aligned (C = 196: 392-byte pixels, the last unit is a half unit)
template <int VEC>
__device__ __forceinline__ u32x4 load_frag_raw(const char* p, int c0, int C) {
  if (VEC == 16) return *(const u32x4*)p;
  const u32x2 lo = *(const u32x2*)p;
  const u32x2 hi = *(const u32x2*)(p + ((c0 + 4 < C) ? 8 : 0));
  return u32x4{lo[0], lo[1], hi[0], hi[1]};
}
template <int VEC>
__device__ __forceinline__ u32x4 mask_frag(u32x4 v, bool ok, int c0, int C) {
  const unsigned m = ok ? 0xffffffffu : 0u, mh = (ok && (VEC == 16 || c0 + 4 < C)) ? 0xffffffffu : 0u;
  return u32x4{v[0] & m, v[1] & m, v[2] & mh, v[3] & mh};
}

}  // namespace ocf
