// internal launch entry points shared between the correlation translation units
#pragma once
#include <hip/hip_runtime.h>
#include "ocflow_hip.h"

namespace ocf {
// LeakyReLU slope the forward kernels apply to the scaled cost volume (1 = none): set by ocf_corr_fwd for the duration of ONE call on the
// calling thread and read by the launch functions of that call (host side only; the kernels take it as an argument)
inline float& corr_fwd_lrelu() {
  static thread_local float slope = 1.f;
  return slope;
}
struct ScopedCorrLrelu {
  float prev;
  explicit ScopedCorrLrelu(float s) : prev(corr_fwd_lrelu()) { corr_fwd_lrelu() = s; }
  ~ScopedCorrLrelu() { corr_fwd_lrelu() = prev; }
};
// bf16 forward on the matrix cores (corr_mfma.hip)
int corr_fwd_bf16_mfma(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ,
                       const ocf_tensor4* out, int B, int C, int H, int W, int d, float fs, int ac, int policy, hipStream_t st);
// which kernel a bf16 forward request gets (enum ocf_corr_variant), without launching anything
int corr_fwd_bf16_variant(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ,
                          const ocf_tensor4* out, int B, int C, int H, int W, int d, int policy);
// bf16 operands without 8-/16-byte channel runs (NCHW, odd strides): the vector-ALU kernel, element loads (corr.hip)
int corr_fwd_bf16_vector(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ,
                         const ocf_tensor4* out, int B, int C, int H, int W, int d, float fs, int ac, hipStream_t st);
// the row / direct kernels (corr_rows.hip): dense channels_last bf16 output, d = 4, no in-kernel warp; 0 = not eligible
int corr_rows_variant(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* out, int C, int H, int W,
                      int d);
int corr_fwd_bf16_rows(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* occ, const ocf_tensor4* out, int B, int C, int H,
                       int W, hipStream_t st, const ocf_tensor4* flow = nullptr, float fs = 1.f, int ac = 0);
// the wave kernel (corr_wave.hip): small levels (C > 64), any output pixel pitch, optional in-register warp gather of the second operand
bool corr_wave_eligible(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ, const ocf_tensor4* out,
                        int B, int C, int H, int W, int d);
int corr_fwd_bf16_wave(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ, const ocf_tensor4* out,
                       int B, int C, int H, int W, float fs, int ac, hipStream_t st);
// matrix-core backward (corr_bwd_mfma.hip): bf16, d = 4, channel-contiguous operands with 8-byte aligned pixels
bool corr_bwd_mfma_eligible(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* g, const ocf_tensor4* g1, const ocf_tensor4* g2,
                            int B, int C, int H, int W, int d, int policy);
int corr_bwd_bf16_mfma(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* g, const ocf_tensor4* g1, const ocf_tensor4* g2,
                       int B, int C, int H, int W, hipStream_t st, const ocf_tensor4* occ, const ocf_tensor4* gocc);
}  // namespace ocf
