// internal launch entry points shared between the correlation translation units
#pragma once
#include <hip/hip_runtime.h>
#include "ocflow_hip.h"

namespace ocf {
// bf16 forward on the matrix cores (corr_mfma.hip)
int corr_fwd_bf16_mfma(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ,
                       const ocf_tensor4* out, int B, int C, int H, int W, int d, float fs, int ac, hipStream_t st);
}  // namespace ocf
