// Adam over FLAT buffers: parameters (fp32 masters), gradients, both moments and the bf16 compute copies of a whole network as five
// arrays with one element order, updated by ONE launch.  The framework's multi-tensor Adam walks the 229 parameter tensors of
// FlowOccNet in chunked launches (6 x 45 us + 3 copies of the masters into the bf16 weights: 349 us per step for 6.9 M parameters,
// ~240 MB of traffic = 35 us at the HBM rate).
// Update rule = torch.optim.Adam (reference: models/flow_model.py:105-106 `torch.optim.Adam(self.parameters(), lr=...)`), in the form
// of the framework's fused kernel: m <- lerp(m, g, 1 - b1); v <- b2 v + (1 - b2) g^2; p <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// (weight decay, when given, is added to the gradient first: Adam, not AdamW).  t is read from device memory (already incremented).
#include "ocf_common.h"

namespace ocf {
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16* __restrict__ pb, long n4, float lr, double beta1,
                                                        double beta2, float eps, float wd, const float* __restrict__ step) {
  const double t = (double)*step;
  const float bc1 = (float)(1.0 - pow(beta1, t)), bc2s = sqrtf((float)(1.0 - pow(beta2, t)));
  const float step_size = lr / bc1, b1 = (float)beta1, b2 = (float)beta2, omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 pp = ((const f32x4*)p)[i], gg = ((const f32x4*)g)[i], mm = ((const f32x4*)m)[i], vv = ((const f32x4*)v)[i];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float gk = gg[k];
      if (wd != 0.f) gk = __builtin_fmaf(pp[k], wd, gk);
      mm[k] = mm[k] + omb1 * (gk - mm[k]);
      vv[k] = b2 * vv[k] + omb2 * gk * gk;
      const float denom = sqrtf(vv[k]) / bc2s + eps;
      pp[k] = pp[k] - step_size * (mm[k] / denom);
    }
    ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
    if (pb) {
      const bf16x2 lo = {(bf16)pp[0], (bf16)pp[1]}, hi = {(bf16)pp[2], (bf16)pp[3]};
      ((u32x2*)pb)[i] = u32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    }
  }
}
}  // namespace ocf

extern "C" int ocf_adam_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* params_bf16, int64_t n, float lr,
                             double beta1, double beta2, float eps, float weight_decay, const float* step, void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!params || !grads || !exp_avg || !exp_avg_sq || !step || n <= 0) return OCF_EINVAL;
  if (n % 4 != 0 || ((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 != 0 ||
      (params_bf16 && (uintptr_t)params_bf16 % 8 != 0))
    return OCF_EUNSUPPORTED;
  const long n4 = n / 4, blocks = (n4 + 255) / 256;
  LaunchProf lp;
  lp.launch(true, adam_flat_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
            exp_avg_sq, (bf16*)params_bf16, n4, lr, beta1, beta2, eps, weight_decay, step);
  return launch_status();
}
