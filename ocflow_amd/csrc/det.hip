// Deterministic variants of the two scatter kernels ("race detection / sanitizers": the float-atomic kernels sum
// in arrival order, so two runs differ in the last bits).  Same arithmetic per contribution; the accumulation goes through 64-bit
// FIXED-POINT integer atomics (2^-36 resolution, range +-1.3e8), which are associative: the result is bit-identical from run to
// run and independent of dispatch order.  A second pass converts the integer image to fp32.  Slower than the float-atomic kernels
// (twice the atomic bytes + a conversion pass): selected explicitly (ops.set_deterministic / ocf_*_det), never by default.
// reference semantics: warp backward = autograd of F.grid_sample (cost_volume_flow_net.py:121-151, models/model.py:191-221);
//                      range map = models/model.py:243-305
#include "ocf_common.h"

namespace ocf {

constexpr double DET_SCALE = 68719476736.0;  // 2^36
typedef unsigned long long u64;

__device__ __forceinline__ void det_add(u64* p, float v) {
  if (v != 0.f) atomicAdd(p, (u64)(long long)__double2ll_rn((double)v * DET_SCALE));  // two's complement wrap == signed add
}

template <typename T>
__global__ __launch_bounds__(256) void warp_bwd_det_kernel(T4 img, T4 flow, T4 gout, u64* wimg, u64* wflow, int B, int C, int H, int W,
                                                           float fs, WarpGeom wg, long total) {
  const float* fl = (const float*)flow.p;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    // (pixel, channel), channel fastest for channel-contiguous tensors, x fastest otherwise
    int c, x, y, b;
    long r = idx;
    if (img.sc == 1) { c = r % C; r /= C; x = r % W; r /= W; y = r % H; b = r / H; }
    else             { x = r % W; r /= W; y = r % H; r /= H; c = r % C; b = r / C; }
    const long fo = (long)b * flow.sb + (long)y * flow.sh + (long)x * flow.sw;
    const float uu = fl[fo] * fs, vv = fl[fo + flow.sc] * fs;
    const Bilin s = bilin_setup(((float)x + uu) * wg.ax + wg.bx, ((float)y + vv) * wg.ay + wg.by, H, W);
    const float g = ldf<T>(gout.p, (long)b * gout.sb + (long)c * gout.sc + (long)y * gout.sh + (long)x * gout.sw);
    float v[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int ddx = n & 1, ddy = n >> 1;
      const bool ok = (ddx ? s.vx1 : s.vx0) & (ddy ? s.vy1 : s.vy0);
      v[n] = 0.f;
      if (ok) {
        const int yy = s.y0 + ddy, xx = s.x0 + ddx;
        v[n] = ldf<T>(img.p, (long)b * img.sb + (long)c * img.sc + (long)yy * img.sh + (long)xx * img.sw);
        if (wimg) {
          const float w = (ddx ? s.wx1 : 1.f - s.wx1) * (ddy ? s.wy1 : 1.f - s.wy1);
          det_add(wimg + (((long)b * C + c) * H + yy) * W + xx, w * g);
        }
      }
    }
    if (wflow) {
      const float gx = g * ((1.f - s.wy1) * (v[1] - v[0]) + s.wy1 * (v[3] - v[2]));
      const float gy = g * ((1.f - s.wx1) * (v[2] - v[0]) + s.wx1 * (v[3] - v[1]));
      u64* wf = wflow + ((long)b * 2 * H + y) * W + x;
      det_add(wf, gx * wg.ax * fs);
      det_add(wf + (long)H * W, gy * wg.ay * fs);
    }
  }
}

// fixed-point image [B, C, H, W] (dense NCHW order) -> fp32 tensor of any strides; occ_out (optional): 1 - clamp(value, 0, 1)
__global__ __launch_bounds__(256) void det_convert_kernel(const u64* w, T4 dst, float* occ_out, int C, int H, int W, long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int x = r % W; r /= W;
    const int y = r % H; r /= H;
    const int c = r % C;
    const int b = r / C;
    const float v = (float)((double)(long long)w[idx] * (1.0 / DET_SCALE));
    ((float*)dst.p)[(long)b * dst.sb + (long)c * dst.sc + (long)y * dst.sh + (long)x * dst.sw] = v;
    if (occ_out) occ_out[idx] = 1.f - fminf(fmaxf(v, 0.f), 1.f);
  }
}

__global__ __launch_bounds__(256) void range_splat_det_kernel(T4 flow, u64* wr, int B, int H, int W, long npix) {
  const float* fl = (const float*)flow.p;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < npix; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W;
    long r = idx / W;
    const int y = r % H, b = r / H;
    const long fo = (long)b * flow.sb + (long)y * flow.sh + (long)x * flow.sw;
    const float cx = (float)x + fl[fo], cy = (float)y + fl[fo + flow.sc];
    if (!(cx == cx) || !(cy == cy)) continue;
    float fx = floorf(cx), fy = floorf(cy);
    const float rx = cx - fx, ry = cy - fy;
    fx = fminf(fmaxf(fx, -2.f), (float)W + 1.f);
    fy = fminf(fmaxf(fy, -2.f), (float)H + 1.f);
    const int x0 = (int)fx, y0 = (int)fy;
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int di = n & 1, dj = n >> 1;
      const int tx = x0 + di, ty = y0 + dj;
      if (tx >= 0 && tx < W && ty >= 0 && ty < H) det_add(wr + ((long)b * H + ty) * W + tx, (di ? rx : 1.f - rx) * (dj ? ry : 1.f - ry));
    }
  }
}

static inline int dgrid(long n) {
  long blocks = (n + 255) / 256;
  return (int)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
}

}  // namespace ocf

using namespace ocf;

extern "C" int ocf_warp_bwd_det_workspace_bytes(int B, int C, int H, int W, int64_t* bytes) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !bytes) return OCF_EINVAL;
  *bytes = 8LL * ((int64_t)B * C * H * W + 2LL * B * H * W);
  return OCF_OK;
}

extern "C" int ocf_range_map_det_workspace_bytes(int B, int H, int W, int64_t* bytes) {
  if (B <= 0 || H <= 0 || W <= 0 || !bytes) return OCF_EINVAL;
  *bytes = 8LL * (int64_t)B * H * W;
  return OCF_OK;
}

extern "C" int ocf_warp_bwd_det(const ocf_tensor4* img, const ocf_tensor4* flow, const ocf_tensor4* gout, const ocf_tensor4* gimg,
                                const ocf_tensor4* gflow, void* workspace, int B, int C, int H, int W, float flow_scale,
                                int align_corners, int dtype, void* stream) {
  clear_stale_error();
  if (!img || !flow || !gout || !workspace || !img->data || !flow->data || !gout->data) return OCF_EINVAL;
  if ((gimg && !gimg->data) || (gflow && !gflow->data) || (!gimg && !gflow)) return OCF_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if (dtype != OCF_F32 && dtype != OCF_BF16) return OCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long nimg = (long)B * C * H * W, nfl = 2L * B * H * W;
  u64* wimg = (u64*)workspace;
  u64* wflow = wimg + nimg;
  if (hipMemsetAsync(workspace, 0, (size_t)(nimg + nfl) * 8, st) != hipSuccess) return OCF_ELAUNCH;
  const WarpGeom wg(H, W, align_corners);
  if (dtype == OCF_F32)
    hipLaunchKernelGGL(warp_bwd_det_kernel<float>, dim3(dgrid(nimg)), dim3(256), 0, st, mk(img), mk(flow), mk(gout), gimg ? wimg : nullptr,
                       gflow ? wflow : nullptr, B, C, H, W, flow_scale, wg, nimg);
  else
    hipLaunchKernelGGL(warp_bwd_det_kernel<bf16>, dim3(dgrid(nimg)), dim3(256), 0, st, mk(img), mk(flow), mk(gout), gimg ? wimg : nullptr,
                       gflow ? wflow : nullptr, B, C, H, W, flow_scale, wg, nimg);
  if (gimg) hipLaunchKernelGGL(det_convert_kernel, dim3(dgrid(nimg)), dim3(256), 0, st, (const u64*)wimg, mk(gimg), (float*)nullptr, C, H, W, nimg);
  if (gflow) hipLaunchKernelGGL(det_convert_kernel, dim3(dgrid(nfl)), dim3(256), 0, st, (const u64*)wflow, mk(gflow), (float*)nullptr, 2, H, W, nfl);
  return launch_status();
}

extern "C" int ocf_range_map_det(const ocf_tensor4* flow, float* range, float* occ, void* workspace, int B, int H, int W, void* stream) {
  clear_stale_error();
  if (!flow || !flow->data || !range || !workspace) return OCF_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long npix = (long)B * H * W;
  if (hipMemsetAsync(workspace, 0, (size_t)npix * 8, st) != hipSuccess) return OCF_ELAUNCH;
  hipLaunchKernelGGL(range_splat_det_kernel, dim3(dgrid(npix)), dim3(256), 0, st, mk(flow), (u64*)workspace, B, H, W, npix);
  ocf_tensor4 dense = {range, (int64_t)H * W, (int64_t)H * W, (int64_t)W, 1};
  hipLaunchKernelGGL(det_convert_kernel, dim3(dgrid(npix)), dim3(256), 0, st, (const u64*)workspace, mk(&dense), occ, 1, H, W, npix);
  return launch_status();
}
