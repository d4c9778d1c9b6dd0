// Bilinear resize of an fp32 image batch that needs no gradient: the unsupervised PWC step's quarter-resolution copy of the first image for
// the smoothness terms (reference models/model.py:63-69: F.interpolate(img1, scale_factor=0.25, mode="bilinear", align_corners=True) on a
// channel slice of the [B, 6, H, W] pair tensor).  The framework's kernel takes 75 us for it at 8 x 3 x 384 x 512 -> 96 x 128 (one thread
// per output pixel looping over batch and channels of a strided slice: 12288 threads); here one thread per output ELEMENT.
// Same arithmetic as the framework's kernel (fp32: source index = scale * dst [align_corners] or max(scale * (dst + 0.5) - 0.5, 0); the four
// taps weighted h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d)).
#include "ocf_common.h"

namespace ocf {
__global__ __launch_bounds__(256) void resize_bilinear_kernel(T4 x, T4 y, int B, int C, int H, int W, int h, int w, float rh, float rw, int ac) {
  const long total = (long)B * C * h * w;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ox = (int)(i % w);
  long r = i / w;
  const int oy = (int)(r % h);
  r /= h;
  const int c = (int)(r % C), b = (int)(r / C);
  const float sy = ac ? rh * oy : fmaxf(rh * (oy + 0.5f) - 0.5f, 0.f);
  const float sx = ac ? rw * ox : fmaxf(rw * (ox + 0.5f) - 0.5f, 0.f);
  const int y1 = (int)sy, x1 = (int)sx;
  const int yp = y1 < H - 1 ? 1 : 0, xp = x1 < W - 1 ? 1 : 0;
  const float l1y = sy - y1, l0y = 1.f - l1y, l1x = sx - x1, l0x = 1.f - l1x;
  const float* src = (const float*)x.p + (long)b * x.sb + (long)c * x.sc + (long)y1 * x.sh + (long)x1 * x.sw;
  const float v00 = src[0], v01 = src[(long)xp * x.sw], v10 = src[(long)yp * x.sh], v11 = src[(long)yp * x.sh + (long)xp * x.sw];
  ((float*)y.p)[(long)b * y.sb + (long)c * y.sc + (long)oy * y.sh + (long)ox * y.sw] = l0y * (l0x * v00 + l1x * v01) + l1y * (l0x * v10 + l1x * v11);
}
}  // namespace ocf

extern "C" int ocf_resize_bilinear(const ocf_tensor4* x, const ocf_tensor4* y, int B, int C, int H, int W, int h, int w, int align_corners,
                                   void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !y || !x->data || !y->data || B <= 0 || C <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return OCF_EINVAL;
  const long total = (long)B * C * h * w;
  if (total >= (1L << 31) * 256) return OCF_EUNSUPPORTED;
  const float rh = align_corners ? (h > 1 ? (float)(H - 1) / (float)(h - 1) : 0.f) : (float)H / (float)h;
  const float rw = align_corners ? (w > 1 ? (float)(W - 1) / (float)(w - 1) : 0.f) : (float)W / (float)w;
  LaunchProf lp;
  lp.launch(true, resize_bilinear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mk(x), mk(y), B, C, H, W, h, w,
            rh, rw, align_corners ? 1 : 0);
  return launch_status();
}
