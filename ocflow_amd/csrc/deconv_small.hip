// The up-sampling layers of the estimators: nn.ConvTranspose2d(Ci, Co, kernel 3, stride 2, padding 1) with Co <= 4 and Ci <= 64 -- `upconv1`
// (flow 2 -> 2, occlusion 16 -> 1) and `upconv2` (features 32 -> 2, occlusion 1 -> 1) of flow_net.py:38-48, flow_occ_net.py:46-66 -- forward,
// input gradient, weight + bias gradient, bf16 pixel-major activations.  With one or two output channels these are memory-bound gathers
// (<= 4 input pixels per output pixel); the library serves them with its vector-1 implicit-GEMM and `naive_conv_*` kernels plus a
// workspace fill and a cast per weight gradient (~90 launches per step for 16 layers).
//
//   y[b,co,oy,ox] = bias[co] + sum_{ci,ky,kx : oy = 2 iy - 1 + ky, ox = 2 ix - 1 + kx} x[b,ci,iy,ix] * w[ci,co,ky,kx]
#include "ocf_common.h"

namespace ocf {
namespace dcv {
constexpr int MAXCO = 4, MAXCI = 64;

struct Args {
  const bf16* x;    // [B][H][W][xp]
  const void* w;    // [Ci][Co][3][3], element strides wsi / wso / wsy / wsx, fp32 or bf16
  const float* bias;
  bf16* y;          // [B][Ho][Wo][yp]
  const bf16* gy;   // backward: gradient of y
  bf16* gx;         // backward: gradient of x [B][H][W][gxp]
  float* gw;        // fp32, element strides gsi / gso / gsy / gsx (accumulated)
  float* gbias;     // fp32 [Co] (accumulated) or null
  const bf16* ya;   // backward with act: the forward's output (sigmoid values), pixel pitch yap
  long xp, yp, gyp, gxp, wsi, wso, wsy, wsx, gsi, gso, gsy, gsx, yap;
  int B, H, W, Ho, Wo, Ci, Co, w_f32;
  int act;          // 0: none; 1: sigmoid applied to the bf16-rounded result (forward) / its derivative applied to gy (backward)
};

__device__ __forceinline__ float wload(const Args& a, int ci, int co, int ky, int kx) {
  const long off = ci * a.wsi + co * a.wso + ky * a.wsy + kx * a.wsx;
  return a.w_f32 ? ((const float*)a.w)[off] : (float)((const bf16*)a.w)[off];
}

// weights to LDS as fp32 [ky][kx][co][ci]
__device__ __forceinline__ void stage_w(const Args& a, float* sw) {
  const int n = 9 * a.Co * a.Ci;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int ci = i % a.Ci, r = i / a.Ci, co = r % a.Co, t = r / a.Co;
    sw[i] = (float)(bf16)wload(a, ci, co, t / 3, t % 3);  // (fp32 master weights: the bf16 rounding autocast applies)
  }
  __syncthreads();
}

// gradient w.r.t. the transposed convolution's own result at element `off` of gy: gy itself, or with act the bf16-rounded gy * (1 - y) * y
// (the framework's sigmoid_backward, rounded to the feature dtype before the convolution's backward reads it)
__device__ __forceinline__ float dz_at(const Args& a, long gy_off, long ya_off) {
  const float g = (float)a.gy[gy_off];
  if (!a.act) return g;
  const float y = (float)a.ya[ya_off];
  return (float)(bf16)((g * (1.f - y)) * y);
}

// thread = one output pixel, all Co channels; VEC: x pixels are 16-byte aligned and Ci % 8 == 0 (8 channels per load)
template <int CO, bool VEC>
__global__ __launch_bounds__(256) void deconv_fwd_kernel(const Args a) {
  __shared__ float sw[9 * MAXCO * MAXCI];
  stage_w(a, sw);
  const unsigned npix = (unsigned)(a.B * a.Ho * a.Wo);  // (pixel and element counts fit 31 bits: checked by the launcher)
  for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < npix; p += gridDim.x * 256u) {
    const unsigned row = p / (unsigned)a.Wo;
    const int ox = (int)(p - row * a.Wo), b = (int)(row / (unsigned)a.Ho), oy = (int)(row - b * a.Ho);
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; co++) acc[co] = a.bias ? a.bias[co] : 0.f;
    // taps with oy + 1 - ky even: ky has the parity of oy + 1
    for (int ky = (oy + 1) & 1; ky < 3; ky += 2) {
      const int iy = (oy + 1 - ky) >> 1;
      if (iy < 0 || iy >= a.H) continue;
      for (int kx = (ox + 1) & 1; kx < 3; kx += 2) {
        const int ix = (ox + 1 - kx) >> 1;
        if (ix < 0 || ix >= a.W) continue;
        const bf16* xr = a.x + (((long)b * a.H + iy) * a.W + ix) * a.xp;
        const float* wr = sw + (ky * 3 + kx) * CO * a.Ci;
        if (VEC) {
          for (int c8 = 0; c8 < a.Ci; c8 += 8) {
            const uint4 v = *(const uint4*)(xr + c8);
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const float lo = __uint_as_float(u[j] << 16), hi = __uint_as_float(u[j] & 0xffff0000u);
#pragma unroll
              for (int co = 0; co < CO; co++) {
                acc[co] = __builtin_fmaf(lo, wr[co * a.Ci + c8 + 2 * j], acc[co]);
                acc[co] = __builtin_fmaf(hi, wr[co * a.Ci + c8 + 2 * j + 1], acc[co]);
              }
            }
          }
        } else {
          for (int ci = 0; ci < a.Ci; ci++) {
            const float xv = (float)xr[ci];
#pragma unroll
            for (int co = 0; co < CO; co++) acc[co] = __builtin_fmaf(xv, wr[co * a.Ci + ci], acc[co]);
          }
        }
      }
    }
    bf16* yr = a.y + (long)p * a.yp;
#pragma unroll
    for (int co = 0; co < CO; co++) {
      float v = acc[co];
      if (a.act) v = 1.f / (1.f + expf(-(float)(bf16)v));  // (sigmoid of the bf16 value the unfused chain would have stored)
      yr[co] = (bf16)v;
    }
  }
}

// thread = one input pixel x CPT channels (8 with 16-byte stores when the gradient's pixels allow, else 1):
// gx[b,ci,iy,ix] = sum_{co,ky,kx} gy[b,co,2iy-1+ky,2ix-1+kx] * w[ci,co,ky,kx]
template <int CO, int CPT>
__global__ __launch_bounds__(256) void deconv_bwd_data_kernel(const Args a) {
  __shared__ float sw[9 * MAXCO * MAXCI];
  stage_w(a, sw);
  const unsigned cg = (unsigned)(a.Ci / CPT), n = (unsigned)(a.B * a.H * a.W) * cg;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const unsigned p = i / cg;
    const int c0 = (int)(i - p * cg) * CPT;
    const unsigned row = p / (unsigned)a.W;
    const int ix = (int)(p - row * a.W), b = (int)(row / (unsigned)a.H), iy = (int)(row - b * a.H);
    float acc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; j++) acc[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
      const int oy = 2 * iy - 1 + ky;
      if (oy < 0 || oy >= a.Ho) continue;
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const int ox = 2 * ix - 1 + kx;
        if (ox < 0 || ox >= a.Wo) continue;
        const long op = ((long)b * a.Ho + oy) * a.Wo + ox;
        const float* wr = sw + (ky * 3 + kx) * CO * a.Ci + c0;
#pragma unroll
        for (int co = 0; co < CO; co++) {
          const float gv = dz_at(a, op * a.gyp + co, op * a.yap + co);
#pragma unroll
          for (int j = 0; j < CPT; j++) acc[j] = __builtin_fmaf(gv, wr[co * a.Ci + j], acc[j]);
        }
      }
    }
    bf16* o = a.gx + (long)p * a.gxp + c0;
    if (CPT == 8) {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = (bf16)acc[j < CPT ? j : 0];
      *(bf16x8*)o = v;
    } else {
      o[0] = (bf16)acc[0];
    }
  }
}

// gw[ci,co,ky,kx] += sum x[b,ci,iy,ix] * gy[b,co,2iy-1+ky,2ix-1+kx];  gbias[co] += sum gy.
// lane = (pixel within a group of 64 / cil, input channel), cil = Ci rounded up to a power of two: a wave walks groups of 64 / cil
// input pixels with 9 * CO accumulators per lane; partial sums are reduced over the pixel lanes by shuffles, over the block's waves
// through per-wave LDS copies, and ONE atomic per (block, weight) lands in the fp32 gradient (a few hundred blocks: no same-address pile-up).
template <int CO, int NW>
__global__ __launch_bounds__(64 * NW) void deconv_bwd_weight_kernel(const Args a, int cil, int groups_per_wave, int opix_per_block) {
  extern __shared__ float red[];  // [NW][9 CO Ci]: one copy per wave, summed by the block (LDS float atomics retire one wave
  __shared__ float redb[NW][CO];  // instruction per ~120 cycles on this part: 18 of them x 16 waves were a third of the kernel)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n = 9 * CO * a.Ci;
  const int sub = lane / cil, ci = lane % cil, spw = 64 / cil;
  const int wv = blockIdx.x * NW + wid;
  const int npix = a.B * a.H * a.W;
  float acc[9][CO];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int co = 0; co < CO; co++) acc[t][co] = 0.f;
  const int g0 = wv * groups_per_wave;
  int ng = (npix - g0 * spw + spw - 1) / spw;  // groups this wave has (wave-uniform)
  if (ng > groups_per_wave) ng = groups_per_wave;
  const bool pair = CO == 2 && !a.act && (a.gyp & 1) == 0 && ((uintptr_t)a.gy & 3) == 0;  // both channels of a gy pixel in one 4-byte load
  constexpr int UNR = CO <= 2 ? 4 : 1;  // (3 and 4 output channels: the accumulators leave no room)
#pragma unroll UNR
  for (int g = 0; g < ng; g++) {
    const int pp = (g0 + g) * spw + sub;
    const bool valid = pp < npix;
    const unsigned p = valid ? pp : 0;
    const unsigned row = p / (unsigned)a.W;
    const int ix = (int)(p - row * a.W), b = (int)(row / (unsigned)a.H), iy = (int)(row - b * a.H);
    const float xv = (valid && ci < a.Ci) ? (float)a.x[(long)p * a.xp + ci] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
      const int oy = 2 * iy - 1 + ky;
      const bool oky = oy >= 0 && oy < a.Ho;
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const int ox = 2 * ix - 1 + kx;
        const bool ok = oky && ox >= 0 && ox < a.Wo;
        const long op = ((long)b * a.Ho + (oky ? oy : 0)) * a.Wo + (ok ? ox : 0);
        const bf16* gr = a.gy + op * a.gyp;
        if (pair) {
          const unsigned u = ok ? *(const unsigned*)gr : 0u;
          acc[ky * 3 + kx][0] = __builtin_fmaf(xv, __uint_as_float(u << 16), acc[ky * 3 + kx][0]);
          acc[ky * 3 + kx][CO - 1] = __builtin_fmaf(xv, __uint_as_float(u & 0xffff0000u), acc[ky * 3 + kx][CO - 1]);
        } else {
#pragma unroll
          for (int co = 0; co < CO; co++) {
            const float gv = ok ? dz_at(a, op * a.gyp + co, op * a.yap + co) : 0.f;
            acc[ky * 3 + kx][co] = __builtin_fmaf(xv, gv, acc[ky * 3 + kx][co]);
          }
        }
      }
    }
  }
  // over the pixel lanes; LDS in the weight's own [ci][co][ky][kx] order
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int co = 0; co < CO; co++) {
      float v = acc[t][co];
      for (int o = 32; o >= cil; o >>= 1) v += __shfl_xor(v, o, 64);
      if (sub == 0 && ci < a.Ci) red[wid * n + (ci * CO + co) * 9 + t] = v;
    }
  // bias gradient: the block's share of the output pixels, lanes over pixels
  if (a.gbias) {
    const int nop = a.B * a.Ho * a.Wo;
    const int q0 = blockIdx.x * opix_per_block, q1 = q0 + opix_per_block < nop ? q0 + opix_per_block : nop;
    float s[CO];
#pragma unroll
    for (int co = 0; co < CO; co++) s[co] = 0.f;
    for (int q = q0 + threadIdx.x; q < q1; q += 64 * NW) {
#pragma unroll
      for (int co = 0; co < CO; co++) s[co] += dz_at(a, (long)q * a.gyp + co, (long)q * a.yap + co);
    }
#pragma unroll
    for (int co = 0; co < CO; co++) {
      const float t = wave_sum(s[co]);
      if (lane == 0) redb[wid][co] = t;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 64 * NW) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) v += red[w * n + i];
    const int t = i % 9, r = i / 9, co = r % CO, c = r / CO;  // (consecutive lanes -> consecutive floats of a dense weight gradient)
    atomicAdd(a.gw + c * a.gsi + co * a.gso + (t / 3) * a.gsy + (t % 3) * a.gsx, v);
  }
  if (a.gbias && threadIdx.x < CO) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) v += redb[w][threadIdx.x];
    atomicAdd(a.gbias + threadIdx.x, v);
  }
}

// ---- backward in ONE launch (round 5): input gradient + weight gradient + bias gradient -----------------------------------------------
// The two-kernel form above spent 29-63 us per layer on megabytes of data (16 layers, 0.48 ms per step): its weight-gradient kernel gave
// a wave 64 / Ci pixels per trip (two at Ci = 32) behind nine dependent 4-byte loads, and every one of up to 256 blocks of 1024 threads
// ended in 9 Co Ci same-address atomics; the data-gradient kernel re-read the same nine taps with 2-byte loads.
// Here a thread owns (input pixel, CPT channels) and walks the grid's share of them: per trip it loads the pixel's CPT channels (one
// 16-byte load) and its nine output taps ONCE (Co channels each: one 4-byte load when Co = 2), forms the CPT input gradients from the
// weights in LDS and adds its CPT x 9 Co products to register accumulators.  At the end the lanes that hold the same channel group
// (lane % (Ci / CPT)) are summed by shuffles, the four waves through LDS, and a block adds its 9 Co Ci sums with one atomic each --
// at most 128 blocks.  The bias gradient is the sum of the taps (ky, kx) in {1, 2}^2 -- every output pixel is exactly one of those --
// taken by the threads of channel group 0.
template <int CO, int CPT, int CG>
__global__ __launch_bounds__(256) void deconv_bwd_fused_kernel(const Args a, int need_gx) {
  constexpr int cg = CG;  // channel groups per pixel = Ci / CPT (compile time: the shuffle tree below must unroll into independent chains)
  extern __shared__ float smem[];  // [9][CO][Ci] weights, then [4][9 CO Ci] wave sums
  __shared__ float redb[4][CO];
  const int n = 9 * CO * a.Ci;
  float* sw = smem;
  float* red = smem + ((n + 3) & ~3);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (need_gx) stage_w(a, sw);
  const unsigned npix = (unsigned)(a.B * a.H * a.W), nitem = npix * (unsigned)cg;
  const bool pair = CO == 2 && (a.gyp & 1) == 0 && ((uintptr_t)a.gy & 3) == 0 && (!a.act || ((a.yap & 1) == 0 && ((uintptr_t)a.ya & 3) == 0));
  float acc[CPT][9 * CO];
#pragma unroll
  for (int j = 0; j < CPT; j++)
#pragma unroll
    for (int t = 0; t < 9 * CO; t++) acc[j][t] = 0.f;
  float bsum[CO];
#pragma unroll
  for (int co = 0; co < CO; co++) bsum[co] = 0.f;
  const unsigned stride = gridDim.x * 256u;
  const unsigned trips = (nitem + stride - 1) / stride;  // the same for every thread: the shuffles below need whole waves
  for (unsigned it = 0; it < trips; it++) {
    const unsigned i = it * stride + blockIdx.x * 256u + tid;
    const bool valid = i < nitem;
    const unsigned iv = valid ? i : 0u;
    const unsigned p = iv / (unsigned)cg;
    const int c0 = (int)(iv - p * cg) * CPT;
    const unsigned row = p / (unsigned)a.W;
    const int ix = (int)(p - row * a.W), b = (int)(row / (unsigned)a.H), iy = (int)(row - b * a.H);
    // ---- loads: all issued before the first use, from always-valid addresses (masked by `ok`)
    float xv[CPT];
    const bf16* xr = a.x + (long)p * a.xp + c0;
    if (CPT == 8) {
      const uint4 v = *(const uint4*)xr;
      const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; j++) { xv[(2 * j) % CPT] = __uint_as_float(u[j] << 16); xv[(2 * j + 1) % CPT] = __uint_as_float(u[j] & 0xffff0000u); }
    } else if (CPT == 4) {
      const uint2 v = *(const uint2*)xr;
      const unsigned u[2] = {v.x, v.y};
#pragma unroll
      for (int j = 0; j < 2; j++) { xv[(2 * j) % CPT] = __uint_as_float(u[j] << 16); xv[(2 * j + 1) % CPT] = __uint_as_float(u[j] & 0xffff0000u); }
    } else {
#pragma unroll
      for (int j = 0; j < CPT; j++) xv[j] = (float)xr[j];
    }
    unsigned graw[9], yraw[9];
    bool okt[9];
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
      const int oy = 2 * iy - 1 + ky;
      const bool oky = oy >= 0 && oy < a.Ho;
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const int ox = 2 * ix - 1 + kx;
        const bool ok = valid && oky && ox >= 0 && ox < a.Wo;
        const long op = ((long)b * a.Ho + (oky ? oy : 0)) * a.Wo + ((ox >= 0 && ox < a.Wo) ? ox : 0);
        const int t = ky * 3 + kx;
        okt[t] = ok;
        if (pair) {
          graw[t] = *(const unsigned*)(a.gy + op * a.gyp);
          yraw[t] = a.act ? *(const unsigned*)(a.ya + op * a.yap) : 0u;
        } else {
          unsigned g = 0u, y = 0u;
#pragma unroll
          for (int co = 0; co < CO; co++) {
            g |= (unsigned)__builtin_bit_cast(unsigned short, a.gy[op * a.gyp + co]) << (16 * (co & 1));
            if (a.act) y |= (unsigned)__builtin_bit_cast(unsigned short, a.ya[op * a.yap + co]) << (16 * (co & 1));
          }
          graw[t] = g; yraw[t] = y;  // (CO <= 2: both channels fit one dword)
        }
      }
    }
    if (!valid) {
#pragma unroll
      for (int j = 0; j < CPT; j++) xv[j] = 0.f;
    }
    float gv[9][CO];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int co = 0; co < CO; co++) {
        float g = co == 0 ? __uint_as_float(graw[t] << 16) : __uint_as_float(graw[t] & 0xffff0000u);
        if (a.act) {
          const float y = co == 0 ? __uint_as_float(yraw[t] << 16) : __uint_as_float(yraw[t] & 0xffff0000u);
          g = (float)(bf16)((g * (1.f - y)) * y);  // (dz_at: the framework's sigmoid_backward, rounded to the feature dtype)
        }
        gv[t][co] = okt[t] ? g : 0.f;
      }
    // ---- weight gradient
#pragma unroll
    for (int j = 0; j < CPT; j++)
#pragma unroll
      for (int t = 0; t < 9; t++)
#pragma unroll
        for (int co = 0; co < CO; co++) acc[j][t * CO + co] = __builtin_fmaf(xv[j], gv[t][co], acc[j][t * CO + co]);
    // ---- bias gradient: taps (1,1) (1,2) (2,1) (2,2) partition the output pixels
    if (c0 == 0) {
#pragma unroll
      for (int co = 0; co < CO; co++) bsum[co] += (gv[4][co] + gv[5][co]) + (gv[7][co] + gv[8][co]);
    }
    // ---- input gradient
    if (need_gx) {
      float gx[CPT];
#pragma unroll
      for (int j = 0; j < CPT; j++) gx[j] = 0.f;
#pragma unroll
      for (int t = 0; t < 9; t++)
#pragma unroll
        for (int co = 0; co < CO; co++) {
          const float* wr = sw + (t * CO + co) * a.Ci + c0;
#pragma unroll
          for (int j = 0; j < CPT; j++) gx[j] = __builtin_fmaf(gv[t][co], wr[j], gx[j]);
        }
      if (valid) {
        bf16* o = a.gx + (long)p * a.gxp + c0;
        if (CPT == 8) {
          bf16x8 v;
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] = (bf16)gx[j % CPT];
          *(bf16x8*)o = v;
        } else if (CPT == 4) {
          typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
          bf16x4v v;
#pragma unroll
          for (int j = 0; j < 4; j++) v[j] = (bf16)gx[j % CPT];
          *(bf16x4v*)o = v;
        } else {
#pragma unroll
          for (int j = 0; j < CPT; j++) o[j] = (bf16)gx[j];
        }
      }
    }
  }
  // ---- sum over the lanes of the same channel group, then over the block's waves.  One tree LEVEL at a time over a whole row of 9 CO
  // values: the exchanges of a level are independent and go out back to back (written value by value, the compiler waited for every
  // single ds_bpermute before issuing the next: 576 dependent round trips, 25 us of a 31 us launch)
  const int grp = lane & (cg - 1);
#pragma unroll
  for (int j = 0; j < CPT; j++) {
#pragma unroll
    for (int o = 32; o >= cg; o >>= 1) {
      float tmp[9 * CO];
#pragma unroll
      for (int k = 0; k < 9 * CO; k++) tmp[k] = __shfl_xor(acc[j][k], o, 64);
#pragma unroll
      for (int k = 0; k < 9 * CO; k++) acc[j][k] += tmp[k];
    }
    if (lane < cg) {
#pragma unroll
      for (int t = 0; t < 9; t++)
#pragma unroll
        for (int co = 0; co < CO; co++) red[wid * n + ((grp * CPT + j) * CO + co) * 9 + t] = acc[j][t * CO + co];  // [ci][co][ky][kx]
    }
  }
  if (a.gbias) {
#pragma unroll
    for (int co = 0; co < CO; co++) {
      const float t = wave_sum(bsum[co]);
      if (lane == 0) redb[wid][co] = t;
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    const float v = (red[i] + red[n + i]) + (red[2 * n + i] + red[3 * n + i]);
    const int t = i % 9, r = i / 9, co = r % CO, c = r / CO;
    atomicAdd(a.gw + c * a.gsi + co * a.gso + (t / 3) * a.gsy + (t % 3) * a.gsx, v);
  }
  if (a.gbias && tid < CO) atomicAdd(a.gbias + tid, (redb[0][tid] + redb[1][tid]) + (redb[2][tid] + redb[3][tid]));
}
}  // namespace dcv
}  // namespace ocf

static bool dcv_fill(ocf::dcv::Args& a, const ocf_tensor4* x, int Ci, int H, int W, const ocf_tensor4* y, int Co, int Ho, int Wo, int B, long* xp,
                     long* yp) {
  auto pitch = [&](const ocf_tensor4* t, int C, int h, int w) { return w > 1 ? t->sw : (h > 1 ? t->sh : (B > 1 ? t->sb : (int64_t)C)); };
  auto pm = [&](const ocf_tensor4* t, int C, int h, int w) {
    const int64_t p = pitch(t, C, h, w);
    return (C == 1 || t->sc == 1) && p >= C && (h == 1 || t->sh == p * w) && (B == 1 || t->sb == p * w * h);
  };
  if (!pm(x, Ci, H, W) || !pm(y, Co, Ho, Wo)) return false;
  *xp = pitch(x, Ci, H, W);
  *yp = pitch(y, Co, Ho, Wo);
  a.B = B; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.Ci = Ci; a.Co = Co;
  return true;
}
static bool dcv_shape_ok(int B, int Ci, int Co, int H, int W, int Ho, int Wo) {
  if ((long)B * H * W * (Ci > 0 ? Ci : 1) >= (1L << 31) || (long)B * Ho * Wo * 4 >= (1L << 31)) return false;
  return B > 0 && Ci > 0 && Co > 0 && H > 0 && W > 0 && (Ho == 2 * H - 1 || Ho == 2 * H) && (Wo == 2 * W - 1 || Wo == 2 * W);
}

extern "C" int ocf_deconv3x3s2_fwd(const ocf_tensor4* x, const void* w, int64_t w_si, int64_t w_so, int64_t w_sy, int64_t w_sx, int w_is_f32,
                                   const float* bias, const ocf_tensor4* y, int B, int Ci, int Co, int H, int W, int Ho, int Wo, int act,
                                   int dtype, void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !y || !w || !x->data || !y->data) return OCF_EINVAL;
  if (!dcv_shape_ok(B, Ci, Co, H, W, Ho, Wo)) return OCF_EINVAL;
  if (dtype != OCF_BF16 || Co > dcv::MAXCO || Ci > dcv::MAXCI) return OCF_EUNSUPPORTED;
  dcv::Args a = {};
  if (!dcv_fill(a, x, Ci, H, W, y, Co, Ho, Wo, B, &a.xp, &a.yp)) return OCF_EUNSUPPORTED;
  if (act != 0 && act != 1) return OCF_EINVAL;
  a.act = act;
  a.x = (const bf16*)x->data; a.y = (bf16*)y->data; a.w = w; a.bias = bias; a.w_f32 = w_is_f32;
  a.wsi = w_si; a.wso = w_so; a.wsy = w_sy; a.wsx = w_sx;
  const long npix = (long)B * Ho * Wo, blocks = (npix + 255) / 256;
  LaunchProf lp;
  const bool vec = Ci % 8 == 0 && a.xp % 8 == 0 && ((uintptr_t)a.x & 15) == 0;
  const dim3 grid((unsigned)(blocks < 4096 ? blocks : 4096));
#define OCF_DCV_FWD(CO)                                                                                          \
  case CO:                                                                                                       \
    if (vec) lp.launch(true, dcv::deconv_fwd_kernel<CO, true>, grid, dim3(256), 0, (hipStream_t)stream, a);      \
    else lp.launch(true, dcv::deconv_fwd_kernel<CO, false>, grid, dim3(256), 0, (hipStream_t)stream, a);         \
    break;
  switch (Co) { OCF_DCV_FWD(1) OCF_DCV_FWD(2) OCF_DCV_FWD(3) OCF_DCV_FWD(4) }
#undef OCF_DCV_FWD
  return launch_status();
}

extern "C" int ocf_deconv3x3s2_bwd(const ocf_tensor4* x, const void* w, int64_t w_si, int64_t w_so, int64_t w_sy, int64_t w_sx, int w_is_f32,
                                   const ocf_tensor4* gy, const ocf_tensor4* gx, float* gw, int64_t g_si, int64_t g_so, int64_t g_sy, int64_t g_sx,
                                   float* gbias, const ocf_tensor4* y_act, int B, int Ci, int Co, int H, int W, int Ho, int Wo, int dtype,
                                   void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !gy || !w || !x->data || !gy->data) return OCF_EINVAL;
  if (!dcv_shape_ok(B, Ci, Co, H, W, Ho, Wo)) return OCF_EINVAL;
  if (dtype != OCF_BF16 || Co > dcv::MAXCO || Ci > dcv::MAXCI) return OCF_EUNSUPPORTED;
  dcv::Args a = {};
  if (!dcv_fill(a, x, Ci, H, W, gy, Co, Ho, Wo, B, &a.xp, &a.gyp)) return OCF_EUNSUPPORTED;
  if (y_act) {  // sigmoid output of the fused forward: same shape as gy
    if (!y_act->data) return OCF_EINVAL;
    dcv::Args t = a;
    long xp_, yap_;
    if (!dcv_fill(t, x, Ci, H, W, y_act, Co, Ho, Wo, B, &xp_, &yap_)) return OCF_EUNSUPPORTED;
    a.act = 1; a.ya = (const bf16*)y_act->data; a.yap = yap_;
  }
  a.x = (const bf16*)x->data; a.gy = (const bf16*)gy->data; a.w = w; a.w_f32 = w_is_f32;
  a.wsi = w_si; a.wso = w_so; a.wsy = w_sy; a.wsx = w_sx;
  a.gw = gw; a.gsi = g_si; a.gso = g_so; a.gsy = g_sy; a.gsx = g_sx; a.gbias = gbias;
  LaunchProf lp;
  bool launched = false;
  // ONE launch for all three gradients where the fused kernel takes the layer: a weight gradient is wanted, Co <= 2, Ci in {1, 2} or
  // a multiple of 8 whose channel groups divide a wave (8, 16, 32, 64) with 16-byte aligned pixels
  if (gw && Co <= 2) {
    const bool wide = Ci % 8 == 0 && (Ci == 8 || Ci == 16 || Ci == 32 || Ci == 64) && a.xp % 8 == 0 && ((uintptr_t)a.x & 15) == 0;
    bool gx_ok = true;
    long gxp = 0, dummy = 0;
    if (gx) {
      if (!gx->data) return OCF_EINVAL;
      dcv::Args t = a;
      gx_ok = dcv_fill(t, gx, Ci, H, W, gy, Co, Ho, Wo, B, &gxp, &dummy) && (!wide || (gxp % 8 == 0 && ((uintptr_t)gx->data & 15) == 0));
    }
#ifdef OCF_TUNING_BUILD
    static const bool off = getenv("OCF_DECONV_BWD_SPLIT") != nullptr;  // (A/B: the two-kernel form)
#else
    constexpr bool off = false;
#endif
    if (!off && gx_ok && (wide || Ci <= 2)) {
      // channels per thread: 4 for the wide layers (72 accumulators at Co = 2: two waves per SIMD; with 8 the kernel sat at 256 registers,
      // one wave per SIMD, and every trip's load round trip was exposed), the whole pixel for Ci <= 2
#ifdef OCF_TUNING_BUILD
      static const int cpt_env = getenv("OCF_DECONV_CPT") ? atoi(getenv("OCF_DECONV_CPT")) : 0;
#else
      constexpr int cpt_env = 0;  // (measured: 264.6 us with 8 channels per thread against 245.5 with 4 over the 16 layers of scripts/bench_deconv.py)
#endif
      const int cpt = wide ? ((cpt_env == 8 || Ci == 8) ? 8 : 4) : Ci, cg = Ci / cpt;
      if (gx) { a.gx = (bf16*)gx->data; a.gxp = gxp; }
      const long nitem = (long)B * H * W * cg;
      // at most 256 blocks (every block ends in 9 Co Ci atomics on the same addresses)
      long blocks = (nitem + 255) / 256;
      if (blocks > 256) blocks = 256;
      if (blocks < 1) blocks = 1;
      const int n = 9 * Co * Ci;
      const size_t lds = (size_t)(((n + 3) & ~3) + 4 * n) * sizeof(float);
      const dim3 grid((unsigned)blocks);
      const int need_gx = gx ? 1 : 0;
#define OCF_DCV_FUSED(CO, CPT, CG) lp.launch(true, dcv::deconv_bwd_fused_kernel<CO, CPT, CG>, grid, dim3(256), lds, (hipStream_t)stream, a, need_gx)
#define OCF_DCV_FUSED_CO(CO)                                                                                       \
  if (cpt == 8) {                                                                                                  \
    if (cg == 1) OCF_DCV_FUSED(CO, 8, 1); else if (cg == 2) OCF_DCV_FUSED(CO, 8, 2);                                \
    else if (cg == 4) OCF_DCV_FUSED(CO, 8, 4); else OCF_DCV_FUSED(CO, 8, 8);                                        \
  } else if (cpt == 4) {                                                                                           \
    if (cg == 4) OCF_DCV_FUSED(CO, 4, 4); else if (cg == 8) OCF_DCV_FUSED(CO, 4, 8); else OCF_DCV_FUSED(CO, 4, 16); \
  } else if (cpt == 2) OCF_DCV_FUSED(CO, 2, 1); else OCF_DCV_FUSED(CO, 1, 1);
      if (Co == 1) { OCF_DCV_FUSED_CO(1) } else { OCF_DCV_FUSED_CO(2) }
#undef OCF_DCV_FUSED_CO
#undef OCF_DCV_FUSED
      return launch_status();
    }
  }
  if (gx) {
    if (!gx->data) return OCF_EINVAL;
    long gxp, dummy;
    dcv::Args t = a;
    if (!dcv_fill(t, gx, Ci, H, W, gy, Co, Ho, Wo, B, &gxp, &dummy)) return OCF_EUNSUPPORTED;
    a.gx = (bf16*)gx->data; a.gxp = gxp;
    const long n = (long)B * H * W * Ci, blocks = (n + 255) / 256;
    const bool vec = Ci % 8 == 0 && gxp % 8 == 0 && ((uintptr_t)gx->data & 15) == 0;
    const long nthr = vec ? n / 8 : n, nb = (nthr + 255) / 256;
    const dim3 grid((unsigned)(nb < 4096 ? nb : 4096));
#define OCF_DCV_BWD(CO)                                                                                           \
  case CO:                                                                                                        \
    if (vec) lp.launch(!gw, dcv::deconv_bwd_data_kernel<CO, 8>, grid, dim3(256), 0, (hipStream_t)stream, a);      \
    else lp.launch(!gw, dcv::deconv_bwd_data_kernel<CO, 1>, grid, dim3(256), 0, (hipStream_t)stream, a);          \
    break;
    switch (Co) { OCF_DCV_BWD(1) OCF_DCV_BWD(2) OCF_DCV_BWD(3) OCF_DCV_BWD(4) }
#undef OCF_DCV_BWD
    launched = true;
  }
  if (gw) {
    int cil = 1;
    while (cil < Ci) cil <<= 1;
    const long npix = (long)B * H * W, nop = (long)B * Ho * Wo, spw = 64 / cil;
    const long groups = (npix + spw - 1) / spw;
    // ~2 pixel groups per wave (the loop is load-latency bound: many waves); every block ends in 9 * Co * Ci same-address atomics
    // (measured ~22 ns per block), so large maps use 16-wave blocks, at most 256 of them
    const int nred = 9 * Co * Ci;  // floats of one wave's partial sums (16 copies must fit 48 KB of LDS)
    const int nw = (groups >= 2 * 16 * 128 && nred * 16 * 4 <= 48 * 1024) ? 16 : 4;
    long blocks = (groups + nw * 2 - 1) / (nw * 2);
    if (blocks < nop / (512 * nw)) blocks = nop / (512 * nw);  // (the bias sum: <= 8 output pixels per thread)
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    const int gpw = (int)((groups + blocks * nw - 1) / (blocks * nw)), opb = (int)((nop + blocks - 1) / blocks);
    const dim3 grid((unsigned)blocks);
#define OCF_DCV_WG(CO)                                                                                                              \
  case CO:                                                                                                                          \
    if (nw == 16) lp.launch(true, dcv::deconv_bwd_weight_kernel<CO, 16>, grid, dim3(1024), (size_t)nred * 64, (hipStream_t)stream, a, cil, gpw, opb); \
    else lp.launch(true, dcv::deconv_bwd_weight_kernel<CO, 4>, grid, dim3(256), (size_t)nred * 16, (hipStream_t)stream, a, cil, gpw, opb);            \
    break;
    switch (Co) { OCF_DCV_WG(1) OCF_DCV_WG(2) OCF_DCV_WG(3) OCF_DCV_WG(4) }
#undef OCF_DCV_WG
    launched = true;
  }
  return launched ? launch_status() : OCF_EINVAL;
}
