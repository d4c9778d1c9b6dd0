// Channel concatenation of pixel-major (channels_last) bf16 tensors into one dense channels_last tensor, zero-padded to `Cout` channels:
// the decoder inputs  cat(cost volume 81, features C, flow 2, flow features 2) / cat(f1, warped f2, occlusion features 1, occlusion 1)  of
// flow_net.py:104-121, flow_occ_net.py:153-176 with the alignment channels of `cat_aligned` (models/networks/_epilogue.py) written in the
// same pass.  The framework's cat falls to its generic gather kernel for these operands (views of a paired batch, 81- and 2-channel
// parts: 4 launches of ~68 us at 8 x 109 x 256) and needs a separate zero block for the padding.
// A workgroup owns NP = 32 consecutive pixels: it assembles their output rows (NP x Cout bf16, <= 36 KB) in LDS and writes them out as
// one contiguous run of aligned 16-byte stores.  A DENSE part (pixel pitch == its channel count: the cost volume, the up-sampled flow,
// ...) is then a contiguous byte range too -- NP x C_k x 2 bytes, 16-byte aligned because NP = 32 -- and is streamed in with aligned
// 16-byte loads whatever C_k is; the regrouping into output pixels happens in the LDS writes (one ds_write_b128 when part and offset are
// 8-channel aligned, eight ds_write_b16 otherwise).  Parts with a wider pitch (channel slices) are read row by row.  The first version,
// one thread per output unit, read the 162-byte cost-volume pixels with 16-byte loads at 2-byte alignment and gathered every unit that
// straddles two parts element by element: 69 us for the 128-channel decoder input of 8 x 109 x 256 and 32 us for the 96-channel one,
// against 12 us for an all-aligned concatenation of the same size (profiles/r4_bench_concat.txt).
#include "ocf_common.h"

namespace ocf {
namespace cc {
constexpr int MAXP = 6;
struct __attribute__((packed, aligned(2))) U16B { u32x4 v; };  // a 16-byte load at 2-byte alignment (one global_load_dwordx4)
struct Args {
  const char* p[MAXP];
  long pitch[MAXP];  // elements between consecutive pixels of part k
  int c0[MAXP + 1];  // first output channel of part k; c0[n] = total channels
  int vec[MAXP];     // part k can be read in aligned 16-byte units
  char* out;
  int n, Cout;       // Cout % 8 == 0
  long opitch;       // elements between consecutive output pixels (>= Cout, a multiple of 8: the output may be a channel slice)
  long npix;
};

constexpr int NP = 32;

__global__ __launch_bounds__(256) void concat_kernel(const Args a) {
  extern __shared__ u32x4 s_tile[];  // [NP][Cout] bf16
  unsigned short* s_e = (unsigned short*)s_tile;
  const int tid = threadIdx.x;
  const long pix0 = (long)blockIdx.x * NP;
  const int np = a.npix - pix0 < NP ? (int)(a.npix - pix0) : NP;
  const int G = a.Cout >> 3, ctot = a.c0[a.n];
  // alignment channels: zero the tail units of every pixel row (the parts overwrite what they cover)
  if (ctot < a.Cout) {
    const int g0 = ctot >> 3, ng = G - g0;
    for (int i = tid; i < np * ng; i += 256) {
      const int px = i / ng, g = g0 + (i - px * ng);
      s_tile[px * G + g] = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();  // (a part may end inside the first zeroed unit)
  }
#pragma unroll
  for (int k = 0; k < MAXP; k++) {
    if (k >= a.n) break;
    const int Ck = a.c0[k + 1] - a.c0[k], c0 = a.c0[k];
    const char* src = a.p[k] + pix0 * a.pitch[k] * 2;
    if (a.pitch[k] == Ck && ((uintptr_t)a.p[k] & 15) == 0) {
      // dense part: NP x Ck contiguous elements (16-byte aligned start: NP * Ck * 2 is a multiple of 64)
      const int nel = np * Ck, nfull = nel >> 3;
      const bool al = (Ck & 7) == 0 && (c0 & 7) == 0;
      for (int u = tid; u < nfull; u += 256) {
        const u32x4 v = *(const u32x4*)(src + u * 16);
        const int e = u * 8, px = e / Ck, c = e - px * Ck;
        if (al) {
          *(u32x4*)(s_e + px * a.Cout + c0 + c) = v;
        } else {
          int pp = px, cc = c;
          const unsigned w[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
          for (int j = 0; j < 8; j++) {
            s_e[pp * a.Cout + c0 + cc] = (unsigned short)(w[j >> 1] >> ((j & 1) * 16));
            if (++cc == Ck) { cc = 0; pp++; }
          }
        }
      }
      for (int e = nfull * 8 + tid; e < nel; e += 256) {  // (a partial last block: fewer than 8 elements left over)
        const int px = e / Ck, c = e - px * Ck;
        s_e[px * a.Cout + c0 + c] = ((const unsigned short*)src)[e];
      }
    } else if (a.vec[k] && (Ck & 7) == 0) {
      // channel slice of a wider tensor, 16-byte aligned rows: one unit per thread
      const int upr = Ck >> 3;
      for (int i = tid; i < np * upr; i += 256) {
        const int px = i / upr, u = i - px * upr;
        *(u32x4*)(s_e + px * a.Cout + c0 + u * 8) = *(const u32x4*)(src + ((long)px * a.pitch[k] + u * 8) * 2);
      }
    } else {
      for (int i = tid; i < np * Ck; i += 256) {
        const int px = i / Ck, c = i - px * Ck;
        s_e[px * a.Cout + c0 + c] = ((const unsigned short*)src)[(long)px * a.pitch[k] + c];
      }
    }
  }
  __syncthreads();
  if (a.opitch == a.Cout) {
    u32x4* dst = (u32x4*)(a.out + pix0 * a.Cout * 2);
    for (int i = tid; i < np * G; i += 256) dst[i] = s_tile[i];
  } else {  // channel slice of a wider channels_last buffer (the dense block's input range): one run of G units per pixel
    const long og = a.opitch >> 3;
    u32x4* dst = (u32x4*)(a.out) + pix0 * og;
    for (int i = tid; i < np * G; i += 256) {
      const int px = i / G, g = i - px * G;
      dst[px * og + g] = s_tile[i];
    }
  }
}
}  // namespace cc
}  // namespace ocf

extern "C" int ocf_concat_channels(const ocf_tensor4* const* parts, const int* channels, int n, const ocf_tensor4* out, int Cout, int B, int H, int W,
                                   int dtype, void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!parts || !channels || !out || !out->data || n <= 0 || B <= 0 || H <= 0 || W <= 0) return OCF_EINVAL;
  if (dtype != OCF_BF16 || n > cc::MAXP || Cout % 8 != 0) return OCF_EUNSUPPORTED;
  auto pitch = [&](const ocf_tensor4* t, int C) { return W > 1 ? t->sw : (H > 1 ? t->sh : (B > 1 ? t->sb : (int64_t)C)); };
  auto pixel_major = [&](const ocf_tensor4* t, int C) {
    const int64_t p = pitch(t, C);
    return (C == 1 || t->sc == 1) && p >= C && (H == 1 || t->sh == p * W) && (B == 1 || t->sb == p * W * H);
  };
  cc::Args a;
  int c = 0;
  for (int k = 0; k < n; k++) {
    if (!parts[k] || !parts[k]->data || channels[k] <= 0) return OCF_EINVAL;
    if (!pixel_major(parts[k], channels[k])) return OCF_EUNSUPPORTED;
    a.p[k] = (const char*)parts[k]->data;
    a.pitch[k] = pitch(parts[k], channels[k]);
    a.c0[k] = c;
    a.vec[k] = (c % 8 == 0) && (a.pitch[k] % 8 == 0) && ((uintptr_t)parts[k]->data % 16 == 0);
    c += channels[k];
  }
  for (int k = n; k <= cc::MAXP; k++) a.c0[k] = c;
  for (int k = n; k < cc::MAXP; k++) { a.p[k] = a.p[0]; a.pitch[k] = 0; a.vec[k] = 0; }
  if (c > Cout) return OCF_EINVAL;
  if (!pixel_major(out, Cout) || pitch(out, Cout) % 8 != 0 || (uintptr_t)out->data % 16 != 0) return OCF_EUNSUPPORTED;
  a.out = (char*)out->data; a.n = n; a.Cout = Cout; a.opitch = pitch(out, Cout); a.npix = (long)B * H * W;
  const long blocks = (a.npix + cc::NP - 1) / cc::NP;
  const size_t lds = (size_t)cc::NP * Cout * 2;
  if (blocks > 0x7fffffffL || lds > 48 * 1024) return OCF_EUNSUPPORTED;  // (Cout <= 768)
  LaunchProf lp;
  lp.launch(true, cc::concat_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
  return launch_status();
}
