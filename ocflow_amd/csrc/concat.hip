// Channel concatenation of pixel-major (channels_last) bf16 tensors into one dense channels_last tensor, zero-padded to `Cout` channels:
// the decoder inputs  cat(cost volume 81, features C, flow 2, flow features 2) / cat(f1, warped f2, occlusion features 1, occlusion 1)  of
// flow_net.py:104-121, flow_occ_net.py:153-176 with the alignment channels of `cat_aligned` (models/networks/_epilogue.py) written in the
// same pass.  The framework's cat falls to its generic gather kernel for these operands (views of a paired batch, 81- and 2-channel
// parts: 4 launches of ~68 us at 8 x 109 x 256) and needs a separate zero block for the padding.
// Thread = one 16-byte unit (8 channels) of one output pixel: a unit inside ONE part is one 16-byte load -- at whatever 2-byte
// alignment the part's pixels have (the 162-byte pixels of the cost volume: gfx950 serves unaligned global loads; the first version
// gathered those units 2 bytes at a time and took 62 us for the 128-channel decoder input of 8 x 109 x 256, against 17 us for an
// all-aligned concatenation of 96 channels) -- and only a unit that straddles a part boundary is gathered element by element.
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
  int n, Cout;       // Cout % 8 == 0, dense output
  long npix;
};

__global__ __launch_bounds__(256) void concat_kernel(const Args a) {
  const int G = a.Cout >> 3;
  const long total = a.npix * G;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long pix = i / G;
    const int ch = (int)(i - pix * G) << 3;
    u32x4 v = {0u, 0u, 0u, 0u};
    int k = 0;
#pragma unroll
    for (int q = 1; q < MAXP; q++) k += (q < a.n && ch >= a.c0[q]) ? 1 : 0;  // the part holding channel ch (if any)
    if (ch < a.c0[a.n]) {
      if (ch + 8 <= a.c0[k + 1]) {
        v = ((const U16B*)(a.p[k] + (pix * a.pitch[k] + (ch - a.c0[k])) * 2))->v;
      } else {
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int c = ch + j;
          int kk = k;
          while (kk + 1 < a.n && c >= a.c0[kk + 1]) kk++;
          e[j] = c < a.c0[a.n] ? ((const unsigned short*)a.p[kk])[pix * a.pitch[kk] + (c - a.c0[kk])] : (unsigned short)0;
        }
        v = u32x4{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16),
                  (unsigned)e[6] | ((unsigned)e[7] << 16)};
      }
    }
    *(u32x4*)(a.out + i * 16) = v;
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
  if (!pixel_major(out, Cout) || pitch(out, Cout) != Cout || (uintptr_t)out->data % 16 != 0) return OCF_EUNSUPPORTED;
  a.out = (char*)out->data; a.n = n; a.Cout = Cout; a.npix = (long)B * H * W;
  const long total = a.npix * (Cout / 8);
  const long blocks = (total + 255) / 256;
  LaunchProf lp;
  lp.launch(true, cc::concat_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}
