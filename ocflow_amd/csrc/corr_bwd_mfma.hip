// Cost-volume backward on the matrix cores (bf16, channels_last, d = 4): both gradients as BAND GEMMs along image rows.
//   g1[y][x][c] = 1/C sum_{dy} sum_{x'} G1_dy[x][x'] * w2[y+dy][x'][c],   G1_dy[x][x'] = gout[y][x][(dy+4)*9 + (x'-x)+4],  |x'-x| <= 4
//   g2[y][x][c] = 1/C sum_{dy} sum_{x'} G2_dy[x][x'] * f1[y-dy][x'][c],   G2_dy[x][x'] = gout[y-dy][x'][(dy+4)*9 + (x-x')+4], |x-x'| <= 4
// (reference semantics: autograd of compute_cost_volume, cost_volume_flow_net.py:153-181 / flow_occ_net.py, as the vector kernel
// corr_bwd_tiled in corr.hip, which remains the path for fp32, other layouts and small problems.)
// Per 16-pixel strip of a row and per dy that is ONE v_mfma_f32_16x16x32_bf16 per 16 channels: D[c][x] = sum_k F^T[c][k] * G[k][x]
// with k = 32 source pixels around the strip (24 can be non-zero).  The vector kernel spent 648 FMAs + 81 LDS unit reads per thread on
// it and was VALU/LDS bound at 47-50 us per pass at 8x32x109x256; the MFMA count here is 1.6 us worth.  What remains is operand
// delivery:
//   F side (w2 / f1, [pixel][channel] in memory, k = pixel is the STRIDED direction): staged in LDS as [halo row][pixel][32 ch] and
//       read with ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, transposed on the way out);
//   G side, g1: the 8 consecutive k of a lane are 8 consecutive taps of ITS pixel -- one unaligned 16-byte global load straight into
//       the operand registers, out-of-band taps masked (no LDS, no staging pass);
//   G side, g2: the 8 k of a lane are 8 different source pixels (160 bytes apart): the 36-tap windows of the halo pixels are copied
//       to LDS once per tile and the operand is assembled from 2-byte LDS reads (6 per fragment: k is PERMUTED, k = 8 G + j <->
//       source pixel 4 j + G, so that the four lane groups of a read sit on four different bank octets, and j = 6, 7 are never in
//       the band).  The F side applies the same permutation through the addresses of its transposing reads.
// One workgroup = 4 rows x 32 pixels x one 32-channel block (grid.y = channel blocks); wave w owns row w (2 strips x 9 dy x 2
// channel sub-blocks = 36 MFMAs).  D puts 4 consecutive channels of one pixel in a lane: 8-byte stores.
#include "corr_internal.h"
#include "ocf_common.h"

namespace ocf {
namespace bm {
constexpr int D = 4, N = 9, NK = 81;
constexpr int TH = 4, TW = 32, HR = TH + 2 * D, HWP = TW + 2 * D;  // halo: 12 rows x 40 pixels
constexpr int FP1 = 96;  // g1: bytes per staged halo pixel (64 of data).  A transposing read takes 4 CONSECUTIVE pixels: 24 dwords apart
                         // = bank octets 0, 24, 16, 8
constexpr int FP2 = 72;  // g2: the 4 pixels of a read are 4 apart (permuted k): 72 dwords apart = bank octets 0, 8, 16, 24
constexpr int WT = TH * N;   // taps of the window kept per halo pixel in the g2 pass (4 tap rows)
constexpr int GP = 80;       // bytes per halo pixel of that window (72 of data; 8-byte aligned chunks)
// g1 with the gradient delivered through LDS (GL = true): the 81 taps of the tile's own 4 x 32 pixels, 168 bytes per pixel (21 eight-byte
// chunks, 162 of data), a front pad for the windows that start up to 15 taps before a pixel and a tail pad for those that end after it
constexpr int GIP = 168, GI_CH = GIP / 8, GI_FRONT = 32, GI_TAIL = 48;
constexpr int GI_BYTES = GI_FRONT + TH * TW * GIP + GI_TAIL;
}  // namespace bm

typedef short s4v_ __attribute__((ext_vector_type(4)));
typedef short s8v_ __attribute__((ext_vector_type(8)));
struct __attribute__((packed, aligned(2))) U16B { u32x4 v; };
struct __attribute__((packed, aligned(2))) U8B { u32x2 v; };

// two transposing reads -> the 8 k values of this lane's channel; a0 / a1: this lane's chunk address for k 0-3 / 4-7
__device__ __forceinline__ bf16x8 tr_pair(const char* a0, const char* a1) {
  typedef s4v_ __attribute__((address_space(3))) * lds_s4;
  const s4v_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)a0);
  const s4v_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)a1);
  const s8v_ v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// GL (g1 only): the gradient operand goes global -> LDS in coalesced 8-byte chunks (consecutive lanes = consecutive chunks of a pixel) and
// the fragments are assembled from aligned LDS reads, instead of one unaligned 16-byte global load per lane and fragment.  Inside the
// model the gradient is an 81-channel slice at a 256-byte pixel pitch: there the per-lane loads of the direct form run at half their
// dense-tensor speed (52 vs 26 us at 8 x 32 x 109 x 256, tools/kbench), while the chunked staging the g2 pass has always used does not
// care about the pitch (24 us either way).
namespace bm {
constexpr int SF1 = HR * HWP * FP1, SF2 = HR * HWP * FP2, SG2 = HR * HWP * GP + 64;  // LDS regions: other operand (g1 / g2 layout), g2 windows
}

// one pass (G2 = false: g1, G2 = true: g2) of one tile; s_f / s_g: this pass's LDS regions (SF1 (+ GI_BYTES with GL) / SF2 + SG2 bytes)
template <bool G2, bool GL>
__device__ __forceinline__ void corr_bwd_pass(char* s_f, char* s_g, T4 other, T4 g, T4 dst, T4 occ, T4 w2own, T4 gocc, int B, int C, int H, int W,
                                              int tiles_x, int tiles_y, int ablate) {
  using namespace bm;
  constexpr int FP = G2 ? FP2 : FP1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, G = lane >> 4, q = i >> 2, p = i & 3;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tpi = tiles_x * tiles_y;
  const int b = tile / tpi;
  const int trem = tile - b * tpi;
  const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
  const int cbase = (int)blockIdx.y * 32;
  const int y = ty0 + w;
  const float invC = 1.f / (float)C;

  // Address arithmetic is what this kernel would otherwise be bound by (the first version spent ~2000 instructions per wave, most of
  // them 64-bit index math, and took 21 us with every memory access and the MFMA loop switched off): all per-image offsets are 32-bit
  // (host-checked), the batch base is scalar, and the staging loops give a lane a FIXED column and walk rows that are uniform.
  const int osh = (int)other.sh, osw = (int)other.sw, gsh = (int)g.sh, gsw = (int)g.sw;
  const char* obase = other.p + (long)b * other.sb * 2;
  const char* gbase = g.p + (long)b * g.sb * 2;

  // ---- g1: the G operands straight from global memory.  Lane (x = i, group G) of (strip s, dy): taps 8G - i + 0..7 of tap row dy
  // of pixel (y, x): 16 bytes at a 2-byte aligned address.  Taps outside 0..8 are other data and are masked; in the workgroups that
  // hold the first / last pixel of the tensor such a window can leave the tensor: there the address is clamped into it and the
  // window shifted back.
  u32x4 bfr[G2 ? 1 : 2][G2 ? 1 : N], msk = {0u, 0u, 0u, 0u};
  int dlt[G2 ? 1 : 2][G2 ? 1 : N];  // edge workgroups: bytes each window was moved
  const bool edge = !G2 && !GL && ((b == 0 && ty0 == 0 && tx0 == 0) || (b == B - 1 && ty0 + TH >= H && tx0 + TW >= W));
  // ---- g1, GL: the tile's own gradient pixels, 21 eight-byte chunks each (the last one holds 2 bytes of data: loaded as a halfword so
  // that nothing past the pixel's 81 taps is ever read) -- 2688 chunks, 10.5 per thread; values are committed to LDS with the other
  // operand's below, after every global load of the tile has been issued
  constexpr int GIN = GL ? (TH * TW * GI_CH + 255) / 256 : 1;
  u32x2 gi[GIN];
  if (!G2 && GL) {
#pragma unroll
    for (int k = 0; k < GIN; k++) {
      const int it = tid + 256 * k;
      const int px = it / GI_CH, ch = it - px * GI_CH;
      const int pr = px >> 5, pc = px & 31;  // (TW == 32)
      const int gy = ty0 + pr, gx = tx0 + pc;
      const bool ok = it < TH * TW * GI_CH && gy < H && gx < W;
      const char* src = gbase + ((ok ? gy : 0) * gsh + (ok ? gx : 0) * gsw) * 2 + ch * 8;
      u32x2 v;
      if (ch == GI_CH - 1) { v = u32x2{(unsigned)*(const unsigned short*)src, 0u}; }
      else { v = ((const U8B*)src)->v; }
      const unsigned m = ok ? 0xffffffffu : 0u;
      gi[k] = u32x2{v[0] & m, v[1] & m};
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int t0 = 8 * G + 2 * r - i, t1 = t0 + 1;
      msk[r] = ((t0 >= 0 && t0 <= 8) ? 0xffffu : 0u) | ((t1 >= 0 && t1 <= 8) ? 0xffff0000u : 0u);
    }
    if (y >= H || G == 3) msk = u32x4{0u, 0u, 0u, 0u};
  }
  if (!G2 && !GL) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int t0 = 8 * G + 2 * r - i, t1 = t0 + 1;
      msk[r] = ((t0 >= 0 && t0 <= 8) ? 0xffffu : 0u) | ((t1 >= 0 && t1 <= 8) ? 0xffff0000u : 0u);
    }
    if (y >= H || G == 3 || (ablate & 1)) msk = u32x4{0u, 0u, 0u, 0u};
    const int yc = y < H ? y : H - 1;
    // (values are only loaded here; masks and the edge fix-up are applied after the staging loads below have been issued too)
    if (!edge) {
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int x = tx0 + 16 * s + i;
        const int xc = x < W ? x : W - 1;
        const char* pix = gbase + ((ablate & 1) ? 0 : (yc * gsh + xc * gsw + (G < 3 ? 8 * G - i : 0)) * 2);
#pragma unroll
        for (int dyi = 0; dyi < N; dyi++) bfr[s][dyi] = ((const U16B*)(pix + dyi * (N * 2)))->v;
      }
    } else {
      const char* lo = g.p;
      const char* hi = g.p + (((long)(B - 1) * g.sb + (long)(H - 1) * g.sh + (long)(W - 1) * g.sw + NK) * 2 - 16);
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int x = tx0 + 16 * s + i;
        const int xc = x < W ? x : W - 1;
        const char* pix = gbase + (yc * gsh + xc * gsw + 8 * G - i) * 2;
#pragma unroll
        for (int dyi = 0; dyi < N; dyi++) {
          const char* a = pix + dyi * (N * 2);
          const char* ac = a < lo ? lo : (a > hi ? hi : a);
          bfr[s][dyi] = ((const U16B*)ac)->v;
          dlt[s][dyi] = (int)(ac - a);
        }
      }
    }
  }

  // ---- g2: the tap windows of the halo pixels -> LDS.  Halo row hr (source row ty0 - 4 + hr) serves the targets w = 0..3 with tap
  // row dyi = w + 8 - hr: the window keeps tap rows ws .. ws + 3, ws = clamp(8 - hr, 0, 5).  360 eight-byte chunks per halo row:
  // chunk tid of every row (lane-fixed pixel and chunk, uniform row), then the remaining 104 per row spread over the workgroup.
  constexpr int WCH = WT * 2 / 8, WRC = HWP * WCH;  // 9 chunks per pixel, 360 per row
  constexpr int WNB = HR * (WRC - 256), WSB = (WNB + 255) / 256;
  u32x2 va[G2 ? HR : 1], vb[G2 ? WSB : 1];
  int ob[G2 ? WSB : 1];
  const int wpA = tid / WCH, wcA = tid - wpA * WCH;
  if (G2) {
    constexpr int CH = WCH, RC = WRC;
    const int pA = wpA, cA = wcA;
    const int gxA = tx0 - D + pA;
    const bool okA = gxA >= 0 && gxA < W;
    const int offA = ((ablate & 1) ? 0 : (gxA < 0 ? 0 : (gxA >= W ? W - 1 : gxA)) * gsw * 2) + cA * 8;
#pragma unroll
    for (int r = 0; r < HR; r++) {
      const int gy = ty0 - D + r;
      const int ws = 8 - r < 0 ? 0 : (8 - r > 5 ? 5 : 8 - r);
      const int gyc = (ablate & 1) ? 0 : (gy < 0 ? 0 : (gy >= H ? H - 1 : gy));
      const u32x2 ld = ((const U8B*)(gbase + gyc * gsh * 2 + ws * (N * 2) + offA))->v;
      const unsigned m = (okA && gy >= 0 && gy < H) ? 0xffffffffu : 0u;
      va[r] = u32x2{ld[0] & m, ld[1] & m};
    }
    constexpr int NB = WNB, SB = WSB;
#pragma unroll
    for (int j = 0; j < SB; j++) {
      const int it = tid + 256 * j;
      const int itc = it < NB ? it : 0;
      const int r = itc / (RC - 256), ch = 256 + itc - r * (RC - 256);
      const int px = ch / CH, c = ch - px * CH;
      const int gy = ty0 - D + r, gx = tx0 - D + px;
      const int ws = 8 - r < 0 ? 0 : (8 - r > 5 ? 5 : 8 - r);
      const int gyc = (ablate & 1) ? 0 : (gy < 0 ? 0 : (gy >= H ? H - 1 : gy)), gxc = (ablate & 1) ? 0 : (gx < 0 ? 0 : (gx >= W ? W - 1 : gx));
      const u32x2 ld = ((const U8B*)(gbase + (gyc * gsh + gxc * gsw + ws * N) * 2 + c * 8))->v;
      const unsigned m = (it < NB && gy >= 0 && gy < H && gx >= 0 && gx < W) ? 0xffffffffu : 0u;
      vb[j] = u32x2{ld[0] & m, ld[1] & m};
      ob[j] = it < NB ? (r * HWP + px) * GP + c * 8 : -1;
    }
  }

  // ---- the halo tile of the other operand for this channel block: [halo row][pixel][32 ch] in 8-byte chunks, 320 per halo row:
  // chunk tid of every row (lane-fixed pixel and channels, uniform row), then chunks 256..319: wave w takes rows w, w + 4, w + 8.
  {
    const int u8 = tid & 7, c0 = cbase + 4 * u8;
    const int c0c = c0 < C ? c0 : 0;
    const int pA = tid >> 3, pB = 32 + ((tid & 63) >> 3);
    const int gxA = tx0 - D + pA, gxB = tx0 - D + pB;
    const bool okA = gxA >= 0 && gxA < W && c0 < C, okB = gxB >= 0 && gxB < W && c0 < C;
    const int gxAc = (ablate & 2) ? 0 : (gxA < 0 ? 0 : (gxA >= W ? W - 1 : gxA)), gxBc = (ablate & 2) ? 0 : (gxB >= W ? W - 1 : gxB);
    const int offA = (gxAc * osw + c0c) * 2, offB = (gxBc * osw + c0c) * 2;
    const bool use_occ = !G2 && occ.p;
    const float* occb = (const float*)occ.p + (long)b * occ.sb;
    const int csh = (int)occ.sh;
    u32x2 sv[HR + 3];
    float so[HR + 3];
#pragma unroll
    for (int r = 0; r < HR + 3; r++) {
      const int hr = r < HR ? r : w + 4 * (r - HR);
      const int gy = ty0 - D + hr;
      const int gyc = (ablate & 2) ? 0 : (gy < 0 ? 0 : (gy >= H ? H - 1 : gy));
      const u32x2 ld = *(const u32x2*)(obase + gyc * osh * 2 + (r < HR ? offA : offB));
      const unsigned m = ((r < HR ? okA : okB) && gy >= 0 && gy < H) ? 0xffffffffu : 0u;
      sv[r] = u32x2{ld[0] & m, ld[1] & m};
      so[r] = 1.f;
      if (use_occ) so[r] = occb[gyc * csh + (r < HR ? gxAc : gxBc)];
    }
    if (!G2 && edge) {
#pragma unroll
      for (int s = 0; s < 2; s++)
#pragma unroll
        for (int dyi = 0; dyi < N; dyi++) {
          unsigned __int128 x128 = __builtin_bit_cast(unsigned __int128, bfr[s][dyi]);
          const int dl = dlt[s][dyi];
          if (dl > 0) x128 = dl >= 16 ? 0 : x128 << (8 * dl);
          if (dl < 0) x128 = dl <= -16 ? 0 : x128 >> (-8 * dl);
          bfr[s][dyi] = __builtin_bit_cast(u32x4, x128);
        }
    }
    if (!G2 && GL) {
#pragma unroll
      for (int k = 0; k < GIN; k++) {
        const int it = tid + 256 * k;
        if (it < TH * TW * GI_CH) *(u32x2*)(s_g + GI_FRONT + it * 8) = gi[k];  // chunk `it` of the tile: pixel-major, 21 chunks per pixel
      }
    }
    if (G2) {  // (stored only now: every global load of the tile is in flight before the first wait)
#pragma unroll
      for (int r = 0; r < HR; r++) *(u32x2*)(s_g + r * (HWP * GP) + wpA * GP + wcA * 8) = va[r];
#pragma unroll
      for (int j = 0; j < WSB; j++)
        if (ob[j] >= 0) *(u32x2*)(s_g + ob[j]) = vb[j];
    }
#pragma unroll
    for (int r = 0; r < HR + 3; r++) {
      u32x2 v = sv[r];
      if (use_occ) {  // rounded like the materialised product w2 * occ
        const float o = so[r];
        const float f0 = __builtin_bit_cast(float, v[0] << 16) * o, f1 = __builtin_bit_cast(float, v[0] & 0xffff0000u) * o;
        const float f2 = __builtin_bit_cast(float, v[1] << 16) * o, f3 = __builtin_bit_cast(float, v[1] & 0xffff0000u) * o;
        const bf16x2 lo2 = {(bf16)f0, (bf16)f1}, hi2 = {(bf16)f2, (bf16)f3};
        v = u32x2{__builtin_bit_cast(unsigned, lo2), __builtin_bit_cast(unsigned, hi2)};
      }
      const int hr = r < HR ? r : w + 4 * (r - HR);
      *(u32x2*)(s_f + (hr * HWP + (r < HR ? pA : pB)) * FP + u8 * 8) = v;
    }
  }
  __syncthreads();

  f32x4 acc[2][2];
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int nb = 0; nb < 2; nb++) acc[s][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (ablate & 4) {
  } else if (!G2) {
#pragma unroll
    for (int dyi = 0; dyi < N; dyi++) {
      const char* frow = s_f + (w + dyi) * (HWP * FP) + p * 8;
#pragma unroll
      for (int s = 0; s < 2; s++) {
        // k = 8G + 4h + q <-> halo pixel 16 s + k (clamped: k >= 24 is outside the band and, for s = 1, partly outside the halo)
        const int h0 = 16 * s + 8 * G + q, h1 = h0 + 4;
        const int p0 = h0 < HWP ? h0 : HWP - 1, p1 = h1 < HWP ? h1 : HWP - 1;
        const bool xin = tx0 + 16 * s + i < W;
        u32x4 raw;
        if (GL) {
          // taps 8G - i + 9 dyi + 0..7 of pixel (w, 16 s + i): 16 bytes at a 2-byte aligned LDS address = three aligned 8-byte reads
          // and a funnel shift (lanes of group 3 are outside the band: masked, any address)
          const int addr = GI_FRONT + (w * TW + 16 * s + i) * GIP + ((G < 3 ? 8 * G - i : 0) + N * dyi) * 2;
          const char* a8 = s_g + (addr & ~7);
          const u32x2 d0 = *(const u32x2*)a8, d1 = *(const u32x2*)(a8 + 8), d2 = *(const u32x2*)(a8 + 16);
          const bool up = (addr & 4) != 0;
          const unsigned q0 = up ? d0[1] : d0[0], q1 = up ? d1[0] : d0[1], q2 = up ? d1[1] : d1[0], q3 = up ? d2[0] : d1[1],
                         q4 = up ? d2[1] : d2[0];
          const bool half = (addr & 2) != 0;
          raw = half ? u32x4{__builtin_amdgcn_alignbyte(q1, q0, 2), __builtin_amdgcn_alignbyte(q2, q1, 2), __builtin_amdgcn_alignbyte(q3, q2, 2),
                             __builtin_amdgcn_alignbyte(q4, q3, 2)}
                     : u32x4{q0, q1, q2, q3};
        } else {
          raw = bfr[s][dyi];
        }
        const bf16x8 bq = __builtin_bit_cast(bf16x8, raw & (xin ? msk : u32x4{0u, 0u, 0u, 0u}));
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
          const bf16x8 a = tr_pair(frow + p0 * FP + nb * 32, frow + p1 * FP + nb * 32);
          acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq, acc[s][nb], 0, 0, 0);
        }
      }
    }
  } else {
    // element j of the lane: source pixel 4 j + G of the strip's 32, tap (x - x') + 4 = i - 4 j - G + 8
    unsigned gm[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int t0 = i - 8 * r - G + 8, t1 = t0 - 4;
      gm[r] = ((t0 >= 0 && t0 <= 8) ? 0xffffu : 0u) | ((t1 >= 0 && t1 <= 8) ? 0xffff0000u : 0u);
    }
    const char* glane = s_g + G * GP + (i - G + 8) * 2;
#pragma unroll
    for (int dyi = 0; dyi < N; dyi++) {
      const int hr = w + 8 - dyi;  // source row y - dy
      const int ws = dyi - w < 0 ? 0 : (dyi - w > 5 ? 5 : dyi - w);  // = clamp(8 - hr, 0, 5)
      const char* grow = glane + hr * (HWP * GP) + (dyi - ws) * (N * 2);
      const char* frow = s_f + hr * (HWP * FP) + p * 8;
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const char* gs = grow + 16 * s * GP;
        u32x4 bv;
#pragma unroll
        for (int r = 0; r < 3; r++) {
          const unsigned e0 = *(const unsigned short*)(gs + (2 * r) * (4 * GP - 8));
          const unsigned e1 = *(const unsigned short*)(gs + (2 * r + 1) * (4 * GP - 8));
          bv[r] = (e0 | (e1 << 16)) & gm[r];
        }
        bv[3] = 0u;
        const bf16x8 bq = __builtin_bit_cast(bf16x8, bv);
        const int h0 = 16 * s + 4 * q + G, h1 = h0 + 16;
        const int p1 = h1 < HWP ? h1 : HWP - 1;
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
          const bf16x8 a = tr_pair(frow + h0 * FP + nb * 32, frow + p1 * FP + nb * 32);
          acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq, acc[s][nb], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: lane (i, G) holds channels cbase + 16 nb + 4 G + 0..3 of pixel (y, tx0 + 16 s + i)
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int x = tx0 + 16 * s + i;
    const bool inimg = y < H && x < W;
    float o = 1.f, part = 0.f;
    if (G2 && occ.p && inimg) o = ((const float*)occ.p)[(long)b * occ.sb + y * (int)occ.sh + x * (int)occ.sw];
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
      const int c0 = cbase + 16 * nb + 4 * G;
      const bool live = inimg && c0 < C;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = acc[s][nb][r] * invC;
      if (G2 && occ.p) {
        if (gocc.p && live) {
          // the reference chain rounds the gradient w.r.t. (w2*occ) to the feature dtype before this product
          const u32x2 wv = *(const u32x2*)(w2own.p + (long)b * w2own.sb * 2 + (y * (int)w2own.sh + x * (int)w2own.sw + c0) * 2);
          part = __builtin_fmaf((float)(bf16)v[0], __builtin_bit_cast(float, wv[0] << 16), part);
          part = __builtin_fmaf((float)(bf16)v[1], __builtin_bit_cast(float, wv[0] & 0xffff0000u), part);
          part = __builtin_fmaf((float)(bf16)v[2], __builtin_bit_cast(float, wv[1] << 16), part);
          part = __builtin_fmaf((float)(bf16)v[3], __builtin_bit_cast(float, wv[1] & 0xffff0000u), part);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = (float)(bf16)v[r] * o;
      }
      if (live && !(ablate & 8)) {
        const bf16x2 lo2 = {(bf16)v[0], (bf16)v[1]}, hi2 = {(bf16)v[2], (bf16)v[3]};
        *(u32x2*)(dst.p + (long)b * dst.sb * 2 + (y * (int)dst.sh + x * (int)dst.sw + c0) * 2) =
            u32x2{__builtin_bit_cast(unsigned, lo2), __builtin_bit_cast(unsigned, hi2)};
      }
    }
    if (G2 && occ.p && gocc.p) {  // one fp32 atomic per pixel and channel block (the four lane groups reduced by shuffle)
      part += __shfl_xor(part, 16, 64);
      part += __shfl_xor(part, 32, 64);
      if (inimg && G == 0) atomicAdd((float*)gocc.p + (long)b * gocc.sb + (y * (int)gocc.sh + x * (int)gocc.sw), part);
    }
  }
}

template <bool G2, bool GL = false>
__global__ __launch_bounds__(256, (G2 || GL) ? 2 : 3) void corr_bwd_mfma(T4 other, T4 g, T4 dst, T4 occ, T4 w2own, T4 gocc, int B, int C, int H,
                                                                int W, int tiles_x, int tiles_y, int ablate) {
  using namespace bm;
#ifndef OCF_TUNING_BUILD
  ablate = 0;  // phase-skipping switches of the tuning build (scripts/ablate_corr_bwd.sh)
#endif
  __shared__ __attribute__((aligned(16))) char s_f[G2 ? SF2 : SF1];
  __shared__ __attribute__((aligned(16))) char s_g[G2 ? SG2 : (GL ? GI_BYTES : 16)];
  if (ablate & 16) {  // dispatch floor: the grid with its LDS allocation and nothing else
    if (threadIdx.x == 0) s_f[0] = 1;
    return;
  }
  corr_bwd_pass<G2, GL>(s_f, s_g, other, g, dst, occ, w2own, gocc, B, C, H, W, tiles_x, tiles_y, ablate);
}

// BOTH gradients of a tile in ONE launch (round 3): the g1 pass, a barrier, the g2 pass, sharing one 73 KB LDS allocation (two
// workgroups per CU).  What it buys over the two launches: one dispatch per level instead of two (five levels per step), and the second
// read of the tile's gradient rows -- g2 re-stages them as the halo windows of the SAME workgroup a few microseconds after g1 read them
// -- is an L2 hit instead of a second trip to HBM (rocprofv3 round 2: 188.8 MB of traffic for 96.9 MB of algorithmic bytes at
// 8 x 32 x 109 x 256, the gradient being fetched by both launches at its 256-byte in-model pitch).
template <bool GL>
__global__ __launch_bounds__(256, 2) void corr_bwd_mfma_both(T4 f1, T4 w2, T4 g, T4 g1, T4 g2, T4 occ, T4 gocc, int B, int C, int H, int W, int tiles_x,
                                                            int tiles_y) {
  using namespace bm;
  constexpr int BYTES = (SF1 + (GL ? GI_BYTES : 16)) > (SF2 + SG2) ? (SF1 + (GL ? GI_BYTES : 16)) : (SF2 + SG2);
  __shared__ __attribute__((aligned(16))) char smem[BYTES];
  const T4 none = {nullptr, 0, 0, 0, 0};
  corr_bwd_pass<false, GL>(smem, smem + SF1, w2, g, g1, occ, none, none, B, C, H, W, tiles_x, tiles_y, 0);
  __syncthreads();  // every fragment read of the g1 pass is done before the g2 pass re-uses the allocation
  corr_bwd_pass<true, false>(smem, smem + SF2, f1, g, g2, occ, w2, gocc, B, C, H, W, tiles_x, tiles_y, 0);
}

// Both gradients in ONE launch with the two passes in DIFFERENT workgroups (blockIdx.z = pass): for the small pyramid levels, where one
// pass does not fill the 256 CUs and a launch is a single latency chain (load, commit, compute, store) -- run one after the other inside
// a workgroup the two chains add up (12.6 us at 8 x 196 x 7 x 16 against 8.0 / 6.0 for the passes alone), side by side they overlap.
template <bool GL>
__global__ __launch_bounds__(256, 2) void corr_bwd_mfma_par(T4 f1, T4 w2, T4 g, T4 g1, T4 g2, T4 occ, T4 gocc, int B, int C, int H, int W, int tiles_x,
                                                           int tiles_y) {
  using namespace bm;
  constexpr int BYTES = (SF1 + (GL ? GI_BYTES : 16)) > (SF2 + SG2) ? (SF1 + (GL ? GI_BYTES : 16)) : (SF2 + SG2);
  __shared__ __attribute__((aligned(16))) char smem[BYTES];
  const T4 none = {nullptr, 0, 0, 0, 0};
  if (blockIdx.z == 0)
    corr_bwd_pass<false, GL>(smem, smem + SF1, w2, g, g1, occ, none, none, B, C, H, W, tiles_x, tiles_y, 0);
  else
    corr_bwd_pass<true, false>(smem, smem + SF2, f1, g, g2, occ, w2, gocc, B, C, H, W, tiles_x, tiles_y, 0);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The phases of the two passes as separate functions (round 3): issue (global loads into registers), commit (registers -> LDS), compute
// (fragments, matrix-core loop, epilogue).  Same LDS layouts, fragments and epilogues as corr_bwd_pass above, bit-identical results;
// differences: the gradient of the g1 pass is staged with lane-fixed chunk columns and a branch-free tail chunk, and every global
// address is a uniform base + an unsigned 32-bit lane offset (the scalar-base addressing mode: no 64-bit vector address arithmetic).
// They were written for a PIPELINED kernel -- a workgroup owning a run of (tile, channel block) items, the loads of the next pass issued
// before the current pass computes.  That kernel was correct and SLOWER (8 x 32 x 109 x 256, gradient at a 256-byte pitch, +occ: 77.8 us
// against 62.3; profiles/r3_h_ab_bwd_variants.txt): the passes are bound by instruction issue at two waves per SIMD, not by exposed
// load latency, and the loop form adds instructions (lane constants re-derived per phase to keep the compiler from hoisting 250 VGPRs
// of addresses) and loses the dynamic balance of one workgroup per tile.  It was removed; the phase functions stayed.
// ---------------------------------------------------------------------------------------------------------------------------------
namespace bp {
using namespace bm;
constexpr int GPX = 12;                          // gradient pixels staged per iteration: 12 x 21 chunks = 252 lanes
constexpr int GIT = (TH * TW + GPX - 1) / GPX;   // 11 iterations
constexpr int WCH = WT * 2 / 8, WRC = HWP * WCH; // g2 windows: 9 chunks per pixel, 360 per halo row
constexpr int WNB = HR * (WRC - 256), WSB = (WNB + 255) / 256;

struct Lane {
  int tid, lane, w, i, G, q, p;
  int gsh, gsw;
};
struct Item { int b, ty0, tx0, cbase; };
struct R1 { u32x2 gi[GIT]; u32x2 sv[HR + 3]; float so[HR + 3]; };
struct R2 { u32x2 va[HR]; u32x2 vb[WSB]; u32x2 sv[HR + 3]; };

__device__ __forceinline__ Item item_of(int it, int tiles_x, int tiles_y, int cblocks) {
  const int tile = it / cblocks, cb = it - tile * cblocks;
  const int tpi = tiles_x * tiles_y;
  const int b = tile / tpi, trem = tile - b * tpi;
  return Item{b, (trem / tiles_x) * TH, (trem % tiles_x) * TW, cb * 32};
}

// halo tile of the other operand (w2 for g1, f1 for g2): chunk tid of every halo row, then chunks 256..319 (wave w: rows w, w + 4, w + 8)
// (The issue functions only LOAD -- raw values, clamped addresses; validity masks, the tail-chunk shift and the occlusion product are
// applied by the commit functions, which recompute the conditions.  With the mask applied right after each load the compiler kept
// program order and put an `s_waitcnt vmcnt(0)` behind every one of the twelve window loads of the g2 pass: twelve dependent round
// trips instead of one.)
template <bool OCC>
__device__ __forceinline__ void issue_other(const Lane& L, const Item& t, const T4& other, const T4& occ, int C, int H, int W, u32x2* sv, float* so) {
  const int osh = (int)other.sh, osw = (int)other.sw;
  const char* obase = other.p + (long)t.b * other.sb * 2;
  const int u8 = L.tid & 7, c0 = t.cbase + 4 * u8;
  const int c0c = c0 < C ? c0 : 0;
  const int pA = L.tid >> 3, pB = 32 + ((L.tid & 63) >> 3);
  const int gxA = t.tx0 - D + pA, gxB = t.tx0 - D + pB;
  const int gxAc = gxA < 0 ? 0 : (gxA >= W ? W - 1 : gxA), gxBc = gxB >= W ? W - 1 : gxB;
  const int offA = (gxAc * osw + c0c) * 2, offB = (gxBc * osw + c0c) * 2;
  const float* occb = OCC ? (const float*)occ.p + (long)t.b * occ.sb : nullptr;
  const int csh = OCC ? (int)occ.sh : 0;
#pragma unroll
  for (int r = 0; r < HR + 3; r++) {
    const int hr = r < HR ? r : L.w + 4 * (r - HR);
    const int gy = t.ty0 - D + hr;
    const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
    sv[r] = *(const u32x2*)(obase + (unsigned)(gyc * osh * 2 + (r < HR ? offA : offB)));
    if (OCC) so[r] = occb[(unsigned)(gyc * csh + (r < HR ? gxAc : gxBc))];
  }
}

template <bool OCC, int FP>
__device__ __forceinline__ void commit_other(const Lane& L, const Item& t, char* s_f, const u32x2* sv, const float* so, int C, int H, int W) {
  const int u8 = L.tid & 7, c0 = t.cbase + 4 * u8;
  const int pA = L.tid >> 3, pB = 32 + ((L.tid & 63) >> 3);
  const int gxA = t.tx0 - D + pA, gxB = t.tx0 - D + pB;
  const bool okA = gxA >= 0 && gxA < W && c0 < C, okB = gxB >= 0 && gxB < W && c0 < C;
#pragma unroll
  for (int r = 0; r < HR + 3; r++) {
    const int hr = r < HR ? r : L.w + 4 * (r - HR);
    const int gy = t.ty0 - D + hr;
    const unsigned m = ((r < HR ? okA : okB) && gy >= 0 && gy < H) ? 0xffffffffu : 0u;
    u32x2 v = u32x2{sv[r][0] & m, sv[r][1] & m};
    if (OCC) {  // rounded like the materialised product w2 * occ
      const float o = so[r];
      const float f0 = __builtin_bit_cast(float, v[0] << 16) * o, f1 = __builtin_bit_cast(float, v[0] & 0xffff0000u) * o;
      const float f2 = __builtin_bit_cast(float, v[1] << 16) * o, f3 = __builtin_bit_cast(float, v[1] & 0xffff0000u) * o;
      const bf16x2 lo2 = {(bf16)f0, (bf16)f1}, hi2 = {(bf16)f2, (bf16)f3};
      v = u32x2{__builtin_bit_cast(unsigned, lo2), __builtin_bit_cast(unsigned, hi2)};
    }
    *(u32x2*)(s_f + (hr * HWP + (r < HR ? pA : pB)) * FP + u8 * 8) = v;
  }
}

// g1: the 81 taps of the tile's own pixels, 21 eight-byte chunks each; lane = (pixel tid / 21 of the iteration's 12, chunk tid % 21).
// The last chunk holds 2 bytes of data: it is fetched as the 8 bytes that END at the pixel's last tap and shifted down, so that nothing
// past the 81 taps is read and there is no branch.
__device__ __forceinline__ void issue_g1(const Lane& L, const Item& t, const T4& g, int H, int W, u32x2* gi) {
  const char* gbase = g.p + (long)t.b * g.sb * 2;
  const int p0 = L.tid / GI_CH, ch = L.tid - p0 * GI_CH;
  const int coff = ch == GI_CH - 1 ? NK * 2 - 8 : ch * 8;
#pragma unroll
  for (int k = 0; k < GIT; k++) {
    const int px = p0 + GPX * k;
    const int pr = px >> 5, pc = px & 31;  // (TW == 32)
    const int gy = t.ty0 + pr, gx = t.tx0 + pc;
    const bool ok = L.tid < GPX * GI_CH && px < TH * TW && gy < H && gx < W;
    gi[k] = ((const U8B*)(gbase + (unsigned)(((ok ? gy : 0) * L.gsh + (ok ? gx : 0) * L.gsw) * 2 + coff)))->v;
  }
}

__device__ __forceinline__ void commit_g1(const Lane& L, const Item& t, char* s_g, const u32x2* gi, int H, int W) {
  const int p0 = L.tid / GI_CH, ch = L.tid - p0 * GI_CH;
  const bool tail = ch == GI_CH - 1;
#pragma unroll
  for (int k = 0; k < GIT; k++) {
    const int px = p0 + GPX * k;
    const int pr = px >> 5, pc = px & 31;
    const bool ok = t.ty0 + pr < H && t.tx0 + pc < W;
    const unsigned m = ok ? 0xffffffffu : 0u;
    const u32x2 v = tail ? u32x2{(gi[k][1] >> 16) & m, 0u} : u32x2{gi[k][0] & m, gi[k][1] & m};
    if (L.tid < GPX * GI_CH && px < TH * TW) *(u32x2*)(s_g + GI_FRONT + px * GIP + ch * 8) = v;
  }
}

// g2: the tap windows (4 tap rows) of the halo pixels
__device__ __forceinline__ void issue_g2(const Lane& L, const Item& t, const T4& g, int H, int W, u32x2* va, u32x2* vb) {
  const char* gbase = g.p + (long)t.b * g.sb * 2;
  const int pA = L.tid / WCH, cA = L.tid - pA * WCH;
  const int gxA = t.tx0 - D + pA;
  const int offA = (gxA < 0 ? 0 : (gxA >= W ? W - 1 : gxA)) * L.gsw * 2 + cA * 8;
#pragma unroll
  for (int r = 0; r < HR; r++) {
    const int gy = t.ty0 - D + r;
    const int ws = 8 - r < 0 ? 0 : (8 - r > 5 ? 5 : 8 - r);
    const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
    va[r] = ((const U8B*)(gbase + (unsigned)(gyc * L.gsh * 2 + ws * (N * 2) + offA)))->v;
  }
#pragma unroll
  for (int j = 0; j < WSB; j++) {
    const int it = L.tid + 256 * j;
    const int itc = it < WNB ? it : 0;
    const int r = itc / (WRC - 256), chn = 256 + itc - r * (WRC - 256);
    const int px = chn / WCH, c = chn - px * WCH;
    const int gy = t.ty0 - D + r, gx = t.tx0 - D + px;
    const int ws = 8 - r < 0 ? 0 : (8 - r > 5 ? 5 : 8 - r);
    const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
    vb[j] = ((const U8B*)(gbase + (unsigned)((gyc * L.gsh + gxc * L.gsw + ws * N) * 2 + c * 8)))->v;
  }
}

__device__ __forceinline__ void commit_g2(const Lane& L, const Item& t, char* s_g, const u32x2* va, const u32x2* vb, int H, int W) {
  const int pA = L.tid / WCH, cA = L.tid - pA * WCH;
  const int gxA = t.tx0 - D + pA;
  const bool okA = gxA >= 0 && gxA < W;
#pragma unroll
  for (int r = 0; r < HR; r++) {
    const int gy = t.ty0 - D + r;
    const unsigned m = (okA && gy >= 0 && gy < H) ? 0xffffffffu : 0u;
    *(u32x2*)(s_g + r * (HWP * GP) + pA * GP + cA * 8) = u32x2{va[r][0] & m, va[r][1] & m};
  }
#pragma unroll
  for (int j = 0; j < WSB; j++) {
    const int it = L.tid + 256 * j;
    const int r = it / (WRC - 256), chn = 256 + it - r * (WRC - 256);
    const int px = chn / WCH, c = chn - px * WCH;
    const int gy = t.ty0 - D + r, gx = t.tx0 - D + px;
    const unsigned m = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? 0xffffffffu : 0u;
    if (it < WNB) *(u32x2*)(s_g + (r * HWP + px) * GP + c * 8) = u32x2{vb[j][0] & m, vb[j][1] & m};
  }
}

__device__ __forceinline__ void compute_g1(const Lane& L, const Item& t, const char* s_f, const char* s_g, const T4& dst, int C, int H, int W) {
  const int i = L.i, G = L.G, w = L.w;
  const int y = t.ty0 + w;
  u32x4 msk;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int t0 = 8 * G + 2 * r - i, t1 = t0 + 1;
    msk[r] = ((t0 >= 0 && t0 <= 8) ? 0xffffu : 0u) | ((t1 >= 0 && t1 <= 8) ? 0xffff0000u : 0u);
  }
  if (y >= H || G == 3) msk = u32x4{0u, 0u, 0u, 0u};
  f32x4 acc[2][2];
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int nb = 0; nb < 2; nb++) acc[s][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dyi = 0; dyi < N; dyi++) {
    const char* frow = s_f + (w + dyi) * (HWP * FP1) + L.p * 8;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int h0 = 16 * s + 8 * G + L.q, h1 = h0 + 4;
      const int p0 = h0 < HWP ? h0 : HWP - 1, p1 = h1 < HWP ? h1 : HWP - 1;
      const bool xin = t.tx0 + 16 * s + i < W;
      const int addr = GI_FRONT + (w * TW + 16 * s + i) * GIP + ((G < 3 ? 8 * G - i : 0) + N * dyi) * 2;
      const char* a8 = s_g + (addr & ~7);
      const u32x2 d0 = *(const u32x2*)a8, d1 = *(const u32x2*)(a8 + 8), d2 = *(const u32x2*)(a8 + 16);
      const bool up = (addr & 4) != 0;
      const unsigned q0 = up ? d0[1] : d0[0], q1 = up ? d1[0] : d0[1], q2 = up ? d1[1] : d1[0], q3 = up ? d2[0] : d1[1], q4 = up ? d2[1] : d2[0];
      const bool half = (addr & 2) != 0;
      const u32x4 raw = half ? u32x4{__builtin_amdgcn_alignbyte(q1, q0, 2), __builtin_amdgcn_alignbyte(q2, q1, 2),
                                     __builtin_amdgcn_alignbyte(q3, q2, 2), __builtin_amdgcn_alignbyte(q4, q3, 2)}
                                 : u32x4{q0, q1, q2, q3};
      const bf16x8 bq = __builtin_bit_cast(bf16x8, raw & (xin ? msk : u32x4{0u, 0u, 0u, 0u}));
#pragma unroll
      for (int nb = 0; nb < 2; nb++) {
        const bf16x8 a = tr_pair(frow + p0 * FP1 + nb * 32, frow + p1 * FP1 + nb * 32);
        acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq, acc[s][nb], 0, 0, 0);
      }
    }
  }
  const float invC = 1.f / (float)C;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int x = t.tx0 + 16 * s + i;
    const bool inimg = y < H && x < W;
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
      const int c0 = t.cbase + 16 * nb + 4 * G;
      if (inimg && c0 < C) {
        const bf16x2 lo2 = {(bf16)(acc[s][nb][0] * invC), (bf16)(acc[s][nb][1] * invC)};
        const bf16x2 hi2 = {(bf16)(acc[s][nb][2] * invC), (bf16)(acc[s][nb][3] * invC)};
        *(u32x2*)(dst.p + (long)t.b * dst.sb * 2 + (y * (int)dst.sh + x * (int)dst.sw + c0) * 2) =
            u32x2{__builtin_bit_cast(unsigned, lo2), __builtin_bit_cast(unsigned, hi2)};
      }
    }
  }
}

template <bool OCC>
__device__ __forceinline__ void compute_g2(const Lane& L, const Item& t, const char* s_f, const char* s_g, const T4& dst, const T4& occ,
                                           const T4& w2own, const T4& gocc, int C, int H, int W, bool gocc_plain) {
  const int i = L.i, G = L.G, w = L.w;
  const int y = t.ty0 + w;
  f32x4 acc[2][2];
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int nb = 0; nb < 2; nb++) acc[s][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the epilogue's own operands (this pixel's mask value, this lane's four channels of the warped features for d/docc) are requested
  // BEFORE the matrix-core loop: loaded inside the epilogue they were four dependent round trips after the last MFMA
  float oo[2] = {1.f, 1.f};
  u32x2 wv[2][2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int x = t.tx0 + 16 * s + i;
    const int yc = y < H ? y : H - 1, xc = x < W ? x : W - 1;
    if (OCC) oo[s] = ((const float*)occ.p)[(long)t.b * occ.sb + (unsigned)(yc * (int)occ.sh + xc * (int)occ.sw)];
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
      const int c0 = t.cbase + 16 * nb + 4 * G;
      wv[s][nb] = u32x2{0u, 0u};  // (raw, from a clamped address: the epilogue only uses it for live pixels / channels)
      if (OCC && gocc.p)
        wv[s][nb] = *(const u32x2*)(w2own.p + (long)t.b * w2own.sb * 2 + (unsigned)((yc * (int)w2own.sh + xc * (int)w2own.sw + (c0 < C ? c0 : 0)) * 2));
    }
  }
  // the epilogue's ADDRESSES (this lane's two output pixels in dst and in d/docc) are formed here, into vector registers: their
  // tensor descriptors (pointer + three strides each) then do not stay live in scalar registers across the unrolled matrix-core
  // loop -- that cost nine SGPR spills in the <OCC = true> instantiation; the kernel has vector registers to spare (92 of 256)
  char* dp[2];
  float* gop[2];
  const bool has_gocc = OCC && gocc.p != nullptr;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int x = t.tx0 + 16 * s + i;
    dp[s] = dst.p + (long)t.b * dst.sb * 2 + (y * (int)dst.sh + x * (int)dst.sw + t.cbase + 4 * G) * 2;
    gop[s] = has_gocc ? (float*)gocc.p + (long)t.b * gocc.sb + (y * (int)gocc.sh + x * (int)gocc.sw) : nullptr;
    asm volatile("" : "+v"(dp[s]), "+v"(gop[s]));
  }
  unsigned gm[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int t0 = i - 8 * r - G + 8, t1 = t0 - 4;
    gm[r] = ((t0 >= 0 && t0 <= 8) ? 0xffffu : 0u) | ((t1 >= 0 && t1 <= 8) ? 0xffff0000u : 0u);
  }
  const char* glane = s_g + G * GP + (i - G + 8) * 2;
#pragma unroll
  for (int dyi = 0; dyi < N; dyi++) {
    const int hr = w + 8 - dyi;  // source row y - dy
    const int ws = dyi - w < 0 ? 0 : (dyi - w > 5 ? 5 : dyi - w);
    const char* grow = glane + hr * (HWP * GP) + (dyi - ws) * (N * 2);
    const char* frow = s_f + hr * (HWP * FP2) + L.p * 8;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const char* gs = grow + 16 * s * GP;
      u32x4 bv;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const unsigned e0 = *(const unsigned short*)(gs + (2 * r) * (4 * GP - 8));
        const unsigned e1 = *(const unsigned short*)(gs + (2 * r + 1) * (4 * GP - 8));
        bv[r] = (e0 | (e1 << 16)) & gm[r];
      }
      bv[3] = 0u;
      const bf16x8 bq = __builtin_bit_cast(bf16x8, bv);
      const int h0 = 16 * s + 4 * L.q + G, h1 = h0 + 16;
      const int p1 = h1 < HWP ? h1 : HWP - 1;
#pragma unroll
      for (int nb = 0; nb < 2; nb++) {
        const bf16x8 a = tr_pair(frow + h0 * FP2 + nb * 32, frow + p1 * FP2 + nb * 32);
        acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq, acc[s][nb], 0, 0, 0);
      }
    }
  }
  // the prefetched epilogue operands are CONSUMED here, while no store is outstanding: the stores below are conditional, the compiler
  // cannot count them, and a wait for these loads placed after the first store becomes `vmcnt(0)` = wait for the stores as well
  if (OCC) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      asm volatile("" : "+v"(oo[s]));
#pragma unroll
      for (int nb = 0; nb < 2; nb++) asm volatile("" : "+v"(wv[s][nb][0]), "+v"(wv[s][nb][1]));
    }
  }
  const float invC = 1.f / (float)C;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int x = t.tx0 + 16 * s + i;
    const bool inimg = y < H && x < W;
    const float o = oo[s];
    float part = 0.f;
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
      const int c0 = t.cbase + 16 * nb + 4 * G;
      const bool live = inimg && c0 < C;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = acc[s][nb][r] * invC;
      if (OCC) {
        if (has_gocc && live) {  // the reference chain rounds the gradient w.r.t. (w2*occ) to the feature dtype before this product
          const u32x2 w4 = wv[s][nb];
          part = __builtin_fmaf((float)(bf16)v[0], __builtin_bit_cast(float, w4[0] << 16), part);
          part = __builtin_fmaf((float)(bf16)v[1], __builtin_bit_cast(float, w4[0] & 0xffff0000u), part);
          part = __builtin_fmaf((float)(bf16)v[2], __builtin_bit_cast(float, w4[1] << 16), part);
          part = __builtin_fmaf((float)(bf16)v[3], __builtin_bit_cast(float, w4[1] & 0xffff0000u), part);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = (float)(bf16)v[r] * o;
      }
      if (live) {
        const bf16x2 lo2 = {(bf16)v[0], (bf16)v[1]}, hi2 = {(bf16)v[2], (bf16)v[3]};
        *(u32x2*)(dp[s] + 32 * nb) = u32x2{__builtin_bit_cast(unsigned, lo2), __builtin_bit_cast(unsigned, hi2)};
      }
    }
    if (has_gocc) {  // one fp32 atomic per pixel and channel block (the four lane groups reduced by shuffle)
      part += __shfl_xor(part, 16, 64);
      part += __shfl_xor(part, 32, 64);
      if (inimg && G == 0) {
        float* go = gop[s];
        if (gocc_plain) *go = part;  // (one channel block: this lane owns the pixel's value; the launcher did not zero the tensor)
        else atomicAdd(go, part);
      }
    }
  }
}
}  // namespace bp

// the two passes side by side (blockIdx.z), built from the phase functions above; the gradient of the g1 pass through LDS
template <bool OCC>
__global__ __launch_bounds__(256, 2) void corr_bwd_mfma_par_lds(T4 f1, T4 w2, T4 g, T4 g1, T4 g2, T4 occ, T4 gocc, int B, int C, int H, int W,
                                                               int tiles_x, int tiles_y, int ncb) {
  using namespace bp;
  constexpr int BYTES = (SF1 + GI_BYTES) > (SF2 + SG2) ? (SF1 + GI_BYTES) : (SF2 + SG2);
  __shared__ __attribute__((aligned(16))) char smem[BYTES];
  Lane L;
  L.tid = threadIdx.x; L.lane = L.tid & 63;
  L.w = __builtin_amdgcn_readfirstlane(L.tid >> 6);
  L.i = L.lane & 15; L.G = L.lane >> 4; L.q = L.i >> 2; L.p = L.i & 3;
  L.gsh = (int)g.sh; L.gsw = (int)g.sw;
  // ONE grid dimension: logical id = (tile, channel block, pass), pass fastest, and logical ids are contiguous per XCD (xcd_remap): the
  // 2 * ncb workgroups that stage the SAME grad_out tile are dispatched back to back on one XCD, so one of them brings it in from HBM
  // and the others hit that XCD's L2.  (With the pass in blockIdx.z and the channel block in blockIdx.y they ran a whole grid apart:
  // FETCH 1.95x the algorithmic bytes at level 2.)
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int pass = lid & 1, tc = lid >> 1;
  const int tile = tc / ncb, cb = tc - tile * ncb;
  const int tpi = tiles_x * tiles_y;
  const int b = tile / tpi, trem = tile - b * tpi;
  const Item t = {b, (trem / tiles_x) * TH, (trem % tiles_x) * TW, cb * 32};
  if (pass == 0) {
    R1 r1;
    issue_g1(L, t, g, H, W, r1.gi);
    issue_other<OCC>(L, t, w2, occ, C, H, W, r1.sv, r1.so);
    commit_g1(L, t, smem + SF1, r1.gi, H, W);
    commit_other<OCC, FP1>(L, t, smem, r1.sv, r1.so, C, H, W);
    __syncthreads();
    compute_g1(L, t, smem, smem + SF1, g1, C, H, W);
  } else {
    R2 r2;
    issue_g2(L, t, g, H, W, r2.va, r2.vb);
    issue_other<false>(L, t, f1, occ, C, H, W, r2.sv, nullptr);
    commit_g2(L, t, smem + SF2, r2.va, r2.vb, H, W);
    commit_other<false, FP2>(L, t, smem, r2.sv, nullptr, C, H, W);
    __syncthreads();
    compute_g2<OCC>(L, t, smem, smem + SF2, g2, occ, w2, gocc, C, H, W, ncb == 1);
  }
}

static bool px8(const ocf_tensor4* t) {  // channel-contiguous, pixels and rows 8-byte aligned, forward strides
  return t->sc == 1 && t->sw > 0 && t->sh > 0 && t->sb > 0 && t->sw % 4 == 0 && t->sh % 4 == 0 && t->sb % 4 == 0 && (uintptr_t)t->data % 8 == 0;
}

bool corr_bwd_mfma_eligible(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* g, const ocf_tensor4* g1, const ocf_tensor4* g2,
                            int B, int C, int H, int W, int d, int policy) {
  if (policy == OCF_CORR_BWD_POLICY_VECTOR || d != bm::D || C % 4 != 0) return false;
  if (!px8(f1) || !px8(w2) || (g1 && !px8(g1)) || (g2 && !px8(g2))) return false;
  if (!(g->sc == 1 && g->sw >= bm::NK && g->sh > 0 && g->sb > 0)) return false;
  // byte offsets inside one image are 32-bit in the kernel
  auto span = [&](const ocf_tensor4* t, long c) { return ((long)t->sh * H + (long)t->sw * W + c) * 2; };
  if (span(f1, C) >= (1L << 31) || span(w2, C) >= (1L << 31) || span(g, bm::NK) >= (1L << 31)) return false;
  if ((g1 && span(g1, C) >= (1L << 31)) || (g2 && span(g2, C) >= (1L << 31))) return false;
  return true;  // measured faster than the vector kernel at every pyramid level of the benchmark (profiles/r2_e_kbench.txt)
}

int corr_bwd_bf16_mfma(const ocf_tensor4* f1, const ocf_tensor4* w2, const ocf_tensor4* g, const ocf_tensor4* g1, const ocf_tensor4* g2,
                       int B, int C, int H, int W, hipStream_t st, const ocf_tensor4* occ, const ocf_tensor4* gocc) {
  using namespace bm;
  const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;
  const dim3 grid((unsigned)(B * tx * ty), (unsigned)((C + 31) / 32)), block(256);
#ifdef OCF_TUNING_BUILD
  const int ablate = getenv("OCF_ABLATE_BWD") ? atoi(getenv("OCF_ABLATE_BWD")) : 0;
#else
  constexpr int ablate = 0;
#endif
  // the gradient through LDS when it arrives as a slice of a wider tensor (pixel pitch != 81 elements): see the note above the kernel
  bool g1_lds = g->sw != NK;
#ifdef OCF_TUNING_BUILD
  if (const char* e = getenv("OCF_BWD_G1_LDS")) g1_lds = atoi(e) != 0;
#endif
  bool both = g1 && g2;
#ifdef OCF_TUNING_BUILD
  if (const char* e = getenv("OCF_BWD_FUSED")) both = both && atoi(e) != 0;
#endif
  // both gradients: the two passes side by side in different workgroups of ONE launch (measured faster than one after the other inside
  // a workgroup at every pyramid level: 9.0 / 9.8 / 17.7 / 32.9 / 61.0 us against 14.2 / 15.5 / 18.0 / 35.2 / 62.3)
  bool par = both;
#ifdef OCF_TUNING_BUILD
  if (const char* e = getenv("OCF_BWD_PAR")) par = both && atoi(e) != 0;
  const bool par_phase = getenv("OCF_BWD_PAR_PHASE") ? atoi(getenv("OCF_BWD_PAR_PHASE")) != 0 : true;
#else
  constexpr bool par_phase = true;
#endif
  // d/docc: with ONE channel block every pixel's value is written by exactly one lane of the phase-function kernel (plain store, no
  // zero-fill launch); with several channel blocks -- and in the other kernels -- it is accumulated with float atomics into zeroed memory
  const bool gocc_store = par && g1_lds && par_phase && C <= 32;
  if (gocc && gocc->data && !gocc_store && hipMemsetAsync(gocc->data, 0, sizeof(float) * (size_t)B * H * W, st) != hipSuccess) return OCF_ELAUNCH;
  LaunchProf lp;
  if (par) {
    const dim3 grid2((unsigned)(B * tx * ty), (unsigned)((C + 31) / 32), 2u);
    const bool has_occ = occ && occ->data;
    const int ncb = (C + 31) / 32;
    const dim3 grid1d((unsigned)(B * tx * ty) * (unsigned)ncb * 2u);
    if (g1_lds && par_phase && has_occ)
      lp.launch(true, corr_bwd_mfma_par_lds<true>, grid1d, block, 0, st, mk(f1), mk(w2), mk(g), mk(g1), mk(g2), mk(occ), mk(gocc), B, C, H, W, tx, ty,
                ncb);
    else if (g1_lds && par_phase)
      lp.launch(true, corr_bwd_mfma_par_lds<false>, grid1d, block, 0, st, mk(f1), mk(w2), mk(g), mk(g1), mk(g2), mk(occ), mk(gocc), B, C, H, W, tx, ty,
                ncb);
    else if (g1_lds)
      lp.launch(true, corr_bwd_mfma_par<true>, grid2, block, 0, st, mk(f1), mk(w2), mk(g), mk(g1), mk(g2), mk(occ), mk(gocc), B, C, H, W, tx, ty);
    else
      lp.launch(true, corr_bwd_mfma_par<false>, grid2, block, 0, st, mk(f1), mk(w2), mk(g), mk(g1), mk(g2), mk(occ), mk(gocc), B, C, H, W, tx, ty);
    return launch_status();
  }
  if (both) {
    const dim3 grid1((unsigned)(B * tx * ty), (unsigned)((C + 31) / 32));
    if (g1_lds)
      lp.launch(true, corr_bwd_mfma_both<true>, grid1, block, 0, st, mk(f1), mk(w2), mk(g), mk(g1), mk(g2), mk(occ), mk(gocc), B, C, H, W, tx, ty);
    else
      lp.launch(true, corr_bwd_mfma_both<false>, grid1, block, 0, st, mk(f1), mk(w2), mk(g), mk(g1), mk(g2), mk(occ), mk(gocc), B, C, H, W, tx, ty);
    return launch_status();
  }
  if (g1 && g1_lds)
    lp.launch(!g2, corr_bwd_mfma<false, true>, grid, block, 0, st, mk(w2), mk(g), mk(g1), mk(occ), mk((const ocf_tensor4*)nullptr),
              mk((const ocf_tensor4*)nullptr), B, C, H, W, tx, ty, ablate);
  else if (g1)
    lp.launch(!g2, corr_bwd_mfma<false, false>, grid, block, 0, st, mk(w2), mk(g), mk(g1), mk(occ), mk((const ocf_tensor4*)nullptr),
              mk((const ocf_tensor4*)nullptr), B, C, H, W, tx, ty, ablate);
  if (g2) lp.launch(true, corr_bwd_mfma<true, false>, grid, block, 0, st, mk(f1), mk(g), mk(g2), mk(occ), mk(w2), mk(gocc), B, C, H, W, tx, ty, ablate);
  return launch_status();
}

}  // namespace ocf
