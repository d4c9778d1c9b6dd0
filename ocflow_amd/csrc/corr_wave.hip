// K1(+K2) forward, bf16, channels_last features, d = 4: the "wave" kernel for the small pyramid levels (C > 64: 14,336 / 3,584 / 896
// pixels per level at B = 8, 436 x 1024) and for cost volumes written into a wider buffer (any output pixel pitch).
//
// Those launches are ONE latency chain on a fraction of the chip (rocprofv3, round 2: 7.4-8.0 us each for 0.85-8 MB).  The direct
// kernel (corr_rows.hip) kept a nine-wave workgroup per tile row -- 56 / 112 / 448 workgroups, i.e. 56 of 256 CUs busy at the top
// level, every CU pulling 9 x 21 KB of fragments through one texture path -- then an LDS row image, a barrier and a copy-out.
// Here the unit of work is one WAVE = (image row y, 16*S-pixel strip, displacement row dy) and nothing else is shared:
//   load      the wave's f1 and f2 MFMA fragments for ALL channel blocks straight from global memory in one burst
//             (f2 rows are re-read by the nine dy waves of nine neighbouring rows -- from L2; a whole level is a few MB);
//   contract  2 * KB * S  v_mfma_f32_16x16x32_bf16  (band GEMM: D[m][n] = sum_c f1[y, x0+m, c] * f2[y+dy, x0-4+n, c]);
//   store     lane (n = l & 15, g = l >> 4) holds pixels m = 4g .. 4g+3 at f2 column n: tap dx = n - m of pixel m.  The 16 lanes of an
//             accumulator register are 16 CONSECUTIVE bf16 of the output (element 81 m + 9 dy + n - m): one 2-byte store per register,
//             32-byte segments, no LDS, no barrier.  (At 0.15-2.3 MB of output per launch the partial-line stores cost nothing; the
//             large levels keep the row kernels and their 16-byte copy-out.)
//   grid      504 / 2016 / 4032 independent waves instead of 56 / 112 / 448 nine-wave workgroups: the whole chip takes part and a
//             wave's chain is  load -> MFMA -> store.
// OCC    : corr(f1, f2*occ)[dy,dx](x) = occ(y+dy, x+dx) * corr(f1,f2)[dy,dx](x): the f2 column is the lane -- one float per lane and tile.
// GATHER : the f2 fragment of lane (column, channel group) is the bilinear sample warp(f2, flow*fs)(y+dy, column) computed in registers
//          from the four neighbour units (address-select loads, same fma order and the same rounding to bf16 as ocf_warp_fwd, so the
//          result equals the two-launch form bit for bit): the warped tensor never exists in HBM (K1+K2 of SURVEY 2.2 in one launch).
// reference semantics: models/networks/correlation_layer.py:7-40; flow_net.py:110-112 (warp -> corr); flow_occ_net.py:167-168 (mask)
#include "ocf_common.h"
#include "corr_internal.h"
#include "corr_frag.h"

namespace ocf {
namespace wv {
constexpr int D = 4, N = 9, NK = 81, WAVES = 4;
}

__device__ u32x4 g_wave_zero16[2];  // out-of-image taps of the gather read zeros from here (address select, see corr_rows.hip)

template <int KB, int S, int VEC, bool OCC, bool GATHER>
__global__ __launch_bounds__(64 * wv::WAVES) void corr_fwd_wave(T4 f1, T4 f2, T4 flow, T4 occ, T4 out, int B, int C, int H, int W,
                                                                 int tiles_x, int nunits, float fs, WarpGeom wg, float lrelu) {
  using namespace wv;
  typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int j = lane & 15, g = lane >> 4;
  const int unit = xcd_remap(blockIdx.x, gridDim.x) * WAVES + wave;
  if (unit >= nunits) return;
  const int tile = unit / N, dy = unit - tile * N;
  const int per_b = H * tiles_x;
  const int b = tile / per_b, rem = tile - b * per_b;
  const int y = rem / tiles_x, tx0 = (rem - y * tiles_x) * (16 * S);
  const int gy2 = y + dy - D;

  // ---- where this lane's accumulator registers go: register r of strip s = pixel tx0 + 16 s + 4 g + r, tap (dy, dx),
  // dx = q - r for an "A lane" (column tile s), q + 16 - r for a "B lane" (column tile s + 1); q = j - 4 g
  const int q = j - 4 * g;
  const bool is_b = q <= -5;
  const int osw = (int)out.sw;
  char* orow = out.p + ((long)b * out.sb + (long)y * out.sh + (long)tx0 * out.sw + N * dy) * 2;
  int ooff[S][4];  // byte offset from orow, -1 = no valid tap / pixel outside the image
#pragma unroll
  for (int s = 0; s < S; s++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int dx = (is_b ? q + 16 : q) - r, px = 16 * s + 4 * g + r;
      ooff[s][r] = (dx >= 0 && dx < N && tx0 + px < W) ? (px * osw + dx) * 2 : -1;
    }
  if (gy2 < 0 || gy2 >= H) {  // (wave-uniform) the whole displacement row lies outside the image: nine zero taps per pixel
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (ooff[s][r] >= 0) *(unsigned short*)(orow + ooff[s][r]) = 0;
    return;
  }

  const char* f1b = f1.p + ((long)b * f1.sb + (long)y * f1.sh) * 2;
  const int f1sw = (int)f1.sw, f2sw = (int)f2.sw;
  u32x4 ar[KB][S], br[KB][S + 1];
  unsigned orw[S + 1];
  bf16x8v bq[KB][S + 1];

  // ---- f1 fragments: lane (i = l & 15, kg = l >> 4) = channels 8 kg .. 8 kg + 7 of pixel tx0 + 16 s + i, per 32-channel block
#pragma unroll
  for (int kb = 0; kb < KB; kb++) {
    const int c0 = kb * 32 + g * 8, cc = c0 < C ? c0 : 0;
#pragma unroll
    for (int s = 0; s < S; s++) {
      const int gx = tx0 + 16 * s + j;
      ar[kb][s] = load_frag_raw<VEC>(f1b + (long)((gx < W ? gx : W - 1) * f1sw + cc) * 2, cc, C);
    }
  }
#pragma unroll
  for (int t = 0; t <= S; t++) {
    orw[t] = 0x3f800000u;
    if (OCC) {
      const int gx = tx0 - D + 16 * t + j;
      orw[t] = ((const unsigned*)occ.p)[(long)b * occ.sb + (long)gy2 * occ.sh + (long)(gx < 0 ? 0 : (gx >= W ? W - 1 : gx)) * occ.sw];
    }
  }

  if (!GATHER) {
    const char* f2b = f2.p + ((long)b * f2.sb + (long)gy2 * f2.sh) * 2;
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      const int c0 = kb * 32 + g * 8, cc = c0 < C ? c0 : 0;
#pragma unroll
      for (int t = 0; t <= S; t++) {
        const int gx = tx0 - D + 16 * t + j;
        br[kb][t] = load_frag_raw<VEC>(f2b + (long)((gx < 0 ? 0 : (gx >= W ? W - 1 : gx)) * f2sw + cc) * 2, cc, C);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // the whole burst is issued before the first mask waits for a load
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      const int c0 = kb * 32 + g * 8;
#pragma unroll
      for (int t = 0; t <= S; t++) {
        const int col = 16 * t + j, gx = tx0 - D + col;
        bq[kb][t] = __builtin_bit_cast(bf16x8v, mask_frag<VEC>(br[kb][t], col < 16 * S + 2 * D && gx >= 0 && gx < W && c0 < C, c0, C));
      }
    }
  } else {
    // ---- the f2 operand through the warp: this lane's column of tile t is the pixel (gy2, gx); its bilinear neighbours are the same
    // for every channel block.  Invalid neighbours read a zero unit (weight kept: w * 0 adds nothing), columns outside the image or
    // beyond the band are masked after the blend.
    const char* f2img = f2.p + (long)b * f2.sb * 2;
    const float* flp = (const float*)flow.p + (long)b * flow.sb + (long)gy2 * flow.sh;
    const char* zp = (const char*)g_wave_zero16;
    const int f2sh = (int)f2.sh;
#pragma unroll
    for (int t = 0; t <= S; t++) {
      const int col = 16 * t + j, gx = tx0 - D + col;
      const int gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
      const float uu = flp[(long)gxc * flow.sw] * fs, vv = flp[(long)gxc * flow.sw + flow.sc] * fs;
      const Bilin sp = bilin_setup(((float)gxc + uu) * wg.ax + wg.bx, ((float)gy2 + vv) * wg.ay + wg.by, H, W);
      float w4[4];
      const char* src[4];
#pragma unroll
      for (int n = 0; n < 4; n++) {
        const int ddx = n & 1, ddy = n >> 1;
        const bool ok = (ddx ? sp.vx1 : sp.vx0) & (ddy ? sp.vy1 : sp.vy0);
        w4[n] = (ddx ? sp.wx1 : 1.f - sp.wx1) * (ddy ? sp.wy1 : 1.f - sp.wy1);
        src[n] = ok ? f2img + (long)((sp.y0 + ddy) * f2sh + (sp.x0 + ddx) * f2sw) * 2 : nullptr;
      }
      const bool colok = col < 16 * S + 2 * D && gx >= 0 && gx < W;
#pragma unroll
      for (int kb = 0; kb < KB; kb++) {
        const int c0 = kb * 32 + g * 8;
        const bool cok = c0 < C;
        u32x4 nb[4];
#pragma unroll
        for (int n = 0; n < 4; n++) nb[n] = *(const u32x4*)((src[n] && cok) ? src[n] + c0 * 2 : zp);
        float a8[8];
#pragma unroll
        for (int e = 0; e < 8; e++) a8[e] = 0.f;
#pragma unroll
        for (int n = 0; n < 4; n++) {
          float f8[8];
          unit_to_f<bf16>(nb[n], f8);
#pragma unroll
          for (int e = 0; e < 8; e++) a8[e] = __builtin_fmaf(w4[n], f8[e], a8[e]);
        }
        const u32x4 v = f_to_unit<bf16>(a8);  // rounded to bf16 like the materialised warp output
        const unsigned m = (colok && cok) ? 0xffffffffu : 0u;
        bq[kb][t] = __builtin_bit_cast(bf16x8v, u32x4{v[0] & m, v[1] & m, v[2] & m, v[3] & m});
      }
    }
  }

  bf16x8v a[KB][S];
#pragma unroll
  for (int kb = 0; kb < KB; kb++) {
    const int c0 = kb * 32 + g * 8;
#pragma unroll
    for (int s = 0; s < S; s++) a[kb][s] = __builtin_bit_cast(bf16x8v, mask_frag<VEC>(ar[kb][s], tx0 + 16 * s + j < W && c0 < C, c0, C));
  }
  float o[S + 1];
#pragma unroll
  for (int t = 0; t <= S; t++) {
    const int gx = tx0 - D + 16 * t + j;
    o[t] = __builtin_bit_cast(float, orw[t] & ((!OCC || (gx >= 0 && gx < W)) ? 0xffffffffu : 0u));
  }

  f32x4 acc[2 * S];
#pragma unroll
  for (int t = 0; t < 2 * S; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < KB; kb++)
#pragma unroll
    for (int s = 0; s < S; s++) {
      acc[2 * s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][s], bq[kb][s], acc[2 * s], 0, 0, 0);
      acc[2 * s + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][s], bq[kb][s + 1], acc[2 * s + 1], 0, 0, 0);
    }

  const float invC = 1.f / (float)C;
#pragma unroll
  for (int s = 0; s < S; s++) {
    f32x4 v = is_b ? acc[2 * s + 1] : acc[2 * s];
    if (OCC) v = v * (is_b ? o[s + 1] : o[s]);
    v = v * invC;
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = v[r] > 0.f ? v[r] : v[r] * lrelu;
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (ooff[s][r] >= 0) *(unsigned short*)(orow + ooff[s][r]) = __builtin_bit_cast(unsigned short, (bf16)v[r]);
  }
}

template <int KB, int S, int VEC, bool OCC, bool GATHER>
static int launch_wave(const T4& f1, const T4& f2, const T4& flow, const T4& occ, const T4& out, int B, int C, int H, int W, float fs, int ac,
                       hipStream_t st) {
  const int tx = (W + 16 * S - 1) / (16 * S);
  const long nunits = (long)B * H * tx * wv::N;
  if (nunits > 0x7fffffffL) return OCF_EUNSUPPORTED;
  const dim3 grid((unsigned)((nunits + wv::WAVES - 1) / wv::WAVES)), block(64 * wv::WAVES);
  LaunchProf lp;
  lp.launch(true, corr_fwd_wave<KB, S, VEC, OCC, GATHER>, grid, block, 0, st, f1, f2, flow, occ, out, B, C, H, W, tx, (int)nunits, fs,
            WarpGeom(H, W, ac), corr_fwd_lrelu());
  return launch_status();
}

template <int KB, int S, int VEC>
static int launch_wave_mode(const T4& f1, const T4& f2, const T4& flow, const T4& occ, const T4& out, int B, int C, int H, int W, float fs,
                            int ac, hipStream_t st) {
  if (flow.p) {
    if (VEC != 16) return OCF_EUNSUPPORTED;
    constexpr int V = 16;
    return occ.p ? launch_wave<KB, S, V, true, true>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, st)
                 : launch_wave<KB, S, V, false, true>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, st);
  }
  return occ.p ? launch_wave<KB, S, VEC, true, false>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, st)
               : launch_wave<KB, S, VEC, false, false>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, st);
}

template <int S, int VEC>
static int launch_wave_kb(const T4& f1, const T4& f2, const T4& flow, const T4& occ, const T4& out, int B, int C, int H, int W, float fs, int ac,
                          hipStream_t st) {
  switch ((C + 31) / 32) {
    case 1: return launch_wave_mode<1, S, VEC>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, st);
    case 2: return launch_wave_mode<2, S, VEC>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, st);
    case 3: return launch_wave_mode<3, S, VEC>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, st);
    case 4: return launch_wave_mode<4, S, VEC>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, st);
    case 5: return launch_wave_mode<5, S, VEC>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, st);
    case 6: return launch_wave_mode<6, S, VEC>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, st);
    case 7: return launch_wave_mode<7, S, VEC>(f1, f2, flow, occ, out, B, C, H, W, fs, ac, st);
    default: return OCF_EUNSUPPORTED;
  }
}

// strips per wave: two wherever the map is wider than one strip (measured at B = 8, tools/kbench: 28 x 64: 6.4 vs 7.0 us, 14 x 32: 4.7 vs 5.1 us
// -- half the f1 fragment loads per output pixel and 3 instead of 4 column tiles of f2 per 32 pixels outweigh the smaller wave count)
static int wave_strips(int B, int H, int W) {
#ifdef OCF_TUNING_BUILD
  if (const char* e = getenv("OCF_WAVE_STRIPS"))
    if (atoi(e) > 0) return atoi(e) == 2 && W > 16 ? 2 : 1;
#endif
  return W > 16 ? 2 : 1;
}

// operands the wave kernel takes: bf16 (checked by the caller), d = 4, channel-contiguous features with 8-byte (16-byte when a flow is
// given) aligned pixels and C % 4 == 0 (C % 8 for 16-byte units), <= 224 channels, a channel-contiguous output of ANY pixel pitch >= 81
bool corr_wave_eligible(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ, const ocf_tensor4* out,
                        int B, int C, int H, int W, int d) {
  if (d != wv::D || C % 4 != 0 || C > 224) return false;
  int vec = min_vec(vec_width<bf16>(f1), vec_width<bf16>(f2));
  if (vec == 16 && C % 8 != 0) vec = 8;
  if (vec < 8 || (flow && vec != 16)) return false;
  if (!(out->sc == 1 && out->sw >= wv::NK && out->sh > 0 && out->sb > 0)) return false;
  if (flow && !(flow->sw > 0 && flow->sh > 0 && flow->sb > 0)) return false;
  auto span = [&](const ocf_tensor4* t, long c) { return (labs(t->sh) * H + labs(t->sw) * W + c) * 2L; };
  if (span(f1, C) >= (1L << 31) || span(f2, C) >= (1L << 31) || span(out, wv::NK) >= (1L << 31)) return false;
  return true;
}

int corr_fwd_bf16_wave(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* occ, const ocf_tensor4* out,
                       int B, int C, int H, int W, float fs, int ac, hipStream_t st) {
  int vec = min_vec(vec_width<bf16>(f1), vec_width<bf16>(f2));
  if (vec == 16 && C % 8 != 0) vec = 8;
  const T4 a = mk(f1), bq = mk(f2), fl = mk(flow), oc = mk(occ), o = mk(out);
  const int S = wave_strips(B, H, W);
  // register budget: S = 2 keeps 5 KB fragment registers (16 bytes each) per lane
  if (S == 2 && (C + 31) / 32 <= 4) return vec == 16 ? launch_wave_kb<2, 16>(a, bq, fl, oc, o, B, C, H, W, fs, ac, st)
                                                     : launch_wave_kb<2, 8>(a, bq, fl, oc, o, B, C, H, W, fs, ac, st);
  return vec == 16 ? launch_wave_kb<1, 16>(a, bq, fl, oc, o, B, C, H, W, fs, ac, st) : launch_wave_kb<1, 8>(a, bq, fl, oc, o, B, C, H, W, fs, ac, st);
}

}  // namespace ocf
