// K1 forward, bf16, dense channels_last, d = 4, C <= 64: the "row" kernel (gfx950, wave64).
//
// What the previous band-GEMM kernel (corr_mfma.hip, kept for C > 64 / strided layouts / d < 4 / the in-kernel warp gather)
// spent its time on at the 8 x 32 x 109 x 256 level was not HBM (PMC traffic == algorithmic bytes) but instruction issue: tile index
// math in the persistent loop, a padded LDS band image, and a copy-out that gathered eight 2-byte LDS values per 16-byte store
// (DESIGN.md section 4).  This kernel removes those instruction streams instead of overlapping them:
//
//   product     NON-transposed band GEMM: D[m][n] = sum_c f1[y, x0+m, c] * f2[y+dy, x0+n, c] with the f1 fragment as the MFMA A
//               operand.  Lane (n = l&15, g = l>>4) then holds, in its four accumulator registers, four PIXELS m = 4g..4g+3 at one
//               f2 column n.  In the output's own memory order (pixel-major, 81 taps per pixel) tap (dy, dx = n - m) of pixel m sits
//               at element  81 m + 9 dy + (n - m) = 80 m + 9 dy + n :  the 16 lanes of a register write 16 CONSECUTIVE bf16.
//   LDS image   of a tile row is therefore kept in exactly the global layout (32 px x 81 taps = 5184 contiguous bytes); a band
//               element is ONE ds_write_b16 whose address is a per-lane constant plus an immediate (strip, register, dy are
//               unrolled), bank conflicts are at most 2-way (free for 16/32-bit LDS stores, MI355X_MICROARCH "LDS").
//   copy-out    = ds_read_b128 + global_store_dwordx4 of whole contiguous rows, no index arithmetic, no gathers.
//   f1          goes global -> MFMA fragment registers directly (each f1 value is used by exactly one wave); only the f2 halo tile is
//               staged in LDS ([k-block][row][x][4 units], unit order XOR-permuted: conflict-free ds_write_b128 / ds_read_b128).
//   schedule    one workgroup = a 32 x TH pixel tile, TH*WPR waves; wave = (tile row, 1/WPR of the nine dy).  No persistent loop:
//               54 KB of LDS and <= 168 VGPRs let three workgroups share a CU, so one workgroup's global loads, another's MFMA /
//               band phase and a third's stores overlap through the hardware scheduler instead of a hand-built pipeline.
//   occlusion   MODE 1: corr(f1, f2*occ)[dy,dx](x) = occ(x+d) * corr(f1,f2)[dy,dx](x) -- the f2 column is the lane, so the mask is one
//               LDS float per lane and (dy, column tile).
// reference semantics: models/networks/correlation_layer.py:7-40; flow_occ_net.py:167-168 (mask)
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_ext.h>
#include "ocf_common.h"
#include "corr_internal.h"
#include "corr_frag.h"

namespace ocf {
namespace rw {

constexpr int D = 4, N = 9, NK = 81, TW = 32, NUM = 4, ROWPX = TW + 2 * D;
constexpr int ROW_ELEMS = TW * NK;      // 2592 bf16 = 5184 bytes = 324 sixteen-byte vectors per tile row
constexpr int ROW_VECS = ROW_ELEMS / 8;
constexpr int OCC_W = ROWPX;            // occlusion halo pitch (floats); the third column tile may read 8 floats past a row
constexpr int TRASH_BYTES = 320;        // where the band stores of lanes without a valid tap go (2 bytes per lane + the dy immediate <= 144)
constexpr int LDS_GRANULE = 1280, LDS_PER_CU = 160 * 1024;  // measured: three 53,888-byte workgroups do NOT share a CU, three of 52,992 do

// P "passes": one staged halo tile serves P consecutive groups of TH rows (the band image and the waves are those of a TH-row tile).  At
// the C = 64 level (KB = 2) a 2-row tile stages 10 halo rows for 2 rows of output -- 51 KB through LDS per 64 pixels, and rocprof showed
// that launch skeleton-bound; two passes over a 12-row halo stage 61 KB per 128 pixels, still fit two workgroups per CU (74 KB), and the
// level's 448 tiles are resident at once instead of being walked two per workgroup.
template <int KB, int TH, int WPR, int P = 1> struct G {
  static constexpr int HH = TH * P + 2 * D;
  static constexpr int PLANE = HH * ROWPX * NUM + (KB > 1 ? 4 : 0);  // 16-byte slots per 32-channel block; with two blocks an odd
                                                                     // multiple of 64 bytes: a pixel's blocks land on different bank halves
  static constexpr int NW = TH * WPR, NT = 64 * NW;
  static constexpr int ROW_ITEMS = ROWPX * NUM * KB, SWEEPS = (ROW_ITEMS + 63) / 64;
  static constexpr int ROWS_PER_WAVE = (HH + NW - 1) / NW;
  static constexpr int DY_PER = (N + WPR - 1) / WPR;
  static constexpr size_t lds_bytes(int mode) {
    return (size_t)PLANE * KB * 16 + (size_t)TH * ROW_ELEMS * 2 + TRASH_BYTES + (mode ? (size_t)(HH * OCC_W) * 4 : 0);
  }
};

__device__ __forceinline__ int slot(int row, int x, int u) { return (row * ROWPX + x) * NUM + (((x >> 2) & 3) ^ ((3 * u) & 3)); }

}  // namespace rw

// out-of-image operands are read from here instead of being masked after the load (address select: no clamping arithmetic, and no
// load whose only use sits behind a condition -- hipcc sinks such a load into a branch and then waits for EVERY outstanding load)
__device__ u32x4 g_zero16[2];  // zero-initialised; deliberately NOT const (a constant-address-space pointer would turn the selected loads into flat_load)

// GATHER (round 3): the second operand is warp(f2, flow * fs) -- K1 + K2 in ONE launch at the two large pyramid levels.  The halo tile is not
// copied from f2 but SAMPLED: every staged 16-byte unit (8 channels of one halo pixel) is the bilinear blend of the four neighbour units
// around the pixel's sample position, loaded with address-select (invalid taps read a zero unit), blended in fp32 in ocf_warp_fwd's own
// fma order and rounded to bf16 -- bit-identical to the materialised warp.  With a smooth flow the 16 pixels x 4 units of a wave
// instruction sample 16 neighbouring source pixels, i.e. a tap load is as coalesced as the plain staging load, and the four taps of
// neighbouring pixels share cache lines (L1): the cost over the plain kernel is 4x the staging load INSTRUCTIONS and ~80 VALU operations
// per staged unit, not 4x the traffic.  The flow of the NEXT tile's halo pixels is prefetched with its f1 fragments; the tap loads are
// issued one halo row (3-5 units per lane) at a time at the top of a tile -- three workgroups per CU cover each other's waits.
template <int KB, int TH, int WPR, int MODE, int P = 1, bool GATHER = false>
__global__ __launch_bounds__(64 * TH * WPR, KB == 1 ? 3 : 2) void corr_fwd_rows(T4 f1, T4 f2, T4 occ, T4 out, int B, int C, int H, int W, int tiles_x,
                                                                   int tiles_y, int ablate, T4 flow, float fs, WarpGeom wg, float lrelu) {
#ifndef OCF_TUNING_BUILD
  ablate = 0;  // phase-skipping switches (wrong results) exist only in the tuning build tools/kbench makes for itself
#endif
  using namespace rw;
  using GG = G<KB, TH, WPR, P>;
  typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
  extern __shared__ u32x4 smem[];
  u32x4* s_f2 = smem;
  unsigned short* s_out = (unsigned short*)(smem + GG::PLANE * KB);
  unsigned short* s_trash = s_out + TH * ROW_ELEMS;
  float* s_occ = (float*)(s_trash + TRASH_BYTES / 2);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  const int ry = wave / WPR, part = wave - ry * WPR;
  const int tpi = tiles_x * tiles_y, ntiles = B * tpi;
  const int f1sh = (int)f1.sh, f1sw = (int)f1.sw, f2sh = (int)f2.sh, f2sw = (int)f2.sw;
  const char* zp = (const char*)g_zero16;
  constexpr int OCC_PER = (GG::HH * OCC_W + GG::NT - 1) / GG::NT;

  // PERSISTENT, XCD-affine walk (<= 3 workgroups per CU): blocks b and b+8 share an XCD, XCD x owns the contiguous tile range
  // [x * per_xcd, (x+1) * per_xcd) -- neighbouring tiles share halo rows in that XCD's L2.  The operands of the NEXT tile are
  // loaded into registers before the current tile is computed, so a workgroup's memory latency hides behind its own MFMA / band /
  // copy-out phase (the one-tile-per-workgroup form of this kernel spent 14 of its 21 us waiting for the first byte).
  const int per_xcd = (ntiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, slots = gridDim.x >> 3;
  int ti = blockIdx.x >> 3;

  u32x4 st[GATHER ? 1 : GG::ROWS_PER_WAVE][GATHER ? 1 : GG::SWEEPS], an[P][KB][2];
  unsigned flw[GATHER ? GG::ROWS_PER_WAVE : 1][GATHER ? GG::SWEEPS : 1][2];  // GATHER: (u, v) of the halo pixel of every staged unit
  unsigned ocr[OCC_PER];
  const int fsh = (int)flow.sh, fsw = (int)flow.sw, fsc = (int)flow.sc;
  // ---- issue every global load of tile (b, ty0, tx0): f2 halo rows (one wave per row, lanes sweep the 16-byte units in memory
  // order), this wave's f1 fragments (lane (i = l&15, kg = l>>4): channels 8kg..8kg+7 of pixel i of a strip), the occlusion halo
  auto issue = [&](int b, int ty0, int tx0) __attribute__((always_inline)) {
    const char* f1b = f1.p + (long)b * f1.sb * 2;
    const char* f2b = f2.p + (long)b * f2.sb * 2;
#pragma unroll
    for (int rr = 0; rr < GG::ROWS_PER_WAVE; rr++) {
      const int r = wave + rr * GG::NW;
      const int gy = ty0 - D + r;
      const bool rowok = r < GG::HH && gy >= 0 && gy < H;
      if (GATHER) {  // the flow at the (clamped) halo pixel; the taps themselves are loaded when the tile is staged
        const float* flb = (const float*)flow.p + (long)b * flow.sb;
        const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
#pragma unroll
        for (int k = 0; k < GG::SWEEPS; k++) {
          const int x = (lane + 64 * k) / (NUM * KB);
          const int gx = tx0 - D + x, gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
          const unsigned* fp = (const unsigned*)(flb + gyc * fsh + gxc * fsw);
          flw[rr][k][0] = fp[0];
          flw[rr][k][1] = fp[fsc];
        }
        continue;
      }
#pragma unroll
      for (int k = 0; k < GG::SWEEPS; k++) {
        const int it = lane + 64 * k;
        const int uu = it & (NUM * KB - 1), x = it / (NUM * KB);
        const int gx = tx0 - D + x, c0 = uu * 8;
        const bool ok = rowok && it < GG::ROW_ITEMS && gx >= 0 && gx < W && c0 < C && !(ablate & 8);
        st[rr][k] = *(const u32x4*)(ok ? f2b + (long)(gy * f2sh + gx * f2sw + c0) * 2 : zp);
      }
    }
#pragma unroll
    for (int ps = 0; ps < P; ps++) {
      const int gy1 = ty0 + TH * ps + ry;
#pragma unroll
      for (int kb = 0; kb < KB; kb++)
#pragma unroll
        for (int s = 0; s < 2; s++) {
          const int gx = tx0 + 16 * s + j, c0 = kb * 32 + g * 8;
          const bool ok = gy1 < H && gx < W && c0 < C && !(ablate & 8);
          an[ps][kb][s] = *(const u32x4*)(ok ? f1b + (long)(gy1 * f1sh + gx * f1sw + c0) * 2 : zp);
        }
    }
    if (MODE) {
      const char* ocb = occ.p + (long)b * occ.sb * 4;
#pragma unroll
      for (int k = 0; k < OCC_PER; k++) {
        const int e = tid + k * GG::NT;
        const int r = e / OCC_W, x = e - r * OCC_W;
        const int gy = ty0 - D + r, gx = tx0 - D + x;
        const bool ok = e < GG::HH * OCC_W && gy >= 0 && gy < H && gx >= 0 && gx < W;
        ocr[k] = *(const unsigned*)(ok ? ocb + (long)(gy * (int)occ.sh + gx * (int)occ.sw) * 4 : zp);
      }
    }
  };

  // ---- lane constants of the band write.  q = n - m0 (m0 = 4g: first pixel of the lane's registers): register r holds tap
  // dx = q - r of column tile A (columns x0..x0+15) or dx = q + 16 - r of tile B (x0+16..x0+31); a lane is an "A lane" (q in [0,11]),
  // a "B lane" (q <= -5) or idle, never both.  Store pointer per (strip, register): element 1296 s + 80 (4g + r) + n of the tile row
  // image (n = j, +16 for tile B) at this wave's first dy; registers without a valid tap store into a scratch area (no exec
  // masking, no branches in the loop: hipcc turns `if (valid) store` into a branch per store)
  const int q = j - 4 * g;
  const bool is_b = q <= -5;
  const int dy_lo = part * GG::DY_PER;
  unsigned short* dstp[2][4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int dx = (is_b ? q + 16 : q) - r;
    const bool ok = dx >= 0 && dx < N;
#pragma unroll
    for (int s = 0; s < 2; s++)
      dstp[s][r] = ok ? s_out + ry * ROW_ELEMS + 1296 * s + 320 * g + 80 * r + j + (is_b ? 16 : 0) + N * dy_lo : s_trash + lane;
  }
  const float invC = 1.f / (float)C;
  // fragment read pointers of (this wave's row, column tile t) at dy_lo; the halo row hr = ry + dy only adds an immediate
  const u32x4* fb[3];
#pragma unroll
  for (int t = 0; t < 3; t++) fb[t] = s_f2 + slot(ry + dy_lo, 16 * t + (t == 2 ? (j & 7) : j), g);  // columns 40..47 are never a valid tap
  const float* ob[3];
#pragma unroll
  for (int t = 0; t < 3; t++) ob[t] = s_occ + (ry + dy_lo) * OCC_W + 16 * t + (t == 2 ? (j & 7) : j);

  auto coords = [&](int t, int& b, int& ty0, int& tx0) __attribute__((always_inline)) {
    const int tile = xcd * per_xcd + t;
    b = tile / tpi;
    const int trem = tile - b * tpi;
    const int tyi = trem / tiles_x;
    ty0 = tyi * (TH * P); tx0 = (trem - tyi * tiles_x) * TW;
  };
  bool have = ti < per_xcd && xcd * per_xcd + ti < ntiles;
  int b = 0, ty0 = 0, tx0 = 0;
  if (have) {
    coords(ti, b, ty0, tx0);
    issue(b, ty0, tx0);
  }
  while (have) {
    // ---- commit the prefetched operands of this tile to LDS (the previous tile's readers passed the barrier at the loop's end)
    if (GATHER) {
      // sample the halo tile through the warp, one halo row of this wave at a time: all tap loads of the row in flight, then the blends
      const char* f2img = f2.p + (long)b * f2.sb * 2;
#pragma unroll
      for (int rr = 0; rr < GG::ROWS_PER_WAVE; rr++) {
        const int r = wave + rr * GG::NW;
        if (r < GG::HH) {
          const int gy = ty0 - D + r;
          const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
          u32x4 nb[GG::SWEEPS][4];
          float w4[GG::SWEEPS][4];
#pragma unroll
          for (int k = 0; k < GG::SWEEPS; k++) {
            const int it = lane + 64 * k;
            const int uu = it & (NUM * KB - 1), x = it / (NUM * KB);
            const int gx = tx0 - D + x, c0 = uu * 8;
            const int gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
            const bool ok = it < GG::ROW_ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W && c0 < C;
            const float uf = __builtin_bit_cast(float, flw[rr][k][0]) * fs, vf = __builtin_bit_cast(float, flw[rr][k][1]) * fs;
            const Bilin sp = bilin_setup(((float)gxc + uf) * wg.ax + wg.bx, ((float)gyc + vf) * wg.ay + wg.by, H, W);
#pragma unroll
            for (int n = 0; n < 4; n++) {
              const int ddx = n & 1, ddy = n >> 1;
              const bool okn = ok && ((ddx ? sp.vx1 : sp.vx0) & (ddy ? sp.vy1 : sp.vy0));
              w4[k][n] = (ddx ? sp.wx1 : 1.f - sp.wx1) * (ddy ? sp.wy1 : 1.f - sp.wy1);
              nb[k][n] = *(const u32x4*)(okn ? f2img + (long)((sp.y0 + ddy) * f2sh + (sp.x0 + ddx) * f2sw + c0) * 2 : zp);
            }
          }
#pragma unroll
          for (int k = 0; k < GG::SWEEPS; k++) {
            const int it = lane + 64 * k;
            const int uu = it & (NUM * KB - 1), x = it / (NUM * KB);
            float a8[8];
#pragma unroll
            for (int e = 0; e < 8; e++) a8[e] = 0.f;
#pragma unroll
            for (int n = 0; n < 4; n++) {
              float f8[8];
              unit_to_f<bf16>(nb[k][n], f8);
#pragma unroll
              for (int e = 0; e < 8; e++) a8[e] = __builtin_fmaf(w4[k][n], f8[e], a8[e]);
            }
            if (it < GG::ROW_ITEMS) s_f2[(uu >> 2) * GG::PLANE + slot(r, x, uu & 3)] = f_to_unit<bf16>(a8);
          }
        }
      }
    } else {
#pragma unroll
    for (int rr = 0; rr < GG::ROWS_PER_WAVE; rr++) {
      const int r = wave + rr * GG::NW;
      if (r < GG::HH) {
#pragma unroll
        for (int k = 0; k < GG::SWEEPS; k++) {
          const int it = lane + 64 * k;
          const int uu = it & (NUM * KB - 1), x = it / (NUM * KB);
          if (it < GG::ROW_ITEMS) s_f2[(uu >> 2) * GG::PLANE + slot(r, x, uu & 3)] = st[rr][k];
        }
      }
    }
    }
    if (MODE) {
#pragma unroll
      for (int k = 0; k < OCC_PER; k++) {
        const int e = tid + k * GG::NT;
        if (e < GG::HH * OCC_W) ((unsigned*)s_occ)[e] = ocr[k];
      }
    }
    bf16x8v a[P][KB][2];
#pragma unroll
    for (int ps = 0; ps < P; ps++)
#pragma unroll
      for (int kb = 0; kb < KB; kb++)
#pragma unroll
        for (int s = 0; s < 2; s++) a[ps][kb][s] = __builtin_bit_cast(bf16x8v, an[ps][kb][s]);
    const int cb = b, cty0 = ty0, ctx0 = tx0;
    __syncthreads();
    // ---- next tile of this workgroup: its loads fly under everything below
    const int nti = ti + slots;
    const bool nhave = nti < per_xcd && xcd * per_xcd + nti < ntiles;
    if (nhave) {
      coords(nti, b, ty0, tx0);
      issue(b, ty0, tx0);
    }
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int ps = 0; ps < P; ps++) {
    // ---- band GEMM + band write of this wave's row and dy share (pass ps: rows TH * ps .. TH * ps + TH - 1 of the staged tile)
    constexpr int RO = ROWPX * NUM;  // slots per halo row
    const int gy1 = cty0 + TH * ps + ry;
    const bool row_live = gy1 < H;
    if (row_live) {
#pragma unroll
      for (int i = 0; i < GG::DY_PER; i++) {
        const int dy = dy_lo + i;
        if (dy < N) {
          f32x4 acc[4];
#pragma unroll
          for (int t = 0; t < 4; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kb = 0; kb < KB; kb++) {
            const bf16x8v b0 = __builtin_bit_cast(bf16x8v, fb[0][kb * GG::PLANE + (i + TH * ps) * RO]);
            const bf16x8v b1 = __builtin_bit_cast(bf16x8v, fb[1][kb * GG::PLANE + (i + TH * ps) * RO]);
            const bf16x8v b2 = __builtin_bit_cast(bf16x8v, fb[2][kb * GG::PLANE + (i + TH * ps) * RO]);
            if (ablate & 1) { acc[0][0] += (float)b0[0]; acc[1][0] += (float)b1[0]; acc[3][0] += (float)b2[0]; continue; }
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ps][kb][0], b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ps][kb][0], b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ps][kb][1], b1, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ps][kb][1], b2, acc[3], 0, 0, 0);
          }
          float o[3] = {1.f, 1.f, 1.f};
          if (MODE) {
#pragma unroll
            for (int t = 0; t < 3; t++) o[t] = ob[t][(i + TH * ps) * OCC_W];
          }
#pragma unroll
          for (int s = 0; s < 2; s++) {
            f32x4 v = is_b ? acc[2 * s + 1] : acc[2 * s];
            if (MODE) v = v * (is_b ? o[s + 1] : o[s]);
            v = v * invC;
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = v[r] > 0.f ? v[r] : v[r] * lrelu;  // (ocf_corr_fwd's leaky_slope; 1 = none)
            if (ablate & 2) { if (v[0] + v[1] + v[2] + v[3] == 123.456f) dstp[s][0][0] = 1; continue; }
            // two packed converts; the four 2-byte stores take the low / high halves (ds_write_b16 / ds_write_b16_d16_hi); N * i is an immediate
            typedef float f32x2v __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
            const unsigned p01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v){v[0], v[1]}, bf16x2v));
            const unsigned p23 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v){v[2], v[3]}, bf16x2v));
            dstp[s][0][N * i] = (unsigned short)p01;
            dstp[s][1][N * i] = (unsigned short)(p01 >> 16);
            dstp[s][2][N * i] = (unsigned short)p23;
            dstp[s][3][N * i] = (unsigned short)(p23 >> 16);
          }
        }
      }
    }
    if (WPR > 1) __syncthreads();  // WPR == 1: a row's band image is written and read by one wave (LDS is in order per wave)

    // ---- copy-out: the tile row image IS the global byte order; wave (row, part) stores its share of the row's 16-byte vectors
    if (row_live && !(ablate & 4)) {
      const int vcols = min(TW, W - ctx0);
      char* gp = out.p + ((long)cb * out.sb + (long)gy1 * out.sh + (long)ctx0 * out.sw) * 2;
      const u32x4* lp = (const u32x4*)(s_out + ry * ROW_ELEMS);
      constexpr int VPW = (ROW_VECS + WPR - 1) / WPR;
      const int v_lo = part * VPW;
      if (vcols == TW) {  // full tile row (wave-uniform): compile-time trip count, only the last sweep is partial
#pragma unroll
        for (int k = 0; k < (VPW + 63) / 64; k++) {
          const int v = v_lo + lane + 64 * k;
          if (64 * (k + 1) <= VPW && v_lo + VPW <= ROW_VECS) ((u32x4*)gp)[v] = lp[v];
          else if (lane + 64 * k < VPW && v < ROW_VECS) ((u32x4*)gp)[v] = lp[v];
        }
      } else {
        const int row_elems = vcols * NK, nvec = row_elems >> 3;
        const int v_hi = min(nvec, v_lo + VPW);
        for (int v = v_lo + lane; v < v_hi; v += 64) ((u32x4*)gp)[v] = lp[v];
        if (part == WPR - 1)
          for (int e = nvec * 8 + lane; e < row_elems; e += 64) ((unsigned short*)gp)[e] = s_out[ry * ROW_ELEMS + e];
      }
    }
    if (P > 1 && WPR > 1 && ps + 1 < P) __syncthreads();  // the row image is rewritten by the next pass: every wave has copied its share out
    }  // passes
    ti = nti; have = nhave;
    if (have) __syncthreads();  // every fragment / occlusion read of this tile is done before the next commit rewrites the LDS images
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// "Direct" variant for the small pyramid levels (C > 64: 3..7 channel blocks, a few hundred tile rows in the whole batch).  Those
// launches are one latency chain on a fraction of the chip, so the chain is made as short as it can be: one workgroup = one 32 (or
// 16) pixel tile of ONE image row, nine waves = the nine dy; every wave loads its f1 AND f2 MFMA fragments for ALL channel blocks
// straight from global memory in one burst (no LDS staging, no per-block barrier: the band kernel paid fetch -> LDS -> barrier ->
// MFMA once per 32 channels), runs its 2*KB*STRIPS MFMAs and writes its dy slice of the row image; one barrier, vector copy-out.
// f2 rows are re-read by the nine dy waves of nine neighbouring rows -- from L2, and the whole level is a few MB.
// VEC = 16: 16-byte aligned pixels (C % 8 == 0); VEC = 8: 8-byte aligned (C = 196: 392-byte pixels, the last unit is a half unit).
// ---------------------------------------------------------------------------------------------------------------------------------
// (waves per SIMD: a nine-wave workgroup puts three waves on one SIMD)
template <int KB, int STRIPS, int VEC, int MODE>
__global__ __launch_bounds__(64 * rw::N) void corr_fwd_direct(T4 f1, T4 f2, T4 occ, T4 out, int B, int C, int H, int W, int tiles_x, float lrelu) {
  using namespace rw;
  typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
  constexpr int TWD = 16 * STRIPS, ROW_E = TWD * NK;
  __shared__ __attribute__((aligned(16))) unsigned short s_out[ROW_E + TRASH_BYTES / 2];
  unsigned short* s_trash = s_out + ROW_E;

  const int tid = threadIdx.x;
  const int dy = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int per_b = H * tiles_x;
  const int b = tile / per_b, rem = tile - b * per_b;
  const int y = rem / tiles_x, tx0 = (rem - y * tiles_x) * TWD;
  const char* f1b = f1.p + ((long)b * f1.sb + (long)y * f1.sh) * 2;
  const int gy2 = y + dy - D;
  const bool row2 = gy2 >= 0 && gy2 < H;
  const int gy2c = gy2 < 0 ? 0 : (gy2 >= H ? H - 1 : gy2);
  const char* f2b = f2.p + ((long)b * f2.sb + (long)gy2c * f2.sh) * 2;
  const int f1sw = (int)f1.sw, f2sw = (int)f2.sw;

  // ---- every fragment of the wave in ONE burst of loads (raw, clamped addresses), masks applied behind a scheduling barrier: one
  // exposed memory latency per wave (left to itself hipcc interleaves load / wait / select and pays it once per fragment)
  u32x4 ar[KB][STRIPS], br[KB][STRIPS + 1];
  unsigned orw[STRIPS + 1];
#pragma unroll
  for (int kb = 0; kb < KB; kb++) {
    const int c0 = kb * 32 + g * 8, cc = c0 < C ? c0 : 0;
#pragma unroll
    for (int s = 0; s < STRIPS; s++) {
      const int gx = tx0 + 16 * s + j;
      ar[kb][s] = load_frag_raw<VEC>(f1b + (long)((gx < W ? gx : W - 1) * f1sw + cc) * 2, cc, C);
    }
#pragma unroll
    for (int t = 0; t <= STRIPS; t++) {
      const int gx = tx0 - D + 16 * t + j;
      br[kb][t] = load_frag_raw<VEC>(f2b + (long)((gx < 0 ? 0 : (gx >= W ? W - 1 : gx)) * f2sw + cc) * 2, cc, C);
    }
  }
#pragma unroll
  for (int t = 0; t <= STRIPS; t++) {
    orw[t] = 0x3f800000u;
    if (MODE) {
      const int gx = tx0 - D + 16 * t + j;
      orw[t] = ((const unsigned*)occ.p)[(long)b * occ.sb + (long)gy2c * occ.sh + (long)(gx < 0 ? 0 : (gx >= W ? W - 1 : gx)) * occ.sw];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  bf16x8v a[KB][STRIPS], bq[KB][STRIPS + 1];
  float o[STRIPS + 1];
#pragma unroll
  for (int kb = 0; kb < KB; kb++) {
    const int c0 = kb * 32 + g * 8;
#pragma unroll
    for (int s = 0; s < STRIPS; s++) a[kb][s] = __builtin_bit_cast(bf16x8v, mask_frag<VEC>(ar[kb][s], tx0 + 16 * s + j < W && c0 < C, c0, C));
#pragma unroll
    for (int t = 0; t <= STRIPS; t++) {
      const int col = 16 * t + j, gx = tx0 - D + col;
      bq[kb][t] = __builtin_bit_cast(bf16x8v, mask_frag<VEC>(br[kb][t], row2 && col < TWD + 2 * D && gx >= 0 && gx < W && c0 < C, c0, C));
    }
  }
#pragma unroll
  for (int t = 0; t <= STRIPS; t++) {
    const int gx = tx0 - D + 16 * t + j;
    o[t] = __builtin_bit_cast(float, orw[t] & ((!MODE || (row2 && gx >= 0 && gx < W)) ? 0xffffffffu : 0u));
  }
  f32x4 acc[2 * STRIPS];
#pragma unroll
  for (int t = 0; t < 2 * STRIPS; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < KB; kb++)
#pragma unroll
    for (int s = 0; s < STRIPS; s++) {
      acc[2 * s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][s], bq[kb][s], acc[2 * s], 0, 0, 0);
      acc[2 * s + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][s], bq[kb][s + 1], acc[2 * s + 1], 0, 0, 0);
    }
  // ---- band write (see corr_fwd_rows): element 1296 s + 80 (4g + r) + 9 dy + n of the row image, n = j (+16 for column tile B)
  const int q = j - 4 * g;
  const bool is_b = q <= -5;
  const float invC = 1.f / (float)C;
#pragma unroll
  for (int s = 0; s < STRIPS; s++) {
    f32x4 v = is_b ? acc[2 * s + 1] : acc[2 * s];
    if (MODE) v = v * (is_b ? o[s + 1] : o[s]);
    v = v * invC;
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = v[r] > 0.f ? v[r] : v[r] * lrelu;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int dx = (is_b ? q + 16 : q) - r;
      unsigned short* dst = (dx >= 0 && dx < N) ? s_out + 1296 * s + 320 * g + 80 * r + N * dy + j + (is_b ? 16 : 0) : s_trash + lane;
      *dst = __builtin_bit_cast(unsigned short, (bf16)v[r]);
    }
  }
  __syncthreads();
  const int vcols = min(TWD, W - tx0);
  const int row_elems = vcols * NK, nvec = row_elems >> 3;
  char* gp = out.p + ((long)b * out.sb + (long)y * out.sh + (long)tx0 * out.sw) * 2;
  for (int v = tid; v < nvec; v += 64 * N) ((u32x4*)gp)[v] = ((const u32x4*)s_out)[v];
  for (int e = nvec * 8 + tid; e < row_elems; e += 64 * N) ((unsigned short*)gp)[e] = s_out[e];
}

template <int KB, int STRIPS, int VEC, int MODE>
static int launch_direct(const T4& f1, const T4& f2, const T4& occ, const T4& out, int B, int C, int H, int W, hipStream_t st) {
  const int tx = (W + 16 * STRIPS - 1) / (16 * STRIPS);
  const long ntiles = (long)B * H * tx;
  if (ntiles > 0x7fffffffL) return OCF_EUNSUPPORTED;
  const dim3 grid((unsigned)ntiles), block(64 * rw::N);
  ProfileSlot& ps = profile_slot();
  if (ps.start && ps.stop) {
    hipEvent_t e0 = ps.start, e1 = ps.stop;
    ps.start = nullptr; ps.stop = nullptr;
    hipExtLaunchKernelGGL((corr_fwd_direct<KB, STRIPS, VEC, MODE>), grid, block, 0, st, e0, e1, 0, f1, f2, occ, out, B, C, H, W, tx, corr_fwd_lrelu());
  } else {
    hipLaunchKernelGGL((corr_fwd_direct<KB, STRIPS, VEC, MODE>), grid, block, 0, st, f1, f2, occ, out, B, C, H, W, tx, corr_fwd_lrelu());
  }
  return launch_status();
}

template <int VEC, int MODE>
static int launch_direct_kb(const T4& f1, const T4& f2, const T4& occ, const T4& out, int B, int C, int H, int W, hipStream_t st) {
  const int kb = (C + 31) / 32;
  if (W <= 16) {
    switch (kb) {
      case 3: return launch_direct<3, 1, VEC, MODE>(f1, f2, occ, out, B, C, H, W, st);
      case 4: return launch_direct<4, 1, VEC, MODE>(f1, f2, occ, out, B, C, H, W, st);
      case 5: return launch_direct<5, 1, VEC, MODE>(f1, f2, occ, out, B, C, H, W, st);
      case 6: return launch_direct<6, 1, VEC, MODE>(f1, f2, occ, out, B, C, H, W, st);
      case 7: return launch_direct<7, 1, VEC, MODE>(f1, f2, occ, out, B, C, H, W, st);
      default: return OCF_EUNSUPPORTED;
    }
  }
  switch (kb) {
    case 3: return launch_direct<3, 2, VEC, MODE>(f1, f2, occ, out, B, C, H, W, st);
    case 4: return launch_direct<4, 2, VEC, MODE>(f1, f2, occ, out, B, C, H, W, st);
    default: return OCF_EUNSUPPORTED;
  }
}

static int rows_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    n = (n / 8) * 8;
    if (n < 8) n = 8;
  }
  return n;
}

template <int KB, int TH, int WPR, int MODE, int P = 1, bool GATHER = false>
static int launch_rows(const T4& f1, const T4& f2, const T4& occ, const T4& out, int B, int C, int H, int W, hipStream_t st, const T4& flow = T4{nullptr, 0, 0, 0, 0},
                       float fs = 1.f, int ac = 0) {
  using GG = rw::G<KB, TH, WPR, P>;
  const int tx = (W + rw::TW - 1) / rw::TW, ty = (H + TH * P - 1) / (TH * P);
  const long ntiles = (long)B * tx * ty;
  if (ntiles > 0x7fffffffL) return OCF_EUNSUPPORTED;
  size_t lds = GG::lds_bytes(MODE);
#ifdef OCF_TUNING_BUILD
  if (getenv("OCF_LDS_PAD")) lds += atoi(getenv("OCF_LDS_PAD"));  // residency experiments
#endif
  static int wg_per_cu = 0;
  if (wg_per_cu == 0) {
    int nb = 0;
    (void)hipFuncSetAttribute((const void*)corr_fwd_rows<KB, TH, WPR, MODE, P, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, corr_fwd_rows<KB, TH, WPR, MODE, P, GATHER>, GG::NT, lds);
    wg_per_cu = (e == hipSuccess && nb >= 1) ? (nb > (KB == 1 ? 3 : 2) ? (KB == 1 ? 3 : 2) : nb) : 1;
    // the occupancy API is advisory: LDS is handed out in 1280-byte granules (a persistent grid sized for a workgroup that is not
    // resident would run that workgroup's tiles after everyone else has finished)
    const int by_lds = (int)(rw::LDS_PER_CU / (((lds + rw::LDS_GRANULE - 1) / rw::LDS_GRANULE) * rw::LDS_GRANULE));
    if (by_lds >= 1 && wg_per_cu > by_lds) wg_per_cu = by_lds;
  }
  // persistent grid: a multiple of 8 (XCD-affine walk), at most what is co-resident
  long gsz = ((ntiles + 7) / 8) * 8;
  long cap = (long)wg_per_cu * rows_num_cus();
#ifdef OCF_TUNING_BUILD
  if (getenv("OCF_ROWS_CAP")) cap = atol(getenv("OCF_ROWS_CAP"));  // grid-balance experiments
#endif
  if (gsz > cap) gsz = cap;
  const dim3 grid((unsigned)gsz), block(GG::NT);
  ProfileSlot& ps = profile_slot();
  if (ps.start && ps.stop) {  // one-shot: the events receive this dispatch's own begin / end timestamps
    hipEvent_t e0 = ps.start, e1 = ps.stop;
    ps.start = nullptr; ps.stop = nullptr;
    hipExtLaunchKernelGGL((corr_fwd_rows<KB, TH, WPR, MODE, P, GATHER>), grid, block, lds, st, e0, e1, 0, f1, f2, occ, out, B, C, H, W, tx, ty, 0, flow, fs,
                          WarpGeom(H, W, ac), corr_fwd_lrelu());
  } else {
    int ablate = 0;
#ifdef OCF_TUNING_BUILD
    static const int ab = getenv("OCF_ABLATE") ? atoi(getenv("OCF_ABLATE")) : 0;
    ablate = ab;
    static bool told = false;
    if (!told) {
      int nb = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, corr_fwd_rows<KB, TH, WPR, MODE, P, GATHER>, GG::NT, lds);
      fprintf(stderr, "[tuning] corr_fwd_rows<%d,%d,%d,%d,%d,%d>: %zu B LDS, %d workgroups per CU, ablate %d\n", KB, TH, WPR, MODE, P, (int)GATHER, lds, nb, ab);
      told = true;
    }
#endif
    hipLaunchKernelGGL((corr_fwd_rows<KB, TH, WPR, MODE, P, GATHER>), grid, block, lds, st, f1, f2, occ, out, B, C, H, W, tx, ty, ablate, flow, fs,
                       WarpGeom(H, W, ac), corr_fwd_lrelu());
  }
  return launch_status();
}

// 0: not for these kernels; OCF_CORR_ROWS_K1 / _K2 / OCF_CORR_DIRECT otherwise
int corr_rows_variant(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* out, int C, int H, int W,
                      int d) {
  if (d != rw::D || C % 4 != 0) return 0;
  const int vec = min_vec(vec_width<bf16>(f1), vec_width<bf16>(f2));
  // with a flow (in-kernel warp gather): the two row kernels only, 16-byte units, a flow with forward strides that fit 32-bit offsets
  if (flow && (C > 64 || vec != 16 || C % 8 != 0 || flow->sw <= 0 || flow->sh <= 0 || flow->sc <= 0 ||
               flow->sh * (long)H + flow->sc * 2 >= (1L << 31)))
    return 0;
  if (vec < 8 || (vec == 8 && C <= 64) || (vec == 16 && C % 8 != 0)) return 0;
  auto span = [&](const ocf_tensor4* t) { return (labs(t->sh) * H + labs(t->sw) * W + C) * 2L; };
  if (span(f1) >= (1L << 31) || span(f2) >= (1L << 31)) return 0;
  // dense channels_last cost volume whose tile rows start on 16-byte boundaries
  if (!(out->sc == 1 && out->sw == rw::NK && ((uintptr_t)out->data % 16 == 0) && (out->sh * 2) % 16 == 0 && (out->sb * 2) % 16 == 0))
    return 0;
  if (C <= 32) return OCF_CORR_ROWS_K1;
  if (C <= 64) return OCF_CORR_ROWS_K2;
  const int kb = (C + 31) / 32;
  if (kb <= 4 || (kb <= 7 && W <= 16)) return OCF_CORR_DIRECT;  // (register budget: 2 strips x 7 blocks of fragments do not fit)
  return 0;
}

int corr_fwd_bf16_rows(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* occ, const ocf_tensor4* out, int B, int C, int H,
                       int W, hipStream_t st, const ocf_tensor4* flow, float fs, int ac) {
  const T4 a = mk(f1), bq = mk(f2), oc = mk(occ), o = mk(out);
  if (flow) {  // K1 + K2 in one launch (C <= 64: corr_rows_variant)
    const T4 fl = mk(flow);
    if (C <= 32)
      return occ ? launch_rows<1, 4, 1, 1, 1, true>(a, bq, oc, o, B, C, H, W, st, fl, fs, ac) : launch_rows<1, 4, 1, 0, 1, true>(a, bq, oc, o, B, C, H, W, st, fl, fs, ac);
    return occ ? launch_rows<2, 2, 2, 1, 2, true>(a, bq, oc, o, B, C, H, W, st, fl, fs, ac) : launch_rows<2, 2, 2, 0, 2, true>(a, bq, oc, o, B, C, H, W, st, fl, fs, ac);
  }
  if (C > 64) {
    if (min_vec(vec_width<bf16>(f1), vec_width<bf16>(f2)) == 16)
      return occ ? launch_direct_kb<16, 1>(a, bq, oc, o, B, C, H, W, st) : launch_direct_kb<16, 0>(a, bq, oc, o, B, C, H, W, st);
    return occ ? launch_direct_kb<8, 1>(a, bq, oc, o, B, C, H, W, st) : launch_direct_kb<8, 0>(a, bq, oc, o, B, C, H, W, st);
  }
#ifdef OCF_TUNING_BUILD
  if (C <= 32) {  // tile-shape experiments for the one-block kernel
    static const int kv = getenv("OCF_ROWS_K1_SHAPE") ? atoi(getenv("OCF_ROWS_K1_SHAPE")) : 0;
    if (kv == 1) return occ ? launch_rows<1, 2, 1, 1>(a, bq, oc, o, B, C, H, W, st) : launch_rows<1, 2, 1, 0>(a, bq, oc, o, B, C, H, W, st);
    if (kv == 2) return occ ? launch_rows<1, 2, 2, 1>(a, bq, oc, o, B, C, H, W, st) : launch_rows<1, 2, 2, 0>(a, bq, oc, o, B, C, H, W, st);
    if (kv == 3) return occ ? launch_rows<1, 4, 2, 1>(a, bq, oc, o, B, C, H, W, st) : launch_rows<1, 4, 2, 0>(a, bq, oc, o, B, C, H, W, st);
    if (kv == 4) return occ ? launch_rows<1, 2, 2, 1, 2>(a, bq, oc, o, B, C, H, W, st) : launch_rows<1, 2, 2, 0, 2>(a, bq, oc, o, B, C, H, W, st);
  }
#endif
  if (C <= 32) return occ ? launch_rows<1, 4, 1, 1>(a, bq, oc, o, B, C, H, W, st) : launch_rows<1, 4, 1, 0>(a, bq, oc, o, B, C, H, W, st);
#ifdef OCF_TUNING_BUILD
  static const int p1 = getenv("OCF_ROWS_K2_PASSES") ? atoi(getenv("OCF_ROWS_K2_PASSES")) : 2;
  if (p1 == 1) return occ ? launch_rows<2, 2, 2, 1>(a, bq, oc, o, B, C, H, W, st) : launch_rows<2, 2, 2, 0>(a, bq, oc, o, B, C, H, W, st);
#endif
  // two passes of two rows over one 12-row halo tile (see struct G)
  return occ ? launch_rows<2, 2, 2, 1, 2>(a, bq, oc, o, B, C, H, W, st) : launch_rows<2, 2, 2, 0, 2>(a, bq, oc, o, B, C, H, W, st);
}

}  // namespace ocf
