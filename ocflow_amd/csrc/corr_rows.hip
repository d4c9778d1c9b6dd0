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
#include <hip/hip_ext.h>
#include "ocf_common.h"
#include "corr_internal.h"

namespace ocf {
namespace rw {

constexpr int D = 4, N = 9, NK = 81, TW = 32, NUM = 4, ROWPX = TW + 2 * D;
constexpr int ROW_ELEMS = TW * NK;      // 2592 bf16 = 5184 bytes = 324 sixteen-byte vectors per tile row
constexpr int ROW_VECS = ROW_ELEMS / 8;
constexpr int OCC_W = ROWPX;            // occlusion halo pitch (floats); the third column tile may read 8 floats past a row
constexpr int TRASH_BYTES = 512;        // where the band stores of lanes without a valid tap go (2 bytes per lane + the dy immediate)

template <int KB, int TH, int WPR> struct G {
  static constexpr int HH = TH + 2 * D;
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

template <int KB, int TH, int WPR, int MODE>
__global__ __launch_bounds__(64 * TH * WPR, 3) void corr_fwd_rows(T4 f1, T4 f2, T4 occ, T4 out, int B, int C, int H, int W, int tiles_x,
                                                                   int tiles_y) {
  using namespace rw;
  using GG = G<KB, TH, WPR>;
  typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
  extern __shared__ u32x4 smem[];
  u32x4* s_f2 = smem;
  unsigned short* s_out = (unsigned short*)(smem + GG::PLANE * KB);
  unsigned short* s_trash = s_out + TH * ROW_ELEMS;
  float* s_occ = (float*)(s_trash + TRASH_BYTES / 2);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tpi = tiles_x * tiles_y;
  const int b = tile / tpi, trem = tile - b * tpi;
  const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const char* f1b = f1.p + (long)b * f1.sb * 2;
  const char* f2b = f2.p + (long)b * f2.sb * 2;
  const int f1sh = (int)f1.sh, f1sw = (int)f1.sw, f2sh = (int)f2.sh, f2sw = (int)f2.sw;

  // ---- f2 halo tile: one wave per halo row (scalar row math), lanes sweep the row's 16-byte units in memory order; every load of
  // a thread is issued before the first LDS store; out-of-image units are clamped loads whose result is replaced by zero
  u32x4 st[GG::ROWS_PER_WAVE][GG::SWEEPS];
#pragma unroll
  for (int rr = 0; rr < GG::ROWS_PER_WAVE; rr++) {
    const int r = wave + rr * GG::NW;
    const int gy = ty0 - D + r;
    const bool rowok = r < GG::HH && gy >= 0 && gy < H;
    const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
#pragma unroll
    for (int k = 0; k < GG::SWEEPS; k++) {
      const int it = lane + 64 * k;
      const int uu = it & (NUM * KB - 1), x = it / (NUM * KB);
      const int gx = tx0 - D + x, c0 = uu * 8;
      const bool ok = rowok && it < GG::ROW_ITEMS && gx >= 0 && gx < W && c0 < C;
      const int gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx), cc = c0 < C ? c0 : 0;
      const u32x4 v = *(const u32x4*)(f2b + (long)(gyc * f2sh + gxc * f2sw + cc) * 2);
      st[rr][k] = ok ? v : zero4;
    }
  }
  // ---- f1 fragments of this wave's row straight from global memory: lane (i = l&15, kg = l>>4) holds channels 8kg..8kg+7 of
  // pixel i of the strip (A operand: row = f1 pixel)
  const int j = lane & 15, g = lane >> 4;
  const int ry = wave / WPR, part = wave - ry * WPR;
  const int gy1 = ty0 + ry;
  const bool row_live = gy1 < H;
  bf16x8v a[KB][2];
  {
    const int gyc = gy1 < H ? gy1 : H - 1;
#pragma unroll
    for (int kb = 0; kb < KB; kb++)
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int gx = tx0 + 16 * s + j, c0 = kb * 32 + g * 8;
        const bool ok = row_live && gx < W && c0 < C;
        const int gxc = gx < W ? gx : W - 1, cc = c0 < C ? c0 : 0;
        const u32x4 v = *(const u32x4*)(f1b + (long)(gyc * f1sh + gxc * f1sw + cc) * 2);
        a[kb][s] = __builtin_bit_cast(bf16x8v, ok ? v : zero4);
      }
  }
  if (MODE) {
    const float* ocb = (const float*)occ.p + (long)b * occ.sb;
    for (int e = tid; e < GG::HH * OCC_W; e += GG::NT) {
      const int r = e / OCC_W, x = e - r * OCC_W;
      const int gy = ty0 - D + r, gx = tx0 - D + x;
      const bool ok = r < GG::HH && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
      const float o = ocb[gyc * (int)occ.sh + gxc * (int)occ.sw];
      s_occ[e] = ok ? o : 0.f;
    }
  }
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
  __syncthreads();

  // ---- band GEMM + band write.  q = n - m0 (m0 = 4g: first pixel of the lane's registers): register r holds tap dx = q - r of
  // column tile A (columns x0..x0+15) or dx = q + 16 - r of tile B (x0+16..x0+31); a lane is an "A lane" (q in [0,11]), a "B lane"
  // (q <= -5) or idle, never both.
  const int q = j - 4 * g;
  const bool is_b = q <= -5;
  const int dy_lo = part * GG::DY_PER;
  // store pointer per (strip, register): element 1296 s + 80 (4g + r) + n of the tile row image (n = j, +16 for tile B) at this
  // wave's first dy; registers without a valid tap store into a scratch area instead (no exec masking, no branches in the loop:
  // hipcc turns `if (valid) store` into a branch per store)
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
  if (row_live) {
#pragma unroll
    for (int i = 0; i < GG::DY_PER; i++) {
      const int dy = dy_lo + i;
      if (dy < N) {
        const int hr = ry + dy;
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
          const u32x4* pl = s_f2 + kb * GG::PLANE;
          const bf16x8v b0 = __builtin_bit_cast(bf16x8v, pl[slot(hr, j, g)]);
          const bf16x8v b1 = __builtin_bit_cast(bf16x8v, pl[slot(hr, 16 + j, g)]);
          const bf16x8v b2 = __builtin_bit_cast(bf16x8v, pl[slot(hr, 32 + (j & 7), g)]);  // columns 40..47 are never a valid tap
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][0], b0, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][0], b1, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][1], b1, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][1], b2, acc[3], 0, 0, 0);
        }
        float o[3] = {1.f, 1.f, 1.f};
        if (MODE) {
#pragma unroll
          for (int t = 0; t < 3; t++) o[t] = s_occ[hr * OCC_W + 16 * t + (t == 2 ? (j & 7) : j)];
        }
#pragma unroll
        for (int s = 0; s < 2; s++) {
          f32x4 v = is_b ? acc[2 * s + 1] : acc[2 * s];
          if (MODE) v = v * (is_b ? o[s + 1] : o[s]);
          v = v * invC;
#pragma unroll
          for (int r = 0; r < 4; r++) dstp[s][r][N * i] = __builtin_bit_cast(unsigned short, (bf16)v[r]);  // N * i: an immediate
        }
      }
    }
  }
  if (WPR > 1) __syncthreads();  // WPR == 1: a row's band image is written and read by one wave (LDS is in order per wave)

  // ---- copy-out: the tile row image IS the global byte order; wave (row, part) stores its share of the row's 16-byte vectors
  if (row_live) {
    const int vcols = min(TW, W - tx0);
    const int row_elems = vcols * NK, nvec = row_elems >> 3;
    char* gp = out.p + ((long)b * out.sb + (long)gy1 * out.sh + (long)tx0 * out.sw) * 2;
    const u32x4* lp = (const u32x4*)(s_out + ry * ROW_ELEMS);
    constexpr int VPW = (ROW_VECS + WPR - 1) / WPR;
    const int v_lo = part * VPW, v_hi = min(nvec, v_lo + VPW);
#pragma unroll
    for (int k = 0; k < (VPW + 63) / 64; k++) {
      const int v = v_lo + lane + 64 * k;
      if (v < v_hi) ((u32x4*)gp)[v] = lp[v];
    }
    if (part == WPR - 1)
      for (int e = nvec * 8 + lane; e < row_elems; e += 64) ((unsigned short*)gp)[e] = s_out[ry * ROW_ELEMS + e];
  }
}

template <int KB, int TH, int WPR, int MODE>
static int launch_rows(const T4& f1, const T4& f2, const T4& occ, const T4& out, int B, int C, int H, int W, hipStream_t st) {
  using GG = rw::G<KB, TH, WPR>;
  const int tx = (W + rw::TW - 1) / rw::TW, ty = (H + TH - 1) / TH;
  const long ntiles = (long)B * tx * ty;
  if (ntiles > 0x7fffffffL) return OCF_EUNSUPPORTED;
  const size_t lds = GG::lds_bytes(MODE);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)corr_fwd_rows<KB, TH, WPR, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const dim3 grid((unsigned)ntiles), block(GG::NT);
  ProfileSlot& ps = profile_slot();
  if (ps.start && ps.stop) {  // one-shot: the events receive this dispatch's own begin / end timestamps
    hipEvent_t e0 = ps.start, e1 = ps.stop;
    ps.start = nullptr; ps.stop = nullptr;
    hipExtLaunchKernelGGL((corr_fwd_rows<KB, TH, WPR, MODE>), grid, block, lds, st, e0, e1, 0, f1, f2, occ, out, B, C, H, W, tx, ty);
  } else {
    hipLaunchKernelGGL((corr_fwd_rows<KB, TH, WPR, MODE>), grid, block, lds, st, f1, f2, occ, out, B, C, H, W, tx, ty);
  }
  return launch_status();
}

bool corr_rows_eligible(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* flow, const ocf_tensor4* out, int C, int H,
                        int W, int d) {
  if (flow || d != rw::D || C > 64 || C % 8 != 0) return false;
  if (vec_width<bf16>(f1) != 16 || vec_width<bf16>(f2) != 16) return false;
  auto span = [&](const ocf_tensor4* t) { return (labs(t->sh) * H + labs(t->sw) * W + C) * 2L; };
  if (span(f1) >= (1L << 31) || span(f2) >= (1L << 31)) return false;
  // dense channels_last cost volume whose tile rows start on 16-byte boundaries
  return out->sc == 1 && out->sw == rw::NK && ((uintptr_t)out->data % 16 == 0) && (out->sh * 2) % 16 == 0 && (out->sb * 2) % 16 == 0;
}

int corr_fwd_bf16_rows(const ocf_tensor4* f1, const ocf_tensor4* f2, const ocf_tensor4* occ, const ocf_tensor4* out, int B, int C, int H,
                       int W, hipStream_t st) {
  const T4 a = mk(f1), bq = mk(f2), oc = mk(occ), o = mk(out);
  if (C <= 32) return occ ? launch_rows<1, 4, 1, 1>(a, bq, oc, o, B, C, H, W, st) : launch_rows<1, 4, 1, 0>(a, bq, oc, o, B, C, H, W, st);
  return occ ? launch_rows<2, 2, 2, 1>(a, bq, oc, o, B, C, H, W, st) : launch_rows<2, 2, 2, 0>(a, bq, oc, o, B, C, H, W, st);
}

}  // namespace ocf
