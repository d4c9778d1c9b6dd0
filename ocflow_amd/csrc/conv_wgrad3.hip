// Weight (and bias) gradient of the 3x3 / stride-1 / padding == dilation convolutions of the estimator and context networks, third
// version (reference layers flow_net.py:11-37, flow_occ_net.py:11-67, context_net.py:11-17), bf16 channels_last, gfx950:
//
//   dW[co][ky][kx][ci] += sum_{b,y,x} dz[b,y,x,co] * x[b, y + (ky-1) d, x + (kx-1) d, ci]          (zero outside the image)
//   db[co]             += sum_{b,y,x} dz[b,y,x,co]
//
// straight into the caller's fp32 gradient buffer (any strides), so the framework's per-layer workspace fill + split-K kernel + bf16
// cast + bias reduction become ONE launch.
//
// GEMM view: M = co, N = ci, K = pixels, once per tap.  Both operands are stored pixel-major (a pixel's channels are contiguous) while a
// matrix-core fragment wants 8 consecutive PIXELS of one channel, so the LDS tiles keep the memory layout and the fragments are read
// with ds_read_b64_tr_b16 (the transposing LDS read of gfx950), two per v_mfma_f32_32x32x16_bf16 operand.  32x32 tiles because of the
// epilogue: one accumulator register of a 32x32 tile is 32 consecutive ci of two co rows = two 128-byte runs per float-atomic wave
// instruction, the shape that runs at the chip's full atomic rate; the first build used 16x16 tiles (four 64-byte runs per instruction)
// and spent 170 of its 207 us at 8 x 128 -> 128 x 109 x 256 in the atomics (~0.3 TB/s).
//
// Decomposition.  A workgroup owns one tap ROW ky and a (32 WM) x (64 WN) block of (co, ci) -- blockIdx enumerates (pixel chunk, ky, ci
// block, co block) -- so a stage needs ONE row segment of dz and ONE (shifted) row segment of x: no halo rows, any dilation.  Its 8 waves
// are WM x WN x WK: each wave accumulates a 32 x 64 x 3-tap tile (six 32x32 accumulator tiles) and WK = 8 / (WM WN) waves split the pixels of a
// stage between them (summed through LDS at the end), which keeps all 8 waves busy on the narrow layers.  Smaller per-workgroup tiles
// are deliberate: every workgroup ends with float atomics on its tile, and the chip adds ~1.3 TB/s of them however they are spread.
//
// Staging: LDS-DMA (global_load_lds_dwordx4), two buffers, one barrier per stage, zero page for everything outside the image / channel
// range.  LDS rows are one pixel's channels (64 / 128 / 256 bytes); the 16-byte chunks of a row are XOR-permuted by pixel bits so that
// the 4 pixel rows a half-wave's transposing read touches land on 4 different 64-byte bank ranges (swz()); the permutation is applied
// on the DMA's per-lane source address.
#include "ocf_common.h"
#include <mutex>
#include <set>
#include <utility>

namespace ocf {
namespace wg3 {
constexpr int NTHR = 512, NWAVE = 8, DMA_ITERS = 10;
__device__ u32x4 g_zero[64];

struct Args {
  const char* x;   // [B][H][W][xp] bf16
  const char* dz;  // [B][H][W][zp] bf16
  float* gw;       // fp32, element strides so / si / sky / skx
  float* gbias;    // fp32 [Co] or null
  long so, si, sky, skx, xp, zp;
  int B, H, W, Ci, Co, d;
  int nseg_x, nseg, segs_per_chunk, n_ci_blk, n_co_blk;
  int tune;  // ablation (policy / 100000): 1 = no atomics, 2 = no fragment reads / MFMAs, 4 = no DMA
  int nbuf;  // LDS stage buffers (2)
};

typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef s4v __attribute__((address_space(3))) * lds_s4;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void dma16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
#define OCF_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    OCF_VM(0) OCF_VM(1) OCF_VM(2) OCF_VM(3) OCF_VM(4) OCF_VM(5) OCF_VM(6) OCF_VM(7) OCF_VM(8) OCF_VM(9) OCF_VM(10) OCF_VM(11) OCF_VM(12)
    OCF_VM(13) OCF_VM(14) OCF_VM(15) OCF_VM(16) OCF_VM(17) OCF_VM(18) OCF_VM(19) OCF_VM(20) OCF_VM(21) OCF_VM(22) OCF_VM(23) OCF_VM(24)
    OCF_VM(25) OCF_VM(26) OCF_VM(27) OCF_VM(28) OCF_VM(29) OCF_VM(30)
#undef OCF_VM
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// chunk permutation of pixel row `pixel` in a tile with RB-byte rows: a multiple of 4 chunks, so the 64 bytes (32 channels) the two lane
// groups of a half-wave read from one pixel stay one aligned run
template <int RB>
__device__ __forceinline__ int swz(int pixel) {
  if (RB == 256) return (pixel & 3) * 4;
  if (RB == 128) return ((pixel >> 1) & 1) * 4;
  return 0;
}

__device__ __forceinline__ bf16x8 tr_pair(const char* a0, const char* a1) {
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)a0);
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)a1);
  const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int WM, int WN>
__global__ __launch_bounds__(NTHR, 2) void conv3x3_wgrad3_kernel(const Args a) {
  constexpr int WK = NWAVE / (WM * WN), SEG = WK <= 2 ? 64 : 32, KSTEPS = SEG / 16;
  constexpr int COT = 32 * WM, CIT = 64 * WN, RBZ = COT * 2, RBX = CIT * 2;
  constexpr int DZ_BYTES = SEG * RBZ, N_DZ = DZ_BYTES / 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = a.d, H = a.H, W = a.W;
  const int n_x = ((SEG + 2 * d) * RBX + 1023) >> 10;  // wave instructions of one x row segment
  const int seg_bytes = DZ_BYTES + (n_x << 10), buf_bytes = WK * seg_bytes;
  const int n_per_seg = N_DZ + n_x;

  // ---- block -> (pixel chunk, ky, ci block, co block); the tiles of one chunk read the same dz / x rows: same XCD (ids 8 apart) -----
  const int ntile = 3 * a.n_ci_blk * a.n_co_blk;
  const int bid = blockIdx.x;
  const int chunk = ((bid >> 3) / ntile) * 8 + (bid & 7), tile = (bid >> 3) % ntile;
  const int ky = tile % 3, cib = (tile / 3) % a.n_ci_blk, cob = tile / (3 * a.n_ci_blk);
  const int co0 = cob * COT, ci0 = cib * CIT;
  const int s_begin = min(a.nseg, chunk * a.segs_per_chunk), s_end = min(a.nseg, s_begin + a.segs_per_chunk);
  const int nstage = (s_end - s_begin + WK - 1) / WK;
  const int dy = (ky - 1) * d;

  const int ks = wave / (WM * WN), wrem = wave % (WM * WN), wm = wrem / WN, wn = wrem % WN;
  const int l15 = lane & 15, q4 = l15 >> 2, p4 = l15 & 3, grp = (lane >> 4) & 1, hh = lane >> 5, l31 = lane & 31;
  const char* zsrc = (const char*)g_zero + lane * 16;

  // ---- DMA bookkeeping.  The WM*WN waves that will CONSUME segment slot ks also load it: wave (ks, wrem) issues the 1-KiB pieces
  // r = wrem, wrem + WM*WN, ... of its slot, so a wave needs the (row, x0) arithmetic of ONE segment per stage.  Stage-invariant per lane:
  // the element offset of its 16 bytes inside the row segment and the pixel index (-1: never valid)
  constexpr int WPS = WM * WN;  // waves per slot
  int doff[DMA_ITERS], dpix[DMA_ITERS];
#pragma unroll
  for (int k = 0; k < DMA_ITERS; k++) {
    const int r = k * WPS + wrem;
    doff[k] = 0; dpix[k] = -1;
    if (r < n_per_seg) {
      const int o = (r < N_DZ ? r : r - N_DZ) * 1024 + lane * 16;  // byte offset inside the dz / x tile
      if (r < N_DZ) {
        const int pix = o / RBZ, ch = ((((o % RBZ) >> 4) ^ swz<RBZ>(pix)) << 3);
        doff[k] = (int)(pix * a.zp) + ch;
        dpix[k] = (co0 + ch < a.Co) ? pix : -1;
      } else {
        const int pix = o / RBX, ch = ((((o % RBX) >> 4) ^ swz<RBX>(pix)) << 3);
        doff[k] = (int)(pix * a.xp) + ch;
        dpix[k] = (pix < SEG + 2 * d && ci0 + ch < a.Ci) ? pix : -1;
      }
    }
  }
  // the segment a wave loads for stage t: scalars once per stage (stage_of), then one 1-KiB piece per call (piece)
  struct Stage {
    const char* zbase;
    const char* xbase;
    char* buf;
    int x0;
    bool seg_ok, xrow_ok;
  };
  auto stage_of = [&](int t, int slot_buf) __attribute__((always_inline)) {
    Stage st;
    st.buf = smem + slot_buf * buf_bytes + ks * seg_bytes;
    const int sg = s_begin + t * WK + ks;
    const int row = sg / a.nseg_x;  // = b * H + y
    st.x0 = (sg - row * a.nseg_x) * SEG;
    const int yy = row % H + dy;
    st.seg_ok = sg < s_end;
    st.xrow_ok = st.seg_ok && yy >= 0 && yy < H;
    st.zbase = a.dz + (((long)row * W + st.x0) * a.zp + co0) * 2;
    st.xbase = a.x + ((((long)row + dy) * W + (st.x0 - d)) * a.xp + ci0) * 2;
    return st;
  };
  auto piece = [&](const Stage& st, int k) __attribute__((always_inline)) {
    const int r = k * WPS + wrem;
    if (r < n_per_seg) {
      const char* src;
      if (r < N_DZ) {
        const bool ok = st.seg_ok && dpix[k] >= 0 && st.x0 + dpix[k] < W;
        src = ok ? st.zbase + (long)doff[k] * 2 : zsrc;
      } else {
        const int gx = st.x0 - d + dpix[k];
        const bool ok = st.xrow_ok && dpix[k] >= 0 && gx >= 0 && gx < W;
        src = ok ? st.xbase + (long)doff[k] * 2 : zsrc;
      }
      dma16(src, st.buf + r * 1024);
    }
  };

  // ---- fragment read offsets (bytes inside a segment slot) -------------------------------------------------------------------------------
  // 32x32x16 operand: lane = channel 16 grp + 4 p4 .. of the 32-channel block, pixel rows 8 hh + q4 (first read) and + 4 (second read:
  // same permutation).  A (dz): the wave's 32 co.  B (x): tap kx -> rows shifted by kx*d; the second 32-ci block nt flips address bit 6.
  int aoff;
  {
    const int pix = 8 * hh + q4, chunk_l = wm * 4 + 2 * grp + (p4 >> 1);
    aoff = pix * RBZ + ((chunk_l ^ swz<RBZ>(pix)) << 4) + (p4 & 1) * 8;
  }
  int boff[3][2];
#pragma unroll
  for (int kx = 0; kx < 3; kx++)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int pix = kx * d + 8 * hh + q4 + 4 * h, chunk_l = wn * 8 + 2 * grp + (p4 >> 1);
      boff[kx][h] = DZ_BYTES + pix * RBX + ((chunk_l ^ swz<RBX>(pix)) << 4) + (p4 & 1) * 8;
    }

  f32x16 acc[2][3], accb;
#pragma unroll
  for (int e = 0; e < 16; e++) {
    accb[e] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++) acc[nt][kx][e] = 0.f;
  }
  const bool bias_wave = a.gbias != nullptr && ky == 1 && cib == 0 && wn == 0;  // wave-uniform
  const bf16 one = (bf16)1.0f;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};

  // Two stage buffers: the DMA of stage t + 1 is issued in front of the matrix work of stage t.  Measured and NOT kept (each parity-green):
  // a ring of three / four buffers with counted waits (115 vs 103 us at 8 x 128 -> 128 x 109 x 256: the stage is not latency-bound, the
  // per-CU load path is -- ablation: skeleton 16 us + DMA 35 + MFMA 39 + atomics 22, additive), the DMA pieces dealt between the MFMAs
  // behind scheduling fences (123 us), the younger half of the workgroup issuing after its matrix phase (150 us: the phase loop cost
  // more registers and scalar spills than the overlap returned).
  if (nstage > 0 && !(a.tune & 4)) {
    const Stage st = stage_of(0, 0);
#pragma unroll
    for (int k = 0; k < DMA_ITERS; k++) piece(st, k);
  }
  for (int t = 0; t < nstage; t++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // stage t has landed for every wave; every wave is done with the other buffer
    if (t + 1 < nstage && !(a.tune & 4)) {
      const Stage st = stage_of(t + 1, (t + 1) & 1);
#pragma unroll
      for (int k = 0; k < DMA_ITERS; k++) piece(st, k);
    }
    if (s_begin + t * WK + ks < s_end && !(a.tune & 2)) {  // wave-uniform: this wave's segment of the stage exists
      const char* seg = smem + (t & 1) * buf_bytes + ks * seg_bytes;
      // software pipeline over the 16-pixel k-steps: the 14 transposing reads of step k2 + 1 are in flight under the 6-7 MFMAs of step
      // k2 (written out because the compiler otherwise re-uses one fragment register set: read, wait, MFMA, read, wait, MFMA ...)
      bf16x8 fa[2], fb[2][6];
      auto load = [&](int k2, int set) __attribute__((always_inline)) {
        const char* sz = seg + k2 * 16 * RBZ + aoff;
        const char* sx = seg + k2 * 16 * RBX;
        fa[set] = tr_pair(sz, sz + 4 * RBZ);
#pragma unroll
        for (int kx = 0; kx < 3; kx++)
#pragma unroll
          for (int nt = 0; nt < 2; nt++) fb[set][kx * 2 + nt] = tr_pair(sx + (boff[kx][0] ^ (nt << 6)), sx + (boff[kx][1] ^ (nt << 6)));
      };
      load(0, 0);
#pragma unroll
      for (int k2 = 0; k2 < KSTEPS; k2++) {
        const int fc = k2 & 1;
        if (k2 + 1 < KSTEPS) load(k2 + 1, fc ^ 1);
        if (bias_wave) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[fc], ones, accb, 0, 0, 0);
#pragma unroll
        for (int kx = 0; kx < 3; kx++)
#pragma unroll
          for (int nt = 0; nt < 2; nt++) acc[nt][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[fc], fb[fc][kx * 2 + nt], acc[nt][kx], 0, 0, 0);
      }
    }
  }

  // ---- sum the WK pixel slices of a (wm, wn) tile through LDS: halve the number of live slices per round ------------------------------
  if (WK > 1) {
#pragma unroll
    for (int stride = WK / 2; stride >= 1; stride >>= 1) {
      __syncthreads();  // (first round: every wave is done with the stage buffers; later rounds: the readers are done with the slabs)
      if (ks >= stride && ks < 2 * stride) {
        char* dst = smem + ((ks - stride) * (WM * WN) + wrem) * (28 * 1024) + lane * 16;
#pragma unroll
        for (int v = 0; v < 4; v++) {
#pragma unroll
          for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++)
              *(f32x4*)(dst + ((nt * 3 + kx) * 4 + v) * 1024) = f32x4{acc[nt][kx][4 * v], acc[nt][kx][4 * v + 1], acc[nt][kx][4 * v + 2], acc[nt][kx][4 * v + 3]};
          *(f32x4*)(dst + (24 + v) * 1024) = f32x4{accb[4 * v], accb[4 * v + 1], accb[4 * v + 2], accb[4 * v + 3]};
        }
      }
      __syncthreads();
      if (ks < stride) {
        const char* src = smem + (ks * (WM * WN) + wrem) * (28 * 1024) + lane * 16;
#pragma unroll
        for (int v = 0; v < 4; v++) {
#pragma unroll
          for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
              const f32x4 r = *(const f32x4*)(src + ((nt * 3 + kx) * 4 + v) * 1024);
#pragma unroll
              for (int e = 0; e < 4; e++) acc[nt][kx][4 * v + e] += r[e];
            }
          const f32x4 rb = *(const f32x4*)(src + (24 + v) * 1024);
#pragma unroll
          for (int e = 0; e < 4; e++) accb[4 * v + e] += rb[e];
        }
      }
    }
  }

  // ---- float atomics into the gradient: one accumulator register = 32 consecutive ci (128 bytes) of two co rows ------------------------
  if (ks == 0 && nstage > 0 && !(a.tune & 1)) {
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int co = co0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (co < a.Co) {
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
          const int ci = ci0 + wn * 64 + nt * 32 + l31;
          if (ci < a.Ci) {
#pragma unroll
            for (int kx = 0; kx < 3; kx++) atomicAdd(a.gw + (long)co * a.so + (long)ci * a.si + ky * a.sky + kx * a.skx, acc[nt][kx][e]);
          }
        }
        if (bias_wave && l31 == 0) atomicAdd(a.gbias + co, accb[e]);
      }
    }
  }
}

typedef void (*kern_t)(const Args);
static bool lds_attr_done(kern_t k) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair((const void*)k, dev);
  if (done.count(key)) return true;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
  done.insert(key);
  return true;
}

}  // namespace wg3
}  // namespace ocf

namespace ocf {
namespace wg3 {
// launch geometry of ocf_conv3x3_wgrad_bias for a request (shared by the launcher and ocf_conv3x3_wgrad_plan)
struct Plan {
  int wm, wn, wk, seg, nseg_x, nseg, n_co_blk, n_ci_blk, ntile, chunks, segs_per_chunk, n_x, n_dz, capped;
  size_t lds;
};
static int make_plan(Plan& p, int B, int Ci, int Co, int H, int W, int dilation, int policy) {
  // policy (tuning): chunks + 1000 * (ci waves: 1 / 2) + 10000 * (co waves: 1 / 2 / 4); 0 = automatic
  const int p_chunks = policy % 1000, p_wn = (policy / 1000) % 10, p_wm = (policy / 10000) % 10;
  if ((p_wn != 0 && p_wn != 1 && p_wn != 2) || (p_wm != 0 && p_wm != 1 && p_wm != 2 && p_wm != 4)) return OCF_EINVAL;
  p.wm = p_wm ? p_wm : (Co <= 32 ? 1 : (Co <= 64 ? 2 : 4));
  p.wn = p_wn ? p_wn : 1;  // (64-ci tiles: half the atomic bytes of 128-ci tiles; 103 vs 127 us at 8 x 128 -> 128 x 109 x 256)
  p.wk = 8 / (p.wm * p.wn);
  p.seg = p.wk <= 2 ? 64 : 32;
  p.nseg_x = (W + p.seg - 1) / p.seg;
  p.nseg = B * H * p.nseg_x;
  p.n_co_blk = (Co + 32 * p.wm - 1) / (32 * p.wm);
  p.n_ci_blk = (Ci + 64 * p.wn - 1) / (64 * p.wn);
  p.ntile = 3 * p.n_ci_blk * p.n_co_blk;
  // pixel chunks: about one workgroup per compute unit (every chunk adds a full copy of its tiles with float atomics); at least two
  // stages per workgroup so that the DMA of one overlaps the matrix work of the other
  const int stages_total = (p.nseg + p.wk - 1) / p.wk;
  // (512-thread workgroups with ~190 registers: ONE is resident per compute unit, so a 257th workgroup would run as a second round)
  int chunks = p_chunks > 0 ? p_chunks : (256 / p.ntile) / 8 * 8;
  if (chunks < 8) chunks = 8;
  p.capped = 0;
  if (chunks > stages_total / 2) { chunks = stages_total / 2 > 0 ? stages_total / 2 : 1; p.capped = 1; }  // small maps: fewer chunks than CUs
  int spc = ((p.nseg + chunks - 1) / chunks + p.wk - 1) / p.wk * p.wk;
  chunks = ((p.nseg + spc - 1) / spc + 7) / 8 * 8;  // a multiple of 8 for the block -> chunk map (surplus chunks own no segments)
  p.chunks = chunks;
  p.segs_per_chunk = spc;
  p.n_x = ((p.seg + 2 * dilation) * 128 * p.wn + 1023) / 1024;
  p.n_dz = p.seg * 64 * p.wm / 1024;
  if (p.n_dz + p.n_x > DMA_ITERS * p.wm * p.wn) return OCF_EUNSUPPORTED;
  const size_t buf_bytes = (size_t)p.wk * (size_t)(p.n_dz + p.n_x) * 1024;
  const size_t red = p.wk > 1 ? (size_t)(p.wk / 2) * p.wm * p.wn * 28 * 1024 : 0;
  p.lds = 2 * buf_bytes;
  if (red > p.lds) p.lds = red;
  if (p.lds > 160 * 1024) return OCF_EUNSUPPORTED;
  return OCF_OK;
}
}  // namespace wg3
}  // namespace ocf

extern "C" int ocf_conv3x3_wgrad_plan(int B, int Ci, int Co, int H, int W, int dilation, int policy, int* workgroups, int* chunks, int* tiles,
                                      int* segments_per_chunk, int* chunks_capped) {
  using namespace ocf;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || dilation < 1 || policy < 0) return OCF_EINVAL;
  if (dilation > 16) return OCF_EUNSUPPORTED;
  wg3::Plan p;
  const int st = wg3::make_plan(p, B, Ci, Co, H, W, dilation, policy);
  if (st != OCF_OK) return st;
  if (workgroups) *workgroups = p.chunks * p.ntile;
  if (chunks) *chunks = p.chunks;
  if (tiles) *tiles = p.ntile;
  if (segments_per_chunk) *segments_per_chunk = p.segs_per_chunk;
  if (chunks_capped) *chunks_capped = p.capped;
  return OCF_OK;
}

extern "C" int ocf_conv3x3_wgrad_bias(const ocf_tensor4* x, const ocf_tensor4* dz, float* gw, int64_t gw_so, int64_t gw_si, int64_t gw_sky,
                                      int64_t gw_skx, float* gbias, int B, int Ci, int Co, int H, int W, int dilation, int dtype, int policy,
                                      void* stream) {
  using namespace ocf;
  clear_stale_error();
  if (!x || !dz || !gw || !x->data || !dz->data) return OCF_EINVAL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || dilation < 1 || policy < 0) return OCF_EINVAL;
  if (dtype != OCF_BF16 || dilation > 16) return OCF_EUNSUPPORTED;
  auto pitch = [&](const ocf_tensor4* t, int C) { return W > 1 ? t->sw : (H > 1 ? t->sh : (B > 1 ? t->sb : (int64_t)C)); };
  auto cl = [&](const ocf_tensor4* t, int C) {
    const int64_t p = pitch(t, C);
    return (C == 1 || t->sc == 1) && p >= C && (H == 1 || t->sh == p * W) && (B == 1 || t->sb == p * W * H) && (uintptr_t)t->data % 16 == 0 &&
           (p * 2) % 16 == 0;
  };
  // channels are read in 16-byte chunks: a channel count that is not a multiple of 8 needs the tail chunk inside the pixel
  // (the alignment channels of a concatenation: their products land beyond Ci / Co and are never added)
  if (!cl(x, Ci) || !cl(dz, Co) || pitch(x, Ci) < (Ci + 7) / 8 * 8 || pitch(dz, Co) < (Co + 7) / 8 * 8) return OCF_EUNSUPPORTED;
  if ((long)B * H * W * (pitch(x, Ci) > pitch(dz, Co) ? pitch(x, Ci) : pitch(dz, Co)) >= (1L << 30)) return OCF_EUNSUPPORTED;
  wg3::Plan p;
  const int pst = wg3::make_plan(p, B, Ci, Co, H, W, dilation, policy % 100000);
  if (pst != OCF_OK) return pst;
  const int wm = p.wm, wn = p.wn;
  wg3::Args a;
  a.x = (const char*)x->data; a.dz = (const char*)dz->data; a.gw = gw; a.gbias = gbias;
  a.so = gw_so; a.si = gw_si; a.sky = gw_sky; a.skx = gw_skx; a.xp = pitch(x, Ci); a.zp = pitch(dz, Co);
  a.B = B; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.d = dilation; a.tune = policy / 100000;
  a.nseg_x = p.nseg_x;
  a.nseg = p.nseg;
  a.n_co_blk = p.n_co_blk;
  a.n_ci_blk = p.n_ci_blk;
  a.segs_per_chunk = p.segs_per_chunk;
  a.nbuf = 2;
  wg3::kern_t k = nullptr;
  if (wm == 4 && wn == 2) k = wg3::conv3x3_wgrad3_kernel<4, 2>;
  else if (wm == 4 && wn == 1) k = wg3::conv3x3_wgrad3_kernel<4, 1>;
  else if (wm == 2 && wn == 2) k = wg3::conv3x3_wgrad3_kernel<2, 2>;
  else if (wm == 2 && wn == 1) k = wg3::conv3x3_wgrad3_kernel<2, 1>;
  else if (wm == 1 && wn == 2) k = wg3::conv3x3_wgrad3_kernel<1, 2>;
  else k = wg3::conv3x3_wgrad3_kernel<1, 1>;
  if (!wg3::lds_attr_done(k)) return OCF_ELAUNCH;
  LaunchProf lp;
  lp.launch(true, k, dim3((unsigned)(p.chunks * p.ntile)), dim3(wg3::NTHR), p.lds, (hipStream_t)stream, a);
  return launch_status();
}
