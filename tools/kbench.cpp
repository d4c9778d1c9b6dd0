// kbench: GPU-side micro-benchmark of the C-ABI kernels (no Python in the loop): back-to-back launches bracketed by
// HIP events on one stream.  Build: hipcc -O2 --offload-arch=gfx950 -I include tools/kbench.cpp -L ocflow_amd/lib
//        -locflow_hip -Wl,-rpath,$PWD/ocflow_amd/lib -o tools/kbench      Run: tools/kbench [B] [bf16|fp32] [cl|nchw]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ocflow_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static ocf_tensor4 mkt(void* p, int B, int C, int H, int W, bool cl) {
  ocf_tensor4 t; t.data = p;
  if (cl) { t.sb = (int64_t)C * H * W; t.sc = 1; t.sh = (int64_t)W * C; t.sw = C; }
  else    { t.sb = (int64_t)C * H * W; t.sc = (int64_t)H * W; t.sh = W; t.sw = 1; }
  return t;
}

template <typename F> static float time_us(F fn, hipStream_t st, int iters = 200) {
  for (int i = 0; i < 10; i++) fn();
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; i++) fn();
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 8;
  bool bf = argc > 2 ? !strcmp(argv[2], "bf16") : true;
  bool cl = argc > 3 ? !strcmp(argv[3], "cl") : true;
  int es = bf ? 2 : 4, dt = bf ? OCF_BF16 : OCF_F32;
  const int L[5][3] = {{196, 7, 16}, {128, 14, 32}, {96, 28, 64}, {64, 55, 128}, {32, 109, 256}};
  hipStream_t st; CK(hipStreamCreate(&st));
  printf("B=%d %s %s  (ablate=%s)\n", B, bf ? "bf16" : "fp32", cl ? "cl" : "nchw", getenv("OCF_ABLATE") ? getenv("OCF_ABLATE") : "0");
  double tb = 0, tt = 0;
  for (int l = 0; l < 5; l++) {
    int C = L[l][0], H = L[l][1], W = L[l][2];
    size_t nf = (size_t)B * C * H * W, no = (size_t)B * 81 * H * W, np = (size_t)B * H * W;
    void *f1, *f2, *out, *g1, *g2, *flow, *occ, *gimg, *gflow;
    CK(hipMalloc(&f1, nf * es)); CK(hipMalloc(&f2, nf * es)); CK(hipMalloc(&out, no * es)); CK(hipMalloc(&g1, nf * es)); CK(hipMalloc(&g2, nf * es));
    CK(hipMalloc(&flow, np * 8)); CK(hipMalloc(&occ, np * 4)); CK(hipMalloc(&gimg, nf * 4)); CK(hipMalloc(&gflow, np * 8));
    std::vector<unsigned short> hb(nf); std::vector<float> hf(nf);
    for (size_t i = 0; i < nf; i++) { float v = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f; hf[i] = v; unsigned u; memcpy(&u, &v, 4); hb[i] = (unsigned short)(u >> 16); }
    if (bf) { CK(hipMemcpy(f1, hb.data(), nf * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(f2, hb.data(), nf * 2, hipMemcpyHostToDevice)); }
    else    { CK(hipMemcpy(f1, hf.data(), nf * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(f2, hf.data(), nf * 4, hipMemcpyHostToDevice)); }
    std::vector<float> hfl(np * 2), ho(np);
    for (size_t i = 0; i < np * 2; i++) hfl[i] = (float)((i * 40503u) % 4001) / 1000.f - 2.f;
    for (size_t i = 0; i < np; i++) ho[i] = (float)((i * 9973u) % 1000) / 1000.f;
    CK(hipMemcpy(flow, hfl.data(), np * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(occ, ho.data(), np * 4, hipMemcpyHostToDevice));
    CK(hipMemset(out, 0, no * es));
    ocf_tensor4 t1 = mkt(f1, B, C, H, W, cl), t2 = mkt(f2, B, C, H, W, cl), to = mkt(out, B, 81, H, W, cl), tg1 = mkt(g1, B, C, H, W, cl), tg2 = mkt(g2, B, C, H, W, cl);
    ocf_tensor4 tf = mkt(flow, B, 2, H, W, false), tc = mkt(occ, B, 1, H, W, false), tgi = mkt(gimg, B, C, H, W, cl), tgf = mkt(gflow, B, 2, H, W, false);
    float a = time_us([&] { ocf_corr_fwd(&t1, &t2, nullptr, nullptr, &to, B, C, H, W, 4, 1.f, 0, dt, st); }, st);
    float b = time_us([&] { ocf_corr_fwd(&t1, &t2, nullptr, &tc, &to, B, C, H, W, 4, 1.f, 0, dt, st); }, st);
    float c = time_us([&] { ocf_corr_fwd(&t1, &t2, &tf, &tc, &to, B, C, H, W, 4, 1.25f, 0, dt, st); }, st);
    float d = time_us([&] { ocf_warp_fwd(&t2, &tf, &tg1, B, C, H, W, 1.25f, 0, dt, st); }, st);
    float e = time_us([&] { ocf_corr_bwd(&t1, &t2, &to, &tg1, &tg2, B, C, H, W, 4, dt, st); }, st, 50);
    float f = time_us([&] { ocf_warp_bwd(&t2, &tf, &tg1, &tgi, &tgf, B, C, H, W, 1.25f, 0, dt, st); }, st, 50);
    double px = (double)np, bytes_plain = px * (2.0 * C * es + 81.0 * es);
    printf("%3dx%3dx%4d | corr %7.1f us %6.0f GB/s | +occ %7.1f us %6.0f | fused %7.1f us %6.0f | warp %6.1f us %6.0f | corr_bwd %7.1f us %6.0f | warp_bwd %7.1f us\n",
           C, H, W, a, bytes_plain / a / 1e3, b, (bytes_plain + px * 4) / b / 1e3, c, (bytes_plain + px * 12) / c / 1e3, d, px * (2.0 * C * es + 8) / d / 1e3,
           e, px * (4.0 * C * es + 81.0 * es) / e / 1e3, f);
    tb += bytes_plain + px * 4; tt += b;
    for (void* p : {f1, f2, out, g1, g2, flow, occ, gimg, gflow}) CK(hipFree(p));
  }
  printf("corr+occ all levels: %.1f MB in %.1f us = %.0f GB/s (%.1f%% of 8 TB/s)\n", tb / 1e6, tt, tb / tt / 1e3, tb / tt / 1e3 / 80.0);
  return 0;
}
