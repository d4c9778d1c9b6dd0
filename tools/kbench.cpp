// kbench: GPU-side micro-benchmark of the C-ABI kernels (no Python in the loop): back-to-back launches bracketed by
// HIP events on one stream.  Build: hipcc -O2 --offload-arch=gfx950 -I include tools/kbench.cpp -L ocflow_amd/lib
//        -locflow_hip -Wl,-rpath,$PWD/ocflow_amd/lib -o tools/kbench      Run: tools/kbench [B] [bf16|fp32] [cl|nchw]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <cstring>
#include <algorithm>
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
    // A/B of the per-call kernel policies (same results, different kernels) + a bitwise cross-check of the outputs
    const int var_auto = ocf_corr_fwd_variant(&t1, &t2, nullptr, &tc, &to, B, C, H, W, 4, dt, OCF_CORR_POLICY_AUTO);
    const int var_nowave = ocf_corr_fwd_variant(&t1, &t2, nullptr, &tc, &to, B, C, H, W, 4, dt, OCF_CORR_POLICY_NO_WAVE);
    const int var_band = ocf_corr_fwd_variant(&t1, &t2, nullptr, &tc, &to, B, C, H, W, 4, dt, OCF_CORR_POLICY_BAND_ONLY);
    auto fwd = [&](const ocf_tensor4* occp, int policy, const ocf_tensor4* outp) {
      return time_us([&] { ocf_corr_fwd(&t1, &t2, nullptr, occp, outp, B, C, H, W, 4, 1.f, 0, 0.f, dt, policy, st); }, st);
    };
    const float a_band = fwd(nullptr, OCF_CORR_POLICY_BAND_ONLY, &to), b_band = fwd(&tc, OCF_CORR_POLICY_BAND_ONLY, &to);
    std::vector<unsigned char> ref(no * es), got(no * es);
    CK(hipStreamSynchronize(st)); CK(hipMemcpy(ref.data(), out, no * es, hipMemcpyDeviceToHost));
    const float a_nw = fwd(nullptr, OCF_CORR_POLICY_NO_WAVE, &to), b_nw = fwd(&tc, OCF_CORR_POLICY_NO_WAVE, &to);
    const float a_wv = fwd(nullptr, OCF_CORR_POLICY_WAVE, &to), b_wv = fwd(&tc, OCF_CORR_POLICY_WAVE, &to);
    const float a = fwd(nullptr, OCF_CORR_POLICY_AUTO, &to), b = fwd(&tc, OCF_CORR_POLICY_AUTO, &to);
    CK(hipStreamSynchronize(st)); CK(hipMemcpy(got.data(), out, no * es, hipMemcpyDeviceToHost));
    size_t ndiff = 0;
    for (size_t i = 0; i < no * es; i++) ndiff += ref[i] != got[i];
    printf("  corr fwd (plain / +occ us): auto[%d] %.1f / %.1f | no-wave[%d] %.1f / %.1f | wave-forced %.1f / %.1f | band[%d] %.1f / %.1f | bytes differing auto vs band (+occ): %zu\n",
           var_auto, a, b, var_nowave, a_nw, b_nw, a_wv, b_wv, var_band, a_band, b_band, ndiff);
    // cost volume written into a 128-channel buffer (the decoder input): pixel pitch 256 bytes
    void* wide; CK(hipMalloc(&wide, np * 128 * es));
    ocf_tensor4 tw = mkt(wide, B, 128, H, W, cl);
    const float b_wide = fwd(&tc, OCF_CORR_POLICY_AUTO, &tw);
    const int var_wide = ocf_corr_fwd_variant(&t1, &t2, nullptr, &tc, &tw, B, C, H, W, 4, dt, OCF_CORR_POLICY_AUTO);
    // K1+K2 in one launch (flow given) vs the warp kernel alone
    const int var_fused = ocf_corr_fwd_variant(&t1, &t2, &tf, &tc, &to, B, C, H, W, 4, dt, OCF_CORR_POLICY_AUTO);
    float c = time_us([&] { ocf_corr_fwd(&t1, &t2, &tf, &tc, &to, B, C, H, W, 4, 1.25f, 0, 0.f, dt, OCF_CORR_POLICY_AUTO, st); }, st);
    float d = time_us([&] { ocf_warp_fwd(&t2, &tf, &tg1, B, C, H, W, 1.25f, 0, dt, st); }, st);
    printf("  +occ into a 128-channel buffer: variant %d %.1f us | fused warp+corr(+occ): variant %d %.1f us vs warp %.1f + corr %.1f = %.1f us\n", var_wide,
           b_wide, var_fused, c, d, b, d + b);
    // corr backward: vector kernel vs matrix-core kernel (max abs difference of the two g1 / g2 / d occ results, relative to the max); the
    // gradient dense, and as the 81-channel slice of a 128-channel tensor (what autograd hands over inside the model)
    ocf_tensor4 tgo_w = tw; tgo_w.data = wide;
    float e = 0, eg1 = 0, eg2 = 0, e2 = 0;
    {
      ocf_tensor4 tgo1 = mkt(gflow, B, 1, H, W, false);
      const size_t ng = np * C;
      std::vector<unsigned short> r1(ng), r2(ng), m1(ng), m2(ng);
      std::vector<float> ro(np), mo(np);
      auto bwd = [&](const ocf_tensor4* gop, const ocf_tensor4* pg1, const ocf_tensor4* pg2, int policy) {
        return time_us([&] { ocf_corr_bwd(&t1, &t2, gop, pg1, pg2, B, C, H, W, 4, dt, policy, st); }, st, 50);
      };
      auto bwdo = [&](const ocf_tensor4* gop, int policy) {
        return time_us([&] { ocf_corr_occ_bwd(&t1, &t2, &tc, gop, &tg1, &tg2, &tgo1, B, C, H, W, 4, dt, policy, st); }, st, 50);
      };
      const float v1 = bwd(&to, &tg1, nullptr, OCF_CORR_BWD_POLICY_VECTOR), v2 = bwd(&to, nullptr, &tg2, OCF_CORR_BWD_POLICY_VECTOR);
      const float vo = bwdo(&to, OCF_CORR_BWD_POLICY_VECTOR);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(r1.data(), g1, ng * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), g2, ng * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ro.data(), gflow, np * 4, hipMemcpyDeviceToHost));
      eg1 = bwd(&to, &tg1, nullptr, OCF_CORR_BWD_POLICY_AUTO); eg2 = bwd(&to, nullptr, &tg2, OCF_CORR_BWD_POLICY_AUTO);
      e = bwd(&to, &tg1, &tg2, OCF_CORR_BWD_POLICY_AUTO);
      e2 = bwdo(&to, OCF_CORR_BWD_POLICY_AUTO);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(m1.data(), g1, ng * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(m2.data(), g2, ng * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(mo.data(), gflow, np * 4, hipMemcpyDeviceToHost));
      const float p1 = bwd(&tgo_w, &tg1, nullptr, OCF_CORR_BWD_POLICY_AUTO), p2 = bwd(&tgo_w, nullptr, &tg2, OCF_CORR_BWD_POLICY_AUTO);
      const float po = bwdo(&tgo_w, OCF_CORR_BWD_POLICY_AUTO);
      auto bf = [](unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; };
      double d1 = 0, d2 = 0, dm = 0, mx1 = 0, mx2 = 0, mxo = 0;
      for (size_t i = 0; i < ng; i++) {
        d1 = std::max(d1, (double)fabsf(bf(r1[i]) - bf(m1[i]))); mx1 = std::max(mx1, (double)fabsf(bf(r1[i])));
        d2 = std::max(d2, (double)fabsf(bf(r2[i]) - bf(m2[i]))); mx2 = std::max(mx2, (double)fabsf(bf(r2[i])));
      }
      for (size_t i = 0; i < np; i++) { dm = std::max(dm, (double)fabsf(ro[i] - mo[i])); mxo = std::max(mxo, (double)fabsf(ro[i])); }
      printf("  corr_bwd[variant %d]: vector g1 %.1f g2 %.1f +occ(both) %.1f us | auto g1 %.1f g2 %.1f both %.1f +occ(both) %.1f us | grad_out at a 256-byte pitch: g1 %.1f g2 %.1f +occ(both) %.1f us | max diff / max: g1(occ) %.2e g2(occ) %.2e docc %.2e\n",
             ocf_corr_bwd_variant(&t1, &t2, &to, &tg1, &tg2, B, C, H, W, 4, dt, 0), v1, v2, vo, eg1, eg2, e, e2, p1, p2, po, d1 / (mx1 + 1e-30),
             d2 / (mx2 + 1e-30), dm / (mxo + 1e-30));
    }
    CK(hipFree(wide));
    // warp backward: scatter / window gather / binned gather, on the small-motion flow and on a flow as an untrained model produces it
    // (a quarter of the sources displaced by 8-40 pixels)
    int64_t wsb = 0; ocf_warp_bwd_workspace_bytes(B, C, H, W, &wsb);
    void* wws; CK(hipMalloc(&wws, (size_t)wsb));
    void* flow_big; CK(hipMalloc(&flow_big, np * 8));
    {
      std::vector<float> hb2(np * 2);
      for (size_t i = 0; i < np * 2; i++) {
        const unsigned h = (unsigned)(i * 2654435761u) >> 8;
        const float small = (float)(h % 4001) / 1000.f - 2.f;
        hb2[i] = (h % 4 == 0) ? small * 10.f : small;
      }
      CK(hipMemcpy(flow_big, hb2.data(), np * 8, hipMemcpyHostToDevice));
    }
    ocf_tensor4 tfb = mkt(flow_big, B, 2, H, W, false);
    auto wb = [&](const ocf_tensor4* fl, int policy, void* ws) {
      return time_us([&] { ocf_warp_bwd(&t2, fl, &tg1, &tgi, &tgf, B, C, H, W, 1.25f, 0, dt, policy, ws, st); }, st, 50);
    };
    const float f_scatter = wb(&tf, OCF_WARP_BWD_POLICY_SCATTER, nullptr), f_win = wb(&tf, OCF_WARP_BWD_POLICY_GATHER, nullptr);
    const float f = wb(&tf, OCF_WARP_BWD_POLICY_AUTO, wws);
    const float fb_scatter = wb(&tfb, OCF_WARP_BWD_POLICY_SCATTER, nullptr), fb_win = wb(&tfb, OCF_WARP_BWD_POLICY_GATHER, nullptr);
    const float fb = wb(&tfb, OCF_WARP_BWD_POLICY_AUTO, wws);
    printf("  warp_bwd small motion: auto (binned where eligible) %.1f us | window gather + far scatter %.1f | scatter only %.1f || 25%% large motion: auto %.1f | window %.1f | scatter %.1f\n",
           f, f_win, f_scatter, fb, fb_win, fb_scatter);
    CK(hipFree(wws)); CK(hipFree(flow_big));
    double px = (double)np, bytes_plain = px * (2.0 * C * es + 81.0 * es);
    printf("%3dx%3dx%4d | corr %7.1f us %6.0f GB/s | +occ %7.1f us %6.0f | fused %7.1f us %6.0f | warp %6.1f us %6.0f | corr_bwd %7.1f us %6.0f (g1 %5.1f g2 %5.1f) | +occ %7.1f us | warp_bwd %7.1f us %6.0f\n",
           C, H, W, a, bytes_plain / a / 1e3, b, (bytes_plain + px * 4) / b / 1e3, c, (bytes_plain + px * 12) / c / 1e3, d, px * (2.0 * C * es + 8) / d / 1e3,
           e, px * (4.0 * C * es + 81.0 * es) / e / 1e3, eg1, eg2, e2, f, px * (2.0 * C * es + 4.0 * C + 16) / f / 1e3);
    tb += bytes_plain + px * 4; tt += b;
    for (void* p : {f1, f2, out, g1, g2, flow, occ, gimg, gflow}) CK(hipFree(p));
  }
  printf("corr+occ all levels: %.1f MB in %.1f us = %.0f GB/s (%.1f%% of 8 TB/s)\n", tb / 1e6, tt, tb / tt / 1e3, tb / tt / 1e3 / 80.0);

  // ---- loss kernels at the full image size (B x 3 x 436 x 1024 fp32 images, flow [B,2,..], occlusion [B,1,..]); GB/s = algorithmic
  // bytes per pixel of DESIGN.md section 4 / time
  {
    const int C = 3, H = 436, W = 1024;
    const size_t np = (size_t)B * H * W;
    float *i1, *i2, *fl, *oc, *gfl, *rng, *coef; double* sums;
    CK(hipMalloc(&i1, np * C * 4)); CK(hipMalloc(&i2, np * C * 4)); CK(hipMalloc(&fl, np * 8)); CK(hipMalloc(&oc, np * 4)); CK(hipMalloc(&gfl, np * 8));
    CK(hipMalloc(&rng, np * 4)); CK(hipMalloc(&coef, 16)); CK(hipMalloc(&sums, 64 * sizeof(double)));
    std::vector<float> h(np * C);
    for (size_t i = 0; i < np * C; i++) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    CK(hipMemcpy(i1, h.data(), np * C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(i2, h.data() , np * C * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < np * 2; i++) h[i] = (float)((i * 40503u) % 4001) / 1000.f - 2.f;
    CK(hipMemcpy(fl, h.data(), np * 8, hipMemcpyHostToDevice));
    for (size_t i = 0; i < np; i++) h[i] = (float)((i * 9973u) % 1000) / 1000.f;
    CK(hipMemcpy(oc, h.data(), np * 4, hipMemcpyHostToDevice));
    const float hc[4] = {1e-6f, 1e-6f, 1e-6f, 1e-6f};
    CK(hipMemcpy(coef, hc, 16, hipMemcpyHostToDevice));
    ocf_tensor4 t1 = mkt(i1, B, C, H, W, false), t2 = mkt(i2, B, C, H, W, false), tf = mkt(fl, B, 2, H, W, false), to = mkt(oc, B, 1, H, W, false),
                tg = mkt(gfl, B, 2, H, W, false);
    const double px = (double)np;
    float a = time_us([&] { ocf_photo_loss_fwd(&t1, &t2, &tf, &to, sums, B, C, H, W, st); }, st, 50);
    float b = time_us([&] { ocf_photo_loss_bwd(&t1, &t2, &tf, &to, coef, &tg, nullptr, B, C, H, W, st); }, st, 50);
    float c = time_us([&] { ocf_smooth_loss_fwd(&t1, &tf, sums, B, C, H, W, 100.f, st); }, st, 50);
    float d = time_us([&] { ocf_smooth_loss_bwd(&t1, &tf, coef, &tg, B, C, H, W, 100.f, st); }, st, 50);
    float e_old = time_us([&] { ocf_range_map(&tf, rng, oc, B, H, W, OCF_RANGE_POLICY_SPLAT, st); }, st, 50);
    float e = time_us([&] { ocf_range_map(&tf, rng, oc, B, H, W, OCF_RANGE_POLICY_GATHER, st); }, st, 50);
    printf("range map: gather + far %.1f us | LDS-window splat %.1f us\n", e, e_old);
    float f = time_us([&] { ocf_ssim_fwd(&t1, &t2, sums, B, C, H, W, 11, st); }, st, 20);
    float ms16 = time_us([&] { CK(hipMemsetAsync(sums, 0, 16, st)); }, st, 200);
    float ms14m = time_us([&] { CK(hipMemsetAsync(rng, 0, np * 4, st)); }, st, 50);
    printf("hipMemsetAsync back-to-back on the stream: 16 B %5.1f us, %zu MB %5.1f us\n", ms16, np * 4 >> 20, ms14m);
    printf("losses %dx3x%dx%d fp32 | photo fwd %6.1f us %5.0f GB/s  bwd %6.1f us %5.0f | smooth fwd %6.1f us %5.0f  bwd %6.1f us %5.0f | range map %6.1f us %5.0f | ssim(11) %7.1f us\n",
           B, H, W, a, px * 36 / a / 1e3, b, px * 44 / b / 1e3, c, px * 20 / c / 1e3, d, px * 28 / d / 1e3, e, px * 16 / e / 1e3, f);
    for (void* p : {(void*)i1, (void*)i2, (void*)fl, (void*)oc, (void*)gfl, (void*)rng, (void*)coef, (void*)sums}) CK(hipFree(p));
  }
  // ---- convolution epilogue at the level-2 decoder shapes (bf16 channels_last): in-place bias + LeakyReLU and its backward
  if (bf && cl) {
    const int H = 109, W = 256;
    for (int C : {128, 96, 64, 32}) {
      const size_t n = (size_t)B * C * H * W;
      void *x, *gy, *gx; float *bias, *gb;
      CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&gy, n * 2)); CK(hipMalloc(&gx, n * 2)); CK(hipMalloc(&bias, C * 4)); CK(hipMalloc(&gb, C * 4));
      CK(hipMemset(x, 0x3c, n * 2)); CK(hipMemset(gy, 0x3c, n * 2)); CK(hipMemset(bias, 0, C * 4));
      ocf_tensor4 tx = mkt(x, B, C, H, W, true), tgy = mkt(gy, B, C, H, W, true), tgx = mkt(gx, B, C, H, W, true);
      float a = time_us([&] { ocf_bias_act_fwd(&tx, bias, B, C, H, W, 0.1f, OCF_BF16, st); }, st, 100);
      float b = time_us([&] { ocf_bias_act_bwd(&tx, &tgy, &tgx, gb, B, C, H, W, 0.1f, 0, OCF_BF16, st); }, st, 100);
      printf("epilogue %dx%dx%dx%d bf16 | bias+lrelu fwd %6.1f us %5.0f GB/s | bwd (+bias grad) %6.1f us %5.0f GB/s\n", B, C, H, W, a, n * 4.0 / a / 1e3, b,
             n * 6.0 / b / 1e3);
      for (void* p : {x, gy, gx, (void*)bias, (void*)gb}) CK(hipFree(p));
    }
  }
  return 0;
}
