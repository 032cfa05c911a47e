// tools/fold_probe.hip -- where do the 36 us of the fold product (W_rm = W_gifo_r W_r_m, 3200 x 800 x 512, fp32 MFMA) go?
// Times the product kernel of klstm_kernels.hip under different residency limits (dynamic LDS padding) and epilogues.
#include "../kaldi-lstm_amd/csrc/klstm_kernels.hip"
#define KLSTM_FOLD_TIMING
#include "../kaldi-lstm_amd/csrc/klstm_fold.hip"
#include "../kaldi-lstm_amd/csrc/klstm_fold3.hip"      // (launch_fold of klstm_kernels.hip dispatches to it)
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
using namespace klstm;

int main() {
  const int C = 800, R = 512, I = 40;
  const Dims d{I, C, R, 4, 20};
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  auto dalloc = [&](size_t n) { float *p; CK(hipMalloc(&p, n * 4)); std::vector<float> h(n); for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f; CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p; };
  float *wr = dalloc((size_t)4 * C * R), *wmT = dalloc((size_t)C * R), *out = dalloc((size_t)4 * C * C);
  long nf[2]; pack_sizes_fold(d, nf);
  float *pk[2] = {dalloc((size_t)nf[0] * 4), dalloc((size_t)nf[1] * 4)};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char *name, auto &&launch) {
    for (int i = 0; i < 3; i++) launch();
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; i++) launch();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-60s %.2f us\n", name, ms * 1e3 / 20);
  };
  GemmJob g = make_job(false, true, 4 * C, C, R, wr, R, wmT, R, 0.f, nullptr, C, nullptr);
  g.gperm = C; g.pk1 = reinterpret_cast<float4 *>(pk[0]); g.nch1 = cdiv(C, KCH) + cdiv(I, KCH);
  g.pk2 = reinterpret_cast<float4 *>(pk[1]); g.nch2 = cdiv(4 * C, KCH4);
  const dim3 grid(cdiv(cdiv(C, GT) * cdiv(4 * C, GT), 8) * 8), block(256);
  for (int pad : {0, 14 * 1024, 24 * 1024, 42 * 1024}) {
    char nm[128]; snprintf(nm, sizeof nm, "packed epilogue, +%d KB LDS (=> %d workgroups per CU)", pad / 1024, 160 * 1024 / (40 * 1024 + pad));
    if (pad) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, pad));
    time(nm, [&]() { hipLaunchKernelGGL((k_gemm<false, true>), grid, block, pad, st, g); });
  }
  GemmJob p = make_job(false, true, 4 * C, C, R, wr, R, wmT, R, 0.f, out, C, nullptr);
  for (int pad : {0, 14 * 1024}) {
    char nm[128]; snprintf(nm, sizeof nm, "plain row-major epilogue, +%d KB LDS", pad / 1024);
    time(nm, [&]() { hipLaunchKernelGGL((k_gemm<false, true>), grid, block, pad, st, p); });
  }
  {
    const int nch1 = cdiv(C, KCH) + cdiv(I, KCH), nch2 = cdiv(4 * C, KCH4);
    CK(hipMalloc(&g_fold_dbg, 256 * 64)); CK(hipMemset(g_fold_dbg, 0, 256 * 64));
    time("direct: wave-owned 32x80 tiles, no LDS in the K loop", [&]() { launch_fold_direct(d, wr, wmT, pk, nch1, nch2, st); });
    {
      std::vector<long long> q(256 * 8); CK(hipMemcpy(q.data(), g_fold_dbg, q.size() * 8, hipMemcpyDeviceToHost));
      long long wmin = 1LL << 62, wmax = 0; double kc = 0, kw = 0, tc = 0, tw = 0; int n = 0;
      for (int w = 0; w < 256; w++) if (q[w * 8 + 3]) { n++; kc += q[w * 8]; kw += q[w * 8 + 1]; tc += q[w * 8 + 2]; tw += q[w * 8 + 3]; wmin = std::min(wmin, q[w * 8 + 4]); wmax = std::max(wmax, q[w * 8 + 4] + q[w * 8 + 3]); }
      printf("   per workgroup (mean of %d): K loop %.0f shader clocks = %.2f us (%.0f MHz, %.1f clocks per MFMA), whole kernel %.2f us; first start -> last end %.2f us\n",
             n, kc / n, kw / n / 100, kc / (kw / 100), kc / n / (16 * 80.0), tw / n / 100, (wmax - wmin) / 100.0);
    }
    auto report = [&](const char *what) {
      std::vector<long long> q(256 * 8); CK(hipMemcpy(q.data(), g_fold_dbg, q.size() * 8, hipMemcpyDeviceToHost));
      double kc = 0, kw = 0; int n = 0;
      for (int w = 0; w < 256; w++) if (q[w * 8 + 3]) { n++; kc += q[w * 8]; kw += q[w * 8 + 1]; }
      printf("   %s: K loop %.2f us, %.1f clocks per MFMA\n", what, kw / n / 100, kc / n / (16 * 80.0));
    };
    g_fold_kscale = 0;
    time("direct, every refill re-reads chunk 0 (L1 hits)", [&]() { launch_fold_direct(d, wr, wmT, pk, nch1, nch2, st); });
    report("chunk-0 refills");
    g_fold_kscale = 1;
    // same arrays from both kernels?
    std::vector<float> h0((size_t)nf[0] * 4), h1((size_t)nf[1] * 4), g0(h0.size()), g1(h1.size());
    CK(hipMemset(pk[0], 0, h0.size() * 4)); CK(hipMemset(pk[1], 0, h1.size() * 4));
    launch_fold_direct(d, wr, wmT, pk, nch1, nch2, st); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h0.data(), pk[0], h0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), pk[1], h1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(pk[0], 0, h0.size() * 4)); CK(hipMemset(pk[1], 0, h1.size() * 4));
    hipLaunchKernelGGL((k_gemm<false, true>), grid, block, 0, st, g); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(g0.data(), pk[0], g0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g1.data(), pk[1], g1.size() * 4, hipMemcpyDeviceToHost));
    double e0m = 0, e1m = 0; size_t nz = 0;
    for (size_t i = 0; i < h0.size(); i++) { e0m = std::max(e0m, (double)fabsf(h0[i] - g0[i])); nz += g0[i] != 0.f; }
    for (size_t i = 0; i < h1.size(); i++) e1m = std::max(e1m, (double)fabsf(h1[i] - g1[i]));
    printf("direct vs tiled kernel: max |diff| gates operand %.3g, d_m operand %.3g (%zu non-zero reference entries)\n", e0m, e1m, nz);
  }
  // K sweep (plain epilogue): the fixed cost per tile (first fetch, epilogue) vs the per-K-tile cost
  for (int K : {64, 128, 256, 512}) {
    GemmJob q = make_job(false, true, 4 * C, C, K, wr, R, wmT, R, 0.f, out, C, nullptr);
    char nm[128]; snprintf(nm, sizeof nm, "plain epilogue, K = %d", K);
    time(nm, [&]() { hipLaunchKernelGGL((k_gemm<false, true>), grid, block, 0, st, q); });
  }
  // fewer tiles: M = 2560 (40 x 13 = 520 tiles, ~2 per CU), M = 1216 (19 x 13 = 247 tiles, <= 1 per CU)
  for (int M : {1216, 2560, 3200}) {
    GemmJob q = make_job(false, true, M, C, R, wr, R, wmT, R, 0.f, out, C, nullptr);
    const dim3 gq(cdiv(cdiv(C, GT) * cdiv(M, GT), 8) * 8);
    char nm[128]; snprintf(nm, sizeof nm, "plain epilogue, M = %d (%d tiles)", M, cdiv(C, GT) * cdiv(M, GT));
    time(nm, [&]() { hipLaunchKernelGGL((k_gemm<false, true>), gq, block, 0, st, q); });
  }
  return 0;
}
