// tools/persist_anatomy.hip -- where does a step of the persistent chains (forward, backward) spend its time?  Builds the product
// kernel (klstm_persist.hip) with KLSTM_PERSIST_TIMING: every wave sums shader-clock intervals per phase
// (sweep, barrier 1, contraction, barrier 2, owner epilogue, loop head) over the T-1 steps.  Random weights (timing only).
#define KLSTM_PERSIST_TIMING
#include "../kaldi-lstm_amd/csrc/klstm_persist.hip"
#include "../kaldi-lstm_amd/csrc/klstm_persist_bwd.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
using namespace klstm;

int main(int argc, char **argv) {
  const int C = 800, I = 40, R = 512, S = argc > 1 ? atoi(argv[1]) : 4, T = 200;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int nchm = (C + 31) / 32, nch = nchm + (I + 31) / 32;
  const size_t npk = (size_t)(C / 4) * nch * 128;
  float4 *wpk; CK(hipMalloc(&wpk, npk * 16));
  std::vector<float> h(npk * 4);
  for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.02f;
  CK(hipMemcpy(wpk, h.data(), npk * 16, hipMemcpyHostToDevice));
  auto dalloc = [&](size_t n) { float *p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); return p; };
  float *vecs = dalloc(7 * C), *gifo = dalloc((size_t)(T + 2) * S * 4 * C), *cc = dalloc((size_t)(T + 2) * S * C),
        *hh = dalloc((size_t)(T + 2) * S * C), *mm = dalloc((size_t)(T + 2) * S * C), *x = dalloc((size_t)T * S * I), *cs = dalloc(S * C),
        *wr = dalloc((size_t)4 * C * R), *wx = dalloc((size_t)4 * C * I), *pr0 = dalloc(S * R), *rr = dalloc((size_t)(T + 2) * S * R);
  unsigned long long *gran; CK(hipMalloc(&gran, 32 * C * 8 * 8)); CK(hipMemset(gran, 0, 32 * C * 8 * 8));
  unsigned *ctrl; CK(hipMalloc(&ctrl, 32)); CK(hipMemset(ctrl, 0, 32));
  long long *dbg; CK(hipMalloc(&dbg, 256 * 16 * 10 * 8)); CK(hipMemset(dbg, 0, 256 * 16 * 10 * 8));
  for (int waves : {12}) for (int tpw : {1}) for (int nap0 : {0, 1, 2, 3, 4, 5, 6, 8}) for (int nap : {0}) {
    if (4 * tpw >= waves) continue;
    PersistOpts o; o.waves = waves; o.tpw = tpw;
    PersistFwdArgs a{};
    a.C = C; a.I = I; a.R = R; a.S = S; a.T = T; a.nchm = nchm; a.nch = nch; a.wpk = wpk; a.wr = wr; a.wx = wx; a.prev_r = pr0; a.rr = rr; a.wm = nullptr; a.rin = 0; a.out = nullptr; a.out_stride = 0;
    a.bias = vecs; a.pi = vecs + 4 * C; a.pf = vecs + 5 * C; a.po = vecs + 6 * C;
    a.gifo = gifo; a.cc = cc; a.hh = hh; a.mm = mm; a.x = x; a.x_stride = I; a.prev_c = cs; a.next_c = cs; a.next_r = pr0; a.guard = nullptr; a.gran = gran; a.ctrl = ctrl; a.dbg = dbg; a.nap0 = nap0; a.nap = nap; a.spin_limit = SPIN_LIMIT_DEFAULT; a.test_stall = 0;
    const PGeo g = pick_geo_fwd(o, C, nch, (R + 31) / 32 * 32 + I);
    const size_t shm = (size_t)((S > 4 ? 8 : 4) * (g.maxc * 128 + 16) + (S > 4 ? 8 : 4) * (persist_maxu(g.maxc) * 128 + 16) + 4) * sizeof(float);
    const int grid = C / 4 / g.tpw;
    LaunchProbe pr;
    const int ng = S > 4 ? 2 : 1;
    const bool xbat = false;
    const bool il = S > 4;                           // (5..8 streams: the two groups as interleaved chains, like the engine)
    a.xg = nullptr; a.hstat = nullptr;
    auto go = [&]() -> hipError_t { PDISPATCH_FWD(k_fwd_persist); };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipEventRecord(e0, st)); CK(go()); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
    }
    std::vector<long long> d(256 * 16 * 10); CK(hipMemcpy(d.data(), dbg, d.size() * 8, hipMemcpyDeviceToHost));
    unsigned stw[4]; CK(hipMemcpy(stw, ctrl, 16, hipMemcpyDeviceToHost));
    const double us_step = best * 1e3 / (T - 1);
    // shader clock -> us: total cycles of a wave / kernel time
    auto row = [&](int wg, int w) { return &d[((size_t)wg * 16 + w) * 10]; };
    long long tot = 0; for (int i = 0; i < 6; i++) tot += row(0, 0)[i];
    const double cyc_per_us = tot / (best * 1e3);
    printf("S=%d waves=%d tpw=%d grid=%d maxc=%d pcell=%d nap0=%d nap=%d: %.3f us/step (status %x), ~%.0f MHz shader clock\n", S, g.waves, g.tpw, grid, g.maxc, g.pcell, nap0, nap, us_step, stw[2], cyc_per_us);
    const char *nm[6] = {"sweep+slab", "barrier1", "contract", "barrier2", "epilogue", "loophead"};
    for (int w : {0}) {
      printf("   wg0 wave %2d (%s):", w, w < 4 * g.tpw ? "cell   " : "sweeper");
      for (int i = 0; i < 6; i++) printf(" %s %.2f", nm[i], row(0, w)[i] / cyc_per_us / (T - 1));
      printf("  us/step\n");
    }
    // mean over workgroups of the sweeper wave 1
    double m[6] = {0, 0, 0, 0, 0, 0};
    for (int wg = 0; wg < grid; wg++) for (int i = 0; i < 6; i++) m[i] += row(wg, 4 * g.tpw + 1)[i] / cyc_per_us / (T - 1) / grid;
    printf("   mean over workgroups, first sweeper wave:");
    for (int i = 0; i < 6; i++) printf(" %s %.2f", nm[i], m[i]);
    printf("\n");
  }
  // ---------------- backward (klstm_persist_bwd.hip) ----------------
  {
    const int nchb = (4 * C + 127) / 128;
    const size_t npb = (size_t)(C / 4) * nchb * 128;
    float4 *wpb; CK(hipMalloc(&wpb, npb * 16));
    std::vector<float> hb(npb * 4);
    for (auto &v : hb) v = (rand() / (float)RAND_MAX - 0.5f) * 0.02f;
    CK(hipMemcpy(wpb, hb.data(), npb * 16, hipMemcpyHostToDevice));
    auto rfill = [&](float *p, size_t n, float lo, float hi) {
      std::vector<float> t(n); for (auto &v : t) v = lo + (hi - lo) * (rand() / (float)RAND_MAX);
      CK(hipMemcpy(p, t.data(), n * 4, hipMemcpyHostToDevice));
    };
    rfill(gifo, (size_t)(T + 2) * S * 4 * C, 0.1f, 0.9f); rfill(hh, (size_t)(T + 2) * S * C, -0.5f, 0.5f); rfill(cc, (size_t)(T + 2) * S * C, -1.f, 1.f);
    float *dgifo = dalloc((size_t)(T + 2) * S * 4 * C), *dc = dalloc((size_t)(T + 2) * S * C), *P = dalloc((size_t)T * S * C),
          *dr = dalloc((size_t)(T + 2) * S * R), *od = dalloc((size_t)T * S * R), *idf = dalloc((size_t)T * S * I),
          *wrT = dalloc((size_t)R * 4 * C), *wxT = dalloc((size_t)I * 4 * C), *wmT = dalloc((size_t)C * R);
    rfill(od, (size_t)T * S * R, -0.1f, 0.1f); rfill(wrT, (size_t)R * 4 * C, -0.01f, 0.01f); rfill(wxT, (size_t)I * 4 * C, -0.01f, 0.01f);
    rfill(wmT, (size_t)C * R, -0.01f, 0.01f); rfill(P, (size_t)T * S * C, -0.01f, 0.01f);
    CK(hipMemset(gran, 0, 32 * C * 8 * 8)); CK(hipMemset(ctrl, 0, 32));
    BwdPtrs bp{};
    bp.wrT = wrT; bp.wmT = wmT; bp.wxT = wxT; bp.pi = vecs + 4 * C; bp.pf = vecs + 5 * C; bp.po = vecs + 6 * C;
    bp.gifo = gifo; bp.cc = cc; bp.hh = hh; bp.dgifo = dgifo; bp.dc = dc; bp.dr = dr; bp.pk_fold = wpb;
    bp.wr_nat = wr; bp.wx_nat = wx;                  // (the tail workgroups read the natural matrices)
    bp.pk_fold_gates = wpk; bp.nch_gates = nch;      // (the backward launch takes its columns of W_rm from the gates-order operand)
    const Dims d{I, C, R, S, T};
    const size_t tws_n = persist_bwd_tail_ws_floats(d, true);
    float *tws = dalloc(tws_n);
    // full: 0 bare chain; 1 P + d_r + in_diff on the chain's workgroups ("persist_tail" = 2); 2 P inside, d_r + in_diff on tail workgroups (round 6)
    for (int full : {0, 1, 2}) for (int waves : {16, 12}) for (int nap0 : {0}) {
      if (full == 2 && waves != 16) continue;
      PersistOpts o; o.bwd_waves = waves; o.nap0_bwd = nap0; o.dbg = dbg;
      o.ncu = 256; o.tail_mode = full == 2 ? 1 : 2;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float best = 1e9;
      for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0, st));
        CK(launch_bwd_persist(d, bp, P, full ? od : nullptr, R, full ? idf : nullptr, I, full != 0, gran, ctrl, o, st, {}, tws, tws_n, {}));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
      }
      std::vector<long long> dd(256 * 16 * 10); CK(hipMemcpy(dd.data(), dbg, dd.size() * 8, hipMemcpyDeviceToHost));
      unsigned stw[4]; CK(hipMemcpy(stw, ctrl, 16, hipMemcpyDeviceToHost));
      auto row = [&](int wg, int w) { return &dd[((size_t)wg * 16 + w) * 10]; };
      long long tot = 0; for (int i = 0; i < 10; i++) tot += row(0, 0)[i];
      const double cyc_per_us = tot / (best * 1e3);
      const int ngrp = (S + 3) / 4, nsteps = ngrp * (T - 1);
      const PGeo2 g = pick_geo_bwd2(d, o);
      printf("BWD2 S=%d waves=%d nu=%d %s nap0=%d: %.3f us/step over %d steps (%.1f us per launch, status %x)\n", S, g.nw, g.nu,
             full == 2 ? "P inside, d_r+in_diff on TAIL WORKGROUPS" : full ? "P+d_r+in_diff inside" : "bare chain", nap0, best * 1e3 / nsteps, nsteps, best * 1e3, stw[2]);
      const char *no[6] = {"wait-partials", "combine+publish", "own-rows", "-", "-", "loophead"};
      const char *ns[10] = {"planes+coef", "sweep", "apply+contract", "d-slice", "-", "loophead", "-", "-", "-", "-"};
      printf("   owner wg0 :");
      for (int i : {0, 1, 2, 5}) printf(" %s %.2f", no[i], row(0, 0)[i] / cyc_per_us / nsteps);
      double m[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mx[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      const int grid = C / 4;
      for (int wg = 0; wg < grid; wg++) for (int i = 0; i < 10; i++) {
        const double v = row(wg, 3)[i] / cyc_per_us / nsteps;
        m[i] += v / grid; if (v > mx[i]) mx[i] = v;
      }
      printf("\n   SC wave 1 (mean over workgroups | max):");
      for (int i : {5, 0, 1, 2, 3}) printf(" %s %.2f|%.2f", ns[i], m[i], mx[i]);
      printf("\n   wg0, per SC wave (columns as above):\n");
      for (int w = 2; w < g.nw; w++) {
        printf("     wave %2d:", w);
        for (int i : {5, 0, 1, 2, 3}) printf(" %.2f", row(0, w)[i] / cyc_per_us / nsteps);
        printf("\n");
      }
      const int ntw = full == 2 ? persist_bwd_tail_wgs(d, true, o) : 0;
      if (ntw) {
        // the tail workgroups keep their own clocks.  wave 0 (elementwise side): planes + sweep / apply + tile / barrier / contraction + partial rows / loop head
        const char *nt[6] = {"planes+sweep", "apply+tile", "barrier", "contract+store", "-", "loophead"};
        for (int w : {0, 3}) {
          printf("   %d tail workgroups, wave %d (mean | max over them; a step of theirs has to fit the chain's):", ntw, w);
          double tm[6] = {0, 0, 0, 0, 0, 0}, tx[6] = {0, 0, 0, 0, 0, 0}, sum = 0;
          for (int wg = grid; wg < grid + ntw; wg++) for (int i = 0; i < 6; i++) {
            const double v = row(wg, w)[i] / cyc_per_us / nsteps;
            tm[i] += v / ntw; if (v > tx[i]) tx[i] = v;
          }
          for (int i : {5, 0, 1, 2, 3}) { printf(" %s %.2f|%.2f", nt[i], tm[i], tx[i]); sum += tm[i]; }
          printf("  = %.2f us per step\n", sum);
        }
      }
    }
  }
  return 0;
}
