// tools/kaldi_adapter_bench.cpp -- what a Kaldi build gets: the C++ component of include/klstm_component.hpp, constructed and driven
// exactly as the shim of INTEGRATION.md 2 does it, timed at the shape bench.py measures (BASELINE.json configs[1]: 40 -> 800 / 512,
// NumStream 4, T = 20, 1000-frame synthetic utterances = 50 chained minibatches per pass).
//
//   per minibatch, the trainer's call order (bd-nnet-train-lstm-streams.cc:209-228):
//     Reset(new_utt_flags)                       ...streams.h:212     every minibatch, flags set at utterance starts
//     PropagateFnc(in, &out)                     ...streams.h:222
//     BackpropagateFnc(in, out, out_diff, &in_diff)   :334-335       (SetUpdateFollows(true): Component::Backpropagate calls
//     Update(in, out_diff)                       :501                  Update right behind it)
//   on pitched device matrices (CuMatrix rows are pitched, cu-matrix.cc:67-73), "persist_verify" at the mirror's default (1: every
//   persistent call waits for its launch) unless argv says otherwise.
//
// Needs no HIP headers (g++ -std=c++17 -O2 -Iinclude tools/kaldi_adapter_bench.cpp -Lkaldi-lstm_amd -lklstm): device memory through
// the klstm_malloc / klstm_memcpy_* helpers of the C-ABI.  Prints ONE JSON line; bench.py puts it into its line as `kaldi_adapter`.
//
//   kaldi_adapter_bench [streams=4] [steps=400] [warmup=50] [persist_verify=1] [d2h_per_minibatch=0] [verify_spin=1] [d2h_small=1]
//     verify_spin = 0: the engine's wait for a persistent launch is a hipStreamSynchronize (round 4's) instead of a spin on the
//     host-mapped done word
//     d2h_per_minibatch = 1: a 12-byte pageable device-to-host copy after Update, what Xent::EvalMasked does per minibatch
//     (google/nnet/nnet-loss.cc:110-141)
//     d2h_small = 0: that copy through hipMemcpyAsync + hipStreamSynchronize (rounds 1-5) instead of klstm_memcpy_d2h's small-copy kernel
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "klstm_component.hpp"

using namespace klstm_kaldi;

#define OKC(x) do { if ((x) != KLSTM_OK) { std::fprintf(stderr, "%s: %s\n", #x, klstm_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
  const int S = argc > 1 ? std::atoi(argv[1]) : 4;
  const int steps = argc > 2 ? std::atoi(argv[2]) : 400;
  const int warmup = argc > 3 ? std::atoi(argv[3]) : 50;
  const int verify = argc > 4 ? std::atoi(argv[4]) : 1;
  const int d2h = argc > 5 ? std::atoi(argv[5]) : 0;
  const int vspin = argc > 6 ? std::atoi(argv[6]) : 1;
  const int dsmall = argc > 7 ? std::atoi(argv[7]) : 1;
  const int I = 40, C = 800, R = 512, T = 20, NCHUNK = 50;
  try {
    LstmProjectedStreams c(I, R);
    std::istringstream proto("<CellDim> " + std::to_string(C) + " <ParamScale> 0.01 <NumStream> " + std::to_string(S));   // google/nnet.proto:3
    std::srand(7);
    c.InitData(proto);
    NnetTrainOptions opts;
    opts.learn_rate = 1e-5f; opts.momentum = 0.9f;                  // train_lstm_streams.sh:3-4
    c.SetTrainOptions(opts);
    c.SetUpdateFollows(true);                                       // the shim's constructor
    if (!verify) c.SetPersistVerify(false);
    if (!vspin) c.SetEngineOption("persist_verify_spin", 0);
    if (!dsmall) c.SetEngineOption("d2h_small", 0);

    const int rows = T * S, xs = I + 4, os = R + 8, ds = R + 4, is = I + 12;
    std::mt19937 gen(1234);
    std::normal_distribution<float> nrm(0.f, 1.f);
    std::vector<float> hx((size_t)NCHUNK * rows * xs, 0.f), hod((size_t)NCHUNK * rows * ds, 0.f);
    for (size_t r = 0; r < (size_t)NCHUNK * rows; r++) {
      for (int j = 0; j < I; j++) hx[r * xs + j] = nrm(gen);
      for (int j = 0; j < R; j++) hod[r * ds + j] = 0.1f * nrm(gen);
    }
    void *dx, *dod, *dout, *did, *dscal;
    OKC(klstm_malloc(&dx, hx.size() * 4)); OKC(klstm_malloc(&dod, hod.size() * 4));
    OKC(klstm_malloc(&dout, (size_t)rows * os * 4)); OKC(klstm_malloc(&did, (size_t)rows * is * 4)); OKC(klstm_malloc(&dscal, 64));
    OKC(klstm_memcpy_h2d(dx, hx.data(), hx.size() * 4, nullptr)); OKC(klstm_memcpy_h2d(dod, hod.data(), hod.size() * 4, nullptr));
    OKC(klstm_memset_zero(dscal, 64, nullptr));

    std::vector<int> start(S, 1), none(S, 0);
    float scal[3];
    auto step = [&](int i) {
      const int ck = i % NCHUNK;
      MatrixView in((float *)dx + (size_t)ck * rows * xs, rows, I, xs), out((float *)dout, rows, R, os);
      MatrixView out_diff((float *)dod + (size_t)ck * rows * ds, rows, R, ds), in_diff((float *)did, rows, I, is);
      c.Reset(ck == 0 ? start : none);
      c.PropagateFnc(in, &out);
      c.BackpropagateFnc(in, out, out_diff, &in_diff);
      c.Update(in, out_diff);
      if (d2h) klstm_memcpy_d2h(scal, dscal, sizeof(scal), nullptr);
    };
    for (int i = 0; i < warmup; i++) step(i);
    OKC(klstm_synchronize(c.Engine()));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; i++) step(warmup + i);
    OKC(klstm_synchronize(c.Engine()));
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    long launches = 0, giveups = 0, replayed = 0, dropped = 0;
    double us = 0;
    OKC(klstm_profile_query(c.Engine(), "persist_launches", &us, &launches));
    OKC(klstm_profile_query(c.Engine(), "persist_giveups", &us, &giveups));
    OKC(klstm_profile_query(c.Engine(), "persist_replayed", &us, &replayed));
    OKC(klstm_profile_query(c.Engine(), "persist_dropped", &us, &dropped));
    std::printf("{\"value\": %.1f, \"unit\": \"frames/s\", \"ms_per_step\": %.5f, \"steps\": %d, \"warmup\": %d, \"streams\": %d, "
                "\"persist_verify\": %d, \"verify_spin\": %d, \"d2h_per_minibatch\": %d, \"d2h_small\": %d, \"update\": \"fused (SetUpdateFollows(true))\", "
                "\"persist_launches\": %ld, \"persist_launches_expected\": %ld, \"persist_giveups\": %ld, \"persist_replayed\": %ld, "
                "\"persist_dropped\": %ld, \"driver\": \"tools/kaldi_adapter_bench.cpp: klstm_kaldi::LstmProjectedStreams as in INTEGRATION.md 2, "
                "Reset + PropagateFnc + BackpropagateFnc + Update per minibatch on pitched device matrices\"}\n",
                (double)steps * rows / sec, sec / steps * 1e3, steps, warmup, S, verify, vspin, d2h, dsmall, launches, 2L * (steps + warmup), giveups,
                replayed, dropped);
    klstm_free(dx); klstm_free(dod); klstm_free(dout); klstm_free(did); klstm_free(dscal);
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "EXCEPTION: %s\n", e.what());
    return 3;
  }
}
