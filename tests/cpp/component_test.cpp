// tests/cpp/component_test.cpp -- drives include/klstm_component.hpp (the C++ mirror of the reference
// component) for tests/test_component.py.  Host-only modes need no GPU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/klstm_component.hpp"
#include "../../include/klstm_trainer.hpp"
#include "../../include/klstm_nnet.hpp"

using namespace klstm_kaldi;

static std::vector<float> read_raw(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) KLSTM_ERR("cannot open " << path);
  f.seekg(0, std::ios::end);
  const size_t n = (size_t)f.tellg() / sizeof(float);
  f.seekg(0);
  std::vector<float> v(n);
  f.read(reinterpret_cast<char *>(v.data()), n * sizeof(float));
  return v;
}
static void write_raw(const std::string &path, const float *p, size_t n) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char *>(p), n * sizeof(float));
}
static std::vector<Utterance> read_utts(const std::string &path) {
  // float32 stream: nutt, then per utterance: len, dim, ntargets, feats[len*dim], targets[ntargets]
  const std::vector<float> raw = read_raw(path);
  size_t o = 0;
  const int nutt = (int)raw[o++];
  std::vector<Utterance> utts(nutt);
  for (auto &u : utts) {
    u.num_frames = (int)raw[o++]; u.dim = (int)raw[o++];
    const int nt = (int)raw[o++];
    u.feats.assign(raw.begin() + o, raw.begin() + o + (size_t)u.num_frames * u.dim); o += (size_t)u.num_frames * u.dim;
    for (int i = 0; i < nt; i++) u.targets.push_back((int32)raw[o++]);
  }
  return utts;
}
static LstmProjectedStreams *load_model(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) KLSTM_ERR("cannot open " << path);
  const bool binary = InitKaldiInputStream(f);
  LstmProjectedStreams *c = ReadLstmComponent(f, binary);
  if (!c) KLSTM_ERR("no component in " << path);
  return c;
}
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) KLSTM_ERR(#x << ": " << hipGetErrorString(e_)); } while (0)

int main(int argc, char **argv) {
  try {
    const std::string mode = argc > 1 ? argv[1] : "";
    if (mode == "init_write") {
      // init_write <marker> <in> <out> "<proto tokens>" <binary> <file> <params_raw>
      const std::string marker = argv[2];
      const int I = atoi(argv[3]), O = atoi(argv[4]);
      std::unique_ptr<LstmProjectedStreams> c(marker == "<LstmProjected>" ? new LstmProjected(I, O) : new LstmProjectedStreams(I, O));
      std::istringstream proto(argv[5]);
      c->InitData(proto);
      const bool binary = atoi(argv[6]) != 0;
      std::ofstream f(argv[7], std::ios::binary);
      InitKaldiOutputStream(f, binary);
      c->Write(f, binary);
      std::vector<float> p;
      c->GetParams(&p);
      write_raw(argv[8], p.data(), p.size());
      std::cout << "OK " << c->NumParams() << "\n";
    } else if (mode == "dump_params") {
      // dump_params <model> <params_raw>  -> prints "marker in out cell nstream"
      std::unique_ptr<LstmProjectedStreams> c(load_model(argv[2]));
      std::vector<float> p;
      c->GetParams(&p);
      write_raw(argv[3], p.data(), p.size());
      std::cout << c->Marker() << " " << c->InputDim() << " " << c->OutputDim() << " " << c->CellDim() << " " << c->NumStream() << "\n";
    } else if (mode == "convert") {
      // convert <model_in> <binary> <model_out>   (nnet-copy for one component)
      std::unique_ptr<LstmProjectedStreams> c(load_model(argv[2]));
      const bool binary = atoi(argv[3]) != 0;
      std::ofstream f(argv[4], std::ios::binary);
      InitKaldiOutputStream(f, binary);
      c->Write(f, binary);
      std::cout << "OK\n";
    } else if (mode == "read_vectors") {
      // read_vectors <kaldi text/binary nnet made of vector-parameter components> <out_raw>
      // (a genuine Kaldi-written file, e.g. the reference's feature_transform.nnet.txt: <Nnet> <AddShift> d d [ v ] <Rescale> d d [ v ] </Nnet>)
      std::ifstream f(argv[2], std::ios::binary);
      const bool binary = InitKaldiInputStream(f);
      ExpectToken(f, binary, "<Nnet>");
      std::vector<float> all;
      std::string tok;
      while (true) {
        ReadToken(f, binary, &tok);
        if (tok == "</Nnet>") break;
        int32 dout = 0, din = 0;
        ReadBasicType(f, binary, &dout);
        ReadBasicType(f, binary, &din);
        std::vector<BaseFloat> v;
        ReadVector(f, binary, &v);
        std::cout << tok << " " << dout << " " << din << " " << v.size() << "\n";
        all.insert(all.end(), v.begin(), v.end());
      }
      write_raw(argv[3], all.data(), all.size());
    } else if (mode == "read_matrices") {
      // read_matrices <file with matrices back to back (text, or binary behind a \0B header)> <out_raw>: prints "rows cols" per
      // matrix until the stream is exhausted, dumps all elements
      std::ifstream f(argv[2], std::ios::binary);
      const bool binary = InitKaldiInputStream(f);
      std::vector<float> all;
      while (true) {
        if (!binary) f >> std::ws;
        if (f.peek() == EOF) break;
        std::vector<BaseFloat> m; int32 r = 0, c = 0;
        ReadMatrix(f, binary, &m, &r, &c);
        std::cout << r << " " << c << "\n";
        all.insert(all.end(), m.begin(), m.end());
      }
      write_raw(argv[3], all.data(), all.size());
    } else if (mode == "bad_proto") {
      LstmProjectedStreams c(5, 4);
      std::istringstream proto("<CellDim> 7 <Bogus> 3");
      c.InitData(proto);       // must throw: "Unknown token <Bogus>, a typo in config?"
      std::cout << "NOT THROWN\n";
      return 1;
    } else if (mode == "batcher") {
      // batcher <utts_raw> <S> <T> <delay> <out_raw>
      // utts_raw (float32 stream): nutt, then per utterance: len, dim, ntargets, feats[len*dim], targets[ntargets]
      std::vector<Utterance> utts = read_utts(argv[2]);
      MultiStreamBatcher mb(&utts, atoi(argv[3]), atoi(argv[4]), atoi(argv[5]));
      StreamBatch b;
      std::vector<float> out;       // per batch: feat, target, mask, flags
      int nb = 0;
      while (mb.Next(&b)) {
        out.insert(out.end(), b.feat.begin(), b.feat.end());
        for (int32 v : b.target) out.push_back((float)v);
        out.insert(out.end(), b.frame_mask.begin(), b.frame_mask.end());
        for (int v : b.new_utt_flags) out.push_back((float)v);
        nb++;
      }
      write_raw(argv[6], out.data(), out.size());
      std::cout << "OK " << nb << " " << mb.NumDone() << " " << mb.NumOtherError() << "\n";
    } else if (mode == "nnet_copy") {
      // nnet_copy <nnet_in> <binary> <nnet_out>          (nnet-copy workalike; host only)
      Nnet nnet; nnet.Read(argv[2]);
      nnet.Write(argv[4], atoi(argv[3]) != 0);
      std::cout << "OK " << nnet.NumComponents();
      for (int i = 0; i < nnet.NumComponents(); i++) std::cout << " " << nnet.GetComponent(i).Marker();
      std::cout << "\n";
    } else if (mode == "nnet_train") {
      // nnet_train <nnet_in> <utts_raw> <S> <T> <delay> <lr> <momentum> <crossvalidate> <nnet_out>
      Nnet nnet; nnet.Read(argv[2]);
      std::vector<Utterance> utts = read_utts(argv[3]);
      TrainLstmStreamsOptions o;
      o.num_stream = atoi(argv[4]); o.batch_size = atoi(argv[5]); o.targets_delay = atoi(argv[6]);
      o.trn_opts.learn_rate = (float)atof(argv[7]); o.trn_opts.momentum = (float)atof(argv[8]);
      o.crossvalidate = atoi(argv[9]) != 0;
      std::string report;
      const TrainLstmStreamsStats st = TrainLstmStreams(&nnet, utts, o, &report);
      if (!o.crossvalidate) nnet.Write(argv[10], true);                     // :294-296
      std::cout.precision(10);
      std::cout << "OK " << st.num_done << " " << st.num_minibatches << " " << st.total_frames << " " << st.avg_loss << " "
                << st.frame_accuracy << " " << st.total_frames / st.seconds << "\n" << report << "\n";
    } else if (mode == "nnet_forward") {
      // nnet_forward <nnet> <feats_raw> <rows> <out_raw>     (nnet-forward workalike: one utterance, Feedforward)
      Nnet nnet; nnet.Read(argv[2]);
      const std::vector<float> x = read_raw(argv[3]);
      const int rows = atoi(argv[4]);
      DeviceMatrix in, out;
      in.CopyFromHost(x.data(), rows, nnet.InputDim());
      nnet.Feedforward(in.View(), &out);
      std::vector<float> h;
      out.CopyToHost(&h);
      write_raw(argv[5], h.data(), h.size());
      std::cout << "OK " << out.NumRows() << " " << out.NumCols() << "\n";
    } else if (mode == "run_gpu" || mode == "run_gpu_host") {
      // run_gpu <model> <in_raw> <od_raw> <rows> <lr> <momentum> <nsteps> <out_prefix>
      // run_gpu_host: the four matrices live in HOST memory (CuMatrix with the GPU disabled, cu-matrix.h:479-481)
      const bool host = mode == "run_gpu_host";
      std::unique_ptr<LstmProjectedStreams> c(load_model(argv[2]));
      const std::vector<float> x = read_raw(argv[3]), od = read_raw(argv[4]);
      const int rows = atoi(argv[5]);
      NnetTrainOptions opts;
      opts.learn_rate = (float)atof(argv[6]);
      opts.momentum = (float)atof(argv[7]);
      const int nsteps = atoi(argv[8]);
      const std::string prefix = argv[9];
      c->SetTrainOptions(opts);
      if (const char *f = getenv("KLSTM_TEST_FOLD")) c->SetEngineOption("fold", atoi(f));   // folded recurrence forced on/off
      void *comm = nullptr;
      if (const char *dp = getenv("KLSTM_TEST_DP")) {      // "nranks:rank:idfile" -- data-parallel step through the C-ABI's RCCL calls
        int nranks = 1, rank = 0; char idfile[512] = "";
        sscanf(dp, "%d:%d:%511s", &nranks, &rank, idfile);
        char id[KLSTM_COMM_ID_BYTES];
        if (rank == 0) {
          if (klstm_comm_get_unique_id(id) != KLSTM_OK) KLSTM_ERR("klstm_comm_get_unique_id: " << klstm_last_error());
          if (nranks > 1) { std::ofstream g(std::string(idfile) + ".tmp", std::ios::binary); g.write(id, sizeof(id)); g.close(); rename((std::string(idfile) + ".tmp").c_str(), idfile); }
        } else {
          for (int tries = 0; tries < 600; tries++) { std::ifstream g(idfile, std::ios::binary); if (g.read(id, sizeof(id))) break; usleep(100000); }
        }
        if (klstm_comm_init_rank(0, nranks, rank, id, &comm) != KLSTM_OK) KLSTM_ERR("klstm_comm_init_rank: " << klstm_last_error());
        c->SetDataParallel(comm);
        std::cout << "DP " << nranks << " ranks, rank " << rank << "\n";
      }

      const int I = c->InputDim(), R = c->OutputDim();
      // pitched device matrices, like CuMatrix (cu-matrix.cc:67-73): stride > cols
      const int xs = I + 4, os = R + 8, ds = R + 4, is = I + 12;
      float *dx, *dout, *dod, *did;
      std::vector<float> hx, hodm, houtm, hidm;
      const hipMemcpyKind up = host ? hipMemcpyHostToHost : hipMemcpyHostToDevice;
      const hipMemcpyKind down = host ? hipMemcpyHostToHost : hipMemcpyDeviceToHost;
      if (host) {
        hx.resize((size_t)rows * xs); hodm.resize((size_t)rows * ds); houtm.resize((size_t)rows * os); hidm.resize((size_t)rows * is);
        dx = hx.data(); dod = hodm.data(); dout = houtm.data(); did = hidm.data();
      } else {
        HIPOK(hipMalloc(&dx, (size_t)rows * xs * 4)); HIPOK(hipMalloc(&dout, (size_t)rows * os * 4));
        HIPOK(hipMalloc(&dod, (size_t)rows * ds * 4)); HIPOK(hipMalloc(&did, (size_t)rows * is * 4));
      }
      HIPOK(hipMemcpy2D(dx, xs * 4, x.data(), I * 4, I * 4, rows, up));
      HIPOK(hipMemcpy2D(dod, ds * 4, od.data(), R * 4, R * 4, rows, up));
      MatrixView in(dx, rows, I, xs), out(dout, rows, R, os), out_diff(dod, rows, R, ds), in_diff(did, rows, I, is);
      std::vector<int> flags(c->NumStream(), 1);
      std::vector<float> hout((size_t)rows * R), hid((size_t)rows * I);
      for (int step = 0; step < nsteps; step++) {
        if (step == 0) c->Reset(flags);
        c->PropagateFnc(in, &out);
        c->BackpropagateFnc(in, out, out_diff, &in_diff);
        if (step == nsteps - 1) {
          HIPOK(hipDeviceSynchronize());
          HIPOK(hipMemcpy2D(hout.data(), R * 4, dout, os * 4, R * 4, rows, down));
          HIPOK(hipMemcpy2D(hid.data(), I * 4, did, is * 4, I * 4, rows, down));
          std::ofstream g(prefix + ".gradinfo");
          g << c->InfoGradient() << "\n";
        }
        c->Update(in, out_diff);
        // Copy() after the first minibatch: parameters, momentum buffers, carried c / r state and engine options must
        // all carry over (the reference copy-constructs every member, ...streams.h:38)
        if (getenv("KLSTM_TEST_COPY") && step == 0) c.reset(c->Copy());
      }
      write_raw(prefix + ".out", hout.data(), hout.size());
      write_raw(prefix + ".in_diff", hid.data(), hid.size());
      std::vector<float> p;
      c->GetParams(&p);
      write_raw(prefix + ".params", p.data(), p.size());
      std::ofstream f(prefix + ".model", std::ios::binary);   // trained model, binary Kaldi format
      InitKaldiOutputStream(f, true);
      c->Write(f, true);
      std::cout << "OK\n" << c->Info() << "\n";
      if (comm) klstm_comm_destroy(comm);
    } else if (mode == "run_gpu_full") {
      // run_gpu_full <model> <x_raw> <od_raw> <flags_raw> <rows_per_minibatch> <nmb> <lr> <momentum> <out_prefix>
      // The component exactly as the Kaldi shim of INTEGRATION.md 2 constructs and drives it, at ANY shape (the tests run the
      // benchmarked one: 40/800/512, 4 streams x 20 frames, and <LstmProjected> over a 1000-frame utterance): SetUpdateFollows(true),
      // persist_verify at the mirror's default, pitched device matrices, the trainer's call order per minibatch
      // (bd-nnet-train-lstm-streams.cc:209-228: Reset(new_utt_flags) EVERY minibatch, Propagate, Backpropagate = BackpropagateFnc +
      // Update).  Every minibatch has its own rows; out / in_diff of every minibatch are written out, then the parameters, the
      // momentum buffers, the carried state and the engine's persistent-launch counters.
      std::unique_ptr<LstmProjectedStreams> c(load_model(argv[2]));
      const std::vector<float> x = read_raw(argv[3]), od = read_raw(argv[4]), fl = read_raw(argv[5]);
      const int rows = atoi(argv[6]), nmb = atoi(argv[7]);
      NnetTrainOptions opts;
      opts.learn_rate = (float)atof(argv[8]);
      opts.momentum = (float)atof(argv[9]);
      const std::string prefix = argv[10];
      c->SetTrainOptions(opts);
      c->SetUpdateFollows(true);                               // the shim's constructor
      const int I = c->InputDim(), R = c->OutputDim(), S = c->NumStream();
      const bool streams = std::string(c->Marker()) == "<LstmProjectedStreams>";
      if ((long)x.size() != (long)nmb * rows * I || (long)od.size() != (long)nmb * rows * R || (streams && (long)fl.size() != (long)nmb * S))
        KLSTM_ERR("run_gpu_full: input sizes do not match " << nmb << " minibatches of " << rows << " rows");
      const int xs = I + 4, os = R + 8, ds = R + 4, is = I + 12;       // CuMatrix rows are pitched (cu-matrix.cc:67-73)
      float *dx, *dout, *dod, *did;
      HIPOK(hipMalloc(&dx, (size_t)nmb * rows * xs * 4)); HIPOK(hipMalloc(&dout, (size_t)rows * os * 4));
      HIPOK(hipMalloc(&dod, (size_t)nmb * rows * ds * 4)); HIPOK(hipMalloc(&did, (size_t)rows * is * 4));
      HIPOK(hipMemcpy2D(dx, xs * 4, x.data(), I * 4, I * 4, (size_t)nmb * rows, hipMemcpyHostToDevice));
      HIPOK(hipMemcpy2D(dod, ds * 4, od.data(), R * 4, R * 4, (size_t)nmb * rows, hipMemcpyHostToDevice));
      std::vector<float> hout((size_t)nmb * rows * R), hid((size_t)nmb * rows * I);
      for (int mb = 0; mb < nmb; mb++) {
        MatrixView in(dx + (size_t)mb * rows * xs, rows, I, xs), out(dout, rows, R, os);
        MatrixView out_diff(dod + (size_t)mb * rows * ds, rows, R, ds), in_diff(did, rows, I, is);
        if (streams) {
          std::vector<int> flags(S);
          for (int s = 0; s < S; s++) flags[s] = fl[(size_t)mb * S + s] != 0.f;
          c->Reset(flags);                                       // every minibatch, like nnet.Reset(new_utt_flags) (:209)
        }
        c->PropagateFnc(in, &out);
        c->BackpropagateFnc(in, out, out_diff, &in_diff);
        c->Update(in, out_diff);
        // (what the trainer does next: Xent::EvalMasked of the following minibatch copies scalars to the host, nnet-loss.cc:110-141)
        HIPOK(hipMemcpy2D(hout.data() + (size_t)mb * rows * R, R * 4, dout, os * 4, R * 4, rows, hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy2D(hid.data() + (size_t)mb * rows * I, I * 4, did, is * 4, I * 4, rows, hipMemcpyDeviceToHost));
      }
      write_raw(prefix + ".out", hout.data(), hout.size());
      write_raw(prefix + ".in_diff", hid.data(), hid.size());
      std::vector<float> p;
      c->GetParams(&p);
      write_raw(prefix + ".params", p.data(), p.size());
      std::vector<float> corr(p.size()), sc((size_t)S * c->CellDim()), sr((size_t)S * R);
      if (klstm_get_corr_host(c->Engine(), corr.data()) != KLSTM_OK) KLSTM_ERR("klstm_get_corr_host: " << klstm_last_error());
      if (klstm_get_state_host(c->Engine(), sc.data(), sr.data()) != KLSTM_OK) KLSTM_ERR("klstm_get_state_host: " << klstm_last_error());
      write_raw(prefix + ".corr", corr.data(), corr.size());
      write_raw(prefix + ".state_c", sc.data(), sc.size());
      write_raw(prefix + ".state_r", sr.data(), sr.size());
      std::cout << "OK";
      for (const char *key : {"persist_launches", "persist_giveups", "persist_replayed", "persist_dropped", "fp16_redo"}) {
        double us = 0; long n = 0;
        if (klstm_profile_query(c->Engine(), key, &us, &n) != KLSTM_OK) KLSTM_ERR("klstm_profile_query(" << key << "): " << klstm_last_error());
        std::cout << " " << key << "=" << n;
      }
      std::cout << "\n";
      HIPOK(hipFree(dx)); HIPOK(hipFree(dout)); HIPOK(hipFree(dod)); HIPOK(hipFree(did));
    } else {
      std::cerr << "usage: component_test init_write|dump_params|convert|bad_proto|run_gpu|run_gpu_host|run_gpu_full ...\n";
      return 2;
    }
    return 0;
  } catch (const std::exception &e) {
    std::cout << "EXCEPTION: " << e.what() << "\n";
    return 3;
  }
}
