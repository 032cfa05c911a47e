// tools/fold3_probe.hip -- anatomy of the bf16x3 fold product (klstm_fold3.hip) at 800/512: the split pass, the product at
// different tile shapes / staging depths, and the same instruction stream without the LDS-DMA refills (MFMA + LDS floor).
#define KLSTM_FOLD3_TIMING
#include "../kaldi-lstm_amd/csrc/klstm_fold3.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
using namespace klstm;
static int cdv(int a, int b) { return (a + b - 1) / b; }

int main() {
  const int C = 800, R = 512, I = 40;
  const Dims d{I, C, R, 4, 20};
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  auto dalloc = [&](size_t n) { float *p; CK(hipMalloc(&p, n * 4)); std::vector<float> h(n); for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f; CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p; };
  float *wr = dalloc((size_t)4 * C * R), *wmT = dalloc((size_t)C * R);
  const int nch1 = cdv(C, 32) + cdv(I, 32), nch2 = cdv(4 * C, 128);
  float *pk[2] = {dalloc((size_t)cdv(C, 16) * 4 * nch1 * 128 * 4), dalloc((size_t)cdv(C, 4) * nch2 * 128 * 4)};
  void *scratch; CK(hipMalloc(&scratch, fold_bf16x3_scratch_bytes(d)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char *name, auto &&launch) {
    for (int i = 0; i < 3; i++) launch();
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 50; i++) launch();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    printf("%-72s %.2f us\n", name, ms * 1e3 / 50);
  };
  long long *dbg; CK(hipMalloc(&dbg, 1024 * 8 * 8)); CK(hipMemset(dbg, 0, 1024 * 8 * 8));
  g_fold3_dbg = dbg;
  unsigned short *a3 = static_cast<unsigned short *>(scratch);
  const size_t apl = (size_t)4 * C * R, bpl = (size_t)C * R;
  unsigned short *b3 = a3 + 3 * apl;
  Split3Args s;
  s.src[0] = wr; s.src[1] = wmT; s.dst[0] = a3; s.dst[1] = b3; s.plane[0] = apl; s.plane[1] = bpl; s.n8[0] = apl / 8; s.n8[1] = bpl / 8; s.mode = 1;
  for (unsigned g : {512u, 1024u, 2048u})  {
    char nm[96]; snprintf(nm, sizeof nm, "k_split3, %u workgroups", g);
    time(nm, [&]() { hipLaunchKernelGGL(k_split3, dim3(g), dim3(256), 0, st, s); });
  }
  time("launch_fold_bf16x3 (split + product, as the engine runs it)", [&]() { CK(launch_fold_bf16x3(d, 1, wr, wmT, scratch, pk, nch1, nch2, st, {}, {})); });
  auto anatomy = [&](int nwg) {
    std::vector<long long> h(1024 * 8);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double m[4] = {0, 0, 0, 0}; long long w0 = -1, w1 = 0; int n = 0;
    for (int b = 0; b < nwg; b++) {
      const long long *q = &h[(size_t)b * 8];
      if (!q[3]) continue;
      for (int i = 0; i < 4; i++) m[i] += q[i];
      if (w0 < 0 || q[4] < w0) w0 = q[4];
      if (q[4] + q[3] > w1) w1 = q[4] + q[3];
      n++;
    }
    printf("      mean over %d workgroups: prologue %.0f clk, K loop %.0f clk, epilogue %.0f clk, workgroup %.2f us; first entry -> last exit %.2f us\n",
           n, m[0] / n, m[1] / n, m[2] / n, m[3] / n / 100.0, (w1 - w0) / 100.0);
    CK(hipMemset(dbg, 0, 1024 * 8 * 8));
  };
  Fold3Args a;
  a.dbg = dbg;
  a.C = C; a.R = R; a.a3 = a3; a.b3 = b3; a.a_plane = apl; a.b_plane = bpl;
  a.wr = wr; a.wmT = wmT; a.redo = nullptr; a.wl = nullptr;      // (range guard: no event counter here; logical-row bf16 output: the many-stream launch only)
  a.pk1 = reinterpret_cast<float4 *>(pk[0]); a.nch1 = nch1; a.pk2 = reinterpret_cast<float4 *>(pk[1]); a.nch2 = nch2;
#define VAR(MI, NI, NB, ND) VARL(MI, NI, NB, ND, false)
#define VARL(MI, NI, NB, ND, LW) VARP(MI, NI, NB, ND, LW, 3)
#define VARP(MI, NI, NB, ND, LW, NPL) do { \
    a.nbn = cdv(C, 32 * NI); a.nwg = cdv(4 * C, 32 * MI) * a.nbn; \
    const int shm = NB * NPL * (32 * MI + 32 * NI) * 64; \
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fold_bf16x3<MI, NI, NB, ND, LW, NPL>), hipFuncAttributeMaxDynamicSharedMemorySize, shm)); \
    char nm[128]; snprintf(nm, sizeof nm, "%s product %dx%d tiles, %d buffers (%d KB LDS), %d workgroups%s", NPL == 2 ? "fp16x2" : "bf16x3", 32 * MI, 32 * NI, NB, shm / 1024, a.nwg, ND ? ", NO refills" : LW ? ", 4 loader waves" : ""); \
    time(nm, [&]() { hipLaunchKernelGGL((k_fold_bf16x3<MI, NI, NB, ND, LW, NPL>), dim3((a.nwg + 7) / 8 * 8), dim3(LW ? 512 : 256), shm, st, a); }); \
    anatomy((a.nwg + 7) / 8 * 8); } while (0)
  VAR(4, 3, 2, false);
  VAR(4, 3, 3, false);
  VAR(4, 3, 3, true);
  VARL(4, 3, 3, false, true);
  s.mode = 2; hipLaunchKernelGGL(k_split3, dim3(1024), dim3(256), 0, st, s); CK(hipStreamSynchronize(st));
  VARP(4, 3, 3, false, true, 2);
  VARP(4, 3, 4, false, true, 2);
  VARP(4, 4, 3, false, true, 2);
  VARP(4, 3, 3, true, true, 2);
  s.mode = 1; hipLaunchKernelGGL(k_split3, dim3(1024), dim3(256), 0, st, s); CK(hipStreamSynchronize(st));
  VARL(2, 5, 3, false, true);
  VARL(2, 5, 2, false, true);
  VARL(3, 3, 3, false, true);
  return 0;
}
