// tools/skinny_probe.hip -- where does k_skinny_nn (klstm_fold.hip: in_diff of the output layer, 80 x 512 x 16624) spend its time?
// Per wave: shader clocks entry -> all loads issued -> MFMAs done -> exit.
#include "../kaldi-lstm_amd/csrc/klstm_kernels.hip"
#include "../kaldi-lstm_amd/csrc/klstm_fold3.hip"
#define KLSTM_SKINNY_TIMING
#include "../kaldi-lstm_amd/csrc/klstm_fold.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
using namespace klstm;
int main() {
  const int M = 80, N = 512, K = 16624;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  auto dalloc = [&](size_t n) { float *p; CK(hipMalloc(&p, n * 4)); std::vector<float> h(n); for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f; CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p; };
  float *A = dalloc((size_t)M * K), *B = dalloc((size_t)K * N), *Cm = dalloc((size_t)M * N), *ws = dalloc(skinny_nn_workspace_floats(M, N, K));
  long long *dbg; CK(hipMalloc(&dbg, 1024 * 4 * 8 * 8)); CK(hipMemset(dbg, 0, 1024 * 4 * 8 * 8));
  g_skinny_dbg = dbg;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) CK(launch_skinny_nn(M, N, K, A, K, B, N, Cm, N, ws, st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < 20; i++) CK(launch_skinny_nn(M, N, K, A, K, B, N, Cm, N, ws, st));
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("pair of launches: %.2f us\n", ms * 1e3 / 20);
  std::vector<long long> h(1024 * 4 * 8);
  CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
  double m[4] = {0, 0, 0, 0}; long long w0 = -1, w1 = 0; int n = 0;
  for (int b = 0; b < 1024 * 4; b++) {
    const long long *q = &h[(size_t)b * 8];
    if (!q[3]) continue;
    for (int i = 0; i < 4; i++) m[i] += q[i];
    if (w0 < 0 || q[4] < w0) w0 = q[4];
    if (q[4] + q[3] > w1) w1 = q[4] + q[3];
    n++;
  }
  printf("mean over %d waves: loads issued %.0f clk, MFMAs done +%.0f clk, reduction + stores +%.0f clk, wave %.2f us; first entry -> last exit %.2f us\n",
         n, m[0] / n, m[1] / n, m[2] / n, m[3] / n / 100.0, (w1 - w0) / 100.0);
  return 0;
}
