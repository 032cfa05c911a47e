// tools/kgroup_probe.hip -- the two k-group sums of the 4-row MFMA geometry (ds_bpermute rounds vs v_permlane16/32_swap)
// against a host sum, and the quad-permute stream hand-over of the backward sweep.  Prints the worst deviations.
#include "../kaldi-lstm_amd/csrc/klstm_persist_dev.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
using namespace klstm;
__global__ void k(const float *in, float *o1, float *o2, float *o3) {
  const int lane = threadIdx.x;
  f32x4 v = {in[lane * 4], in[lane * 4 + 1], in[lane * 4 + 2], in[lane * 4 + 3]};
  const f32x4 a = kgroup_sum(v), b = kgroup_sum_pl(v);
  for (int e = 0; e < 4; e++) { o1[lane * 4 + e] = a[e]; o2[lane * 4 + e] = b[e]; }
  // quad permutes: [0,0,2,2] and [1,1,3,3]
  const unsigned u = __float_as_uint(in[lane * 4]);
  o3[lane * 2] = __int_as_float(__builtin_amdgcn_update_dpp(0, (int)u, 0xA0, 0xf, 0xf, true));
  o3[lane * 2 + 1] = __int_as_float(__builtin_amdgcn_update_dpp(0, (int)u, 0xF5, 0xf, 0xf, true));
}
int main() {
  float h[256], *d, *o1, *o2, *o3, r1[256], r2[256], r3[128];
  for (int i = 0; i < 256; i++) h[i] = rand() / (float)RAND_MAX - 0.5f;
  hipMalloc(&d, 1024); hipMalloc(&o1, 1024); hipMalloc(&o2, 1024); hipMalloc(&o3, 512);
  hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o1, o2, o3);
  hipMemcpy(r1, o1, 1024, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 1024, hipMemcpyDeviceToHost); hipMemcpy(r3, o3, 512, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0; int bad = 0;
  for (int j = 0; j < 4; j++) for (int e = 0; e < 4; e++) {
    double s = 0; for (int b = 0; b < 16; b++) s += h[(4 * b + j) * 4 + e];
    for (int row = 0; row < 4; row++) {                       // lanes 12..15 of every row hold the totals
      const int lane = 16 * row + 12 + j;
      e1 = fmax(e1, fabs(r1[lane * 4 + e] - s)); e2 = fmax(e2, fabs(r2[lane * 4 + e] - s));
    }
  }
  for (int l = 0; l < 64; l++) {
    const int q = l & ~3, j = l & 3;
    if (r3[l * 2] != h[(q + (j & 2)) * 4]) bad++;
    if (r3[l * 2 + 1] != h[(q + (j & 2) + 1) * 4]) bad++;
  }
  printf("kgroup_sum max err %.3g, kgroup_sum_pl max err %.3g (lanes 12..15 of all rows), quad-permute mismatches %d\n", e1, e2, bad);
  return (e1 > 1e-5 || e2 > 1e-5 || bad) ? 1 : 0;
}
