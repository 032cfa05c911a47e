// tools/step_anatomy.hip -- what costs time inside a small dependent "step" kernel on MI355X?
// Adds ingredients one at a time to a 200-WG kernel in a 200-launch hipGraph chain. Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE bits: 1 = weight loads (per-WG distinct, WBYTES per WG), 2 = activation loads written by the previous kernel,
//            4 = MFMAs, 8 = LDS combine + barrier, 16 = transcendental epilogue + scattered stores, 32 = write act for next
template <int MODE, int NWAVES, int CH>
__global__ __launch_bounds__(NWAVES * 64) void k_step(const float *__restrict__ w, const float *__restrict__ act_in,
                                                     float *__restrict__ act_out, float *__restrict__ sink, int wstride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *wp = w + (size_t)blockIdx.x * wstride + (size_t)(wave * CH) * 512 + lane * 8;
  float av[CH][8], bv[CH][8];
#pragma unroll
  for (int c = 0; c < CH; c++) {
    if (MODE & 1) {
      const float4 a = *reinterpret_cast<const float4 *>(wp + c * 512), b = *reinterpret_cast<const float4 *>(wp + c * 512 + 4);
      av[c][0] = a.x; av[c][1] = a.y; av[c][2] = a.z; av[c][3] = a.w; av[c][4] = b.x; av[c][5] = b.y; av[c][6] = b.z; av[c][7] = b.w;
    } else { for (int j = 0; j < 8; j++) av[c][j] = 1.f + lane; }
    if (MODE & 2) {
      const float *ap = act_in + ((lane & 3) * 512 + (wave * CH + c) * 32 + (lane >> 4) * 8) % 2048;
      float4 a, b;
      if (MODE & 128) {
        const f32x4 ta = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(ap)), tb = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(ap + 4));
        a = make_float4(ta.x, ta.y, ta.z, ta.w); b = make_float4(tb.x, tb.y, tb.z, tb.w);
      }
      else { a = *reinterpret_cast<const float4 *>(ap); b = *reinterpret_cast<const float4 *>(ap + 4); }
      bv[c][0] = a.x; bv[c][1] = a.y; bv[c][2] = a.z; bv[c][3] = a.w; bv[c][4] = b.x; bv[c][5] = b.y; bv[c][6] = b.z; bv[c][7] = b.w;
    } else { for (int j = 0; j < 8; j++) bv[c][j] = 0.5f; }
  }
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  if (MODE & 4) {
#pragma unroll
    for (int c = 0; c < CH; c++)
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c][j], bv[c][j], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c][j + 1], bv[c][j + 1], acc1, 0, 0, 0);
      }
  } else {
#pragma unroll
    for (int c = 0; c < CH; c++) for (int j = 0; j < 8; j++) { acc0.x += av[c][j] * bv[c][j]; }
  }
  f32x4 v = acc0 + acc1;
  if (MODE & 8) {
    __shared__ f32x4 red[NWAVES][64];
    red[wave][lane] = v;
    __syncthreads();
    if (wave == 0) { for (int ww = 1; ww < NWAVES; ww++) v += red[ww][lane]; }
  }
  if (wave == 0) {
    if (MODE & 16) {
      float x = v.x + v.y;
      x = 1.f / (1.f + expf(-x)); x = x * (1.f / (1.f + expf(-v.z))); x = 1.f - 2.f / (1.f + expf(2.f * x));
      if (lane < 16) { sink[(size_t)(lane & 3) * 3200 + blockIdx.x * 4 + (lane >> 2)] = x; sink[20000 + (size_t)(lane & 3) * 800 + blockIdx.x * 4 + (lane >> 2)] = x; }
    } else if (lane == 0) sink[blockIdx.x] = v.x + v.y + v.z + v.w;
    if ((MODE & 32) && lane < 16) {
      float *dst = act_out + ((lane & 3) * 512 + blockIdx.x * 4 + (lane >> 2)) % 2048;
      if (MODE & 64) __builtin_nontemporal_store(v.x * 1e-9f, dst); else *dst = v.x * 1e-9f;
    }
  }
}

template <class F> float time_chain(hipStream_t st, int n, F launch) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipGraph_t g; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; i++) launch(i);
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
  float best = 1e9;
  for (int rep = 0; rep < 6; rep++) {
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(exec, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(exec));
  return best * 1e3f / n;
}

template <int MODE, int NWAVES, int CH> void run(hipStream_t st, const char *name, float *w, float *a0, float *a1, float *sink, int grid, int wstride) {
  float t = time_chain(st, 200, [&](int i) {
    const float *ain = (MODE & 256) ? (const float *)(w + (size_t)255 * 16 * 1024) : (const float *)(i & 1 ? a1 : a0);
    hipLaunchKernelGGL((k_step<MODE, NWAVES, CH>), dim3(grid), dim3(NWAVES * 64), 0, st, (const float *)w, ain, i & 1 ? a0 : a1, sink, wstride);
  });
  printf("%-58s grid=%3d waves=%d ch/wave=%d : %.2f us/kernel\n", name, grid, NWAVES, CH, t);
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *w, *a0, *a1, *sink;
  const size_t wn = (size_t)256 * 16 * 1024;   // 16 MB of "weights"
  CK(hipMalloc(&w, wn * 4)); CK(hipMalloc(&a0, 2048 * 4)); CK(hipMalloc(&a1, 2048 * 4)); CK(hipMalloc(&sink, 1 << 20));
  CK(hipMemset(w, 0, wn * 4)); CK(hipMemset(a0, 0, 2048 * 4)); CK(hipMemset(a1, 0, 2048 * 4));
  const int WS = 16 * 512;   // per-WG weight slice: 16 rows x 512 = 32 KB
  run<0, 8, 2>(st, "nothing (8 waves)", w, a0, a1, sink, 200, WS);
  run<8, 8, 2>(st, "LDS combine + barrier", w, a0, a1, sink, 200, WS);
  run<8 | 32, 8, 2>(st, "LDS + act store only", w, a0, a1, sink, 200, WS);
  run<2 | 8, 8, 2>(st, "act loads (buffer written 2 kernels ago) + LDS", w, a0, a1, sink, 200, WS);
  run<2 | 8 | 256, 8, 2>(st, "act loads from a never-written buffer + LDS", w, a0, a1, sink, 200, WS);
  run<2 | 8 | 32, 8, 2>(st, "act loads + LDS + act store", w, a0, a1, sink, 200, WS);
  run<2 | 8 | 32 | 64, 8, 2>(st, "act loads + LDS + nontemporal act store", w, a0, a1, sink, 200, WS);
  run<2 | 8 | 32 | 128, 8, 2>(st, "nontemporal act loads + LDS + act store", w, a0, a1, sink, 200, WS);
  run<2 | 8 | 32 | 64 | 128, 8, 2>(st, "nt loads + LDS + nt store", w, a0, a1, sink, 200, WS);
  run<2 | 8 | 32, 8, 2>(st, "act loads + LDS + act store, grid 32", w, a0, a1, sink, 32, WS);
  run<2 | 8 | 32, 1, 2>(st, "act loads + act store, 1 wave, grid 32", w, a0, a1, sink, 32, WS);
  run<1 | 2 | 4 | 8 | 16 | 32, 8, 2>(st, "full", w, a0, a1, sink, 200, WS);
  run<1 | 2 | 4 | 8 | 16 | 32 | 64 | 128, 8, 2>(st, "full, nt act ld/st", w, a0, a1, sink, 200, WS);
  return 0;
}
