// Can v_fma_f64 (VALU) and v_mfma_f64_16x16x4_f64 (matrix pipe) run concurrently on gfx950, i.e. is there fp64
// throughput beyond the 78.6 TF matrix peak?  mode 0: MFMA waves only, 1: VALU waves only, 2: both (2 + 2 per SIMD...).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 1) void k(int iters, int mode, double* sink) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (mode == 0) || (mode == 2 && wave < 4);
  const bool do_valu = (mode == 1) || (mode == 2 && wave >= 4);
  double s = 0.;
  if (do_mfma) {
    d4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = (d4){0., 0., 0., 0.};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  }
  if (do_valu) {
    double x[32];
#pragma unroll
    for (int i = 0; i < 32; i++) x[i] = threadIdx.x * 1e-3 + i;
    double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 8; r++)  // 8 x 32 = 256 FMA wave-instructions per iteration = 1024 cycles at 4 cyc each
#pragma unroll
        for (int i = 0; i < 32; i++) x[i] = __builtin_fma(x[i], a, b);
    }
#pragma unroll
    for (int i = 0; i < 32; i++) s += x[i];
  }
  if (s == 123.456) sink[0] = s;
}

int main() {
  double* sink; hipMalloc((void**)&sink, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256, iters = 4096;
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, iters, mode, sink);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double mf = 0, vf = 0;
    int mw = mode == 0 ? 8 : (mode == 2 ? 4 : 0), vw = mode == 1 ? 8 : (mode == 2 ? 4 : 0);
    mf = (double)blocks * mw * iters * 16 * 2048.0;
    vf = (double)blocks * vw * iters * 256 * 128.0;  // 64 lanes * 2 flop
    printf("mode %d: %.3f ms  mfma %.1f TF  valu %.1f TF  total %.1f TF\n", mode, best, mf / best / 1e9, vf / best / 1e9,
           (mf + vf) / best / 1e9);
  }
  return 0;
}
