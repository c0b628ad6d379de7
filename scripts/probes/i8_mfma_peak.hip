// What the int8 MFMA pipe sustains with nothing else in the way: register-only loops of v_mfma_i32_32x32x32_i8 and
// v_mfma_i32_16x16x64_i8, 1 / 2 waves per SIMD, a few hundred ms (long enough for the power management to settle).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o i8peak.bin i8_mfma_peak.hip && ./i8peak.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void spin(int iters, int* sink, int seed) {
  i32x4 a = {seed + (int)threadIdx.x, seed * 3, seed * 5, seed * 7}, b = {seed * 11, (int)threadIdx.x, seed, 1};
  int x = 0;
  if (SHAPE == 32) {
    i32x16 c[8];
    for (int i = 0; i < 8; i++) c[i] = (i32x16)0;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i], 0, 0, 0);
    for (int i = 0; i < 8; i++)
      for (int r = 0; r < 16; r++) x ^= c[i][r];
  } else {
    i32x4 c[16];
    for (int i = 0; i < 16; i++) c[i] = (i32x4)0;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 16; i++) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
    for (int i = 0; i < 16; i++)
      for (int r = 0; r < 4; r++) x ^= c[i][r];
  }
  if (x == 0x12345678) sink[0] = x;
}

template <int SHAPE>
static void run(int wg_per_cu, int data) {
  int* sink; hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 400000, grid = 256 * wg_per_cu;
  hipLaunchKernelGGL(spin<SHAPE>, dim3(grid), dim3(256), 0, 0, 1000, sink, data);
  hipEventRecord(e0);
  hipLaunchKernelGGL(spin<SHAPE>, dim3(grid), dim3(256), 0, 0, iters, sink, data);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per = SHAPE == 32 ? 8 * 2.0 * 32 * 32 * 32 : 16 * 2.0 * 16 * 16 * 64;
  printf("v_mfma_i32_%s, %d waves/SIMD, operands %s: %.1f ms, %.0f TOP/s\n", SHAPE == 32 ? "32x32x32_i8" : "16x16x64_i8", wg_per_cu,
         data ? "random bits" : "zero", ms, per * iters * 4.0 * grid / ms / 1e9);
}

int main() {
  run<32>(1, 0); run<32>(1, 0x5a17c3); run<32>(2, 0x5a17c3);
  run<16>(1, 0); run<16>(1, 0x5a17c3); run<16>(2, 0x5a17c3);
  return 0;
}
