// What the int8 matrix pipes sustain under the board's power cap when the OPERANDS CHANGE from one MFMA to the next, as they do
// in a real product kernel (i8_mfma_peak.hip feeds every MFMA the same two registers: nothing toggles and the pipe runs at
// full clock).  Register-only loops -- no LDS, no memory -- so what is measured is the MFMA array + register file alone:
// the ceiling any i8 kernel on random data can reach.  Both shapes, 1 and 2 waves per SIMD, ~300 ms each (long enough for the
// power management to settle), random bytes vs all-zero operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o i8power.bin i8_mfma_power.hip && ./i8power.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int SHAPE, int REUSE>
__global__ __launch_bounds__(256) void spin(int iters, int* sink, int random) {
  i32x4 a[4], b[4];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      const unsigned s = (blockIdx.x * 256 + threadIdx.x) * 64 + i * 8 + j;
      a[i][j] = random ? (int)mix(s) : 0;
      b[i][j] = random ? (int)mix(s + 4) : 0;
    }
  int x = 0;
  if (SHAPE == 32) {
    i32x16 c[8];
    for (int i = 0; i < 8; i++) c[i] = (i32x16)0;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 8; i++)
        c[i] = REUSE == 0 ? __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 3], b[(i + (i >> 2)) & 3], c[i], 0, 0, 0)
             : REUSE == 1 ? __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i >> 2], b[i & 3], c[i], 0, 0, 0)            // A kept for 4 MFMAs in a row
                          : __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i >> 2], b[(i >> 1) & 1], c[i], 0, 0, 0);     // A for 4, B for 2
    for (int i = 0; i < 8; i++)
      for (int r = 0; r < 16; r++) x ^= c[i][r];
  } else {
    i32x4 c[16];
    for (int i = 0; i < 16; i++) c[i] = (i32x4)0;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 16; i++) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 3], b[(i + (i >> 2)) & 3], c[i], 0, 0, 0);
    for (int i = 0; i < 16; i++)
      for (int r = 0; r < 4; r++) x ^= c[i][r];
  }
  if (x == 0x12345678) sink[0] = x;
}

template <int SHAPE, int REUSE = 0>
static void run(int wg_per_cu, int random) {
  int* sink; hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 300000, grid = 256 * wg_per_cu;
  hipLaunchKernelGGL((spin<SHAPE, REUSE>), dim3(grid), dim3(256), 0, 0, 20000, sink, random);
  hipEventRecord(e0);
  hipLaunchKernelGGL((spin<SHAPE, REUSE>), dim3(grid), dim3(256), 0, 0, iters, sink, random);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per = SHAPE == 32 ? 8 * 2.0 * 32 * 32 * 32 : 16 * 2.0 * 16 * 16 * 64;
  printf("v_mfma_i32_%s, %d wave(s)/SIMD, operands %s and %s: %.1f ms, %.0f TOP/s = %.2f of 5000\n",
         SHAPE == 32 ? "32x32x32_i8" : "16x16x64_i8", wg_per_cu, random ? "random bytes" : "all zero",
         REUSE == 0 ? "changing every MFMA" : REUSE == 1 ? "A kept for 4 MFMAs in a row" : "A kept for 4, B for 2 MFMAs in a row", ms,
         per * iters * 4.0 * grid / ms / 1e9, per * iters * 4.0 * grid / ms / 1e9 / 5000.);
  hipFree(sink);
}

int main() {
  run<32>(1, 0); run<32>(1, 1); run<32>(2, 1); run<32, 1>(2, 1); run<32, 2>(2, 1);
  run<16>(1, 0); run<16>(1, 1); run<16>(2, 1);
  return 0;
}
