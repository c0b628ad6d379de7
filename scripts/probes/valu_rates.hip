// Issue rates of the VALU instructions the remainder kernels are made of (gfx950): cycles per wave instruction, one wave per SIMD
// two and four, wall clock of 8192 x 64 instructions per wave (four independent chains).    hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
template <int MODE>
__global__ void rate_kernel(unsigned long long* out, double seed, int iters) {
  double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
  unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
  int s0 = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) { REP16(asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (MODE == 1) { REP16(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));) }
    if (MODE == 2) { REP16(asm volatile("v_lshlrev_b32 %0, 16, %0\n v_lshlrev_b32 %1, 16, %1\n v_and_b32 %2, 0xffff0000, %2\n v_and_b32 %3, 0xffff0000, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
    if (MODE == 3) { REP16(asm volatile("v_readlane_b32 %0, %1, 3\n v_readlane_b32 %0, %2, 5\n v_readlane_b32 %0, %3, 7\n v_readlane_b32 %0, %4, 9" : "=s"(s0) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));) }
    if (MODE == 4) { REP16(asm volatile("v_mul_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (MODE == 5) { REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
    if (MODE == 6) { REP16(asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));) }
    if (MODE == 7) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %6, %1\n v_mad_u64_u32 %2, vcc, %6, %7, %2\n v_mad_u64_u32 %3, vcc, %7, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3) : "vcc");) }
    if (MODE == 8) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (MODE == 9) { REP16(asm volatile("v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %4\n v_ldexp_f64 %2, %2, %4\n v_ldexp_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0));) }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + u0 + u1 + u2 + u3 + s0 == 12345.678) out[0] = 0;
}
template <int MODE> void run(const char* name, unsigned long long* d) {
  const int iters = 8192;
  for (int waves : {4, 8, 16}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(64 * waves), 0, 0, d, 1.0000001, 16);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(64 * waves), 0, 0, d, 1.0000001, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)(waves / 4) * iters * 64;     // wave instructions one SIMD issued
    printf("%-26s %d wave(s) per SIMD: %6.2f ns per wave instruction and SIMD  (%.3f ms)\n", name, waves / 4, ms * 1e6 / per_simd, ms);
  }
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 256 * 16 * 8);
  run<0>("v_fma_f64", d); run<4>("v_mul_f64 / v_add_f64", d); run<1>("v_cvt_f64_f32", d); run<6>("v_cvt_f64_i32", d); run<2>("v_lshlrev / v_and b32", d);
  run<3>("v_readlane_b32", d); run<5>("v_fma_f32", d); run<8>("v_pk_fma_f32", d); run<7>("v_mad_u64_u32", d); run<9>("v_ldexp_f64", d);
  return 0;
}
