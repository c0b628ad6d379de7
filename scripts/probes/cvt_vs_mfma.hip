// Does v_cvt_f64_f32 (used to widen staged activations) take fp64-MFMA pipe time on gfx950?  4 MFMA waves + 4 partner
// waves per CU; the partners run (mode 1) v_cvt_f64_f32, (mode 2) 32-bit integer ALU, (mode 3) ds_write_b128.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512, 2) void k(int iters, int mode, int per_iter, double* sink) {
  __shared__ double lds[8192];
  const int wave = threadIdx.x >> 6;
  double s = 0.;
  if (threadIdx.x == 0) *(volatile int*)(lds + 8000) = 0;
  __syncthreads();
  if (wave < 4) {
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
    __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) *(volatile int*)(lds + 8000) = 1;
  } else if (mode > 0) {
    long long count = 0;
    float f = threadIdx.x * 0.5f;
    unsigned u = threadIdx.x;
    double d = 0.;
    volatile int* flag = (volatile int*)(lds + 8000);
    while (*flag == 0) {                   // run until the MFMA waves are done: per_iter instructions, then a nap
      for (int j = 0; j < per_iter; j++) {
        if (mode == 1) { double t; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(t) : "v"(f)); d += 0; s = t; }
        else if (mode == 2) { asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u) : "v"(u)); }
        else { *(d2*)(lds + (threadIdx.x & 255) * 2) = (d2){d, d}; asm volatile("" ::: "memory"); }
      }
      count += per_iter;
    }
    if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long*)(sink + 1), (unsigned long long)count);
    s += u + d;
  }
  if (s == 123.456) sink[0] = s;
}
int main() {
  double* sink; (void)hipMalloc((void**)&sink, 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4096;
  for (int mode = 0; mode < 4; mode++)
    for (int per : {8, 32, 128}) {
      if (mode == 0 && per != 8) continue;
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, iters, mode, per, sink);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      unsigned long long cnt = 0;
      (void)hipMemcpy(&cnt, sink + 1, 8, hipMemcpyDeviceToHost);
      (void)hipMemset(sink + 1, 0, 8);
      double per_wave = (double)cnt / 3 / (256.0 * 4);          // partner instructions per partner wave per launch
      double mfma_per_wave = (double)iters * 16;
      printf("mode %d burst %3d: %.3f ms  mfma %.1f TF  partner instrs per MFMA of the SIMD's consumer: %.2f\n", mode, per,
             best, 256.0 * 4 * iters * 16 * 2048 / best / 1e9, per_wave / mfma_per_wave);
    }
  return 0;
}
