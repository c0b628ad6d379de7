// Does hipExtStreamCreateWithCUMask give a latency-critical single-workgroup kernel its own CUs while a chip-filling kernel runs?
// Stream A: a spin kernel of 512 workgroups x 150 KB LDS (one per CU at a time), mask = all CUs but the reserved ones (or no mask);
// stream B: a chain of 20 dependent single-workgroup kernels needing 134 KB LDS each, launched while A runs.
//   hipcc --offload-arch=gfx950 -O3 -o cumask.bin cu_mask_probe.hip && ./cumask.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(long long cycles, int* sink) {
  extern __shared__ int l[];
  l[threadIdx.x] = threadIdx.x;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (l[(threadIdx.x + 1) & 255] == -1) sink[0] = 1;
}
__global__ void small(long long cycles, int* sink) {
  extern __shared__ int l[];
  l[threadIdx.x] = threadIdx.x;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(2);
  if (l[(threadIdx.x + 1) & 255] == -1) sink[0] = 1;
}

int main() {
  int* sink; CK(hipMalloc(&sink, 4));
  CK(hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)small, hipFuncAttributeMaxDynamicSharedMemorySize, 134 * 1024));
  for (int reserve = 0; reserve <= 16; reserve = reserve ? reserve * 2 : 8) {
    for (int layout = 0; layout < 2; layout++) {
      if (!reserve && layout) continue;
      // 256 CUs -> 8 mask words; layout 0: the reserved CUs are the top bits; layout 1: one CU in every 256 / reserve
      std::vector<uint32_t> ma(8, 0xFFFFFFFFu), mb(8, 0u);
      for (int i = 0; i < reserve; i++) {
        const int cu = layout == 0 ? 255 - i : i * (256 / reserve);
        ma[cu / 32] &= ~(1u << (cu % 32));
        mb[cu / 32] |= 1u << (cu % 32);
      }
      hipStream_t a, b;
      if (reserve) {
        CK(hipExtStreamCreateWithCUMask(&a, 8, ma.data()));
        CK(hipExtStreamCreateWithCUMask(&b, 8, mb.data()));
      } else {
        CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, -1));
      }
      hipEvent_t a0, a1, b0, b1;
      CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a0, a));
      hipLaunchKernelGGL(spin, dim3(512), dim3(256), 150 * 1024, a, 100000 /* 1 ms at 100 MHz */, sink);
      CK(hipEventRecord(a1, a));
      CK(hipEventRecord(b0, b));
      for (int i = 0; i < 20; i++) hipLaunchKernelGGL(small, dim3(1), dim3(256), 134 * 1024, b, 1000 /* 10 us */, sink);
      CK(hipEventRecord(b1, b));
      CK(hipDeviceSynchronize());
      float ta, tb, tab;
      CK(hipEventElapsedTime(&ta, a0, a1)); CK(hipEventElapsedTime(&tb, b0, b1)); CK(hipEventElapsedTime(&tab, a0, b1));
      printf("reserved CUs %2d (%s): chip-filling kernel (2 rounds of 1 ms) %.3f ms; chain of 20 x 10 us single-workgroup kernels %.3f ms, done %.3f ms after the start\n",
             reserve, !reserve ? "no masks, chain on a high-priority stream" : layout == 0 ? "top mask bits" : "evenly spread mask bits", ta, tb, tab);
      CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b));
    }
  }
  return 0;
}
