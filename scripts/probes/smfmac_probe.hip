// v_smfmac_i32_32x32x64_i8 (gfx950 2:4 structured-sparse int8 MFMA): (1) dumps one wave's raw operands and result so that the
// operand layout can be recovered on the host (scripts/probes/smfmac_layout.py), (2) measures the issue rate of sparse against
// dense MFMAs from registers, operands changing every instruction (random bytes / all zero), two waves per SIMD on every CU.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/smfmac_probe.hip -o scripts/probes/smfmac_probe.bin && ./smfmac_probe.bin out.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void one_wave(const i32x4* a, const i32x8* b, const int* idx, i32x16* o) {
  i32x16 c = {};
  c = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a[threadIdx.x], b[threadIdx.x], c, idx[threadIdx.x], 0, 0);
  o[threadIdx.x] = c;
}

__device__ __forceinline__ unsigned hash(unsigned s) { s ^= s >> 16; s *= 0x7feb352du; s ^= s >> 15; s *= 0x846ca68bu; s ^= s >> 16; return s; }

template <int SPARSE>
__global__ __launch_bounds__(256, 2) void rate(int iters, int random, int* sink) {
  i32x4 a[4];
  i32x8 b[4];
  int ix[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) a[i][j] = random ? (int)hash((blockIdx.x * 256 + threadIdx.x) * 64 + i * 8 + j) : 0;
#pragma unroll
    for (int j = 0; j < 8; j++) b[i][j] = random ? (int)hash((blockIdx.x * 256 + threadIdx.x) * 64 + 1000003 + i * 8 + j) : 0;
    ix[i] = 0x4E4E4E4E ^ (random ? (int)(hash(threadIdx.x + i) & 0x11111111u) : 0);   // valid ordered index pairs (0,1)/(1,3)... any bits work for timing
  }
  i32x16 c[8];
#pragma unroll
  for (int i = 0; i < 8; i++) c[i] = (i32x16)0;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (SPARSE)
        c[i] = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a[i & 3], b[(i + (i >> 2)) & 3], c[i], ix[i & 3], 0, 0);
      else {
        const i32x8 bb = b[(i + (i >> 2)) & 3];
        c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 3], (i32x4){bb[0], bb[1], bb[2], bb[3]}, c[i], 0, 0, 0);
      }
    }
  int x = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 16; j++) x ^= c[i][j];
  if (x == 0x12345678) sink[0] = x;
}

int main(int argc, char** argv) {
  // ---- (1) layout dump: A compressed values = distinct small primes-ish per (lane, byte), idx random valid pairs, B random
  std::vector<int> ha(64 * 4), hb(64 * 8), hi(64), ho(64 * 16);
  srand(12345);
  for (auto& v : ha) { unsigned w = 0; for (int k = 0; k < 4; k++) w |= (unsigned)((rand() % 15 + 1) & 255) << (8 * k); v = (int)w; }
  for (auto& v : hb) { unsigned w = 0; for (int k = 0; k < 4; k++) w |= (unsigned)((rand() % 13 - 6) & 255) << (8 * k); v = (int)w; }
  for (auto& v : hi) { unsigned w = 0; for (int g = 0; g < 8; g++) { int p = rand() % 3, q = p + 1 + rand() % (3 - p); w |= (unsigned)(p | (q << 2)) << (4 * g); } v = (int)w; }
  int *da, *db, *di, *dout;
  CK(hipMalloc(&da, ha.size() * 4)); CK(hipMalloc(&db, hb.size() * 4)); CK(hipMalloc(&di, hi.size() * 4)); CK(hipMalloc(&dout, ho.size() * 4));
  CK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(di, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(one_wave, dim3(1), dim3(64), 0, 0, (const i32x4*)da, (const i32x8*)db, di, (i32x16*)dout);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
  FILE* f = fopen(argc > 1 ? argv[1] : "smfmac_dump.bin", "wb");
  fwrite(ha.data(), 4, ha.size(), f); fwrite(hb.data(), 4, hb.size(), f); fwrite(hi.data(), 4, hi.size(), f); fwrite(ho.data(), 4, ho.size(), f);
  fclose(f);
  printf("dumped one wave: A %zu, B %zu, idx %zu, D %zu dwords\n", ha.size(), hb.size(), hi.size(), ho.size());
  // ---- (2) rates
  int n_cu = 0;
  CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
  int* sink; CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = n_cu * 2, iters = 100000;
  for (int sparse = 0; sparse < 2; sparse++)
    for (int random = 0; random < 2; random++) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0, 0));
        if (sparse) hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, iters, random, sink);
        else hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, iters, random, sink);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double inst = (double)blocks * 4 * iters * 8;
      const double logical = inst * 2.0 * 32 * 32 * (sparse ? 64 : 32);
      printf("%s MFMA, %s operands: %.3f ms, %.2f G instr/s, %.0f logical TOP/s (dense-equivalent K = %d per instruction)\n",
             sparse ? "sparse 32x32x64" : "dense  32x32x32", random ? "random" : "zero  ", ms, inst / ms / 1e6, logical / ms / 1e9, sparse ? 64 : 32);
    }
  return 0;
}
