// Feasibility probe for the digit-plane covariance of DESIGN.md section 7 ("Beyond the fp64 pipe"): how fast can ONE kernel
// form all 15 plane-pair products  C_k += A_s[I] * A_t[J]^T  (s + t = k <= 4)  of a 128x128 output tile from a single set of
// LDS fragment reads per k-step?  Planes are int8 [5][n][T] (feature-major, token-contiguous), accumulators int32.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o i8_planes_syrk.bin i8_planes_syrk.hip && ./i8_planes_syrk.bin
// Prints the verification of a small problem against the host, then the rate at n = 14336, T = 32768 (one Llama-3-8B
// calibration batch of sigma_mlp) in int8 TOP/s and as the equivalent fp64 SYRK time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

#ifndef S_PLANES
#define S_PLANES 5
#endif
constexpr int S = S_PLANES;   // digit planes (-DS_PLANES=6 for the 21-pair variant)
// Variants (all 64 x 32 wave tiles of v_mfma_i32_32x32x32_i8):
//   default                  4 waves, 128 x 64 tile, one workgroup per CU, 4-stage ring   (the first product kernel)
//   -DOCC=2 -DNBUF_=2        4 waves, 128 x 64 tile, two workgroups per CU, 2-stage ring   (the 5-plane product kernel now)
//   -DWAVES=8 -DNBUF_=3      8 waves (2 per SIMD), 128 x 128 tile, one workgroup of 512 threads, 3-stage ring: 40 KB per stage
//                            for 16384 outputs instead of 2 x 30 KB -- a third fewer L2->LDS bytes per MFMA
#ifndef WAVES
#define WAVES 4
#endif
#ifndef OCC
#define OCC 1
#endif
#ifndef WBLK
#define WBLK 2      // 32-row blocks per wave along I: 2 = 64 x 32 wave tile; 1 = 32 x 32 (8 waves as 4 x 2 over a 128 x 64 tile:
#endif              //   -DS_PLANES=6 -DWAVES=8 -DWBLK=1 -DNBUF_=4 -- 96 accumulators, two waves per SIMD for the 6-plane route)
#ifndef NBUF_
#define NBUF_ 4
#endif
constexpr int NW = WAVES;
constexpr int TILE = 128;     // workgroup tile: TILE rows of I x TJ rows of J (features)
constexpr int TJ = (NW == 8 && WBLK == 2) ? 128 : 64;
constexpr int WCOLS = TJ / 32;               // waves along J
constexpr int BK = 32;        // tokens per stage = one MFMA k-step
constexpr int PANEL_A = TILE * BK, PANEL_B = TJ * BK;   // bytes of one plane of an operand in a stage
#ifndef KSUB
#define KSUB 1      // MFMA k-steps (32 tokens) per LDS stage: -DKSUB=2 halves the number of barriers / LDS-DMA round trips per token
#endif
constexpr int SUBSTAGE = S * (PANEL_A + PANEL_B);       // 30 KB
constexpr int STAGE = KSUB * SUBSTAGE;
constexpr int NBUF = NBUF_;                  // LDS ring: NBUF - 1 stages in flight behind the one being consumed
constexpr int GA = TILE / 32, GB = TJ / 32;  // 32-row groups per operand
constexpr int PIECES = S * (GA + GB);

__host__ __device__ inline int tiles_of(int n) { return TJ == 64 ? (n / TILE) * (n / TILE + 1) : (n / TILE) * (n / TILE + 1) / 2; }

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l,
                                   16, 0, 0);
}

// planes: [S][n][ldt] int8;  out (optional): [tiles][S][TILE][TJ] int32, tile index = bi*(bi+1) + bj  (bj <= 2 bi + 1)
// 4 waves as 2 x 2, wave tile 64 x 32 = 2 x 1 MFMA blocks, 5 classes -> 160 accumulator registers (the 64 x 64 wave tile's
// 320 do not fit the 256 AGPRs and the compiler shuffles 200 registers per stage)
__global__ __launch_bounds__(64 * NW, OCC) void planes_syrk(const int8_t* __restrict__ planes, int n, int T, int64_t ldt, int* out,
                                                      int* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tile = blockIdx.x;
  int bi, bj;
  if (TJ == 64) {
    bi = (int)((sqrtf(4.f * tile + 1.f) - 1.f) * 0.5f);
    while ((bi + 1) * (bi + 2) <= tile) bi++;
    while (bi * (bi + 1) > tile) bi--;
    bj = tile - bi * (bi + 1);
  } else {
    bi = (int)((sqrtf(8.f * tile + 1.f) - 1.f) * 0.5f);
    while ((bi + 1) * (bi + 2) / 2 <= tile) bi++;
    while (bi * (bi + 1) / 2 > tile) bi--;
    bj = tile - bi * (bi + 1) / 2;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wr = wave / WCOLS, wc = wave % WCOLS;
  const int64_t plane_stride = (int64_t)n * ldt;

  // staging: 30 pieces of 1 KB per stage (A: 5 planes x 4 row groups, B: 5 planes x 2); wave w issues pieces w, w+4, ...
  // lane -> row g*32 + (lane & 31), 16-byte half (lane >> 5): a piece is [half][row][16 B] in LDS, so the 32 lanes that share
  // an MFMA k-half read 512 contiguous bytes
  auto issue_stage = [&](int kt, int buf) {
#pragma unroll
    for (int q = 0; q < (KSUB * PIECES + NW - 1) / NW; q++) {
      const int pu = wave + NW * q;
      const int u = pu / PIECES, p = pu % PIECES;
      if (pu < KSUB * PIECES) {
        const bool isA = p < GA * S;
        const int pp = isA ? p : p - GA * S;
        const int s = isA ? pp / GA : pp / GB, g = isA ? pp % GA : pp % GB;
        // blocked plane layout written by the split pass: [plane][row group of 32][k-step][half][row][16 B] -- a piece is
        // 1 KB contiguous in memory (8 full cache lines per wave instruction instead of 32 quarter-used ones)
        const int64_t G = (isA ? bi * (TILE / 32) : bj * (TJ / 32)) + g;
        const int8_t* src = planes + ((s * (int64_t)(n / 32) + G) * (T / BK) + kt * KSUB + u) * 1024 + lane * 16;
        unsigned char* dst = lds + buf * STAGE + u * SUBSTAGE + (isA ? s * PANEL_A : S * PANEL_A + s * PANEL_B) + g * 1024;
        glds16(src, dst);
      }
    }
  };

  i32x16 acc[S][WBLK];
#pragma unroll
  for (int k = 0; k < S; k++)
#pragma unroll
    for (int a = 0; a < WBLK; a++) acc[k][a] = (i32x16)0;

  const int nk = T / BK / KSUB;   // stages
  for (int p = 0; p < NBUF - 1 && p < nk; p++) issue_stage(p, p);
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt % NBUF;
    // waves 0,1 issue 8 glds per stage, waves 2,3 issue 7; stages kt+1, kt+2 may stay in flight: at most 14 outstanding
    // retires stage kt on every wave (the tail drains everything)
    if (kt + NBUF - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * (KSUB * PIECES / NW)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + NBUF - 1 < nk) issue_stage(kt + NBUF - 1, (kt + NBUF - 1) % NBUF);
#pragma unroll
    for (int u = 0; u < KSUB; u++) {
    const unsigned char* base = lds + buf * STAGE + u * SUBSTAGE;
    const int r = lane & 31, h = lane >> 5;
    i32x4 fa[S][WBLK], fb[S];
#pragma unroll
    for (int s = 0; s < S; s++) {
#pragma unroll
      for (int blk = 0; blk < WBLK; blk++)
        fa[s][blk] = *(const i32x4*)(base + s * PANEL_A + (wr * WBLK + blk) * 1024 + h * 512 + r * 16);
      fb[s] = *(const i32x4*)(base + S * PANEL_A + s * PANEL_B + wc * 1024 + h * 512 + r * 16);
    }
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
      for (int t = 0; t < S - s; t++)
#pragma unroll
        for (int a = 0; a < WBLK; a++)
          acc[s + t][a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s][a], fb[t], acc[s + t][a], 0, 0, 0);
    }
  }

  if (out) {
    int* o = out + (int64_t)tile * S * TILE * TJ;
#pragma unroll
    for (int k = 0; k < S; k++)
#pragma unroll
      for (int a = 0; a < WBLK; a++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
          const int row = wr * 32 * WBLK + a * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
          const int col = wc * 32 + (lane & 31);
          o[(k * TILE + row) * TJ + col] = acc[k][a][reg];
        }
  } else {
    int x = 0;
#pragma unroll
    for (int k = 0; k < S; k++)
#pragma unroll
      for (int a = 0; a < WBLK; a++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) x ^= acc[k][a][reg];
    if (x == 0x7fffffff) sink[0] = x;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  // ---- small verification
  {
    const int n = 256, T = 256;
    std::vector<int8_t> h((size_t)S * n * T);
    srand(1);
    for (auto& v : h) v = (int8_t)(rand() % 256 - 128);
    int8_t* d; int* out; int* sink;
    const int tiles = tiles_of(n);
    CK(hipMalloc(&d, h.size())); CK(hipMalloc(&out, (size_t)tiles * S * TILE * TJ * 4)); CK(hipMalloc(&sink, 4));
    {
      std::vector<int8_t> blk(h.size());
      const int nk = T / BK;
      for (int s_ = 0; s_ < S; s_++)
        for (int row = 0; row < n; row++)
          for (int t = 0; t < T; t++)
            blk[(((size_t)s_ * (n / 32) + row / 32) * nk + t / BK) * 1024 + ((t % BK) / 16) * 512 + (row % 32) * 16 + t % 16] =
                h[((size_t)s_ * n + row) * T + t];
      CK(hipMemcpy(d, blk.data(), blk.size(), hipMemcpyHostToDevice));
    }
    CK(hipFuncSetAttribute((const void*)planes_syrk, hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * STAGE));
    hipLaunchKernelGGL(planes_syrk, dim3(tiles), dim3(64 * NW), NBUF * STAGE, 0, d, n, T, (int64_t)T, out, sink);
    CK(hipDeviceSynchronize());
    std::vector<int> got((size_t)tiles * S * TILE * TJ);
    CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    for (int tile = 0; tile < tiles && bad < 5; tile++) {
      int bi = 0, bj;
      if (TJ == 64) { while ((bi + 1) * (bi + 2) <= tile) bi++; bj = tile - bi * (bi + 1); }
      else { while ((bi + 1) * (bi + 2) / 2 <= tile) bi++; bj = tile - bi * (bi + 1) / 2; }
      for (int k = 0; k < S; k++)
        for (int i = 0; i < TILE; i += 37)
          for (int j = 0; j < TJ; j += 29) {
            long ref = 0;
            for (int s = 0; s <= k; s++) {
              const int t = k - s;
              const int8_t* a = &h[((size_t)s * n + bi * TILE + i) * T];
              const int8_t* b = &h[((size_t)t * n + bj * TJ + j) * T];
              for (int x = 0; x < T; x++) ref += (int)a[x] * (int)b[x];
            }
            if ((int)ref != got[((size_t)tile * S + k) * TILE * TJ + i * TJ + j]) { bad++; if (bad < 5) printf("mismatch tile %d k %d (%d,%d): %ld vs %d\n", tile, k, i, j, ref, got[((size_t)tile * S + k) * TILE * TJ + i * TJ + j]); }
          }
    }
    printf("verification (n=%d, T=%d, 15 plane pairs in 5 classes): %s\n", n, T, bad ? "FAILED" : "ok");
    hipFree(d); hipFree(out); hipFree(sink);
    if (bad) return 2;
  }
  // ---- rate at one calibration batch of sigma_mlp
  {
#ifndef N_FEAT
#define N_FEAT 14336
#endif
    const int n = N_FEAT, T = 32768;
    int8_t* d; int* sink;
    const size_t bytes = (size_t)S * n * T;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4));
    // sparse-ish digits like real planes are not modelled: random bytes (worst case for power)
    std::vector<int8_t> h(1 << 24);
    for (auto& v : h) v = (int8_t)(rand() % 256 - 128);
    for (size_t off = 0; off < bytes; off += h.size()) CK(hipMemcpy(d + off, h.data(), std::min(h.size(), bytes - off), hipMemcpyHostToDevice));
    const int tiles = tiles_of(n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(planes_syrk, dim3(tiles), dim3(64 * NW), NBUF * STAGE, 0, d, n, T, (int64_t)T, (int*)nullptr, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    CK(hipFuncSetAttribute((const void*)planes_syrk, hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * STAGE));
    const double ops = (S * (S + 1) / 2) * 2.0 * (double)tiles * TILE * TJ * T;   // 15 plane pairs, 2 ops per MAC (128 x 64 tiles incl. the diagonal's upper halves)
    printf("planes %d, waves %d, tile 128 x %d, %d workgroup(s) per CU, ring %d x %d tokens: ", S, NW, TJ, OCC, NBUF, 32 * KSUB);
    printf("n=%d T=%d: %.2f ms  %.0f int8 TOP/s (%.1f%% of 5000) incl. the diagonal tiles' upper halves; useful (SYRK count) %.0f TOP/s\n", n, T,
           best, ops / best / 1e9, ops / best / 1e9 / 50.0, (S * (S + 1) / 2) * (double)n * (n + 1) * T / best / 1e9);
  }
  return 0;
}
