// Probe for the next round ("candidates"): the digit-plane SYRK with v_mfma_i32_16x16x64_i8 and a 64 x 48
// wave tile (240 accumulators) instead of 32x32x32 / 64 x 32 (160).  Workgroup tile 128 x 96, k-step 64 tokens, 5 planes,
// 15 plane pairs.  Needs n divisible by 128 and 96: measured at n = 13824 (the 64 x 32 probe i8_planes_syrk.hip takes -DN_FEAT).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o i8p16.bin i8_planes_syrk_16.hip && ./i8p16.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int S = 5, TI = 128, TJ = 96, BK = 64;
constexpr int GA = TI / 16, GB = TJ / 16;                 // 16-row groups per operand
constexpr int PIECES = S * (GA + GB);                      // 70 pieces of 1 KB ([k-quarter][row][16 B]) per stage
constexpr int STAGE = PIECES * 1024;                       // 70 KB
constexpr int NBUF = 2;

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l,
                                   16, 0, 0);
}

// planes: [S][n / 16][T / 64][1 KB];  tile t: bi (128 rows), bj (96 rows), bj <= (128 (bi + 1) - 1) / 96
__global__ __launch_bounds__(256, 1) void planes_syrk16(const int8_t* __restrict__ planes, int n, int T, const int2* tiles, int* out,
                                                        int* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int bi = tiles[blockIdx.x].x, bj = tiles[blockIdx.x].y;
  if (bi < 0) return;   // padding entry of the XCD-interleaved order
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int nk = T / BK;
  const int64_t groups = n / 16;
  auto issue_stage = [&](int kt, int buf) {
#pragma unroll
    for (int q = 0; q < (PIECES + 3) / 4; q++) {
      const int p = wave + 4 * q;
      if (p < PIECES) {
        const bool isA = p < S * GA;
        const int pp = isA ? p : p - S * GA;
        const int s = isA ? pp / GA : pp / GB, g = isA ? pp % GA : pp % GB;
        const int64_t G = (isA ? bi * GA : bj * GB) + g;
        glds16(planes + ((s * groups + G) * (int64_t)nk + kt) * 1024 + lane * 16,
               lds + buf * STAGE + (isA ? (s * GA + g) : (S * GA + s * GB + g)) * 1024);
      }
    }
  };
  i32x4 acc[S][4][3];
#pragma unroll
  for (int k = 0; k < S; k++)
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) acc[k][a][b] = (i32x4)0;
  issue_stage(0, 0);
  for (int kt = 0; kt < nk; kt++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) issue_stage(kt + 1, (kt + 1) & 1);
    const unsigned char* base = lds + (kt & 1) * STAGE;
    const int r = lane & 15, kq = lane >> 4;
    i32x4 fa[S][4], fb[S][3];
#pragma unroll
    for (int s = 0; s < S; s++) {
#pragma unroll
      for (int a = 0; a < 4; a++) fa[s][a] = *(const i32x4*)(base + (s * GA + wr * 4 + a) * 1024 + kq * 256 + r * 16);
#pragma unroll
      for (int b = 0; b < 3; b++) fb[s][b] = *(const i32x4*)(base + (S * GA + s * GB + wc * 3 + b) * 1024 + kq * 256 + r * 16);
    }
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
      for (int t = 0; t < S - s; t++)
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 3; b++)
            acc[s + t][a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[s][a], fb[t][b], acc[s + t][a][b], 0, 0, 0);
  }
  if (out) {
    int* o = out + (int64_t)blockIdx.x * S * TI * TJ;
#pragma unroll
    for (int k = 0; k < S; k++)
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 3; b++)
#pragma unroll
          for (int reg = 0; reg < 4; reg++) {
            const int row = wr * 64 + a * 16 + 4 * (lane >> 4) + reg, col = wc * 48 + b * 16 + (lane & 15);
            o[(k * TI + row) * TJ + col] = acc[k][a][b][reg];
          }
  } else {
    int x = 0;
#pragma unroll
    for (int k = 0; k < S; k++)
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 3; b++)
#pragma unroll
          for (int reg = 0; reg < 4; reg++) x ^= acc[k][a][b][reg];
    if (x == 0x7fffffff) sink[0] = x;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static std::vector<int2> lower_tiles(int n) {
  std::vector<int2> t;
  for (int bi = 0; bi < n / TI; bi++)
    for (int bj = 0; bj <= (TI * (bi + 1) - 1) / TJ; bj++) t.push_back(make_int2(bi, bj));
  return t;
}

// XCD-aware order: workgroup b runs on XCD b % 8; give each XCD whole super-blocks of SBI x SBJ tiles (32 = its CUs at one
// workgroup per CU) so the 4 + 8 operand panels of a super-block are fetched into that XCD's L2 once.
static std::vector<int2> xcd_tiles(int n, int SBI, int SBJ) {
  std::vector<std::vector<int2>> seq(8);
  int sb = 0;
  for (int si = 0; si * SBI < n / TI; si++)
    for (int sj = 0; sj * SBJ * TJ <= TI * std::min(n / TI, (si + 1) * SBI) - 1; sj++, sb++) {
      auto& q = seq[sb % 8];
      for (int a = 0; a < SBI; a++)
        for (int b = 0; b < SBJ; b++) {
          const int bi = si * SBI + a, bj = sj * SBJ + b;
          if (bi < n / TI && bj * TJ <= TI * (bi + 1) - 1 && bj * TJ < n) q.push_back(make_int2(bi, bj));
        }
    }
  size_t len = 0;
  for (auto& q : seq) len = std::max(len, q.size());
  std::vector<int2> t(len * 8, make_int2(-1, -1));
  for (int x = 0; x < 8; x++)
    for (size_t i = 0; i < seq[x].size(); i++) t[i * 8 + x] = seq[x][i];
  return t;
}

int main(int argc, char** argv) {
  {  // verification
    const int n = 384, T = 256, nk = T / BK;
    std::vector<int8_t> h((size_t)S * n * T), blk(h.size());
    srand(1);
    for (auto& v : h) v = (int8_t)(rand() % 256 - 128);
    for (int s = 0; s < S; s++)
      for (int row = 0; row < n; row++)
        for (int t = 0; t < T; t++)
          blk[(((size_t)s * (n / 16) + row / 16) * nk + t / BK) * 1024 + ((t % BK) / 16) * 256 + (row % 16) * 16 + t % 16] = h[((size_t)s * n + row) * T + t];
    auto tiles = lower_tiles(n);
    int8_t* d; int* out; int* sink; int2* dt;
    CK(hipMalloc(&d, blk.size())); CK(hipMalloc(&out, tiles.size() * S * TI * TJ * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&dt, tiles.size() * 8));
    CK(hipMemcpy(d, blk.data(), blk.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dt, tiles.data(), tiles.size() * 8, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)planes_syrk16, hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * STAGE));
    hipLaunchKernelGGL(planes_syrk16, dim3(tiles.size()), dim3(256), NBUF * STAGE, 0, d, n, T, dt, out, sink);
    CK(hipDeviceSynchronize());
    std::vector<int> got(tiles.size() * S * TI * TJ);
    CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    for (size_t ti = 0; ti < tiles.size(); ti++)
      for (int k = 0; k < S; k++)
        for (int i = 0; i < TI; i += 37)
          for (int j = 0; j < TJ; j += 29) {
            const int col = tiles[ti].y * TJ + j;
            if (col >= n) continue;
            long ref = 0;
            for (int s = 0; s <= k; s++)
              for (int x = 0; x < T; x++) ref += (int)h[((size_t)s * n + tiles[ti].x * TI + i) * T + x] * (int)h[((size_t)(k - s) * n + col) * T + x];
            if ((int)ref != got[(ti * S + k) * TI * TJ + i * TJ + j]) bad++;
          }
    printf("verification 16x16x64, 128 x 96 tiles: %s\n", bad ? "FAILED" : "ok");
    if (bad) return 2;
  }
  {
    const int n = 13824, T = 32768;
    const int sbi = argc > 2 ? atoi(argv[1]) : 0, sbj = argc > 2 ? atoi(argv[2]) : 0;
    auto tiles = sbi ? xcd_tiles(n, sbi, sbj) : lower_tiles(n);
    size_t real = 0, want = lower_tiles(n).size();
    for (auto& t : tiles) real += t.x >= 0;
    if (real != want) { printf("tile list wrong: %zu vs %zu\n", real, want); return 3; }
    printf("order: %s %d x %d; ", sbi ? "XCD super-blocks" : "row-major", sbi, sbj);
    int8_t* d; int* sink; int2* dt;
    const size_t bytes = (size_t)S * n * T;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&dt, tiles.size() * 8));
    std::vector<int8_t> h(1 << 24);
    for (auto& v : h) v = (int8_t)(rand() % 256 - 128);
    for (size_t off = 0; off < bytes; off += h.size()) CK(hipMemcpy(d + off, h.data(), std::min(h.size(), bytes - off), hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, tiles.data(), tiles.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(planes_syrk16, dim3(tiles.size()), dim3(256), NBUF * STAGE, 0, d, n, T, dt, (int*)nullptr, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    const double useful = 15.0 * (double)n * (n + 1) * T;   // SYRK count x 15 pairs
    printf("n=%d T=%d, %zu tiles of 128x96: %.2f ms  %.0f useful int8 TOP/s (%.1f%% of 5000)\n", n, T, tiles.size(), best, useful / best / 1e9,
           useful / best / 1e9 / 50.0);
  }
  return 0;
}
