// sigma += X^T X through the int8 matrix cores, exactly: an error-free split of the bf16 activations into digit planes.
//
// A bf16 value is a signed 8-bit significand times a power of two.  Against a per-column scale 2^(E_j - 134), E_j the largest
// exponent in column j of this call, it is a 40-bit fixed-point integer N = sig << (30 - (E_j - e)): five balanced base-256
// digits d_0..d_4 in [-128, 127] (|d_0| <= 64).  Then
//     x_ti x_tj = 2^(E_i + E_j - 328) * sum_{s,t} d_s(t,i) d_t(t,j) 256^(8 - s - t)
// and sum over tokens of d_s d_t is an int8 MFMA product with exact int32 accumulation.  Products with s + t >= 5 carry less
// than 2^-38 of (column maximum)^2 per token and are dropped, 15 plane pairs in 5 scale classes remain; elements more than
// 30 bits below their column's maximum are rounded at 2^-40 of it.  On data whose columns are not dominated by outliers this
// is ~1e-13 of |sigma| after 10^6 tokens (the fp64 route's own rounding is of that order); a column whose maximum towers over
// its typical magnitude loses accuracy, so every call checks a per-column statistic (the share of elements within 2^-5 of the
// column maximum's binade) and hands the batch to the fp64 kernel when it fails.  fp64 reference semantics: src/adapters/
// LlamaAdapter.py:127-147 (sigma += X^T X with X upcast to fp64).
//
// Three kernels per call:
//   i8_colmax_kernel   E_j = max exponent per column
//   i8_split_kernel    digit planes, written in the blocked layout the product kernel streams: [plane][32-row group][k-step]
//                      [k-half][row][16 tokens] -- each 1 KB piece is one contiguous global_load_lds_dwordx4 per wave
//   i8_syrk_kernel     128 x 64 output tiles of the lower triangle; 4 waves, wave tile 64 x 32 (160 int32 accumulators: the
//                      64 x 64 wave tile's 320 exceed the 256 AGPRs); per k-step of 32 tokens ONE set of 15 fragment reads
//                      feeds all 30 MFMAs of the 15 plane pairs (3x less LDS traffic per MFMA than 15 separate GEMMs, which is
//                      what lets it pass the library's int8 rate); LDS ring of 4 stages filled by LDS-DMA three stages ahead,
//                      one raw barrier per stage; every 16384 tokens the int32 classes are folded into sigma in fp64.
#include "common.hpp"

namespace mdg {
namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int NP = 5;            // digit planes
constexpr int TI = 128, TJ = 64; // output tile: TI rows of I x TJ rows of J
constexpr int KS = 32;           // tokens per k-step (one v_mfma_i32_32x32x32_i8)
constexpr int PA = TI * KS, PB = TJ * KS;
constexpr int STAGE_BYTES = NP * (PA + PB);  // 30 KB
constexpr int RING = 4;
constexpr int FLUSH_STEPS = 512;  // 16384 tokens: (k + 1) * 2^14 * 16384 < 2^31 for every class k <= 4
constexpr int NEAR_BINADES = 5;   // an element is "near the column maximum" when its exponent is within this many binades

// bf16 bits -> (signed 9-bit significand, effective exponent >= 1);  value = sig * 2^(ee - 134)
__device__ __forceinline__ void bf16_parts(unsigned b, int& sig, int& ee) {
  const int e = (b >> 7) & 0xFF, m = b & 0x7F;
  sig = e ? (128 | m) : m;
  ee = e ? e : 1;
  if (b & 0x8000) sig = -sig;
}

__global__ __launch_bounds__(256) void i8_colmax_kernel(const bf16_t* x, int64_t ld, int64_t T, int n, int64_t rows_per_block,
                                                        int* emax) {
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t t0 = (int64_t)blockIdx.y * rows_per_block + (threadIdx.x >> 6);
  const int64_t t1 = min(T, (int64_t)(blockIdx.y + 1) * rows_per_block);
  int best = 1;
  if (j < n)
    for (int64_t t = t0; t < t1; t += 4) {
      int sig, ee;
      bf16_parts(x[t * ld + j], sig, ee);
      if (sig != 0) best = max(best, ee);
    }
  if (j < n) atomicMax(emax + j, best);
}

// One thread = one feature row of a 32-row group x one k-step (32 tokens) at a time: two 16-byte stores per plane and
// k-step.  A workgroup walks SPLIT_STEPS k-steps of its row group, 8 at a time.
constexpr int SPLIT_STEPS = 64;
__global__ __launch_bounds__(256) void i8_split_kernel(const bf16_t* x, int64_t ld, int64_t T, int n, int nk, const int* emax,
                                                       signed char* planes, int* near_cnt) {
  __shared__ int near_lds[32];
  const int r = threadIdx.x & 31;
  const int G = blockIdx.x;
  const int j = G * 32 + r;
  const int E = emax[j];
  const int64_t groups = n / 32;
  if (threadIdx.x < 32) near_lds[threadIdx.x] = 0;
  __syncthreads();
  int near = 0;
  for (int kq = 0; kq < SPLIT_STEPS; kq += 8) {
    const int kt = blockIdx.y * SPLIT_STEPS + kq + (threadIdx.x >> 5);
    if (kt >= nk) break;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      unsigned dig[NP][4] = {};  // 16 bytes per plane
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int64_t t = (int64_t)kt * KS + h * 16 + q;
        int sig = 0, ee = 1;
        if (t < T) bf16_parts(x[t * ld + j], sig, ee);
        const int sh = E - ee;
        near += (sig != 0 && sh <= NEAR_BINADES);
        long long N;
        if (sh <= 30) {
          N = (long long)sig << (30 - sh);
        } else {
          const int dn = sh - 30;  // round the magnitude half up; nothing survives a shift by more than 9
          const int mag = dn > 9 ? 0 : ((sig < 0 ? -sig : sig) + (1 << (dn - 1))) >> dn;
          N = sig < 0 ? -mag : mag;
        }
#pragma unroll
        for (int s = NP - 1; s >= 1; s--) {
          const int b = (int)((N + 128) & 255) - 128;  // balanced digit in [-128, 127]
          dig[s][q >> 2] |= (unsigned)(b & 255) << (8 * (q & 3));
          N = (N - b) >> 8;
        }
        dig[0][q >> 2] |= (unsigned)((int)N & 255) << (8 * (q & 3));
      }
#pragma unroll
      for (int s = 0; s < NP; s++) {
        signed char* piece = planes + ((s * groups + G) * (int64_t)nk + kt) * 1024;
        *(i32x4*)(piece + h * 512 + r * 16) = (i32x4){(int)dig[s][0], (int)dig[s][1], (int)dig[s][2], (int)dig[s][3]};
      }
    }
  }
  if (near) atomicAdd(&near_lds[r], near);
  __syncthreads();
  if (threadIdx.x < 32 && near_lds[threadIdx.x]) atomicAdd(near_cnt + G * 32 + threadIdx.x, near_lds[threadIdx.x]);
}

// flag[0] = 1 when some column has fewer than T / 16 elements near its maximum (an outlier-dominated column)
__global__ __launch_bounds__(256) void i8_crest_kernel(const int* near_cnt, int n, int64_t T, int* flag) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n && (int64_t)near_cnt[j] * 16 < T) atomicOr(flag, 1);
}

struct SyrkArgs {
  const signed char* planes;
  const int* emax;
  double* sigma;
  int64_t ld_sigma;
  int n, nk;
};

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16,
                                   0, 0);
}

__global__ __launch_bounds__(256, 1) void i8_syrk_kernel(SyrkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // tile t of the lower region: bi = 128-row block, bj = 64-row block with bj <= 2 bi + 1; bi (bi + 1) tiles precede row bi
  const int tile = blockIdx.x;
  int bi = (int)((sqrtf(4.f * tile + 1.f) - 1.f) * 0.5f);
  while ((bi + 1) * (bi + 2) <= tile) bi++;
  while (bi * (bi + 1) > tile) bi--;
  const int bj = tile - bi * (bi + 1);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t groups = a.n / 32;
  const int nk = a.nk;

  // staging: 30 pieces of 1 KB per stage (A: 5 planes x 4 row groups, B: 5 planes x 2); wave w issues pieces w, w + 4, ...
  auto issue_stage = [&](int kt, int buf) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int p = wave + 4 * q;
      if (p < 30) {
        const bool isA = p < 20;
        const int pp = isA ? p : p - 20;
        const int s = isA ? pp >> 2 : pp >> 1, g = isA ? pp & 3 : pp & 1;
        const int64_t G = (isA ? bi * (TI / 32) : bj * (TJ / 32)) + g;
        const signed char* src = a.planes + ((s * groups + G) * (int64_t)nk + kt) * 1024 + lane * 16;
        unsigned char* dst = lds + buf * STAGE_BYTES + (isA ? s * PA : NP * PA + s * PB) + g * 1024;
        glds16(src, dst);
      }
    }
  };

  i32x16 acc[NP][2];
#pragma unroll
  for (int k = 0; k < NP; k++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[k][b] = (i32x16)0;

  // sigma[i][j] += 2^(E_i + E_j - 328) * sum_k acc_k 256^(8 - k)  =  (sum_k acc_k 2^(64 - 8k)) * 2^(E_i - 164) * 2^(E_j - 164)
  auto flush = [&]() {
    const int col = bj * TJ + wc * 32 + (lane & 31);
    const double sc_j = ldexp(1.0, a.emax[col] - 164);
    // all 32 read-modify-writes of a lane: loads first (independent, in flight together), then the arithmetic and the stores;
    // written as `*p += v` one by one the compiler must keep them in order and every element pays a full memory round trip
    double old[2][16];
    int er[2][16];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const int row = bi * TI + wr * 64 + b * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        old[b][reg] = a.sigma[(int64_t)row * a.ld_sigma + col];
        er[b][reg] = a.emax[row];
      }
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const int row = bi * TI + wr * 64 + b * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        double v = 0.;
#pragma unroll
        for (int k = NP - 1; k >= 0; k--) v += ldexp((double)acc[k][b][reg], 64 - 8 * k);
        if (col <= row) a.sigma[(int64_t)row * a.ld_sigma + col] = old[b][reg] + v * sc_j * ldexp(1.0, er[b][reg] - 164);
      }
#pragma unroll
    for (int k = 0; k < NP; k++)
#pragma unroll
      for (int b = 0; b < 2; b++) acc[k][b] = (i32x16)0;
    // the stores above share the VM counter with the LDS-DMA loads and may retire out of order with them: drain, so that
    // the counted wait of the next stage again counts LDS-DMA loads only
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // two loops: the int32 classes are folded into sigma between runs of FLUSH_STEPS k-steps, outside the MFMA loop (a
  // conditional flush inside it makes the compiler shuttle all 160 accumulators between AGPRs and VGPRs every step)
  for (int p = 0; p < RING - 1 && p < nk; p++) issue_stage(p, p);
  for (int k0 = 0; k0 < nk; k0 += FLUSH_STEPS) {
    const int k1 = min(nk, k0 + FLUSH_STEPS);
    for (int kt = k0; kt < k1; kt++) {
      const int buf = kt % RING;
      // waves 0,1 issue 8 LDS-DMA loads per stage, waves 2,3 issue 7; stages kt+1 and kt+2 may stay in flight: "at most 14
      // outstanding" retires stage kt on every wave; the tail drains everything
      if (kt + RING - 2 < nk) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + RING - 1 < nk) issue_stage(kt + RING - 1, (kt + RING - 1) % RING);
      const unsigned char* base = lds + buf * STAGE_BYTES;
      const int r = lane & 31, h = lane >> 5;
      i32x4 fa[NP][2], fb[NP];
#pragma unroll
      for (int s = 0; s < NP; s++) {
#pragma unroll
        for (int b = 0; b < 2; b++) fa[s][b] = *(const i32x4*)(base + s * PA + (wr * 2 + b) * 1024 + h * 512 + r * 16);
        fb[s] = *(const i32x4*)(base + NP * PA + s * PB + wc * 1024 + h * 512 + r * 16);
      }
#pragma unroll
      for (int s = 0; s < NP; s++)
#pragma unroll
        for (int t = 0; t < NP - s; t++)
#pragma unroll
          for (int b = 0; b < 2; b++)
            acc[s + t][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s][b], fb[t], acc[s + t][b], 0, 0, 0);
    }
    flush();
  }
}

size_t planes_bytes(int64_t T, int64_t n) { return (size_t)NP * (size_t)n * (size_t)ceil_div(T, KS) * KS; }

}  // namespace
}  // namespace mdg

using namespace mdg;

extern "C" size_t mdg_cov_accum_i8_ws_bytes(int64_t n_tokens, int64_t n_feat) {
  if (n_tokens <= 0 || n_feat <= 0) return 0;
  const size_t fallback = mdg_cov_accum_ws_bytes(n_tokens, n_feat, 1);
  return align_up(planes_bytes(n_tokens, n_feat), 256) + align_up((size_t)(2 * n_feat + 4) * sizeof(int), 256) + fallback + 256;
}

extern "C" int mdg_cov_accum_i8(const void* x, int64_t n_tokens, int64_t n_feat, int64_t ld, double* sigma, int64_t ld_sigma,
                                void* ws, size_t ws_bytes, int* used_i8, void* stream) {
  MDG_CLEAR();
  if (used_i8) *used_i8 = 0;
  MDG_CHECK_ARG(n_tokens >= 0 && n_feat > 0, "mdg_cov_accum_i8: bad sizes (tokens=%lld feat=%lld)", (long long)n_tokens,
                (long long)n_feat);
  MDG_CHECK_ARG(n_feat % TI == 0, "mdg_cov_accum_i8: n_feat=%lld must be a multiple of %d (use mdg_cov_accum)",
                (long long)n_feat, TI);
  MDG_CHECK_ARG(ld >= n_feat && ld_sigma >= n_feat, "mdg_cov_accum_i8: leading dimensions too small");
  MDG_CHECK_ARG(n_feat < (1 << 24), "mdg_cov_accum_i8: n_feat too large");
  if (n_tokens == 0) return MDG_OK;
  MDG_CHECK_ARG(x && sigma, "mdg_cov_accum_i8: null pointer");
  const size_t need = mdg_cov_accum_i8_ws_bytes(n_tokens, n_feat);
  MDG_CHECK_ARG(ws && ws_bytes >= need, "mdg_cov_accum_i8: workspace %zu < required %zu", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const int n = (int)n_feat;
  const int nk = (int)ceil_div(n_tokens, KS);
  signed char* planes = (signed char*)ws;
  int* emax = (int*)((char*)ws + align_up(planes_bytes(n_tokens, n_feat), 256));
  int* near_cnt = emax + n;
  int* flag = near_cnt + n;
  void* fb_ws = (char*)emax + align_up((size_t)(2 * n + 4) * sizeof(int), 256);
  MDG_HIP(hipMemsetAsync(emax, 0, (size_t)(2 * n + 4) * sizeof(int), st));
  {
    const int64_t rows_per_block = 2048;
    const dim3 grid((unsigned)ceil_div(n, 64), (unsigned)ceil_div(n_tokens, rows_per_block));
    hipLaunchKernelGGL(i8_colmax_kernel, grid, dim3(256), 0, st, (const bf16_t*)x, ld, n_tokens, n, rows_per_block, emax);
  }
  hipLaunchKernelGGL(i8_split_kernel, dim3((unsigned)(n / 32), (unsigned)ceil_div(nk, SPLIT_STEPS)), dim3(256), 0, st, (const bf16_t*)x, ld,
                     n_tokens, n, nk, emax, planes, near_cnt);
  hipLaunchKernelGGL(i8_crest_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, near_cnt, n, n_tokens, flag);
  MDG_LAUNCH_CHECK();
  int outlier = 0;
  MDG_HIP(hipMemcpyAsync(&outlier, flag, sizeof(int), hipMemcpyDeviceToHost, st));
  MDG_HIP(hipStreamSynchronize(st));
  if (outlier)  // a column dominated by outliers: five planes do not carry fp64-level accuracy there
    return mdg_cov_accum(x, MDG_BF16, n_tokens, n_feat, 1, ld, 0, sigma, ld_sigma, 0, fb_ws,
                         ws_bytes - (size_t)((char*)fb_ws - (char*)ws), stream);
  SyrkArgs a;
  a.planes = planes; a.emax = emax; a.sigma = sigma; a.ld_sigma = ld_sigma; a.n = n; a.nk = nk;
  const int rb = n / TI;
  const size_t lds = (size_t)RING * STAGE_BYTES;
  MDG_HIP(hipFuncSetAttribute((const void*)i8_syrk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(i8_syrk_kernel, dim3((unsigned)(rb * (rb + 1))), dim3(256), lds, st, a);
  MDG_LAUNCH_CHECK();
  if (used_i8) *used_i8 = 1;
  return MDG_OK;
}
