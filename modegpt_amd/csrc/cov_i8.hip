// sigma += X^T X through the int8 matrix cores: an ERROR-FREE SPLIT of the bf16 activations into digit planes and a TRUNCATED
// plane-pair product whose error is bounded per call.
//
// A bf16 value is a signed 8-bit significand times a power of two.  Against a per-column scale 2^(E_j - 172), E_j the largest
// exponent in column j of this call, it is a 48-bit fixed-point integer N = sig << (38 - (E_j - e)): six balanced base-256
// digits d_0..d_5 in [-128, 127] (|d_0| <= 64) -- exactly, for every element within 38 binades of its column maximum.  Then
//     x_ti x_tj = 2^(E_i + E_j - 344) * sum_{s,t} d_s(t,i) d_t(t,j) 256^(10 - s - t)
// and sum over tokens of d_s d_t is an int8 MFMA product with exact int32 accumulation.  The product keeps the plane pairs with
// s + t < P (P = 5: 15 pairs, P = 6: 21 pairs) and DROPS the others.  What the dropped pairs can amount to is bounded from integer
// plane energies the split pass accumulates per column (Cauchy-Schwarz over the tokens; i8_route_kernel below has the derivation):
//     |sigma_ij(P) - sigma_ij| <= (SQ_P + X_P) sqrt(sigma_ii sigma_jj),   entry-wise, for any input,
// and the route is the smallest P with SQ_P <= 1e-12 (the attained part) and X_P <= tau_x <= 1e-11 (cross terms), after at most 32
// columns have been handed to an fp64 column kernel: GUARANTEED <= 1.1e-11, MEASURED <= 1e-12 (scripts/probes/i8_error_bound.py,
// i8_fuzz.py; Gaussian / ReLU columns take five planes at 2e-14 .. 2e-13, SiLU- / GELU-gated products -- the MLP statistic of a
// real Llama -- six at 6e-14; a column whose bulk sits 10+ binades under a few massive activations leaves alone).  Otherwise the
// whole statistic goes through the fp64 kernel (mdg_cov_accum).
// fp64 reference semantics: src/adapters/LlamaAdapter.py:127-147 (sigma += X^T X with X upcast to fp64).
//
// Kernels per call:
//   i8_colmax_kernel   E_j = max exponent per column
//   i8_split_kernel    six digit planes, written in the blocked layout the product kernel streams: [plane][32-row group]
//                      [k-step][k-half][row][16 tokens] -- each 1 KB piece is one contiguous global_load_lds_dwordx4 per wave;
//                      accumulates the per-column integers of the route (sum of d_s^2 per plane, sum of d_0 d_1, nonzero / rounded
//                      counts) on the way, and writes one mask byte per (k-step, 32-row group) saying which planes hold a nonzero
//                      there (an element is two full digits and a carry digit, so whole pieces of the deeper planes are zero on
//                      real activations)
//   i8_route_kernel    the route: bit 0 of the statistic's flag -> six planes, bit 1 -> the fp64 kernel for the whole statistic;
//                      bit 8 of emax[j] -> column j is computed by the fp64 column kernel (read by the launches below)
//   i8_clear_columns_kernel   zeroes the digits of such columns and refreshes their groups' piece masks
//   i8_syrk_kernel<P>  output tiles of the lower triangle, two waves per SIMD inside one workgroup of 8 waves:
//                      P = 5: 128 x 128 tile, wave tile 64 x 32 (160 int32 accumulators; a 64 x 64 wave tile's 320 would not
//                      fit); P = 6: 128 x 64 tile, wave tile 32 x 32 (96).  Per k-step of 32 tokens ONE set of fragment reads
//                      feeds all P (P + 1) / 2 plane-pair products of the wave tile (3x less LDS traffic per MFMA than separate
//                      GEMMs, which is what lets it pass the library's int8 rate); 3- / 4-stage LDS ring filled by LDS-DMA from
//                      SGPR piece descriptors, the two waves of a SIMD in opposite load / multiply order, one raw barrier per
//                      stage; every 2047 k-steps (65504 tokens, the int32 bound) the classes are folded into sigma in fp64.  The
//                      P = 5 variant reads the top five of the six planes (a balanced-digit truncation).  Planes beyond a
//                      32-row group's depth in a k-step (piece masks) are neither written, nor loaded, nor read from LDS, nor
//                      multiplied: no bit of the result changes, and 28 - 37 % of the MFMAs go on SiLU-gated / Gaussian data.
//                      Both instantiations, the column kernel and the fp64 kernel are enqueued for every call; the device picks
//                      (each workgroup of the others exits on its first instruction).  Statistics of 2048 features and more
//                      (everything ops.py sends here) run as a persistent launch: one workgroup per CU pulling tiles from per-XCD
//                      queues, the last, partly filled round cut into k-chunks that fold into fp64 partial tiles
//   i8_tail_combine_kernel<P>  adds the partial tiles of that last round to sigma, in chunk order
//   i8_columns_kernel, i8_columns_reduce_kernel   rows / columns of sigma of the columns that left, in plain fp64
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.hpp"
#include <atomic>

namespace mdg {
namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int NP = 6;            // digit planes written by the split pass; the product kernel uses the top 5 or all 6
constexpr int TI = 128;           // output tile rows (rows of the I operand); its width TJ is 64 or 128, see i8_syrk_kernel
constexpr int KS = 32;           // tokens per k-step (one v_mfma_i32_32x32x32_i8)
constexpr int PA = TI * KS;      // bytes of one plane of the I operand in a stage
// k-steps between folds of the int32 classes into sigma.  An element is an 8-bit significand at some shift, so its balanced
// digits are two full digits and a carry digit at most, and a class sum grows by at most 32768 per token (enumerated over
// every digit vector the split pass can produce: scripts/probes/i8_int32_bound.py) -- 65535 tokens = 2047 k-steps stay below
// 2^31.  (First versions: 512, from the cruder bound 6 pairs x 128 x 128 per token; one fold per launch costs 0.65 ms at the
// sigma_mlp shape.)
constexpr int FLUSH_STEPS = 2047;
constexpr int TOP_SHIFT = 8 * NP - 10;  // 38: the column maximum's significand sits below bit 46 of the 48-bit integer

// Per-column integers the split pass accumulates for the route (i8_route_kernel; host model: tests/i8_model.py), as [NSTAT][n]
// unsigned long long: q_s = sum over tokens of d_s^2 for the six planes, the signed sum of d_0 d_1 (so that the energy of the top
// two digits together, hence a lower bound on the column's norm, is an integer too), and two counters packed into one word.
constexpr int NSTAT = 8;
constexpr int STAT_D0D1 = 6, STAT_COUNTS = 7;     // [7]: (elements rounded to an integer: more than 38 binades down) << 32 | nonzero elements
constexpr int EMAX_COLUMN_OUT = 0x100;            // bit set in emax[j] by the route kernel: column j is computed by the fp64 column kernel

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

// Piece mask: one byte per (k-step, 32-row group), bit s set when the 1 KB piece of plane s there holds any nonzero digit.  An
// element is two full digits and a carry digit (see FLUSH_STEPS), so on real activations whole pieces of the lower planes are
// zero -- Gaussian columns: plane 3 in 90 % of the pieces, planes 4 and 5 always; SiLU-gated: plane 4 in 98 % -- and the
// product kernel neither loads nor multiplies those.  Called with the 32 rows of a piece in the 32 lanes of a half-wave.
__device__ __forceinline__ unsigned write_piece_mask(const unsigned (&any)[NP], unsigned char* zmask, int64_t index) {
  unsigned byte = 0;
#pragma unroll
  for (int s = 0; s < NP; s++) {
    const unsigned long long b = __ballot(any[s] != 0);
    const unsigned half = (threadIdx.x & 32) ? (unsigned)(b >> 32) : (unsigned)b;
    byte |= (half != 0) << s;
  }
  if ((threadIdx.x & 31) == 0) zmask[index] = (unsigned char)byte;
  return byte;     // the mask of the caller's own piece (its half-wave)
}

// Planes the product kernels may read even where the piece mask says "all zero": both load every plane below their MIN_DEPTH
// unconditionally (3 for five planes, 4 for six).  Pieces of the planes from here on are WRITTEN only when they hold a nonzero
// there or in a deeper plane of the same piece (a product kernel that finds plane 5 present loads plane 4 as well) -- on
// Gaussian / ReLU / SiLU-gated activations planes 4 and 5 practically never do: a third of the split pass's writes.
constexpr int ALWAYS_WRITTEN_PLANES = 4;

// One thread = one feature row of a 32-row group x one k-step (32 tokens) at a time: two 16-byte stores per plane and
// k-step.  A workgroup walks SPLIT_STEPS k-steps of its row group, 8 at a time.
constexpr int SPLIT_STEPS = 64;
__global__ __launch_bounds__(256) void i8_split_kernel(const bf16_t* x, int64_t ld, int64_t T, int n, int nk, const int* emax,
                                                       signed char* planes, unsigned long long* stats, unsigned char* zmask) {
  __shared__ unsigned long long st_lds[NSTAT][32];
  const int r = threadIdx.x & 31;
  const int G = blockIdx.x;
  const int j = G * 32 + r;
  const int E = emax[j];
  const int64_t groups = n / 32;
  for (int i = threadIdx.x; i < NSTAT * 32; i += 256) (&st_lds[0][0])[i] = 0;
  __syncthreads();
  long long q[NSTAT] = {};
  for (int kq = 0; kq < SPLIT_STEPS; kq += 8) {
    const int kt = blockIdx.y * SPLIT_STEPS + kq + (threadIdx.x >> 5);
    if (kt >= nk) break;
    unsigned any[NP] = {};       // per plane: does this row hold a nonzero digit in this k-step
#pragma unroll
    for (int h = 0; h < 2; h++) {
      unsigned dig[NP][4] = {};  // 16 bytes per plane
#pragma unroll
      for (int qq = 0; qq < 16; qq++) {
        const int64_t t = (int64_t)kt * KS + h * 16 + qq;
        int sig = 0, ee = 1;
        if (t < T) bf16_parts(x[t * ld + j], sig, ee);
        const int sh = E - ee;
        q[STAT_COUNTS] += (long long)(sig != 0) + ((long long)(sig != 0 && sh > TOP_SHIFT) << 32);
        long long N;
        if (sh <= TOP_SHIFT) {
          N = (long long)sig << (TOP_SHIFT - sh);
        } else {
          const int dn = sh - TOP_SHIFT;  // round the magnitude half up; nothing survives a shift by more than 9
          const int mag = dn > 9 ? 0 : ((sig < 0 ? -sig : sig) + (1 << (dn - 1))) >> dn;
          N = sig < 0 ? -mag : mag;
        }
        int d[NP];
#pragma unroll
        for (int s = NP - 1; s >= 1; s--) {
          d[s] = (int)((N + 128) & 255) - 128;  // balanced digit in [-128, 127]
          dig[s][qq >> 2] |= (unsigned)(d[s] & 255) << (8 * (qq & 3));
          N = (N - d[s]) >> 8;
        }
        d[0] = (int)N;
        dig[0][qq >> 2] |= (unsigned)(d[0] & 255) << (8 * (qq & 3));
#pragma unroll
        for (int s = 0; s < NP; s++) q[s] += d[s] * d[s];
        q[STAT_D0D1] += d[0] * d[1];
      }
#pragma unroll
      for (int s = 0; s < NP; s++) {
        signed char* piece = planes + ((s * groups + G) * (int64_t)nk + kt) * 1024;
        *(i32x4*)(piece + h * 512 + r * 16) = (i32x4){(int)dig[s][0], (int)dig[s][1], (int)dig[s][2], (int)dig[s][3]};
        any[s] |= dig[s][0] | dig[s][1] | dig[s][2] | dig[s][3];
      }
    }
    write_piece_mask(any, zmask, (int64_t)kt * groups + G);   // the 32 lanes of a half-wave hold the 32 rows of the piece
  }
#pragma unroll
  for (int i = 0; i < NSTAT; i++)
    if (q[i]) atomicAdd(&st_lds[i][r], (unsigned long long)q[i]);
  __syncthreads();
  for (int i = threadIdx.x; i < NSTAT * 32; i += 256)
    if (st_lds[i >> 5][i & 31]) atomicAdd(stats + (int64_t)(i >> 5) * n + G * 32 + (i & 31), st_lds[i >> 5][i & 31]);
}

// ---- the same two passes for the usual case of 16-byte addressable rows (ld % 8 == 0, aligned base): 16-byte loads.
// The scalar kernels above read 2 bytes per lane in 64-byte row segments and run at ~2 TB/s; these read whole 256-byte
// segments and are bound by the 6 bytes per element the split writes.
__device__ __forceinline__ int bf16_ee_if_nonzero(unsigned b) {  // effective exponent of a nonzero value, 0 for +-0
  const int e = (b >> 7) & 0xFF;
  return (b & 0x7FFF) ? (e ? e : 1) : 0;
}

#ifndef MDG_COLMAX_WGS
#define MDG_COLMAX_WGS 4096
#endif
__global__ __launch_bounds__(256) void i8_colmax_vec_kernel(const bf16_t* x, int64_t ld, int64_t T, int64_t rows_per_block, int* emax) {
  __shared__ int best_lds[128];
  const int cg = threadIdx.x & 15, tl = threadIdx.x >> 4;  // 16 column groups of 8 columns x 16 token lanes
  const int j0 = blockIdx.x * 128 + cg * 8;
  const int64_t t0 = (int64_t)blockIdx.y * rows_per_block + tl;
  const int64_t t1 = min(T, (int64_t)(blockIdx.y + 1) * rows_per_block);
  if (threadIdx.x < 128) best_lds[threadIdx.x] = 1;
  __syncthreads();
  int best[8] = {1, 1, 1, 1, 1, 1, 1, 1};
  for (int64_t t = t0; t < t1; t += 16) {
    const i32x4 v = *(const i32x4*)(x + t * ld + j0);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const unsigned w = (unsigned)v[q];
      best[2 * q] = max(best[2 * q], bf16_ee_if_nonzero(w & 0xFFFF));
      best[2 * q + 1] = max(best[2 * q + 1], bf16_ee_if_nonzero(w >> 16));
    }
  }
#pragma unroll
  for (int q = 0; q < 8; q++) atomicMax(&best_lds[cg * 8 + q], best[q]);
  __syncthreads();
  if (threadIdx.x < 128) atomicMax(emax + blockIdx.x * 128 + threadIdx.x, best_lds[threadIdx.x]);
}

// workgroup = 128 features (4 row groups) x SPLIT_TILES tiles of 64 tokens (2 k-steps): a tile goes through LDS (the next one's
// loads are in flight meanwhile), one thread then owns one feature of one k-step.  The column statistics of the route are kept
// in registers over the tiles and leave the workgroup as 8 atomics per feature.
constexpr int SPLIT_TILES = 8;
__global__ __launch_bounds__(256) void i8_split_vec_kernel(const bf16_t* x, int64_t ld, int64_t T, int n, int nk, const int* emax,
                                                           signed char* planes, unsigned long long* stats, unsigned char* zmask) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 128];
  __shared__ int st_lds[NSTAT + 1][128];
  const int f0 = blockIdx.x * 128;
  const int f = threadIdx.x & 127, ks = threadIdx.x >> 7;
  const int E = emax[f0 + f];
  const int64_t groups = n / 32;
  const int G = (f0 + f) >> 5, r = f & 31;
  for (int i = threadIdx.x; i < (NSTAT + 1) * 128; i += 256) (&st_lds[0][0])[i] = 0;
  int q[NSTAT + 1] = {};   // q_0 .. q_5, sum d_0 d_1, nonzero elements, rounded elements: < 2^23 each over 8 tiles
  auto load_tile = [&](int tile_index, i32x4 (&v)[4]) {
    const int64_t tok0 = (int64_t)tile_index * 2 * KS;
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
      const int c = threadIdx.x + 256 * c4;  // 16-byte chunk: token c / 16, columns (c % 16) * 8 ..
      const int64_t t = tok0 + (c >> 4);
      v[c4] = (i32x4)0;
      if (t < T) v[c4] = *(const i32x4*)(x + t * ld + f0 + (c & 15) * 8);
    }
  };
  const int tile0 = blockIdx.y * SPLIT_TILES, tiles = (nk + 1) / 2;
  for (int it = 0; it < SPLIT_TILES && tile0 + it < tiles; it++) {
    __syncthreads();              // (the previous tile has been read by every thread)
    {
      i32x4 v[4];
      load_tile(tile0 + it, v);
#pragma unroll
      for (int c4 = 0; c4 < 4; c4++) *(i32x4*)(tile + (threadIdx.x + 256 * c4) * 8) = v[c4];
    }
    __syncthreads();
    const int kt = (tile0 + it) * 2 + ks;
    if (kt >= nk) continue;   // (uniform per wave: a wave holds 64 features of ONE k-step)
    unsigned any[NP] = {};
    unsigned deep_dig[2][NP - ALWAYS_WRITTEN_PLANES][4];   // planes 4, 5 wait for the piece mask; planes 0 - 3 are stored as they are made
    // Balanced base-256 digits without a carry loop: N + 128 (256^0 + ... + 256^4) has the bytes d_i + 128 in its lower five
    // positions -- the addition's own carries are the digit carries -- and the top digit above them; d_i = byte ^ 0x80.  Four
    // elements at a time, byte k of each gathered into one dword by v_perm_b32: ~27 VALU operations per element where the
    // digit-by-digit loop in 64-bit arithmetic took ~60 (the pass was VALU-bound: 1.1 ms at the sigma_mlp shape for 2.8 GB).
#pragma unroll
    for (int h = 0; h < 2; h++) {
      unsigned dig[NP][4];
#pragma unroll
      for (int q4 = 0; q4 < 4; q4++) {
        unsigned lo[4], hi[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          int sig, ee;
          bf16_parts(tile[(ks * 32 + h * 16 + q4 * 4 + e) * 128 + f], sig, ee);
          const int sh = E - ee;
          q[NSTAT - 1] += (sig != 0);
          long long N;
          if (sh <= TOP_SHIFT) {
            N = (long long)sig << (TOP_SHIFT - sh);
          } else {
            const int dn = sh - TOP_SHIFT;
            const int mag = dn > 9 ? 0 : ((sig < 0 ? -sig : sig) + (1 << (dn - 1))) >> dn;
            N = sig < 0 ? -mag : mag;
            q[NSTAT] += (sig != 0);      // rounded to an integer: the remainder term rho of the bound
          }
          const unsigned long long biased = (unsigned long long)N + 0x0000008080808080ull;
          lo[e] = (unsigned)biased;
          hi[e] = (unsigned)(biased >> 32);
        }
#pragma unroll
        for (int s2 = 0; s2 < NP; s2++) {
          constexpr unsigned ZERO_HI = 0x0c0c0000u;            // v_perm_b32 selector 0x0c: constant 0x00
          const int byte = NP - 1 - s2;                        // plane s2 = byte 5 - s2 of the 48-bit integer
          const unsigned sel = ZERO_HI | (unsigned)(byte & 3) | ((4u + (unsigned)(byte & 3)) << 8);   // [byte of src1, byte of src0]
          const unsigned t01 = __builtin_amdgcn_perm(byte < 4 ? lo[1] : hi[1], byte < 4 ? lo[0] : hi[0], sel);
          const unsigned t23 = __builtin_amdgcn_perm(byte < 4 ? lo[3] : hi[3], byte < 4 ? lo[2] : hi[2], sel);
          unsigned w = t01 | (t23 << 16);
          if (s2 > 0) w ^= 0x80808080u;
          dig[s2][q4] = w;
          q[s2] = __builtin_amdgcn_sdot4((int)w, (int)w, q[s2], false);   // sum of the four digits' squares (v_dot4c_i32_i8)
        }
        q[STAT_D0D1] = __builtin_amdgcn_sdot4((int)dig[0][q4], (int)dig[1][q4], q[STAT_D0D1], false);
      }
#pragma unroll
      for (int s2 = 0; s2 < NP; s2++) {
        any[s2] |= dig[s2][0] | dig[s2][1] | dig[s2][2] | dig[s2][3];
        if (s2 < ALWAYS_WRITTEN_PLANES) {
          signed char* piece = planes + ((s2 * groups + G) * (int64_t)nk + kt) * 1024;
          *(i32x4*)(piece + h * 512 + r * 16) = (i32x4){(int)dig[s2][0], (int)dig[s2][1], (int)dig[s2][2], (int)dig[s2][3]};
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++) deep_dig[h][s2 - ALWAYS_WRITTEN_PLANES][i] = dig[s2][i];
        }
      }
    }
    const unsigned present = write_piece_mask(any, zmask, (int64_t)kt * groups + G);
#pragma unroll
    for (int s2 = ALWAYS_WRITTEN_PLANES; s2 < NP; s2++)
      if ((present >> s2) != 0) {   // some plane >= s2 holds a nonzero here: the product kernels load every plane below a
                                    // group's depth (uniform per half-wave = per piece)
        signed char* piece = planes + ((s2 * groups + G) * (int64_t)nk + kt) * 1024;
#pragma unroll
        for (int h = 0; h < 2; h++)
          *(i32x4*)(piece + h * 512 + r * 16) = (i32x4){(int)deep_dig[h][s2 - ALWAYS_WRITTEN_PLANES][0], (int)deep_dig[h][s2 - ALWAYS_WRITTEN_PLANES][1],
                                                        (int)deep_dig[h][s2 - ALWAYS_WRITTEN_PLANES][2], (int)deep_dig[h][s2 - ALWAYS_WRITTEN_PLANES][3]};
      }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i <= NSTAT; i++)
    if (q[i]) atomicAdd(&st_lds[i][f], q[i]);
  __syncthreads();
  if (threadIdx.x < 128) {
    unsigned long long* o = stats + f0 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < STAT_COUNTS; i++) {   // (the sum of d_0 d_1 is signed: sign-extended, the 64-bit sum wraps correctly)
      const int val = st_lds[i][threadIdx.x];
      if (val) atomicAdd(o + (int64_t)i * n, (unsigned long long)(long long)val);
    }
    const unsigned long long counts = (unsigned long long)(unsigned)st_lds[NSTAT - 1][threadIdx.x] |
                                      ((unsigned long long)(unsigned)st_lds[NSTAT][threadIdx.x] << 32);
    if (counts) atomicAdd(o + (int64_t)STAT_COUNTS * n, counts);
  }
}

// ---- the route of a statistic (host model with the derivation: tests/i8_model.py).
// With alpha_s(j) = 256^(5 - s) ||d_s(., j)|| / ||N_j|| (plane energies over the column norm) and rho_j = sqrt(rounded_j) / (2 ||N_j||),
// Cauchy-Schwarz over the tokens bounds the error of the P-plane product entry-wise, for ANY input:
//     |sigma_ij(P) - sigma_ij| / sqrt(sigma_ii sigma_jj)  <=  sum_{s + t >= P} alpha_s(i) alpha_t(j) + rho_i + rho_j + rho_i rho_j  <=  SQ_P + X_P
//     SQ_P = sum_{2 s >= P} A_s^2 + 2 R + R^2   (attained on the diagonal)        X_P = sum_{s != t, s + t >= P} A_s A_t   (cross terms)
// with A_s, R the maxima over the columns that stay on the int8 path.  The route is the smallest P in {5, 6} for which
// SQ_P <= TAU_SQ and X_P <= tau_x(tokens) hold after at most ROUTE_JMAX columns have been handed to the fp64 column kernel -- greedily,
// each time the column whose removal lowers the violation most (a column dominated by a few massive activations carries a bulk
// that lives entirely in the deep planes: it alone sets A_2 .. A_4) -- else the whole statistic goes through mdg_cov_accum.  Columns
// holding an Inf / NaN always leave (only the fp64 arithmetic propagates those the way the reference does).
// ||N_j|| enters through the integer lower bound 2^32 (||256 d_0 + d_1|| - sqrt(nonzeros) / 2): every decision is a function of
// integer sums, hence run-to-run bit-identical.  One workgroup per statistic.
// TAU_SQ bounds the attained part.  The cross part is attained only by columns whose digit sequences are proportional over the
// tokens; for uncorrelated columns the sums behind it grow like sqrt(tokens) where Cauchy-Schwarz allows tokens, so the measured
// error sits ~4.5 / sqrt(tokens) below X_P (0.02 - 0.035 at 32768 tokens on every family of scripts/probes/i8_error_bound.py).
// Short calls have no such averaging (33 tokens: measured / X_P ~ 0.3), and neither have sparse columns (the sums run over a
// column's nonzero elements: 7033 tokens at 1 % density measured 0.23), hence the threshold on X_P grows with the EFFECTIVE token
// count -- the smallest number of nonzero elements any column of the statistic has:
// guaranteed <= TAU_SQ + tau_x <= 1.1e-11 for any input, and <= 1e-12 measured also on the uncorrelated data of a short or sparse call.
constexpr double TAU_SQ = 1e-12, TAU_X_MIN = 1e-12, TAU_X_MAX = 1e-11, TAU_X_TOKENS = 1024.0;
// The `tolerance` argument of mdg_cov_accum_i8 / _multi: one factor on both thresholds (1 = the figures above), per CALL -- the
// library keeps no accuracy state (two host threads with different factors each get the route of their own factor).  A caller
// who accepts `f` times the guarantee gets five planes where the default asks for six (SiLU-gated activations: X_5 = 3.7e-10,
// i.e. f >= 37); the bound every call computes (RouteOut::sq, ::x) says what was guaranteed either way.
__host__ __device__ inline double tau_x_of(int64_t tokens) {
  return fmin(TAU_X_MAX, fmax(TAU_X_MIN, TAU_X_MIN * ((double)tokens / TAU_X_TOKENS)));
}
constexpr int ROUTE_JMAX = MDG_I8_MAX_COLUMNS;   // columns per statistic and call the fp64 column kernel takes (32)
constexpr int NVAL = 7;                     // alpha_0 .. alpha_5, rho
constexpr int ROUTE_THREADS = 512;

struct RouteOut {                           // per statistic, in the workspace (mdg_cov_accum_i8_route reads it back)
  int planes;                               // 5, 6, or 0: the whole statistic goes through the fp64 kernel
  int n_out;                                // columns handed to the fp64 column kernel
  int out[ROUTE_JMAX];                      // ... in the order they were taken
  double sq, x;                             // SQ_P, X_P of the columns that stay (the guaranteed bound is their sum)
  double rho;                               // 2 R + R^2 alone: what is left of the bound when no plane pair is dropped (the exact route)
};

__device__ __forceinline__ void route_terms(const double (&A)[NVAL], int P, double& sq, double& x) {
  sq = 2.0 * A[6] + A[6] * A[6];
  x = 0.0;
#pragma unroll
  for (int s = 0; s < NP; s++)
#pragma unroll
    for (int t = 0; t < NP; t++)
      if (s + t >= P) {
        if (s == t) sq += A[s] * A[t];
        else x += A[s] * A[t];
      }
}
__device__ __forceinline__ double route_violation(const double (&A)[NVAL], int P, double tau_x) {
  double sq, x;
  route_terms(A, P, sq, x);
  return fmax(sq / TAU_SQ, x / tau_x);
}

struct Top2 { double m1; int a1; double m2; };
__device__ __forceinline__ void top2_merge(Top2& a, const Top2& b) {   // (lowest index wins among equals: the model's argmax)
  if (b.m1 > a.m1 || (b.m1 == a.m1 && b.a1 < a.a1)) {
    a.m2 = fmax(a.m1, b.m2);
    a.m1 = b.m1;
    a.a1 = b.a1;
  } else {
    a.m2 = fmax(a.m2, b.m1);
  }
}

// Scratch of the multi-workgroup first pass, per workgroup: the top two of its columns and the maxima of the 64 column classes.
struct RoutePartial {
  Top2 top[NVAL];
  unsigned long long cls[NVAL][64];
  unsigned min_nnz;              // fewest nonzero elements of any (not all-zero) column
};
struct RouteScratch {            // zeroed with the statistics before every call
  int ticket, forced;
};

// Grid: one workgroup per ROUTE_THREADS columns.  Every workgroup turns its columns' integers into alpha_s / rho (kept in `vals`
// for the greedy) and leaves its partial maxima in `partial`; the LAST one to finish (ticket) merges them and decides -- so the
// ~10 fp64 square roots per column are spread over the chip and the common case (nothing has to leave) ends there.
__global__ __launch_bounds__(ROUTE_THREADS) void i8_route_kernel(const unsigned long long* __restrict__ stats, int* emax, int n, int64_t n_tokens,
                                                                 double* __restrict__ vals, RoutePartial* partial, RouteScratch* scratch,
                                                                 int* flag, RouteOut* out, int* route_counts, double tolerance) {
  __shared__ unsigned long long group_max[NVAL][64];   // per quantity: maxima of the 64 column classes j % 64 (bit patterns of doubles >= 0)
  __shared__ Top2 wave_top[ROUTE_THREADS / 64][NVAL];
  __shared__ Top2 top[NVAL];
  __shared__ double floor_of[NVAL];
  __shared__ int decision;   // -1: keep going; 0: accepted; 1: this P cannot be reached
  __shared__ int forced_total, my_ticket;
  __shared__ unsigned min_nnz;
  __shared__ double tau_x_shared;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NVAL * 64; i += ROUTE_THREADS) (&group_max[0][0])[i] = 0ull;
  if (tid == 0) min_nnz = 0xffffffffu;
  __syncthreads();
  // top two of every quantity over the columns still on the int8 path: block reduction of per-thread results into top[]
  auto reduce_top = [&](Top2 (&t)[NVAL]) {
#pragma unroll
    for (int i = 0; i < NVAL; i++) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        Top2 o;
        o.m1 = __shfl_xor(t[i].m1, off);
        o.a1 = __shfl_xor(t[i].a1, off);
        o.m2 = __shfl_xor(t[i].m2, off);
        top2_merge(t[i], o);
      }
      if (lane == 0) wave_top[wave][i] = t[i];
    }
    __syncthreads();
    if (tid < NVAL) {
      Top2 r = wave_top[0][tid];
      for (int w = 1; w < ROUTE_THREADS / 64; w++) top2_merge(r, wave_top[w][tid]);
      top[tid] = r;
    }
    __syncthreads();
  };
  // pass 1 (every workgroup): alpha_s(j), rho_j of its column from the integers; columns with an Inf / NaN (emax 255) leave at once
  Top2 t[NVAL];
#pragma unroll
  for (int i = 0; i < NVAL; i++) t[i] = Top2{-1.0, 0x7fffffff, -1.0};
  {
    const int j = blockIdx.x * ROUTE_THREADS + tid;
    bool nonfinite = false;
    if (j < n) {
      double q[NSTAT];
#pragma unroll
      for (int i = 0; i < NSTAT - 1; i++) q[i] = i == STAT_D0D1 ? (double)(long long)stats[(int64_t)i * n + j] : (double)stats[(int64_t)i * n + j];
      const unsigned long long counts = stats[(int64_t)STAT_COUNTS * n + j];
      const int ex = emax[j];
      const double nnz = (double)(unsigned)counts, rounded = (double)(unsigned)(counts >> 32);
      const double hi2 = 65536.0 * q[0] + 512.0 * q[STAT_D0D1] + q[1];
      const double norm = (sqrt(fmax(hi2, 0.0)) - 0.5 * sqrt(nnz)) * 4294967296.0;
      // (norm <= 0 can only happen for a column of denormals, which has nothing below plane 1; 1e300 keeps the test conservative)
      const double inv = nnz > 0 ? (norm > 0 ? 1.0 / norm : 1e300) : 0.0;
      double a[NVAL];
#pragma unroll
      for (int s2 = 0; s2 < NP; s2++) a[s2] = q[s2] > 0 ? sqrt(q[s2]) * ldexp(1.0, 8 * (NP - 1 - s2)) * inv : 0.0;
      a[6] = rounded > 0 ? 0.5 * sqrt(rounded) * inv : 0.0;
      nonfinite = (ex & 255) == 255;
      if ((unsigned)counts) atomicMin(&min_nnz, (unsigned)counts);
#pragma unroll
      for (int i = 0; i < NVAL; i++) {
        vals[(int64_t)i * n + j] = a[i];
        if (!nonfinite) t[i] = Top2{a[i], j, -1.0};
      }
      if (nonfinite) emax[j] = ex | EMAX_COLUMN_OUT;
    }
#pragma unroll
    for (int i = 0; i < NVAL; i++)
      if (t[i].m1 > 0.0) atomicMax(&group_max[i][lane], (unsigned long long)__double_as_longlong(t[i].m1));   // (column j is in class j % 64 = lane)
    const unsigned long long nf = __ballot(nonfinite);
    if (lane == 0 && nf) atomicAdd(&scratch->forced, __popcll(nf));
  }
  reduce_top(t);
  // hand the partial results over; the last workgroup to arrive goes on
  RoutePartial& mine = partial[blockIdx.x];
  if (tid < NVAL) mine.top[tid] = top[tid];
  if (tid == 0) mine.min_nnz = min_nnz;
  for (int i = tid; i < NVAL * 64; i += ROUTE_THREADS) (&mine.cls[0][0])[i] = (&group_max[0][0])[i];
  __threadfence();
  __syncthreads();
  if (tid == 0) my_ticket = atomicAdd(&scratch->ticket, 1);
  __syncthreads();
  if (my_ticket != (int)gridDim.x - 1) return;
  __threadfence();
  // (the merges take the partials four workgroups at a time: four independent loads in flight instead of a chain of gridDim.x
  //  dependent round trips -- the merge order does not enter the result, top2_merge breaks ties by column index)
  const unsigned nwg = gridDim.x;
  if (tid < NVAL) {
    Top2 r = partial[0].top[tid];
    unsigned w = 1;
    for (; w + 4 <= nwg; w += 4) {
      const Top2 b0 = partial[w].top[tid], b1 = partial[w + 1].top[tid], b2 = partial[w + 2].top[tid], b3 = partial[w + 3].top[tid];
      top2_merge(r, b0); top2_merge(r, b1); top2_merge(r, b2); top2_merge(r, b3);
    }
    for (; w < nwg; w++) top2_merge(r, partial[w].top[tid]);
    r.m1 = fmax(r.m1, 0.0);
    r.m2 = fmax(r.m2, 0.0);
    top[tid] = r;
  }
  for (int i = tid; i < NVAL * 64; i += ROUTE_THREADS) {
    unsigned long long m = 0ull;
    unsigned w = 0;
    for (; w + 4 <= nwg; w += 4) {
      const unsigned long long v0 = (&partial[w].cls[0][0])[i], v1 = (&partial[w + 1].cls[0][0])[i], v2 = (&partial[w + 2].cls[0][0])[i],
                               v3 = (&partial[w + 3].cls[0][0])[i];
      m = max(max(m, v0), max(max(v1, v2), v3));
    }
    for (; w < nwg; w++) m = max(m, (&partial[w].cls[0][0])[i]);
    (&group_max[0][0])[i] = m;
  }
  if (tid == 0) {
    forced_total = scratch->forced;
    unsigned m = 0xffffffffu;
    unsigned w = 0;
    for (; w + 4 <= nwg; w += 4) {
      const unsigned v0 = partial[w].min_nnz, v1 = partial[w + 1].min_nnz, v2 = partial[w + 2].min_nnz, v3 = partial[w + 3].min_nnz;
      m = min(min(m, v0), min(min(v1, v2), v3));
    }
    for (; w < nwg; w++) m = min(m, partial[w].min_nnz);
    tau_x_shared = tau_x_of(min(n_tokens, (int64_t)m));
  }
  __syncthreads();
  const double tau_x = tau_x_shared;
  int n_out = 0;
  if (forced_total > ROUTE_JMAX) {      // too many: the whole statistic goes through the fp64 kernel
    if (tid == 0) {
      out->planes = 0; out->n_out = 0; out->sq = out->x = out->rho = 0.0;
      atomicOr(flag, 2);
    }
    for (int j = tid; j < n; j += ROUTE_THREADS) emax[j] &= 255;
    return;
  }
  if (forced_total) {                   // (rare: listed in index order by one thread)
    if (tid == 0)
      for (int j = 0; j < n; j++)
        if (emax[j] & EMAX_COLUMN_OUT) out->out[n_out++] = j;
    __syncthreads();
  }
  const int n_forced = forced_total;
  // Whatever ROUTE_JMAX columns leave, the (ROUTE_JMAX + 1)-th largest value of every quantity stays.  A lower bound on it without
  // sorting: the (ROUTE_JMAX + 1 - forced)-th largest of the 64 class maxima (that many DISTINCT columns are at least as large).
  if (wave == 0) {
    const int want = ROUTE_JMAX - n_forced;        // 0-based rank among the class maxima
#pragma unroll 1
    for (int i = 0; i < NVAL; i++) {
      const unsigned long long mine = group_max[i][lane];
      int rank = 0;
      for (int k = 0; k < 64; k++) {
        const unsigned long long o = group_max[i][k];
        rank += (o > mine || (o == mine && k < lane));
      }
      if (rank == want) floor_of[i] = __longlong_as_double((long long)mine);
    }
  }
  __syncthreads();
  double fl[NVAL];
#pragma unroll
  for (int i = 0; i < NVAL; i++) fl[i] = floor_of[i];
  bool top_valid = true;
  for (int P = 5; P <= 6; P++) {
    if (n > 64 && route_violation(fl, P, tau_x) > tolerance) continue;   // hopeless for this P (uniform: every thread computes the same)
    n_out = n_forced;
    for (;;) {
      if (!top_valid) {             // (the first look uses pass 1's result)
        top_valid = true;
#pragma unroll
        for (int i = 0; i < NVAL; i++) t[i] = Top2{-1.0, 0x7fffffff, -1.0};
        constexpr int PASS1_COLS = 4;     // (four columns' loads in flight per thread)
        for (int j0 = tid; j0 < n; j0 += ROUTE_THREADS * PASS1_COLS) {
          double v[PASS1_COLS][NVAL];
          int ex[PASS1_COLS];
#pragma unroll
          for (int c = 0; c < PASS1_COLS; c++) {
            const int j = j0 + c * ROUTE_THREADS;
            ex[c] = j < n ? emax[j] : EMAX_COLUMN_OUT;
#pragma unroll
            for (int i = 0; i < NVAL; i++) v[c][i] = j < n ? vals[(int64_t)i * n + j] : 0.0;
          }
#pragma unroll
          for (int c = 0; c < PASS1_COLS; c++)
            if (!(ex[c] & EMAX_COLUMN_OUT)) {
#pragma unroll
              for (int i = 0; i < NVAL; i++) top2_merge(t[i], Top2{v[c][i], j0 + c * ROUTE_THREADS, -1.0});
            }
        }
        reduce_top(t);
      }
      if (tid == 0) {
        double A[NVAL];
#pragma unroll
        for (int i = 0; i < NVAL; i++) {
          top[i].m1 = fmax(top[i].m1, 0.0);
          top[i].m2 = fmax(top[i].m2, 0.0);
          A[i] = top[i].m1;
        }
        decision = -1;
        if (route_violation(A, P, tau_x) <= tolerance) {
          out->planes = P;
          out->n_out = n_out;
          route_terms(A, P, out->sq, out->x);
          out->rho = 2.0 * A[6] + A[6] * A[6];
          if (P == 6) atomicOr(flag, 1);
          if (route_counts && n_out) atomicAdd(route_counts + 3, n_out);
          decision = 0;
        } else if (n_out == ROUTE_JMAX) {
          decision = 1;
        } else {
          int best = -1;
          double best_v = 1e308;
          for (int qi = 0; qi < NVAL; qi++) {      // candidates: the columns that hold a maximum, in quantity order
            const int c = top[qi].a1;
            if (c == 0x7fffffff) continue;
            double A2[NVAL];
#pragma unroll
            for (int i = 0; i < NVAL; i++) A2[i] = top[i].a1 == c ? top[i].m2 : top[i].m1;
            const double v = route_violation(A2, P, tau_x);
            if (v < best_v) { best_v = v; best = c; }
          }
          if (best < 0) decision = 1;              // (no column left)
          else {
            emax[best] |= EMAX_COLUMN_OUT;
            out->out[n_out] = best;
          }
        }
      }
      __syncthreads();
      const int dec = decision;
      __syncthreads();
      if (dec == 0) return;
      if (dec == 1) break;
      n_out++;
      top_valid = false;
    }
    // this P cannot be reached: take the greedy picks back (the forced columns stay out)
    if (n_out > n_forced) {
      for (int j = tid; j < n; j += ROUTE_THREADS)
        if ((emax[j] & EMAX_COLUMN_OUT) && (emax[j] & 255) != 255) emax[j] &= 255;
      top_valid = false;
      __syncthreads();
    }
  }
  if (tid == 0) {
    out->planes = 0; out->n_out = 0; out->sq = out->x = out->rho = 0.0;
    atomicOr(flag, 2);
  }
  for (int j = tid; j < n; j += ROUTE_THREADS) emax[j] &= 255;
}

// Columns the route took off the int8 path no longer matter to the product -- but their digits would still cost it: a bulk 12
// binades under its spikes puts a nonzero into plane 3 of every piece of its 32-row group, and the five-plane kernel then runs
// that group's deep-plane blocks in every k-step of every tile of its row and column block (measured: +2 % on the whole launch
// for four such columns, through the tiles' per-step barrier).  So their rows of the digit planes are zeroed and the piece masks
// of their groups recomputed: one wave per (column, k-step), lane = (half, row) -- a piece is read as the product kernel reads
// it, 1 KB per plane.  Enqueued with every call; every workgroup exits at once when no column left.
__global__ __launch_bounds__(256) void i8_clear_columns_kernel(const RouteOut* route, const int* flag, const int* emax, signed char* planes,
                                                               unsigned char* zmask, int n, int nk) {
  if ((*flag & 2) || (int)blockIdx.x >= route->n_out) return;
  const int j = route->out[blockIdx.x];
  const int64_t groups = n / 32;
  const int G = j >> 5, lane = threadIdx.x & 63, row = lane & 31;
  const bool row_out = (emax[G * 32 + row] & EMAX_COLUMN_OUT) != 0;     // (every column of this group that left, not only j)
  for (int kt = blockIdx.y * 4 + (threadIdx.x >> 6); kt < nk; kt += gridDim.y * 4) {
    const unsigned old_mask = zmask[(int64_t)kt * groups + G];
    unsigned new_mask = 0;
#pragma unroll
    for (int s = 0; s < NP; s++) {
      // planes the split pass did not write here (all-zero pieces of planes 4, 5) are not touched: nothing reads them
      if (s >= ALWAYS_WRITTEN_PLANES && (old_mask >> s) == 0) continue;
      i32x4* p = (i32x4*)(planes + ((s * groups + G) * (int64_t)nk + kt) * 1024) + lane;
      i32x4 v = *p;
      if (row_out) {
        v = (i32x4)0;
        *p = v;
      }
      if (__ballot((v[0] | v[1] | v[2] | v[3]) != 0)) new_mask |= 1u << s;
    }
    if (lane == 0) zmask[(int64_t)kt * groups + G] = (unsigned char)new_mask;
  }
}

template <int V> struct ic { static constexpr int value = V; };

// One statistic of a launch.  block == 0: sigma is n x n, lower triangle; block == 128: sigma is [n / 128][128][128] (per-head
// Grams of a [tokens][heads x 128] activation): only the diagonal 128 x 128 tiles exist, element (row, col) of head row / 128
// lives at sigma[row * ld_sigma + col - 128 (row / 128)] with ld_sigma = 128.
struct SyrkProblem {
  const signed char* planes;
  const int* emax;
  const unsigned char* zmask;          // [nk][n / 32] piece masks written by the split pass (write_piece_mask)
  double* sigma;
  int64_t ld_sigma;
  int n, block;
};
constexpr int MAX_PROBLEMS = 4;        // the four hooks of a layer: sigma_mlp, sigma_x, sigma_q, sigma_k
constexpr int CODE_PROB = 28, CODE_BI = 14;   // tile code = problem << 28 | bi << 14 | bj

struct SyrkArgs {
  SyrkProblem prob[MAX_PROBLEMS];
  int nprob, nk;
  unsigned long long* mfma_count;      // += v_mfma instructions this launch executed (the dense count is known on the host)
  const int* route_flag;               // [nprob] per statistic, written by i8_depth_kernel: bit 0 -> needs six planes, bit 1 -> the fp64 kernel;
                                       // see launch_route()
  int* route_counts;                   // optional device counters [five planes, six planes, fp64 fallback], += 1 by the launch that runs
  const int2* sched;                   // persistent launch: [ngroups][32] entries {tile code (-1 = none), k-chunk code
                                       // (0 = all k-steps; else slot << 10 | Q << 5 | q: chunk q of Q, folded into partial tile `slot`)};
                                       // nullptr = one tile per workgroup
  int ngroups;
  double* partial;                     // [slot][128][TJ] fp64 partial tiles of the k-split last round (zeroed per call; i8_tail_combine_kernel)
  const int4* tail;                    // [n_tail] {tile code, Q, first slot, 0}: the tiles of the split round
  int* xcd_arrive;                     // [8] arrival counters of the round barrier (zeroed per call)
  const int* exact_state;              // nullptr: the exact route is not on offer for this call; else {overflow, ran} written by i8_extract_lo_kernel
#ifdef MDG_I8_STAMPS
  unsigned long long* stamps;          // diagnostic build only: per (workgroup, wave) cycle sums of the k-step phases
#endif
#ifdef MDG_I8_WGTIMES
  unsigned long long* wgtimes;         // diagnostic build only: [256][2 + 32] wall clock (100 MHz) at workgroup start / end / after each tile
#endif
};
// The route of a launch from the per-statistic flags: a statistic with bit 1 set leaves the int8 path ALONE (its tiles are
// skipped here, a gated mdg_cov_accum launch does it); the others share the launch on six planes if any of them asks for six,
// else on five.  Returns 0 / 1 (five / six planes), or -1 when no statistic is left on the int8 path; `fallbacks` = how many left.
__device__ __forceinline__ int launch_route(const SyrkArgs& a, int& live, int& fallbacks) {
  int six = 0;
  live = fallbacks = 0;
  for (int p = 0; p < a.nprob; p++) {
    const int f = a.route_flag[p];
    if (f & 2) fallbacks++;
    else {
      live++;
      six |= f & 1;
    }
  }
  return live ? six : -1;
}

// The EXACT route (section "exact route" below): the three top digit planes through the five-plane kernel with every deeper plane
// masked off -- all nine plane pairs of the 24-bit part, nothing truncated -- and the remainder of the elements that have one
// through the fp64 remainder kernel.  On offer when the remainder lists were built and none overflowed; it then serves the
// statistics of BOTH legacy classes (five and six planes) of the launch.
__device__ __forceinline__ bool exact_route(const SyrkArgs& a) {
  return a.exact_state && a.exact_state[1] == 1 && a.exact_state[0] == 0;
}
// Does the P-plane product launch of this call do the work?  (0: no; 1: the truncated product of P planes; 2: the exact route --
// P = 3: the launch of the three top planes, all nine pairs)
template <int P>
__device__ __forceinline__ int product_launch_runs(const SyrkArgs& a, int& live, int& fallbacks) {
  const int route = launch_route(a, live, fallbacks);
  if (route < 0) return 0;
  if (exact_route(a)) return P == 3 ? 2 : 0;
  if (P == 3) return 0;
  return route == (P == 5 ? 0 : 1) ? 1 : 0;
}

#ifdef MDG_I8_STAMPS
#define MDG_STAMP(x) x = __builtin_amdgcn_s_memtime()
constexpr int STAMP_WGS = 1024;
#else
#define MDG_STAMP(x)
#endif

// Shape and LDS ring per route.  One workgroup of 8 waves per CU (two waves per SIMD, <= 256 registers each).
//   P = 5: 128 x 128 tile, wave tile 64 x 32 (160 accumulators), stages of 40 KB -- 40 KB of L2 -> LDS traffic per k-step for
//          16384 outputs where two 128 x 64 tiles move 60 KB.  Ring of 3 stages, filled two k-steps ahead.
//   P = 6: 128 x 64 tile, wave tile 32 x 32 (96 accumulators), stages of 36 KB.  Ring of 4 stages, filled three k-steps ahead;
//          a stage is therefore complete one barrier before it is multiplied, and a wave reads the next step's fragments right
//          after its last MFMA of this one (their latency runs under its load issue / the barrier).
// What a k-step costs besides its MFMAs, by s_memtime stamps (diagnostic build -DMDG_I8_STAMPS; five planes, Gaussian
// columns, 18.8 MFMAs per wave and step = 1203 matrix-pipe cycles per SIMD): in the first versions (2-stage ring, every wave:
// barrier -> its 5 stage loads -> fragment reads -> MFMAs) a step took ~2500 cycles -- ~700-800 of them spent by all eight
// waves side by side on ~100 instructions of mask decoding (clz / med3 on the VALU), 64-bit address updates and LDS-DMA
// issue while no wave multiplied, then ~1360 on the MFMAs (the younger wave of each SIMD loses the arbitration and finishes
// last; the older one idles ~700 at the next barrier).  Two changes:
//   * the stage loads are driven by per-wave piece descriptors held in SGPRs (base address, LDS offset, mask byte position,
//     plane bits), ~9 scalar instructions per piece, the address an SGPR base + one VGPR offset shared by all pieces
//     (issue_stage): ~420-540 cycles for the 5 loads -- what is left is the LDS-DMA instruction itself, which holds its wave
//     ~85-100 cycles at issue;
//   * the two waves of a SIMD take OPPOSITE orders inside a k-step (roles by wave number >= 4, MI355X_MICROARCH.md "Two waves
//     per SIMD" item 9): waves 4-7 issue their share of the stage loads right after the barrier and multiply afterwards;
//     waves 0-3 multiply first and issue their loads at the end of the step -- one wave's load issue runs under its partner's
//     MFMAs.  A wave waits for its own loads (vmcnt(0)) right before it issues the next ones, a whole k-step after they went
//     out, so the wait is free and needs no load count (the number of pieces a wave loads varies with the zero-plane skipping).
// Five planes 26.7 -> 24.2 ms per sigma_mlp call (~1970 cycles per step), six planes 46.4 -> 37.9 ms.  Measured and dropped on
// the way: a ping-pong with a second barrier per step (one wave of a SIMD only loads while the other only multiplies: 50.8 ms
// -- an LDS-DMA issue beside a partner that issues MFMAs back to back takes 380 cycles instead of 85, s_setprio changes
// nothing); the loads dealt out between a wave's own MFMAs (EXEC = 0 for skipped pieces, the accumulators as asm operands
// to pin the order: 27.0 / 41.6 ms -- in lock-step both waves of a SIMD stall in their load issue together); fragment reads
// ahead of the load issue; static s_setprio 1 for waves 4-7 (26.4 ms); four stages + fragment prefetch for five planes too
// (24.7 ms); super-blocks of 1 / 4 x 4 tiles for six planes (39.4 / 39.0 ms).
// Build-time variants that were measured and dropped (lock-step round barrier, fixed tile lists, returnless atomic fold, the
// narrow five-plane tile, every-wave-loads-first order, deferred MFMAs on six planes, whole tiles in the last round, the
// timing experiments) live in scripts/probes/cov_i8_variants.patch with their numbers; what is compiled here is the shipped
// path.  Two diagnostic builds remain: -DMDG_I8_STAMPS (s_memtime phases of a k-step) and -DMDG_I8_WGTIMES (per-workgroup
// wall clock).
constexpr int SB5 = 2, SB6 = 2;       // one-tile-per-workgroup launches (n < 2048): super-blocks of 2 x 2 (2 x 4) tiles per XCD
constexpr int PERSISTENT_MIN_ROWS = 16;   // statistics of at least this many 128-row blocks (n >= 2048) run as the persistent launch
constexpr int DEFER5 = 4;             // MFMAs a loads-first wave of the five-plane kernel holds back across the barrier (0: 25.3, 2: 26.0, 3: 24.9, 4: 24.6, 5: 25.0, 6: 27.8 ms per call)
constexpr int NW = 8;                 // waves per workgroup
constexpr int RING5 = 3;              // LDS stages of the five-plane kernel (six planes: 4)
// (measured at the sigma_mlp shape, Gaussian / SiLU-gated columns, product launch alone: three stages without fragment prefetch
//  19.5 / 20.3 ms; four stages 19.6 / 20.4; fragment prefetch with four, five or six stages 39 - 40 ms -- hipcc then keeps two sets
//  of fragments beside the 160 accumulators and spills inside the loop.  The five-plane kernel with the deeper planes masked off,
//  which this instantiation replaced: 20.6 / 21.5 ms; the truncated five- / six-plane products: 21.4 / 35.5 ms.)
#ifndef MDG_I8_RING3
#define MDG_I8_RING3 3
#endif
#ifndef MDG_I8_PREFETCH3
#define MDG_I8_PREFETCH3 0
#endif
constexpr int RING3 = MDG_I8_RING3;   // LDS stages of the three-plane (exact route) kernel: 24 KB per k-step each
// k-steps per LDS stage, i.e. per workgroup barrier (three planes only: nothing in that k-step is conditional).  The loads of a stage
// are the same 1 KB pieces, twice as many per issue; what halves is the number of barriers and role switches per MFMA.
// Measured at the sigma_mlp shape, Gaussian columns, product launch alone, one box (scripts/probes/p3_variants.sh,
// profiles/r04_p3_variants.log): one k-step per stage (ring of 3, 4 deferred MFMAs) 19.37 ms; two k-steps (ring of 3 x 48 KB) with
// 0 / 2 / 4 / 6 / 8 / 10 / 12 / 14+ deferred 19.47 / 19.21 / 18.98 / 18.73 / 18.56 / 20.2 / 21.3 / 23.2; three k-steps in a ring of
// two 20.04.
#ifndef MDG_I8_KSS3
#define MDG_I8_KSS3 2
#endif
#ifndef MDG_I8_DEFER3
#define MDG_I8_DEFER3 8     // MFMAs a loads-first wave of the three-plane kernel holds back across the barrier (see DEFER5)
#endif
constexpr int steps_per_stage(int planes) { return planes == 3 ? MDG_I8_KSS3 : 1; }
constexpr bool wide_tile(int planes) { return planes != 6; }   // 128 x 128 tiles (six planes: 128 x 64)
constexpr int ring_depth(int planes) { return planes == 3 ? RING3 : wide_tile(planes) ? RING5 : 4; }
// fragments of the next k-step read right behind this step's MFMAs (needs a stage that is complete a barrier early: RING >= 4)
constexpr bool prefetch_frags(int planes) { return planes == 3 ? (MDG_I8_PREFETCH3 != 0 && RING3 >= 4) : !wide_tile(planes); }
// P = 3 is the product of the EXACT route: planes 0 .. 2 only, ALL nine plane pairs (classes 0 .. 4), no piece masks -- the same tile
// code with nothing conditional left in the k-step (24 KB stages, 36 fragment registers beside the 160 accumulators)
constexpr int classes_of(int planes) { return planes == 3 ? 5 : planes; }

// One output tile (bi, bj) of the lower region: bi = 128-row block, bj = TJ-row block (bj <= bi for 128 x 128 tiles, bj <= 2 bi + 1
// for 128 x 64); the k-steps [kb, ke), then the fold: element (row, col) of the statistic goes to
// fold[(row - fold_row0) * fold_ld + col - fold_col0] (sigma itself, or a partial tile of the k-split last round).
// `executed` += the MFMAs this wave issued.
template <int P>
__device__ __forceinline__ void i8_syrk_tile(const SyrkArgs& a, const SyrkProblem& pr, const int bi, const int bj, const int kb, const int ke, double* const fold,
                                             const int64_t fold_ld, const int fold_row0, const int fold_col0, unsigned char* lds,
                                             unsigned& executed, const unsigned mask_and) {
  constexpr int WB = wide_tile(P) ? 2 : 1;         // 32-row blocks of a wave tile: 64 x 32, or 32 x 32 (96 accumulators at P = 6)
  constexpr int TJ = WB == 2 ? 128 : 64;           // tile columns (rows of the J operand); waves are laid out (128 / 32 WB) x (TJ / 32)
  constexpr int PB = TJ * KS;
  constexpr int GA = TI / 32, GB = TJ / 32;        // 32-row groups (1 KB pieces per plane and stage) of the two operands
  constexpr int WCOLS = TJ / 32;
  constexpr int KSS = steps_per_stage(P);          // k-steps per stage (per barrier)
  constexpr int STEP_BYTES = P * (PA + PB);        // 40 KB (P = 5, 128 x 128) / 36 KB (P = 6, 128 x 64) / 24 KB (P = 3)
  constexpr int STAGE_BYTES = KSS * STEP_BYTES;
  constexpr int PIECES = (GA + GB) * P;            // 1 KB pieces per k-step
  constexpr int RING = ring_depth(P);              // LDS stages
  constexpr bool PREFETCH = prefetch_frags(P);     // the next step's fragments are read before the barrier (needs RING >= 4)
  static_assert(KSS == 1 || (P == 3 && !PREFETCH), "several k-steps per stage: the unconditional three-plane k-step only");
  constexpr int NCLS = classes_of(P);              // digit classes s + t kept: 0 .. NCLS - 1
  // the wave index through readfirstlane: hipcc then knows it is wave-uniform and the staging code becomes scalar (SGPR piece
  // addresses, s_cbranch on the piece tests, M0 from SGPRs) instead of exec-masked branches with a v_readfirstlane per piece
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const bool loads_first = wave >= NW / 2;
  const int wr = wave / WCOLS, wc = wave % WCOLS;
  const int64_t groups = pr.n / 32;
  const int nk = a.nk;

  // staging: (GA + GB) P pieces of 1 KB per stage (A: P planes x 4 row groups, B: P planes x 2 or 4); wave w issues pieces w, w + NW, ...
  // mA / mB: piece masks of the stage's A and B row groups, one byte per 32-row group; planes at or beyond a group's depth
  // (group_depth below) are all-zero there in this k-step and are neither loaded nor multiplied
  constexpr int MIN_DEPTH = P == 3 ? 3 : P - 2;   // planes below this are always staged and multiplied (3 of five, 4 of six; all three of three)
  static_assert(MIN_DEPTH <= ALWAYS_WRITTEN_PLANES, "the split pass leaves all-zero pieces of the deeper planes unwritten");
  auto group_depth = [&](unsigned m, int g) {   // 1 + deepest plane with a nonzero in group g, but at least MIN_DEPTH
    const unsigned byte = (m >> (8 * g)) & 0xFFu;
    return max(MIN_DEPTH, min(P, 32 - __builtin_clz(byte | 1u)));
  };
  // Per-wave piece descriptors, all wave-uniform (SGPRs), set up once: the k-step loop then spends ~8 scalar instructions per
  // piece on the test "does this piece hold a nonzero" + M0 + one LDS-DMA load whose address is SGPR base + one VGPR offset
  // (lane * 16 + k-step * 1024) shared by all pieces.  (First version: a running 64-bit address per piece, depth through
  // clz / med3 on the VALU, exec-masked branches -- ~100 instructions per k-step and wave, 700-800 cycles by s_memtime stamps,
  // during which no wave of the workgroup multiplied.)
  constexpr int NQ = (PIECES + NW - 1) / NW;
  unsigned long long pc_base[NQ];
  unsigned pc_loff[NQ], pc_shift[NQ], pc_cmask[NQ], pc_force[NQ];
  bool pc_valid[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const int p = wave + NW * q;
    pc_valid[q] = (PIECES % NW == 0 && q < PIECES / NW) || p < PIECES;
    const bool isA = p < GA * P;
    const int pp = isA ? p : p - GA * P;
    const int s = isA ? pp / GA : pp / GB, g = isA ? pp % GA : pp % GB;
    const int64_t G = (isA ? bi * (TI / 32) : bj * (TJ / 32)) + g;
    const unsigned long long base = (unsigned long long)(uintptr_t)pr.planes + (unsigned long long)((s * groups + G) * (int64_t)nk) * 1024ull;
    pc_base[q] = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32) |
                 (unsigned)__builtin_amdgcn_readfirstlane((unsigned)base);
    pc_loff[q] = (isA ? s * PA : P * PA + s * PB) + g * 1024;
    pc_shift[q] = (isA ? 0 : 32) + 8 * g;
    pc_cmask[q] = (0xFFu << s) & 0xFFu;          // bits s .. 7 of the group's mask byte: some plane >= s holds a nonzero
    pc_force[q] = s < MIN_DEPTH ? 1u : 0u;   // planes below MIN_DEPTH are always staged (the unconditional MFMA block reads them)
  }
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);
  const unsigned lane16 = lane * 16;
  auto issue_stage = [&](int kt, int buf, unsigned mA, unsigned mB) {
    const unsigned long long m64 = ((unsigned long long)mB << 32) | mA;
    const unsigned lbase = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_BYTES);   // (wave-uniform; says so to the compiler)
#pragma unroll
    for (int kk = 0; kk < KSS; kk++) {
      if (KSS > 1 && kt + kk >= ke) break;   // (the last stage of a tile or k-chunk may hold fewer k-steps)
      const unsigned voff = lane16 + (unsigned)(kt + kk) * 1024u;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        if (!pc_valid[q]) continue;
        const unsigned present = ((unsigned)(m64 >> pc_shift[q]) & pc_cmask[q]) | pc_force[q];
        if (present)   // (an all-zero piece is not loaded: nothing will read it)
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lbase + kk * STEP_BYTES + pc_loff[q]), "v"(voff), "s"(pc_base[q])
                       : "memory");   // (M0 is written; nothing the compiler emits in this kernel reads it)
      }
    }
  };

  i32x16 acc[NCLS][WB];
#pragma unroll
  for (int k = 0; k < NCLS; k++)
#pragma unroll
    for (int b = 0; b < WB; b++) acc[k][b] = (i32x16)0;

  // sigma[i][j] += 2^(E_i + E_j - 344) * sum_k acc_k 256^(10 - k)  =  (sum_k acc_k 2^(80 - 8k)) * 2^(E_i - 172) * 2^(E_j - 172)
  auto flush = [&]() {
    const int col = bj * TJ + wc * 32 + (lane & 31);
    int row0 = bi * TI + wr * 32 * WB + 4 * (lane >> 5);
    // opaque to the optimiser: otherwise the 32 element addresses are computed once, ahead of the MFMA loop, spilled (the
    // accumulators fill the register file there), and reloaded here behind one s_waitcnt vmcnt(0) each -- which turns the 16
    // sigma loads of a block into 16 serialised memory round trips (26 us per flush and tile, 2 x 0.65 ms per launch)
    asm volatile("" : "+v"(row0));
    const int e_col = pr.emax[col];      // bits 0-7: the column's maximum exponent; EMAX_COLUMN_OUT: the fp64 column kernel computes this column
    const double sc_j = ldexp(1.0, (e_col & 255) - 172);
    // all read-modify-writes of a lane: loads first (independent, in flight together), then the arithmetic and the stores;
    // written as `*p += v` one by one the compiler must keep them in order and every element pays a full memory round trip
#pragma unroll
    for (int b = 0; b < WB; b++) {  // one 32-row block at a time: 16 loads in flight per lane
      double* const p = fold + (int64_t)(row0 + b * 32 - fold_row0) * fold_ld + (col - fold_col0);
      const int* const e = pr.emax + row0 + b * 32;
      int er[16];
      double old[16];
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const int off = (reg & 3) + 8 * (reg >> 2);
        old[reg] = p[(int64_t)off * fold_ld];
        er[reg] = e[off];
      }
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const int off = (reg & 3) + 8 * (reg >> 2);
        double v = 0.;
#pragma unroll
        for (int k = NCLS - 1; k >= 0; k--) v += ldexp((double)acc[k][b][reg], 80 - 8 * k);
        // rows and columns the route handed to the fp64 column kernel are not ours: their digit products are computed and dropped
        if (col <= row0 + b * 32 + off && !((e_col | er[reg]) & EMAX_COLUMN_OUT))
          p[(int64_t)off * fold_ld] = old[reg] + v * sc_j * ldexp(1.0, (er[reg] & 255) - 172);
      }
    }
#pragma unroll
    for (int k = 0; k < NCLS; k++)
#pragma unroll
      for (int b = 0; b < WB; b++) acc[k][b] = (i32x16)0;
    // the stores above share the VM counter with the LDS-DMA loads: drain, so that the loop's waits see stage loads only
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // piece masks: uniform-address loads; issued together with a stage's loads, for the stage after it
  const int64_t mgroups = pr.n / 32;
  auto load_masks = [&](int kt, unsigned& va, unsigned& vb) {
    if (P == 3) return;                      // (every piece of the three planes is staged: no masks)
    const unsigned* z = (const unsigned*)(pr.zmask + (int64_t)kt * mgroups);   // n / 32 is a multiple of 4: dword-aligned rows
    va = z[bi];                                                               // groups 4 bi .. 4 bi + 3
    vb = TJ == 128 ? z[bj] : z[bj >> 1];   // groups 4 bj .. + 3; or 2 bj, 2 bj + 1 in one half of the dword (see b_half)
  };
  // 128 x 64 tiles: the B panel's two mask bytes are one half of the loaded dword
  auto b_half = [&](unsigned m) { return TJ == 128 ? m : (m >> ((bj & 1) * 16)) & 0xFFFFu; };
  // mask_and: all ones, or 0x07070707 on the exact route -- planes 3 .. 5 are then "absent" everywhere: neither loaded nor multiplied
  // here (the remainder kernel has them).  Applied where a loaded mask dword is USED, a k-step after its load was issued -- an
  // operation on the freshly loaded value would make hipcc wait for it at once (vmcnt(0) in front of the LDS-DMA issue: 22.5 ->
  // 28.7 ms per sigma_mlp launch when the clamp first sat in load_masks)
  auto wait_loads = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  // masks of the stages kt (being multiplied) .. kt + D (the one this step issues) -- SGPRs; vA / vB: the loaded dwords of the
  // stage after those, in flight
  constexpr int D = RING - 1;
  unsigned mA[D + 1], mB[D + 1], vA = ~0u, vB = ~0u;
#pragma unroll
  for (int i = 0; i <= D; i++) mA[i] = mB[i] = ~0u;
#pragma unroll
  for (int i = 0; i < D; i++)
    if (KSS == 1 && kb + i < ke) {
      unsigned t0, t1;
      load_masks(kb + i, t0, t1);
      mA[i] = __builtin_amdgcn_readfirstlane(t0) & mask_and;
      mB[i] = b_half(__builtin_amdgcn_readfirstlane(t1)) & mask_and;
    }
  if (KSS == 1 && ke - kb > D) load_masks(kb + D, vA, vB);
#pragma unroll
  for (int i = 0; i < D; i++)
    if (kb + i * KSS < ke) issue_stage(kb + i * KSS, i, mA[i], mB[i]);
  wait_loads();
  int buf = 0;                 // (kt - kb) % RING
#ifdef MDG_I8_STAMPS
  unsigned long long ta = 0, tb = 0, tc = 0, td = 0, te = 0, s_wait = 0, s_issue = 0, s_comp = 0, s_tail = 0, t_begin;
  const unsigned executed_before = executed;
  MDG_STAMP(t_begin);
#endif
  const int r = lane & 31, h = lane >> 5;
  // fragments of the planes below MIN_DEPTH (always staged, always multiplied): ONE set of reads feeds all their pairs
  i32x4 fa[MIN_DEPTH][WB], fb[MIN_DEPTH];
  auto load_frags = [&](int stage_buf, int kk = 0) {
    const unsigned char* base = lds + stage_buf * STAGE_BYTES + kk * STEP_BYTES;
#pragma unroll
    for (int s = 0; s < MIN_DEPTH; s++) {
#pragma unroll
      for (int b = 0; b < WB; b++) fa[s][b] = *(const i32x4*)(base + s * PA + (wr * WB + b) * 1024 + h * 512 + r * 16);
      fb[s] = *(const i32x4*)(base + P * PA + s * PB + wc * 1024 + h * 512 + r * 16);
    }
  };
  // the deeper planes of a step, each present one a block of its own (fragment read + its pairs).  A deep plane only pairs with
  // planes 0 (and 1) of the other panel (s + t < P), so the blocks are independent and simply add:
  //   P = 5: 9 pairs + 2 [dA > 3] + 2 [dB > 3] + [dA > 4] + [dB > 4];   P = 6: 15 + 2 [dA > 4] + 2 [dB > 4] + [dA > 5] + [dB > 5]
  // (branching around single MFMAs / fragment reads instead makes hipcc put an lgkmcnt(0) in front of every LDS read; nine
  // straight-line variants behind a switch make it spill the 160 accumulators at the merges)
  auto deep_planes = [&](int stage_buf, unsigned mAk, unsigned mBk) {
    const unsigned char* base = lds + stage_buf * STAGE_BYTES;
    int dAb[WB];
#pragma unroll
    for (int b = 0; b < WB; b++) dAb[b] = group_depth(mAk, wr * WB + b);
    const int dBw = group_depth(mBk, wc);
    int deep_mfmas = 0;
#pragma unroll
    for (int d = MIN_DEPTH; d < P; d++) {
#pragma unroll
      for (int b = 0; b < WB; b++)
        if (dAb[b] > d) {   // plane d of A block b with planes t < P - d of B (all below MIN_DEPTH: already in registers)
          const i32x4 fd = *(const i32x4*)(base + d * PA + (wr * WB + b) * 1024 + h * 512 + r * 16);
#pragma unroll
          for (int t = 0; t < P - d; t++) acc[d + t][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fd, fb[t], acc[d + t][b], 0, 0, 0);
          deep_mfmas += P - d;
        }
      if (dBw > d) {        // plane d of the B block with planes s < P - d of both A blocks
        const i32x4 fd = *(const i32x4*)(base + P * PA + d * PB + wc * 1024 + h * 512 + r * 16);
#pragma unroll
        for (int s2 = 0; s2 < P - d; s2++)
#pragma unroll
          for (int b = 0; b < WB; b++)
            acc[s2 + d][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s2][b], fd, acc[s2 + d][b], 0, 0, 0);
        deep_mfmas += (P - d) * WB;
      }
    }
    executed += deep_mfmas;
  };
  constexpr int UNCOND_PAIRS = P == 6 ? 15 : 9;   // pairs (s, t), s, t < MIN_DEPTH, s + t < NCLS
  constexpr int N_UNCOND = UNCOND_PAIRS * WB;                               // unconditional MFMAs per wave and k-step
  // of them, held back across the barrier by the loads-first waves (six planes: none -- 37.2 ms per call without, 55 ms with 3 - 5
  // deferred: the loads-first waves then lose their fragment prefetch)
  constexpr int DEFER = PREFETCH ? 0 : P == 3 ? MDG_I8_DEFER3 : DEFER5;
  auto rotate = [&]() {
#pragma unroll
    for (int i = 0; i < D; i++) {
      mA[i] = mA[i + 1];
      mB[i] = mB[i + 1];
    }
    buf = buf == RING - 1 ? 0 : buf + 1;
  };
  const auto ahead = [&](int d) { int x = buf + d; return x >= RING ? x - RING : x; };   // (kt + d) % RING
  if (PREFETCH) {
    __builtin_amdgcn_s_barrier();   // stages 0 .. D - 1 complete (every wave waited for its share)
    if (!(DEFER && loads_first)) load_frags(0);
  }
  // two loops: the int32 classes are folded into sigma between runs of FLUSH_STEPS k-steps, outside the MFMA loop (a
  // conditional flush inside it makes the compiler shuttle all 160 accumulators between AGPRs and VGPRs every step)
  constexpr int FOLD_STEPS = FLUSH_STEPS - FLUSH_STEPS % KSS;   // (a fold falls between two stages)
  for (int k0 = kb; k0 < ke; k0 += FOLD_STEPS) {
    const int k1 = min(ke, k0 + FOLD_STEPS);
    // Roles: the two waves of a SIMD take opposite orders inside a k-step.  Waves 4-7 issue their share of stage kt + D right
    // after the barrier and multiply afterwards; waves 0-3 multiply first and issue at the end of the step (after waiting for
    // their previous loads, a whole k-step old by then) -- one wave's ~450 cycles of LDS-DMA issue run under its partner's MFMAs.
    for (int kt = k0; kt < k1; kt += KSS) {
      MDG_STAMP(ta);
      if (loads_first) wait_loads();  // this wave's loads of the previous step
      __builtin_amdgcn_s_barrier();   // stage kt (PREFETCH: kt + 1 too) complete in LDS, stage kt - 1 no longer read
      auto refill = [&]() {           // stage kt + D into the buffer stage kt - 1 just left; masks of the stage after it behind it
        mA[D] = __builtin_amdgcn_readfirstlane(vA) & mask_and;
        mB[D] = b_half(__builtin_amdgcn_readfirstlane(vB)) & mask_and;
        if (kt + D * KSS < ke) issue_stage(kt + D * KSS, ahead(D), mA[D], mB[D]);
        if (KSS == 1 && kt + D + 1 < ke) load_masks(kt + D + 1, vA, vB);
      };
      MDG_STAMP(tb);
      // the unconditional MFMAs of a step, in (s, t, block) order; [lo, hi) selects a run of them
      auto mfma_run = [&](int lo, int hi) {
        int idx = 0;
#pragma unroll
        for (int s = 0; s < MIN_DEPTH; s++)
#pragma unroll
          for (int t = 0; t < MIN_DEPTH; t++)
            if (s + t < NCLS) {
#pragma unroll
              for (int b = 0; b < WB; b++) {
                if (idx >= lo && idx < hi) acc[s + t][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s][b], fb[t], acc[s + t][b], 0, 0, 0);
                idx++;
              }
            }
      };
      if (loads_first) {
        // the loads-first waves keep the last DEFER MFMAs of the previous step back until here: they run while their SIMD
        // partner, which multiplies first, is still waiting for its fragment reads -- the matrix pipe would idle ~150 cycles
        // at every step boundary otherwise (fragments of the previous step are still in this wave's registers: it re-reads
        // them only after its loads are out)
        if (DEFER && kt > k0) {
          mfma_run(N_UNCOND - DEFER, N_UNCOND);
          __builtin_amdgcn_sched_barrier(0);
        }
        refill();
        __builtin_amdgcn_sched_barrier(0);
      }
      MDG_STAMP(tc);
      if (!PREFETCH || (DEFER && loads_first)) load_frags(buf);
      if (KSS > 1) {
        // the stage's k-steps but the last, whole: the fragment registers are re-read once their MFMAs are issued (the SIMD's other
        // wave, a stage's half out of phase, has the matrix pipe meanwhile)
#pragma unroll
        for (int kk = 1; kk < KSS; kk++)
          if (kt + kk < k1) {
            mfma_run(0, N_UNCOND);
            executed += UNCOND_PAIRS * WB;
            __builtin_amdgcn_sched_barrier(0);
            load_frags(buf, kk);
          }
      }
      mfma_run(0, N_UNCOND - DEFER);
      if (DEFER == 0 || !loads_first) mfma_run(N_UNCOND - DEFER, N_UNCOND);
      deep_planes(buf, mA[0], mB[0]);
      executed += UNCOND_PAIRS * WB;
      if (PREFETCH) {
      // next step's fragments: stage kt + 1 has been complete since THIS step's barrier (its loads went out three steps ago
      // and every wave waited for its share before the barrier), so the read latency hides behind the refill / the barrier
      __builtin_amdgcn_sched_barrier(0);   // (not before the MFMAs above are issued: the fragment registers are theirs until then)
      if (kt + 1 < ke && !(DEFER && loads_first)) load_frags(ahead(1));
      }
      MDG_STAMP(td);
      if (!loads_first) {
        __builtin_amdgcn_sched_barrier(0);
        wait_loads();       // this wave's loads of the previous step
        refill();
      }
      MDG_STAMP(te);
#ifdef MDG_I8_STAMPS
      s_wait += tb - ta; s_issue += tc - tb; s_comp += td - tc; s_tail += te - td;
#endif
      rotate();
    }
    if (DEFER && loads_first) {   // the deferred MFMAs of the segment's last step (its fragments are still in registers)
#pragma unroll
      for (int s = 0, idx = 0; s < MIN_DEPTH; s++)
#pragma unroll
        for (int t = 0; t < MIN_DEPTH; t++)
          if (s + t < NCLS) {
#pragma unroll
            for (int b = 0; b < WB; b++) {
              if (idx >= N_UNCOND - DEFER) acc[s + t][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s][b], fb[t], acc[s + t][b], 0, 0, 0);
              idx++;
            }
          }
    }
    flush();
  }
#ifdef MDG_I8_STAMPS
  if (a.stamps && lane == 0 && blockIdx.x < STAMP_WGS) {
    unsigned long long t_end;
    MDG_STAMP(t_end);
    unsigned long long* o = a.stamps + ((size_t)blockIdx.x * NW + wave) * 8;
    o[0] = s_wait; o[1] = s_issue; o[2] = s_comp; o[3] = s_tail; o[4] = t_end - t_begin; o[5] = executed - executed_before; o[6] = ke - kb;   // (of the workgroup's LAST tile or k-chunk)
  }
#endif
}

// Persistent launch (large statistics): 8 x 32 workgroups, one per CU, pulling tiles from a host-built schedule
// (SyrkArgs::sched) instead of one tile per workgroup.  Workgroup b belongs to logical XCD b % 8 (what the dispatcher's
// round-robin gives -- if it ever does not, only locality is lost).  The tiles are dealt out in GROUPS of up to 32 that form
// a compact block of the lower region (4 tile rows x 8 tile columns: 12 distinct panels for 32 tiles instead of 64), one
// group per XCD and round.  Measured on one box, sigma_mlp 32768 x 14336, five / six planes per call:
//   one tile per workgroup, 2 x 2 super-blocks (round 1's launch)      25.3-25.5 / 38.9-39.0 ms   58 / -- GB of L2 misses
//   persistent, every workgroup through a fixed list of its own        24.5-24.6 / 38.2-38.3 ms   53 / 67 GB
//   persistent + a barrier of the XCD's 32 workgroups between rounds   25.4-25.6 / 40.4-40.6 ms   32 / 54 GB
//   persistent, tiles pulled from per-XCD queues (later in round 2, other kernel improvements included; fixed lists at that
//   point: 22.05 ms)                                                   21.3 / 35.5 ms             34 GB             <- shipped
// The lock-step variant halves the L2-miss traffic and is SLOWER: the misses are not what bounds the kernel (the power cap
// is: mdg_probe_mfma_i8), and 32 CUs folding into sigma and refilling their rings at the same instant
// cost more than the hits return.  (The fixed-list and barrier variants: scripts/probes/cov_i8_variants.patch.)
template <int P>  // planes used: 5 or 6
__global__ __launch_bounds__(64 * NW, 1) void i8_syrk_kernel(SyrkArgs a) {
  constexpr int WB = wide_tile(P) ? 2 : 1;         // 32-row blocks of a wave tile: 64 x 32, or 32 x 32 (96 accumulators at P = 6)
  constexpr int TJ = WB == 2 ? 128 : 64;           // tile columns (rows of the J operand); waves are laid out (128 / 32 WB) x (TJ / 32)
  constexpr int PB = TJ * KS;
  constexpr int GA = TI / 32, GB = TJ / 32;        // 32-row groups (1 KB pieces per plane and stage) of the two operands
  constexpr int WCOLS = TJ / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // The route is chosen on the DEVICE: mdg_cov_accum_i8 enqueues the five-plane product, the six-plane product and the fp64
  // kernel back to back, and each exits at once unless the depth statistic of this call (i8_depth_kernel) selects it -- the
  // host never waits for the flag.  The six-plane launch also books the fp64 fallback in the route counters.
  unsigned mask_and = ~0u;
  {
    int live, fallbacks;
    const int runs = product_launch_runs<P>(a, live, fallbacks);
    if (P == 6 && fallbacks && a.route_counts && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.route_counts + 2, fallbacks);
    if (!runs) return;
    if (a.route_counts && blockIdx.x == 0 && threadIdx.x == 0) {
      if (runs == 2) {   // the exact route serves both legacy classes: book them as the route kernel classed them, and the exact count
        int six = 0;
        for (int p = 0; p < a.nprob; p++) six += (a.route_flag[p] & 3) == 1;
        if (live - six) atomicAdd(a.route_counts + 0, live - six);
        if (six) atomicAdd(a.route_counts + 1, six);
        atomicAdd(a.route_counts + 4, live);
      } else {
        atomicAdd(a.route_counts + (P == 5 ? 0 : 1), live);
      }
    }
  }
  unsigned executed = 0;
  const int lane = threadIdx.x & 63;
#ifdef MDG_I8_WGTIMES
  if (a.sched && threadIdx.x == 0) a.wgtimes[blockIdx.x * 64] = wall_clock64();
#endif
  if (a.sched) {
    const int xcd = blockIdx.x & 7;
    auto work = [&](const int2 entry) {
      const int code = __builtin_amdgcn_readfirstlane(entry.x), chunk = __builtin_amdgcn_readfirstlane(entry.y);
      if (code < 0 || (a.route_flag[code >> CODE_PROB] & 2)) return;   // (a statistic that went to the fp64 kernel: not ours)
      const SyrkProblem& pr = a.prob[code >> CODE_PROB];   // (uniform index into the kernel arguments: scalar loads)
      const int bi = (code >> CODE_BI) & ((1 << CODE_BI) - 1), bj = code & ((1 << CODE_BI) - 1);
      if (chunk == 0) {   // per-head statistics: the tile's columns start at the head's first feature
        i8_syrk_tile<P>(a, pr, bi, bj, 0, a.nk, pr.sigma, pr.ld_sigma, 0, pr.block ? bi * TI : 0, lds, executed, mask_and);
      } else {   // the last round: k-chunk q of Q of this tile, folded into its own (zeroed) partial tile
        const int q = chunk & 31, Q = (chunk >> 5) & 31, pslot = chunk >> 10;
        const int kb = (int)((int64_t)a.nk * q / Q), ke = (int)((int64_t)a.nk * (q + 1) / Q);
        if (kb < ke) i8_syrk_tile<P>(a, pr, bi, bj, kb, ke, a.partial + (int64_t)pslot * TI * TJ, TJ, bi * TI, bj * TJ, lds, executed, mask_and);
      }
    };
    // The schedule's groups are QUEUES, one per XCD (XCD x owns groups x, x + 8, ...): a workgroup pulls the next tile of its
    // XCD's queue with one atomic, and when that queue is empty helps the other XCDs with theirs.  The 32 tiles of a group
    // are still taken together by the 32 CUs of one XCD (same panels in the same L2), but a CU that runs a few percent faster
    // -- clocks differ from CU to CU and from board to board under the power cap -- simply takes more tiles, where fixed
    // lists made the whole launch wait for the slowest workgroup.  Which CU computes a tile does not change its result.
    __shared__ int next_entry;
#ifdef MDG_I8_WGTIMES
    int done = 0;
#endif
    for (int victim = 0; victim < 8; victim++) {
      const int x = (xcd + victim) & 7;
      const int entries = ((a.ngroups - x + 7) >> 3) * 32;     // of XCD x's groups
      for (;;) {
        __syncthreads();                                         // the previous tile is complete in every wave
        if (threadIdx.x == 0)
          next_entry = __hip_atomic_fetch_add(a.xcd_arrive + x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int t = __builtin_amdgcn_readfirstlane(next_entry);
        if (t >= entries) break;
        work(a.sched[((t >> 5) * 8 + x) * 32 + (t & 31)]);
#ifdef MDG_I8_WGTIMES
        if (threadIdx.x == 0 && done < 60) a.wgtimes[blockIdx.x * 64 + 2 + done++] = wall_clock64();
#endif
      }
    }
  } else {
    // Tile (bi, bj): bi = 128-row block, bj = TJ-row block of the lower region (bj <= bi for 128 x 128 tiles, bj <= 2 bi + 1 for
    // 128 x 64).  XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2, so workgroup w belongs
    // to XCD w % 8 and is the (w / 8)-th one there.  The tiles are grouped into super-blocks that are square in features
    // (SI x SI tiles of 128 x 128, SI x 2 SI tiles of 128 x 64); a super-block lives on ONE XCD, so per k-step its tiles pull
    // each distinct panel row through that L2 once.  Super-blocks (R, C), C <= R, cover the lower region; tiles of a diagonal
    // super-block that lie above it exit at once.
    int bi, bj;
    constexpr int SI = P == 6 ? SB6 : SB5;   // super-block: SI x SI (or SI x 2 SI) tiles
    {
      constexpr int TPS = TJ == 128 ? SI * SI : SI * 2 * SI;   // tiles per super-block
      constexpr int SJ = TJ == 128 ? SI : 2 * SI;              // tile columns of a super-block
      const int w = blockIdx.x;
      const int q = w >> 3;
      const int sb = q / TPS * 8 + (w & 7), t_in = q % TPS;
      int R = (int)((sqrtf(8.f * sb + 1.f) - 1.f) * 0.5f);
      while ((R + 1) * (R + 2) / 2 <= sb) R++;
      while (R * (R + 1) / 2 > sb) R--;
      const int C = sb - R * (R + 1) / 2;
      bi = SI * R + t_in / SJ;
      bj = SJ * C + t_in % SJ;
    }
    if (bi >= a.prob[0].n / TI || bj * TJ > bi * TI + TI - 1) return;   // (one full-triangle statistic per launch on this path)
    i8_syrk_tile<P>(a, a.prob[0], bi, bj, 0, a.nk, a.prob[0].sigma, a.prob[0].ld_sigma, 0, 0, lds, executed, mask_and);
  }
  if (a.mfma_count && lane == 0) atomicAdd(a.mfma_count, (unsigned long long)executed);
#ifdef MDG_I8_WGTIMES
  if (a.sched && threadIdx.x == 0) a.wgtimes[blockIdx.x * 64 + 1] = wall_clock64();
#endif
}

// The persistent launch's LAST round would keep R = (tiles mod 256) CUs busy for a whole tile while the others idle -- 16 of
// 256 at sigma_x's shape (528 tiles), 184 at sigma_mlp's (6328).  The schedule (schedule_for) therefore cuts each tile of that
// round into Q k-chunks -- R Q pieces worked by all CUs in ceil(R Q / 256) short rounds, 16 x 16 in one round resp. 184 x 4 in
// three -- each folding into its own fp64 partial tile; this kernel then adds a tile's partials to sigma in chunk order (a
// fixed order: the result stays run-to-run bit-identical).  One workgroup per 1024 elements of a split tile.
constexpr int COMBINE_ELEMS = 1024;   // tile elements per workgroup of the combine pass (4 per thread, all chunks' loads in flight together)
template <int P>
__global__ __launch_bounds__(256) void i8_tail_combine_kernel(SyrkArgs a, int n_tail) {
  constexpr int TJ = wide_tile(P) ? 128 : 64;
  constexpr int PARTS = TI * TJ / COMBINE_ELEMS;
  int live, fallbacks;
  if (!product_launch_runs<P>(a, live, fallbacks)) return;     // the product launch of the other route produced the partials, or none did
  const int4 t = a.tail[blockIdx.x / PARTS];
  if (a.route_flag[t.x >> CODE_PROB] & 2) return;
  const SyrkProblem& pr = a.prob[t.x >> CODE_PROB];
  const int bi = (t.x >> CODE_BI) & ((1 << CODE_BI) - 1), bj = t.x & ((1 << CODE_BI) - 1), Q = t.y;
  const double* part = a.partial + (int64_t)t.z * TI * TJ;
#pragma unroll
  for (int i = 0; i < COMBINE_ELEMS / 256; i++) {
    const int e = (blockIdx.x % PARTS) * COMBINE_ELEMS + i * 256 + threadIdx.x;
    const int row = bi * TI + e / TJ, col = bj * TJ + e % TJ;
    if (col > row) continue;
    double* p = pr.sigma + (int64_t)row * pr.ld_sigma + col - (pr.block ? bi * TI : 0);
    double v = *p;
    for (int q = 0; q < Q; q++) v += part[(int64_t)q * TI * TJ + e];   // chunk order: fixed, so the sum is reproducible
    *p = v;
  }
}

// ---- the fp64 column kernel: the rows / columns of sigma that belong to the columns the route took off the int8 path
// (RouteOut::out, at most ROUTE_JMAX per statistic): v_k[c] = sum over tokens of x[t, out[k]] x[t, c] in plain fp64 -- the
// reference's arithmetic (LlamaAdapter.py:127-147) -- for every column c.  One pass over X serves COLK_GROUP such columns: a lane
// owns 8 consecutive columns c (one 16-byte load per token) x the group's columns (64 accumulators), a one-wave workgroup 512
// columns x one of COLK_CHUNKS token chunks; 8 tokens' loads are in flight together, and the group's own values for the next 64
// tokens are fetched while the current 64 are multiplied.  The chunk partials are reduced in chunk order by
// i8_columns_reduce_kernel (run-to-run bit-identical), which adds v_k[c] to sigma[max(c, j)][min(c, j)].  Both launches are
// enqueued with every call and exit at once when the route left every column on the int8 path.  2 x tokens x n flop per column:
// 0.94 GFLOP at the sigma_mlp shape; one pass reads X once (0.94 GB).
constexpr int COLK_GROUP = 8, COLK_CHUNKS = 64, COLK_WG_COLS = 512, COLK_STAGE = 64, COLK_BATCH = 8;
struct ColArgs {
  const bf16_t* x;
  int64_t ld, T;
  int n, vec;                 // vec: rows are 16-byte addressable
  const RouteOut* route;
  const int* flag;            // the statistic's route bits (bit 1: the whole statistic went to the fp64 kernel)
  double* part;               // [ROUTE_JMAX][COLK_CHUNKS][n]
};
__device__ __forceinline__ double bf16_bits_to_f64(unsigned b) { return (double)__uint_as_float(b << 16); }

__global__ __launch_bounds__(64) void i8_columns_kernel(ColArgs a) {
  const int pass = blockIdx.z;
  const int n_out = a.route->n_out;
  if ((*a.flag & 2) || pass * COLK_GROUP >= n_out) return;
  const int nj = min(COLK_GROUP, n_out - pass * COLK_GROUP);
  __shared__ __attribute__((aligned(16))) double xj[COLK_STAGE][COLK_GROUP];
  const int lane = threadIdx.x;
  const int my_k = lane % COLK_GROUP;                       // staging: lane l fetches column l % 8 of the group for tokens l / 8 + 8 i
  const int my_col = my_k < nj ? a.route->out[pass * COLK_GROUP + my_k] : -1;
  const int c0 = blockIdx.x * COLK_WG_COLS + lane * 8;
  const bool active = c0 < a.n;
  const int64_t chunk_len = (a.T + COLK_CHUNKS - 1) / COLK_CHUNKS;
  const int64_t t0 = blockIdx.y * chunk_len, t1 = min(a.T, t0 + chunk_len);
  const unsigned short* xs = (const unsigned short*)a.x;
  auto fetch_group = [&](int64_t t, unsigned short (&g)[COLK_STAGE / 8]) {
#pragma unroll
    for (int i = 0; i < COLK_STAGE / 8; i++) {
      const int64_t tok = t + lane / COLK_GROUP + 8 * i;
      g[i] = (my_col >= 0 && tok < t1) ? xs[tok * a.ld + my_col] : (unsigned short)0;
    }
  };
  double acc[COLK_GROUP][8] = {};
  unsigned short g[COLK_STAGE / 8];
  if (t0 < t1) fetch_group(t0, g);
  for (int64_t t = t0; t < t1; t += COLK_STAGE) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < COLK_STAGE / 8; i++) xj[lane / COLK_GROUP + 8 * i][my_k] = bf16_bits_to_f64(g[i]);
    __syncthreads();
    if (t + COLK_STAGE < t1) fetch_group(t + COLK_STAGE, g);
    if (!active) continue;
    const int steps = (int)min((int64_t)COLK_STAGE, t1 - t);
    for (int tb = 0; tb < steps; tb += COLK_BATCH) {
      unsigned w[COLK_BATCH][4];
#pragma unroll
      for (int i = 0; i < COLK_BATCH; i++) {
        const bool live = t + tb + i < t1;
        const int64_t tok = live ? t + tb + i : t1 - 1;        // (the address stays inside the chunk ...)
        if (a.vec) {
          const i32x4 v = *(const i32x4*)(xs + tok * a.ld + c0);
          w[i][0] = v[0]; w[i][1] = v[1]; w[i][2] = v[2]; w[i][3] = v[3];
        } else {
#pragma unroll
          for (int h = 0; h < 4; h++) w[i][h] = xs[tok * a.ld + c0 + 2 * h] | ((unsigned)xs[tok * a.ld + c0 + 2 * h + 1] << 16);
        }
        // ... and a slot beyond the chunk's end contributes exact zeros: the group's staged values are zero there, but the re-read
        // last token may hold an Inf / NaN -- the very columns this kernel exists for -- and 0 * Inf would turn the +-Inf the
        // reference's fp64 product gives into NaN
        if (!live) w[i][0] = w[i][1] = w[i][2] = w[i][3] = 0u;
      }
#pragma unroll
      for (int i = 0; i < COLK_BATCH; i++) {
        double xc[8];
#pragma unroll
        for (int h = 0; h < 4; h++) {
          xc[2 * h] = bf16_bits_to_f64(w[i][h] & 0xFFFFu);
          xc[2 * h + 1] = bf16_bits_to_f64(w[i][h] >> 16);
        }
#pragma unroll
        for (int k = 0; k < COLK_GROUP; k++) {
          const double xk = xj[tb + i][k];
#pragma unroll
          for (int c = 0; c < 8; c++) acc[k][c] += xk * xc[c];
        }
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int k = 0; k < COLK_GROUP; k++)
    if (k < nj) {
      double* o = a.part + ((int64_t)(pass * COLK_GROUP + k) * COLK_CHUNKS + blockIdx.y) * a.n + c0;
#pragma unroll
      for (int c = 0; c < 8; c++) o[c] = acc[k][c];
    }
}

__global__ __launch_bounds__(256) void i8_columns_reduce_kernel(ColArgs a, const int* emax, double* sigma, int64_t ld_sigma, int block) {
  const int k = blockIdx.y;
  if ((*a.flag & 2) || k >= a.route->n_out) return;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= a.n) return;
  const int j = a.route->out[k];
  if (block && c / block != j / block) return;                 // per-head statistics: only the head's own 128 x 128 block exists
  if ((emax[c] & EMAX_COLUMN_OUT) && c < j) return;            // a pair of two such columns belongs to the pass of the smaller index
  double v = 0.0;
  for (int q = 0; q < COLK_CHUNKS; q++) v += a.part[((int64_t)k * COLK_CHUNKS + q) * a.n + c];   // chunk order: reproducible
  const int row = max(c, j), col = min(c, j);
  sigma[(int64_t)row * ld_sigma + col - (block ? row / block * block : 0)] += v;
}


// ---- the exact route: no plane pair is dropped.
// An element's 48-bit integer N splits into its top three balanced digits and the rest, N = N_d + L with N_d = d_0 2^40 + d_1 2^32 +
// d_2 2^24 and L = d_3 2^16 + d_4 2^8 + d_5 in [-8421504, 8355711]; X = X_d + X_lo accordingly.  On real activations L is zero for
// almost every element: planes 3 .. 5 are reached only by elements 17 binades and more below their column's maximum -- 3e-5 of a
// Gaussian column, 0.5 % of a SiLU-gated one (whose 1 KB pieces of plane 3 nevertheless hold a nonzero 95 % of the time, which is why
// the truncated six-plane product spends 6 of its 15 plane pairs multiplying a 99.5 %-zero operand).  So
//     X^T X = X_d^T X_d  +  X_lo^T X  +  X_d^T X_lo
//   * X_d^T X_d: the NINE plane pairs of the top three planes, all of them (classes 0 .. 4) -- the five-plane product kernel with the
//     planes below masked off (mask_and in i8_syrk_tile): exact int32 class sums as before, 9 instead of 9.4 / 15.1 executed pairs;
//   * the two remainder products: every element with L != 0 is an EVENT (token, column, x_lo = L 2^(E - 172)), listed per COLUMN in
//     token order by i8_extract_lo_kernel / i8_compact_lo_kernel.  Two implementations, picked on the device from the list lengths
//     (i8_lo_mode_kernel):
//       sparse lists (Gaussian columns: two events per column) -- i8_lo_product_kernel, one workgroup per 128 x 128 tile of the lower
//         triangle: the lists of the tile's row block (sigma[r][c] += x_lo(t, r) x(t, c), the partner x in full) and of its column
//         block (+= x_d(t, r) x_lo(t, c), the partner's top three planes recomputed from x where it has deeper digits: x_d = 2^24 q
//         floor(x / (2^24 q) + 8421504 / 2^24), the balanced digits' rounding), summed into an fp64 tile in LDS, sigma read and
//         written once;
//       dense lists (SiLU-gated: 156 per column) -- i8_lo_wide_kernel<false / true>, one wave per (column, block of 512 partner
//         columns), eight partner columns per lane, the sums in registers; the second product reads its partner from a bf16 copy of
//         x in which every listed element is replaced by x_d (i8_copy_xd_kernel, i8_patch_xd_kernel).
//     Either way fp64 products of exact operands, every sum owned by ONE wave that adds its events in list order -- run-to-run
//     bit-identical -- and folded into sigma once per product.
// Nothing is truncated: what is left is fp64 rounding (one rounding per fold of the class sums and per event sum) and the rho term
// of the elements more than 38 binades under their column maximum, which the split rounds to an integer (RouteOut::sq keeps it).
// The route kernel's decisions stay as they are -- which columns leave for the fp64 column kernel, whether the whole statistic
// does -- and the exact route then REPLACES the truncated five- or six-plane product whenever every remainder list fits its
// list (LO_CAP events per column and 2048 tokens = 6.2 % of the elements; cubed Gaussians, Student-t: no -- the truncated
// product with its bound takes those as before).  Cost at the sigma_mlp shape (profiles/r04_exact_route_kernels_*.csv): lists 0.08 /
// 0.25 ms, remainder products 1.0 ms (Gaussian) / 5.4 + 0.5 ms for the copy (SiLU-gated) against 0.4 x 2.1 ... 6.1 x 2.1 ms of
// plane-pair products saved.
constexpr int LO_CHUNK_STEPS = 64;       // k-steps (2048 tokens) per segment of a column's event list
constexpr int LO_CAP = 128;              // events per segment: 6.2 % of its 2048 tokens
constexpr double LO_ROUND = 8421504.0 / 16777216.0;   // (128 (1 + 256 + 65536)) / 2^24: where the balanced digits d_3 d_4 d_5 round
constexpr int EXACT_OVERFLOW = 16, EXACT_RAN = 17, EXACT_MODE = 18;    // ints of the workspace's shared block (MODE: 1 sparse lists, 2 dense)
#ifndef MDG_LO_SPARSE_MEAN
#define MDG_LO_SPARSE_MEAN 32
#endif
#ifndef MDG_LO_SPARSE_MAX
#define MDG_LO_SPARSE_MAX 256
#endif
constexpr int LO_SPARSE_MAX = MDG_LO_SPARSE_MAX;      // sparse lists: at most this many events in any column, LO_SPARSE_MEAN on average
constexpr int LO_SUB = 4, LO_RCAP = 8 * LO_SPARSE_MAX;   // one merged list per (32-column group, column mod 4)
constexpr int LO_TILE = 128, LO_PITCH = LO_TILE + 1;

struct __attribute__((aligned(16))) LoEntry {
  double v;                        // x_lo(token, column) = L 2^(E_column - 172): exact (|L| < 2^24)
  unsigned off, aux;               // byte offset of the token's row in x (token x row pitch x 2);  aux: the same in the x_d copy (token x n x
                                   // 2) -- in a sparse (group, residue) list: the column's index in its group.  The call refuses the exact
                                   // route when x spans 4 GB or more: the walk then spends no 64-bit scalar arithmetic on an address
};
struct LoProblem {
  const bf16_t* x;
  int64_t ld;
  const signed char* planes;
  const unsigned char* zmask;
  const int* emax;
  LoEntry* entries;                // [n][nch][LO_CAP]: one list per COLUMN, written per segment by i8_extract_lo_kernel, then closed up
                                   // to one contiguous list by i8_compact_lo_kernel
  int* counts;                     // [n][nch] segment lengths, then [n] list lengths (totals)
  double* sigma;
  int64_t ld_sigma;
  bf16_t* xd;                      // [tokens][n]: x with every listed element replaced by its top three digit planes x_d = x - x_lo (which is
                                   // a bf16 again: a rounding of 8 significant bits to a coarser grid) -- the partner of the second product
  int n, block;
  LoEntry* rentries;               // sparse mode: [n / 32][LO_SUB][LO_RCAP] merged lists (i8_residue_lo_kernel) and their lengths
  int* rtotals;
  int tile0[3];                    // this statistic's first workgroup in the grids of the two wide products and of the tile kernel
  int pairs;                       // rows of x are 4-byte addressable: a lane of the tile kernel fetches two neighbouring columns with one load
};
struct LoArgs {
  LoProblem prob[MAX_PROBLEMS];
  int nprob, nk, nch, tiles[3];
  int64_t n_tokens;
  int always;                      // MDG_I8_EXACT_ALWAYS: the exact route for launches of the five-plane class too
  const int* route_flag;
  int* state;                      // shared block of the workspace: [EXACT_OVERFLOW], [EXACT_RAN]
};
__device__ __forceinline__ int* lo_totals(const LoProblem& pr, int nch) { return pr.counts + (int64_t)pr.n * nch; }
// Is the exact route on offer for this launch?  Always when the caller asks for it; by default where it is the faster product:
// launches of the six-plane class (9 executed plane pairs + the remainder kernel against 15.1), and five-plane launches of a large
// statistic (9 against 9.4 pairs on a kernel without masks or conditional blocks: 21.6 against 22.2 ms per sigma_mlp call on
// Gaussian columns; with two k-steps per stage 20.6) -- below LO_AUTO_MIN_N features the remainder kernel's fixed costs (a workgroup
// per tile, two list walks, one fold) outweigh 0.4 plane pairs (4096 features, 32768 tokens: 2.13 against 2.15 ms on Gaussian, 3.25
// against 3.42 on SiLU-gated columns; 8192: 7.18 / 7.72 and 10.6 / 12.5 -- profiles/r04_exact_route_timing.log).
constexpr int LO_AUTO_MIN_N = 4096;
__device__ __forceinline__ bool lo_offered(const LoArgs& a) {
  if (a.always) return true;
  if (a.prob[0].n >= LO_AUTO_MIN_N && !a.prob[0].block) return true;
  for (int p = 0; p < a.nprob; p++)
    if ((a.route_flag[p] & 3) == 1) return true;
  return false;
}

// One wave per (32-column group, segment of LO_CHUNK_STEPS k-steps): reads the pieces of planes 3 .. 5 the piece masks say are
// there -- as the product kernel would -- and appends every element with L != 0 to ITS COLUMN's segment, in token order (lane r
// holds tokens 0 .. 15 of a k-step of column r, lane 32 + r tokens 16 .. 31: the second appends behind the first).
__global__ __launch_bounds__(64) void i8_extract_lo_kernel(LoArgs a) {
  const LoProblem& pr = a.prob[blockIdx.z];
  const int64_t groups = pr.n / 32;
  const int G = blockIdx.x, ch = blockIdx.y, lane = threadIdx.x;
  if (!lo_offered(a)) return;
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) a.state[EXACT_RAN] = 1;
  if (G >= groups || (a.route_flag[blockIdx.z] & 2)) return;     // (a statistic that went to the fp64 kernel has no lists)
  const int col = G * 32 + (lane & 31);
  LoEntry* out = pr.entries + ((int64_t)col * a.nch + ch) * LO_CAP;
  const double scale = ldexp(1.0, (pr.emax[col] & 255) - 172);
  int count = 0;                                                  // events of this lane's column so far (the same in both of its lanes)
  const int kt1 = min(a.nk, (ch + 1) * LO_CHUNK_STEPS);
  for (int kt = ch * LO_CHUNK_STEPS; kt < kt1; kt++) {
    const unsigned m = pr.zmask[(int64_t)kt * groups + G];
    if ((m >> 3) == 0) continue;
    const i32x4 zero = (i32x4)0;
    const i32x4 d3 = *((const i32x4*)(pr.planes + ((3 * groups + G) * (int64_t)a.nk + kt) * 1024) + lane);
    const i32x4 d4 = (m >> 4) ? *((const i32x4*)(pr.planes + ((4 * groups + G) * (int64_t)a.nk + kt) * 1024) + lane) : zero;
    const i32x4 d5 = (m >> 5) ? *((const i32x4*)(pr.planes + ((5 * groups + G) * (int64_t)a.nk + kt) * 1024) + lane) : zero;
    const unsigned tok0 = (unsigned)kt * KS + (lane >> 5) * 16;
    int L[16], mine = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int sh = 8 * (q & 3);
      L[q] = (int)(signed char)((unsigned)d3[q >> 2] >> sh) * 65536 + (int)(signed char)((unsigned)d4[q >> 2] >> sh) * 256 +
             (int)(signed char)((unsigned)d5[q >> 2] >> sh);
      mine += L[q] != 0;
    }
    if (__ballot(mine != 0) == 0) continue;
    const int other = __shfl_xor(mine, 32);
    int at = count + ((lane >> 5) ? other : 0);
#pragma unroll
    for (int q = 0; q < 16; q++)
      if (L[q] != 0) {
        if (at < LO_CAP) out[at] = LoEntry{(double)L[q] * scale, (tok0 + q) * (unsigned)(pr.ld * 2), (tok0 + q) * (unsigned)(pr.n * 2)};
        at++;
      }
    count += mine + other;
  }
  if (lane < 32) {
    pr.counts[(int64_t)col * a.nch + ch] = min(count, LO_CAP);
    if (count > LO_CAP) a.state[EXACT_OVERFLOW] = 1;
  }
}

// One wave per column: closes the segments of its list up into one contiguous list, in place (a segment only ever moves towards
// the front, and the wave copies in order), and leaves its length in lo_totals.  The remainder kernel then walks full batches
// whatever the density.
__global__ __launch_bounds__(64) void i8_compact_lo_kernel(LoArgs a) {
  if (!lo_offered(a) || a.state[EXACT_OVERFLOW] != 0) return;
  const LoProblem& pr = a.prob[blockIdx.y];
  const int col = blockIdx.x, lane = threadIdx.x;
  if (col >= pr.n || (a.route_flag[blockIdx.y] & 2)) return;
  const int* counts = pr.counts + (int64_t)col * a.nch;
  LoEntry* base = pr.entries + (int64_t)col * a.nch * LO_CAP;
  int total = 0;
  for (int ch = 0; ch < a.nch; ch++) {
    const int cnt = counts[ch];
    const LoEntry* src = base + (int64_t)ch * LO_CAP;
    if (total != ch * LO_CAP)
      for (int i = 0; i < cnt; i += 64) {
        LoEntry e = LoEntry{0., 0u, 0u};
        if (i + lane < cnt) e = src[i + lane];
        if (i + lane < cnt) base[total + i + lane] = e;      // (the 64 loads of a round are complete before its stores: same wave, in order)
      }
    total += cnt;
  }
  if (lane == 0) lo_totals(pr, a.nch)[col] = total;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// Which remainder kernels run, from the list lengths (one workgroup; read by everything below): SPARSE lists -- at most LO_SPARSE_MAX
// events in any column and LO_SPARSE_MEAN per column on average: Gaussian columns have two per 32768 tokens -- go to the 128 x 128-tile
// kernel, where a tile is one chain of memory round trips around a handful of products and sigma is read and written ONCE for both
// products; anything denser (SiLU-gated: 164 per column and 32768 tokens) to the wide kernels, which need 2.4 x fewer instructions per
// product but pay 0.5 ms for the x_d copy and a dependent chain per (column, partner block).  Measured at the sigma_mlp width, whole
// call, SiLU-gated columns, 2048 / 4096 / 8192 / 16384 / 32768 tokens = 10 / 20 / 41 / 82 / 164 events per column
// (scripts/probes/lo_mode_crossover.sh, profiles/r04_lo_mode_crossover.log): tiles 3.11 / 4.78 / 8.23 / 15.3 / 29.8 ms, wide 3.79 /
// 5.25 / 8.15 / 14.0 / 26.6 -- they cross at ~38; Gaussian columns at 32768 tokens: tiles 1.0 ms, wide 1.8 + 0.5.
constexpr int LO_SPARSE_MEAN = MDG_LO_SPARSE_MEAN;
__global__ __launch_bounds__(1024) void i8_lo_mode_kernel(LoArgs a) {
  if (!lo_offered(a) || a.state[EXACT_OVERFLOW] != 0) return;
  __shared__ long long sums[16];
  __shared__ int maxs[16];
  long long sum = 0, cols = 0;
  int mx = 0;
  for (int p = 0; p < a.nprob; p++) {
    if (a.route_flag[p] & 2) continue;
    const LoProblem& pr = a.prob[p];
    cols += pr.n;
    for (int c = threadIdx.x; c < pr.n; c += 1024) {
      const int t = lo_totals(pr, a.nch)[c];
      sum += t;
      mx = max(mx, t);
    }
  }
  for (int o = 32; o; o >>= 1) {
    sum += __shfl_xor(sum, o);
    mx = max(mx, __shfl_xor(mx, o));
  }
  if ((threadIdx.x & 63) == 0) { sums[threadIdx.x >> 6] = sum; maxs[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; w++) { sum += sums[w]; mx = max(mx, maxs[w]); }
    a.state[EXACT_MODE] = (mx <= LO_SPARSE_MAX && sum <= LO_SPARSE_MEAN * cols) ? 1 : 2;
  }
}

// SPARSE: the lists of the eight columns 4 k + sub of a group, one behind the other, as ONE list per (group, residue) -- the unit a wave
// of the tile kernel walks (it owns the accumulators of those rows / columns); `aux` = the column's index in its group.  One wave
// per list; at most 8 x 64 entries.
__global__ __launch_bounds__(64) void i8_residue_lo_kernel(LoArgs a) {
  if (a.state[EXACT_MODE] != 1) return;
  const LoProblem& pr = a.prob[blockIdx.y];
  const int id = blockIdx.x, lane = threadIdx.x;
  if (id >= pr.n / 32 * LO_SUB || (a.route_flag[blockIdx.y] & 2)) return;
  const int G = id / LO_SUB, sub = id % LO_SUB;
  LoEntry* out = pr.rentries + (int64_t)id * LO_RCAP;
  int at = 0;
  for (int k = 0; k < 8; k++) {
    const int col = G * 32 + 4 * k + sub;
    const int cnt = lo_totals(pr, a.nch)[col];           // <= LO_SPARSE_MAX (the mode says so)
    for (int i = lane; i < cnt; i += 64) {
      LoEntry e = pr.entries[(int64_t)col * a.nch * LO_CAP + i];
      e.aux = 4 * k + sub;
      out[at + i] = e;
    }
    at += cnt;
  }
  if (lane == 0) pr.rtotals[id] = at;
}

// DENSE: x_d starts as a copy of x (16 bytes per thread and step; n is a multiple of 128), then every listed element is replaced by
// x_d = x - x_lo -- exactly (both are multiples of the column's unit, below 2^48 of them), and a bf16 again (a rounding of 8
// significant bits to a coarser grid).  One wave per column for the second step.
__global__ __launch_bounds__(256) void i8_copy_xd_kernel(LoArgs a) {
  if (a.state[EXACT_MODE] != 2) return;
  const LoProblem& pr = a.prob[blockIdx.y];
  if (a.route_flag[blockIdx.y] & 2) return;
  const int64_t per_row = pr.n / 8, total = a.n_tokens * per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / per_row, c = (i - t * per_row) * 8;
    typedef unsigned u32x4u __attribute__((ext_vector_type(4), aligned(2)));
    const u32x4u v = *(const u32x4u*)(pr.x + t * pr.ld + c);
    *(u32x4*)(pr.xd + t * pr.n + c) = (u32x4){v[0], v[1], v[2], v[3]};
  }
}
__global__ __launch_bounds__(64) void i8_patch_xd_kernel(LoArgs a) {
  if (a.state[EXACT_MODE] != 2) return;
  const LoProblem& pr = a.prob[blockIdx.y];
  const int col = blockIdx.x, lane = threadIdx.x;
  if (col >= pr.n || (a.route_flag[blockIdx.y] & 2)) return;
  const int total = lo_totals(pr, a.nch)[col];
  const LoEntry* list = pr.entries + (int64_t)col * a.nch * LO_CAP;
  for (int i = lane; i < total; i += 64) {
    const LoEntry e = list[i];
    const double xd = (double)__uint_as_float((unsigned)*(const bf16_t*)((const char*)pr.x + e.off + 2 * col) << 16) - e.v;
    *(bf16_t*)((char*)pr.xd + e.aux + 2 * col) = (bf16_t)(__float_as_uint((float)xd) >> 16);
  }
}

__device__ __forceinline__ void lds_add_f64(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_f64, returnless: issued in order per wave
}
__device__ __forceinline__ double bf16_to_f64(unsigned bits16) { return (double)__uint_as_float(bits16 << 16); }
// Where row (column) i of the tile sits in the LDS accumulator: even indices in the first half, odd ones in the second.  A lane
// fetches two NEIGHBOURING partner columns with one load; stored side by side its two ds_add_f64 would put the 64 lanes on a 16-byte
// stride -- four lanes per bank pair, a four-way conflict on every atomic of the kernel (the walk was bound by exactly that:
// 10.6 ms at the sigma_mlp shape on SiLU-gated data).  Permuted, the lanes of an atomic cover 512 contiguous bytes (row-event) or
// one 8-byte word per 1032-byte row (column-event): the two passes a 64-lane fp64 access needs anyway.
__device__ __forceinline__ int lo_perm(int i) { return (i >> 1) + 64 * (i & 1); }

// The events of ONE list -- 32-column group G, columns with (column mod 4) == sub -- in list order, by ONE wave, which thereby owns
// the accumulators they touch.  COLS = false: G is row group g of the tile's row block: acc[g 32 + r][c] += x_lo(t, r) x(t, c) for
// the 128 columns c of the tile's column block (lane: c = 2 lane, 2 lane + 1).  COLS = true: G is column group g of the column
// block: acc[r][g 32 + c] += x_d(t, r) x_lo(t, c) for the 128 rows r of the row block (lane: r = 2 lane, 2 lane + 1), x_d the
// partner's top three digit planes -- which IS x for every element within 14 binades of its column maximum (no digit below plane
// 2), so the rounding is taken only when a lane meets a deeper one.  partner0: first column of the partner block.
// The walk is bound by the latency of the partner loads (one 4-byte load per event and lane, rows scattered over the tokens) and
// by the VALU (7 - 14 operations per event): the events go in batches of LO_UN whose loads are all issued before the previous
// batch is multiplied (two batches in flight per wave, sixteen waves per CU).
constexpr int LO_UN = 32;
// The length of list (G, sub) and its first batch of entries, one per lane (lanes beyond the list's end: the last entry's token,
// v = 0: exact zeros) -- fetched for BOTH passes of a tile before the first one starts, so that the second pass does not begin with
// two dependent memory round trips of its own.
struct LoFirst {
  int cnt;
  LoEntry m;
};
__device__ __forceinline__ LoFirst lo_first(const LoProblem& pr, const int G, const int sub, const int lane) {
  const int list_id = G * LO_SUB + sub;
  LoFirst f;
  f.cnt = __builtin_amdgcn_readfirstlane(pr.rtotals[list_id]);
  f.m = LoEntry{0., 0u, 0u};
  if (f.cnt > 0) {
    f.m = pr.rentries[(int64_t)list_id * LO_RCAP + min(lane & (LO_UN - 1), f.cnt - 1)];
    if ((lane & (LO_UN - 1)) >= f.cnt) f.m.v = 0.;
  }
  return f;
}
template <bool COLS>
__device__ __forceinline__ void lo_events(const LoProblem& pr, const int nch, const int G, const int sub, const int partner0, const int g,
                                          const int lane, double* acc, const LoFirst& first) {
  const unsigned short* xs = (const unsigned short*)pr.x;
  unsigned lim_a = 0, lim_b = 0;
  double qa = 1., qb = 1., ia = 1., ib = 1.;
  if (COLS) {   // the partner rows' digit grid: 2^24 units of their own scale
    const int ea = pr.emax[partner0 + 2 * lane] & 255, eb = pr.emax[partner0 + 2 * lane + 1] & 255;
    qa = ldexp(1.0, ea - 148);
    qb = ldexp(1.0, eb - 148);
    ia = 1.0 / qa;
    ib = 1.0 / qb;
    // an element has a digit below plane 2 iff its exponent field is below E - 14 (and it is not zero): 0 < |bits| < (E - 14) << 7
    lim_a = (unsigned)max(ea - 14, 1) << 7;
    lim_b = (unsigned)max(eb - 14, 1) << 7;
  }
  const int list_id = G * LO_SUB + sub;
  const int cnt = first.cnt;
  const LoEntry* list = pr.rentries + (int64_t)list_id * LO_RCAP;
  // batch k: events [k LO_UN, ...) -- one entry per lane (lanes beyond the list's end: the last entry's token, v = 0: exact zeros)
  auto fetch = [&](int k, LoEntry& m) {
    m = list[min(k * LO_UN + (lane & (LO_UN - 1)), cnt - 1)];
    if (k * LO_UN + (lane & (LO_UN - 1)) >= cnt) m.v = 0.;
  };
  auto issue = [&](const LoEntry& m, unsigned (&xv)[LO_UN]) {
#pragma unroll
    for (int u = 0; u < LO_UN; u++) {
      const unsigned off = (unsigned)__builtin_amdgcn_readlane((int)m.off, u);
      const unsigned short* row = (const unsigned short*)((const char*)xs + off) + partner0 + 2 * lane;
      xv[u] = pr.pairs ? *(const unsigned*)row : ((unsigned)row[0] | ((unsigned)row[1] << 16));
    }
  };
  auto multiply = [&](const LoEntry& m, const unsigned (&xv)[LO_UN]) {
#pragma unroll
    for (int u = 0; u < LO_UN; u++) {
      const double v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m.v), u), __builtin_amdgcn_readlane(__double2loint(m.v), u));
      const int col = __builtin_amdgcn_readlane((int)m.aux, u);
      double pa = bf16_to_f64(xv[u] & 0xFFFFu), pb = bf16_to_f64(xv[u] >> 16);
      if (COLS) {
        const bool deep = ((xv[u] & 0x7FFFu) - 1u < lim_a - 1u) || (((xv[u] >> 16) & 0x7FFFu) - 1u < lim_b - 1u);
        if (__ballot(deep)) {
          pa = floor(pa * ia + LO_ROUND) * qa;                             // exact (powers of two, one floor)
          pb = floor(pb * ib + LO_ROUND) * qb;
        }
        lds_add_f64(acc + lane * LO_PITCH + lo_perm(g * 32 + col), v * pa);            // rows 2 lane, 2 lane + 1
        lds_add_f64(acc + (64 + lane) * LO_PITCH + lo_perm(g * 32 + col), v * pb);
      } else {
        lds_add_f64(acc + lo_perm(g * 32 + col) * LO_PITCH + lane, v * pa);            // columns 2 lane, 2 lane + 1
        lds_add_f64(acc + lo_perm(g * 32 + col) * LO_PITCH + 64 + lane, v * pb);
      }
    }
  };
  if (cnt == 0) return;
  const int nb = (cnt + LO_UN - 1) / LO_UN;
  LoEntry mA = first.m, mB;
  unsigned xA[LO_UN], xB[LO_UN];
  issue(mA, xA);
  for (int k = 0; k < nb; k += 2) {
    if (k + 1 < nb) {
      fetch(k + 1, mB);
      issue(mB, xB);
    }
    __builtin_amdgcn_sched_barrier(0);    // (batch B's loads are out before batch A's are waited for: hipcc would sink each load to its use)
    multiply(mA, xA);
    if (k + 1 >= nb) break;
    if (k + 2 < nb) {
      fetch(k + 2, mA);
      issue(mA, xA);
    }
    __builtin_amdgcn_sched_barrier(0);
    multiply(mB, xB);
  }
}

// One workgroup of sixteen waves per 128 x 128 tile of the lower triangle (per-head statistics: the diagonal tiles): wave (g, sub)
// takes the list `sub` of row group g, then of column group g.
constexpr int LO_THREADS = 1024;
__global__ __launch_bounds__(LO_THREADS) void i8_lo_product_kernel(LoArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lo_acc[];     // [128][LO_PITCH]
  if (a.state[EXACT_RAN] != 1 || a.state[EXACT_OVERFLOW] != 0 || a.state[EXACT_MODE] != 1) return;
  int p = 0;
  while (p + 1 < a.nprob && (int)blockIdx.x >= a.prob[p + 1].tile0[2]) p++;
  if (a.route_flag[p] & 2) return;
  const LoProblem& pr = a.prob[p];
  const int t = blockIdx.x - pr.tile0[2];
  int bi, bj;
  if (pr.block) {
    bi = bj = t;
  } else {
    bi = (int)((sqrtf(8.f * t + 1.f) - 1.f) * 0.5f);
    while ((bi + 1) * (bi + 2) / 2 <= t) bi++;
    while (bi * (bi + 1) / 2 > t) bi--;
    bj = t - bi * (bi + 1) / 2;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave & 3, sub = wave >> 2;
  // anything to do?  (the lists of the tile's four row groups and four column groups)
  int any = 0;
  if (tid < 8 * LO_SUB) any = pr.rtotals[(tid < 4 * LO_SUB ? 4 * bi * LO_SUB : 4 * bj * LO_SUB - 4 * LO_SUB) + tid];
  if (!__syncthreads_or(any)) return;
  const LoFirst first_rows = lo_first(pr, 4 * bi + g, sub, lane), first_cols = lo_first(pr, 4 * bj + g, sub, lane);
  for (int i = tid; i < LO_TILE * LO_PITCH; i += LO_THREADS) lo_acc[i] = 0.;
  __syncthreads();
  lo_events<false>(pr, a.nch, 4 * bi + g, sub, bj * LO_TILE, g, lane, lo_acc, first_rows);
  __syncthreads();      // an accumulator changes owner between the two passes: the wave of its row, then the wave of its column
  lo_events<true>(pr, a.nch, 4 * bj + g, sub, bi * LO_TILE, g, lane, lo_acc, first_cols);
  __syncthreads();
  // the tile's 16 elements of a thread: all their sigma loads first, then the additions and the stores (written as `*s += v` behind
  // the tests, every element paid a memory round trip of its own: 16 in series per tile, most of the kernel's time on sparse lists)
  constexpr int PER = LO_TILE * LO_TILE / LO_THREADS;
  const int c = tid % LO_TILE, col = bj * LO_TILE + c;
  const int e_col = col < pr.n ? pr.emax[col] : EMAX_COLUMN_OUT;
  double* s[PER];
  double old[PER], v[PER];
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int r = tid / LO_TILE + u * (LO_THREADS / LO_TILE), row = bi * LO_TILE + r;
    v[u] = lo_acc[lo_perm(r) * LO_PITCH + lo_perm(c)];
    // (rows / columns of the fp64 column kernel are not ours)
    const bool ours = col <= row && row < pr.n && v[u] != 0. && !((pr.emax[row] | e_col) & EMAX_COLUMN_OUT);
    s[u] = ours ? pr.sigma + (int64_t)row * pr.ld_sigma + col - (pr.block ? row / pr.block * pr.block : 0) : nullptr;
    old[u] = ours ? *s[u] : 0.;
  }
#pragma unroll
  for (int u = 0; u < PER; u++)
    if (s[u]) *s[u] = old[u] + v[u];
}

// The two remainder products.  One workgroup = the 16 columns of group G against a block of LW_BLOCK = 512 partner columns; wave w
// owns column G 16 + w, and every lane EIGHT neighbouring partner columns (one 16-byte load
// per event), whose eight sums it keeps in registers while it walks the column's list -- one v_fma_f64 per product, events in list
// order (run-to-run bit-identical; no atomics, no LDS) -- and adds to sigma when the column is done:
//   TR = false   sigma[r][c] += sum_t x_lo(t, r) x(t, c)      for the partner columns c <= r   (X_lo^T X, lower part; a contiguous row)
//   TR = true    sigma[c][r] += sum_t x_lo(t, r) x_d(t, c)    for the partner columns c >= r   (X_d^T X_lo, lower part; partner x_d from
//                the copy, so both products are the same loop: the first versions recomputed x_d from x whenever a lane met an
//                element with digits below plane 2 -- at eight columns per lane nearly every event does)
// Two launches, the second after the first (an entry of sigma gets a sum from each).  Why this shape: the kernel is bound by its
// INSTRUCTION stream -- per event and wave three broadcasts and an address, then per product an unpack, a conversion and the fma; at
// two partner columns per lane (the 128 x 128-tile versions: profiles/r04_exact_route_kernels_silu_gated.csv, 9.3 ms) the fixed part
// and two LDS atomics per event were most of it.  Workgroups run partner block by partner block (P-major): the 256 that are
// resident walk their lists in token order over the SAME 512 columns of x -- 32 MB that stay in the memory-side cache.
#ifndef MDG_LW_UN
#define MDG_LW_UN 8
#endif
#ifndef MDG_LW_OCC
#define MDG_LW_OCC 0      // 8: two workgroups per CU (64 VGPRs)
#endif
#if MDG_LW_OCC
#define LW_OCC_ATTR __attribute__((amdgpu_waves_per_eu(MDG_LW_OCC, MDG_LW_OCC)))
#else
#define LW_OCC_ATTR
#endif
constexpr int LW_COLS = 8, LW_BLOCK = 64 * LW_COLS, LW_UN = MDG_LW_UN, LW_GROUP = 16, LW_TP = LW_GROUP + 1;
template <bool TR>
__global__ __launch_bounds__(LO_THREADS) LW_OCC_ATTR void i8_lo_wide_kernel(LoArgs a) {
  if (a.state[EXACT_RAN] != 1 || a.state[EXACT_OVERFLOW] != 0 || a.state[EXACT_MODE] != 2) return;
  int p = 0;
  while (p + 1 < a.nprob && (int)blockIdx.x >= a.prob[p + 1].tile0[TR]) p++;
  if (a.route_flag[p] & 2) return;
  const LoProblem& pr = a.prob[p];
  const int n = pr.n, nG = n / LW_GROUP;
  constexpr int PER = LW_BLOCK / LW_GROUP;     // groups per partner block
  int t = blockIdx.x - pr.tile0[TR], G, P = 0;
  if (pr.block) {            // per-head statistics: the one partner block is the head
    G = t;
  } else {                   // partner block P of 512 columns, then the groups that have a column on the right side of it
    for (;;) {
      const int cnt = TR ? min(nG, PER * (P + 1)) : nG - PER * P;
      if (t < cnt) break;
      t -= cnt;
      P++;
    }
    G = TR ? t : PER * P + t;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* const xs = TR ? (const char*)pr.xd : (const char*)pr.x;
  typedef unsigned u32x4u __attribute__((ext_vector_type(4), aligned(2)));
  __shared__ double tr_tile[TR ? LW_BLOCK * LW_TP : 1];      // TR: [partner row][column of the group], for the transposed fold
  const int r = G * LW_GROUP + wave;
  const int total = __builtin_amdgcn_readfirstlane(lo_totals(pr, a.nch)[r]);
  const bool r_ours = total != 0 && !(pr.emax[r] & EMAX_COLUMN_OUT);
  const LoEntry* list = pr.entries + (int64_t)r * a.nch * LO_CAP;
  LoEntry m0 = LoEntry{0., 0u, 0u};            // the first 64 entries of the column's list: one per lane
  if (r_ours) {
    m0 = list[min(lane, min(total, 64) - 1)];
    if (lane >= total) m0.v = 0.;              // (padding: the last entry's row, exact zeros)
  }
  {
    const int p0 = pr.block ? G * LW_GROUP / pr.block * pr.block : P * LW_BLOCK;
    const int pend = pr.block ? p0 + pr.block : min(n, p0 + LW_BLOCK);
    const int c0 = p0 + LW_COLS * lane;                                   // this lane's partner columns c0 .. c0 + 7
    // which of them exist, are ours (columns of the fp64 column kernel are not) and lie on this product's side of the diagonal
    unsigned mine = 0;
#pragma unroll
    for (int j = 0; j < LW_COLS; j++)
      if (c0 + j < pend && !(pr.emax[c0 + j] & EMAX_COLUMN_OUT) && (TR ? c0 + j >= r : c0 + j <= r)) mine |= 1u << j;
    const unsigned lane_off = (unsigned)(c0 + LW_COLS <= pend ? c0 : p0) * 2u;   // (lanes beyond the block read its first columns; never used)
    const bool walk = r_ours && __ballot(mine != 0) != 0;
    double acc[LW_COLS];
#pragma unroll
    for (int j = 0; j < LW_COLS; j++) acc[j] = 0.;
    if (walk) {
      // super-batches of 64 entries (one per lane, broadcast with v_readlane), batches of LW_UN events whose partner loads are all
      // issued before the previous batch is multiplied
      for (int sb = 0; sb < total; sb += 64) {
        const int len = min(64, total - sb);
        LoEntry m = m0;
        if (sb) {
          m = list[sb + min(lane, len - 1)];
          if (lane >= len) m.v = 0.;
        }
        const unsigned moff = TR ? m.aux : m.off;
        u32x4 xa[LW_UN], xb[LW_UN];
        auto issue = [&](int b, u32x4 (&xv)[LW_UN]) {
#pragma unroll
          for (int u = 0; u < LW_UN; u++) {
            const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)moff, b * LW_UN + u) + lane_off;
            const u32x4u q = *(const u32x4u*)(xs + o);
            xv[u] = (u32x4){q[0], q[1], q[2], q[3]};
          }
        };
        auto multiply = [&](int b, const u32x4 (&xv)[LW_UN]) {
#pragma unroll
          for (int u = 0; u < LW_UN; u++) {
            const double v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m.v), b * LW_UN + u),
                                              __builtin_amdgcn_readlane(__double2loint(m.v), b * LW_UN + u));
#pragma unroll
            for (int q = 0; q < 4; q++) {
              acc[2 * q] = fma(v, (double)__uint_as_float(xv[u][q] << 16), acc[2 * q]);
              acc[2 * q + 1] = fma(v, (double)__uint_as_float(xv[u][q] & 0xFFFF0000u), acc[2 * q + 1]);
            }
          }
        };
        const int nb = (len + LW_UN - 1) / LW_UN;      // 1 .. 8 batches
        issue(0, xa);
        for (int b = 0; b < nb; b += 2) {
          __builtin_amdgcn_sched_barrier(0);
          if (b + 1 < nb) issue(b + 1, xb);
          __builtin_amdgcn_sched_barrier(0);
          multiply(b, xa);
          if (b + 1 >= nb) break;
          __builtin_amdgcn_sched_barrier(0);
          if (b + 2 < nb) issue(b + 2, xa);
          __builtin_amdgcn_sched_barrier(0);
          multiply(b + 1, xb);
        }
      }
    }
    if (!TR) {
      if (!walk) return;
      // the column's sums into its row of sigma (64 contiguous bytes per lane): all loads first
      double* s[LW_COLS];
      double old[LW_COLS];
#pragma unroll
      for (int j = 0; j < LW_COLS; j++) {
        s[j] = (mine >> j & 1) ? pr.sigma + (int64_t)r * pr.ld_sigma + c0 + j - (pr.block ? r / pr.block * pr.block : 0) : nullptr;
        old[j] = s[j] ? *s[j] : 0.;
      }
#pragma unroll
      for (int j = 0; j < LW_COLS; j++)
        if (s[j]) *s[j] = old[j] + acc[j];
      return;
    }
    // TR: the sums belong to COLUMN r of sigma.  Written from here they are 8-byte accesses a row pitch apart, sixteen waves on the
    // same 128-byte lines one after the other (measured: the fold's L2 requests were 80 % of the walk's); through LDS every thread
    // folds eight neighbouring columns of one partner row, 64 contiguous bytes.
#pragma unroll
    for (int j = 0; j < LW_COLS; j++) tr_tile[(LW_COLS * lane + j) * LW_TP + wave] = (walk && (mine >> j & 1)) ? acc[j] : 0.;
    __syncthreads();
    {
      const int i = p0 + (threadIdx.x >> 1), half = threadIdx.x & 1;      // partner row i, columns G 16 + 8 half ..
      if (i < pend && !(pr.emax[i] & EMAX_COLUMN_OUT)) {
        double v[8], old[8];
        double* s[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int col = G * LW_GROUP + 8 * half + k;
          v[k] = tr_tile[(i - p0) * LW_TP + 8 * half + k];
          s[k] = (v[k] != 0. && col <= i) ? pr.sigma + (int64_t)i * pr.ld_sigma + col - (pr.block ? i / pr.block * pr.block : 0) : nullptr;
          old[k] = s[k] ? *s[k] : 0.;
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (s[k]) *s[k] = old[k] + v[k];
      }
    }
  }
}

// column maxima (n ints, padded to 8 bytes) + the [NSTAT][n] route statistics + the route kernel's ticket: zeroed together per call
size_t ints_bytes(int64_t n) { return (size_t)((n + 1) / 2 * 2) * sizeof(int) + (size_t)(NSTAT * n) * sizeof(unsigned long long) + sizeof(RouteScratch); }
size_t planes_bytes(int64_t T, int64_t n) { return (size_t)NP * (size_t)n * (size_t)ceil_div(T, KS) * KS; }
size_t zmask_bytes(int64_t T, int64_t n) { return align_up((size_t)ceil_div(T, KS) * (size_t)(n / 32), 256); }


// ---- tile schedule of the persistent, XCD-lock-step launch (i8_syrk_kernel with SyrkArgs::sched)
// Groups of up to 32 tiles = one XCD's 32 CUs for one round.  The lower region is cut into macro-rows of 4 tile rows and those
// into chunks of 8 tile columns: a full group is a 4 x 8 block of tiles -- 4 A panels and 8 B panels shared by 32 tiles.  The
// ragged groups along the diagonal are then packed (tiles of the smallest ones fill up the largest), so that ceil(tiles / 32)
// groups -- and as few rounds as the tile count allows -- remain.  Built once per (device, tile-row count, tile shape) on the
// host and kept on the device: a few KB of immutable lookup data, the one allocation the library keeps across calls.
struct Schedule {
  int2* dev = nullptr;     // [ngroups][32] {tile code, k-chunk code}
  int4* tail = nullptr;    // [n_tail] {tile code, Q, first partial slot, 0}
  int ngroups = 0, n_tail = 0, pieces = 0;
};
constexpr int TAIL_MAX_Q = 16;         // k-chunks per tile of the split round(s) (a chunk should stay much longer than the 2-3 k-steps of ring fill)
constexpr int TAIL_MAX_PIECES = 1024;  // partial tiles (chunks of all split tiles together)
constexpr size_t PARTIAL_BYTES = (size_t)TAIL_MAX_PIECES * TI * 128 * sizeof(double);   // partial tiles of at most 128 x 128

// shapes: per statistic {row blocks of 128 features, block (0 = full lower triangle, 128 = per-head diagonal tiles)}
// The one thing the library keeps across calls: device copies of the schedules, a few KB each, keyed by (device, tile shape,
// statistic shapes).  Plain device memory -- no streams, no events (those are the caller's) -- released by mdg_shutdown(); the
// containers' destructors at process exit free host memory only and make no HIP call (the runtime may be gone by then).
std::mutex g_sched_mutex;
std::map<std::vector<int>, Schedule> g_sched_cache;

const Schedule* schedule_for(const std::vector<std::pair<int, int>>& shapes, int cw) {   // cw: tile columns per 128 features (1: 128 x 128 tiles, 2: 128 x 64)
  auto& cache = g_sched_cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_sched_mutex);
  std::vector<int> key = {dev, cw};
  for (auto& sh : shapes) {
    key.push_back(sh.first);
    key.push_back(sh.second);
  }
  auto it = cache.find(key);
  if (it != cache.end()) return &it->second;
  std::vector<std::vector<int>> full, ragged;
  for (size_t pi = 0; pi < shapes.size(); pi++) {
    const int rb = shapes[pi].first, pbits = (int)pi << CODE_PROB;
    if (shapes[pi].second) {   // per-head statistic: the diagonal tiles only
      std::vector<int> g;
      for (int h = 0; h < rb; h++)
        for (int c = 0; c < cw; c++) {
          g.push_back(pbits | (h << CODE_BI) | (h * cw + c));
          if (g.size() == 32) {
            full.push_back(g);
            g.clear();
          }
        }
      if (!g.empty()) ragged.push_back(g);
      continue;
    }
    for (int R = 0; R * 4 < rb; R++) {
      const int r1 = std::min(rb, R * 4 + 4);
      const int ncols = r1 * cw;                       // columns of the macro-row's last tile row
      for (int c0 = 0; c0 < ncols; c0 += 8) {
        std::vector<int> g;
        for (int bi = R * 4; bi < r1; bi++)
          for (int bj = c0; bj < c0 + 8; bj++)
            if (bj < (bi + 1) * cw) g.push_back(pbits | (bi << CODE_BI) | bj);
        if (g.size() == 32) full.push_back(g);
        else if (!g.empty()) ragged.push_back(g);
      }
    }
  }
  std::sort(ragged.begin(), ragged.end(), [](const std::vector<int>& x, const std::vector<int>& y) { return x.size() > y.size(); });
  size_t lo = 0, hi = ragged.size();
  while (lo + 1 < hi) {                               // fill the largest ragged group from the smallest one
    std::vector<int>& big = ragged[lo];
    std::vector<int>& small = ragged[hi - 1];
    while (big.size() < 32 && !small.empty()) {
      big.push_back(small.back());
      small.pop_back();
    }
    if (small.empty()) hi--;
    if (big.size() == 32) lo++;
  }
  std::vector<std::vector<int>> groups(full);        // all groups hold 32 tiles, except possibly the last one
  for (size_t i = 0; i < hi; i++)
    if (!ragged[i].empty()) groups.push_back(ragged[i]);
  size_t tiles = 0;
  for (auto& g : groups) tiles += g.size();
  std::vector<int2> table;
  std::vector<int4> tail;
  auto emit = [&](const std::vector<int>& g) {
    for (int i = 0; i < 32; i++) table.push_back(make_int2(i < (int)g.size() ? g[i] : -1, 0));
  };
  const size_t whole_groups = tiles / 256 * 8;        // the full rounds: 8 groups of 32 whole tiles each
  std::vector<int> rest;                              // tiles of the last, partly filled round
  for (size_t i = whole_groups; i < groups.size(); i++) rest.insert(rest.end(), groups[i].begin(), groups[i].end());
  // Q chunks per tile turn the R left-over tiles into R Q pieces worked in ceil(R Q / 256) short rounds of 1 / Q tile each
  // (+ ~4 % of a tile per round for the ring fill and the fold of a chunk): take the cheapest Q
  const int R = (int)rest.size();
  int Q = 1;
  double best = 1.0;
  for (int q = 2; q <= TAIL_MAX_Q && R > 0; q++) {
    if (R * q > TAIL_MAX_PIECES) break;
    const double cost = (double)((R * q + 255) / 256) * (1.0 / q + 0.04);
    if (cost < best - 0.02) { best = cost; Q = q; }
  }
  Schedule sch;
  if (Q >= 2) {
    for (size_t i = 0; i < whole_groups; i++) emit(groups[i]);
    // piece t Q + q = chunk q of tile t; piece p runs in tail round p / 256 on XCD p % 8; its partial tile is slot p
    const int pieces = R * Q, tail_rounds = (pieces + 255) / 256;
    std::vector<int2> last((size_t)tail_rounds * 256, make_int2(-1, 0));
    for (int t = 0; t < R; t++) {
      tail.push_back(make_int4(rest[t], Q, t * Q, 0));
      for (int q = 0; q < Q; q++) {
        const int piece = t * Q + q, idx = piece % 256;
        last[(size_t)(piece / 256) * 256 + (idx % 8) * 32 + idx / 8] = make_int2(rest[t], (piece << 10) | (Q << 5) | q);
      }
    }
    table.insert(table.end(), last.begin(), last.end());
    sch.n_tail = (int)tail.size();
    sch.pieces = pieces;
  } else {
    for (auto& g : groups) emit(g);
  }
  sch.ngroups = (int)(table.size() / 32);
  if (hipMalloc((void**)&sch.dev, table.size() * sizeof(int2)) != hipSuccess) return nullptr;
  if (hipMemcpy(sch.dev, table.data(), table.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  if (!tail.empty()) {
    if (hipMalloc((void**)&sch.tail, tail.size() * sizeof(int4)) != hipSuccess) return nullptr;
    if (hipMemcpy(sch.tail, tail.data(), tail.size() * sizeof(int4), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  }
  return &cache.emplace(key, sch).first->second;
}

// Workspace layout of a call: [shared block: route flag, executed-MFMA counter, XCD arrival counters][partial tiles]
// then per statistic [digit planes][column maxima, route statistics][alpha / rho][RouteOut][column-kernel partials][piece masks],
// then the fp64 fallback's split-K space.
constexpr size_t SHARED_BYTES = 256;
struct ProblemWs {
  size_t planes, ints, vals, route, colpart, zmask, lo_entries, lo_counts, lo_xd, lo_rentries, lo_rtotals;   // byte offsets
};
int lo_chunks(int64_t T) { return (int)ceil_div(ceil_div(T, (int64_t)KS), (int64_t)LO_CHUNK_STEPS); }
size_t layout(int count, const mdg_cov_problem* pr, ProblemWs* out, size_t* fallback_off) {
  size_t off = SHARED_BYTES + PARTIAL_BYTES, fb = 0;
  for (int i = 0; i < count; i++) {
    const int64_t cols = pr[i].n_feat * pr[i].batch;
    ProblemWs w;
    w.planes = off;
    off += align_up(planes_bytes(pr[i].n_tokens, cols), 256);
    w.ints = off;                                   // column maxima (n ints, padded to 8 bytes), then the [NSTAT][n] route statistics
    off += align_up(ints_bytes(cols), 256);
    w.vals = off;                                   // alpha_s / rho per column, then the route kernel's per-workgroup partial maxima
    off += align_up((size_t)(NVAL * cols) * sizeof(double) + (size_t)ceil_div(cols, (int64_t)ROUTE_THREADS) * sizeof(RoutePartial), 256);
    w.route = off;
    off += align_up(sizeof(RouteOut), 256);
    w.colpart = off;                                // chunk partials of the fp64 column kernel
    off += align_up((size_t)ROUTE_JMAX * COLK_CHUNKS * (size_t)cols * sizeof(double), 256);
    w.zmask = off;
    off += zmask_bytes(pr[i].n_tokens, cols);
    w.lo_entries = off = align_up(off, 256);     // the exact route's event lists: [cols][chunks][LO_CAP] x 16 bytes, then the counts
    off += (size_t)cols * lo_chunks(pr[i].n_tokens) * LO_CAP * sizeof(LoEntry);
    w.lo_counts = off;
    off += align_up((size_t)cols * (lo_chunks(pr[i].n_tokens) + 1) * sizeof(int), 256);
    w.lo_xd = off;                               // the x_d copy of the exact route: [tokens][cols] bf16
    off += align_up((size_t)cols * (size_t)pr[i].n_tokens * sizeof(bf16_t), 256);
    w.lo_rentries = off;                         // sparse mode: the merged (group, residue) lists and their lengths
    off += (size_t)(cols / 32) * LO_SUB * LO_RCAP * sizeof(LoEntry);
    w.lo_rtotals = off;
    off += align_up((size_t)(cols / 32) * LO_SUB * sizeof(int), 256);
    if (out) out[i] = w;
    fb = std::max(fb, mdg_cov_accum_ws_bytes(pr[i].n_tokens, pr[i].n_feat, pr[i].batch));
  }
  if (fallback_off) *fallback_off = off;
  return off + fb + 256;
}

bool problems_ok(int count, const mdg_cov_problem* pr) {
  if (count < 1 || count > MAX_PROBLEMS || !pr) return false;
  for (int i = 0; i < count; i++) {
    if (pr[i].n_tokens != pr[0].n_tokens || pr[i].n_tokens < 0 || pr[i].n_tokens >= (1ll << 28) || pr[i].n_feat <= 0 || pr[i].batch < 1) return false;
    if (pr[i].batch == 1 ? pr[i].n_feat % TI != 0 : pr[i].n_feat != TI) return false;   // per-head statistics: head_dim 128 only
    if (pr[i].n_feat * pr[i].batch >= (1 << 21)) return false;
    if (pr[i].ld < pr[i].n_feat * pr[i].batch || pr[i].ld_sigma < pr[i].n_feat) return false;
    if (pr[i].batch > 1 && (pr[i].ld_sigma != TI || pr[i].sigma_batch_stride != (int64_t)TI * TI)) return false;
  }
  return true;
}

}  // namespace

// mdg_shutdown(): give the cached schedules back.  The caller guarantees no int8 covariance call is in flight.
int release_i8_schedules() {
  std::lock_guard<std::mutex> lock(g_sched_mutex);
  int dev0 = 0;
  const bool have_dev = hipGetDevice(&dev0) == hipSuccess;
  int rc = MDG_OK;
  for (auto& kv : g_sched_cache) {
    if (hipSetDevice(kv.first[0]) != hipSuccess) { rc = MDG_ERR_HIP; continue; }
    if (kv.second.dev && hipFree(kv.second.dev) != hipSuccess) rc = MDG_ERR_HIP;
    if (kv.second.tail && hipFree(kv.second.tail) != hipSuccess) rc = MDG_ERR_HIP;
  }
  g_sched_cache.clear();
  if (have_dev) (void)hipSetDevice(dev0);
  (void)hipGetLastError();
  return rc;
}
}  // namespace mdg

using namespace mdg;

extern "C" size_t mdg_cov_accum_i8_multi_ws_bytes(int count, const mdg_cov_problem* problems) {
  if (!problems_ok(count, problems) || problems[0].n_tokens == 0) return 0;
  return layout(count, problems, nullptr, nullptr);
}

extern "C" int mdg_cov_accum_i8_multi(int count, const mdg_cov_problem* problems, void* ws, size_t ws_bytes, double tolerance, int flags,
                                      int* used_i8, int* route_counts, void* ev_start, void* ev_stop, void* stream) {
  MDG_CLEAR();
  if (used_i8) *used_i8 = 0;
  MDG_CHECK_ARG(tolerance >= 1.0 && tolerance <= 1e6, "mdg_cov_accum_i8_multi: tolerance factor %g outside [1, 1e6] (1 = guaranteed <= 1.1e-11)",
                tolerance);
  MDG_CHECK_ARG((flags & ~(MDG_I8_NO_EXACT | MDG_I8_EXACT_ALWAYS)) == 0 && flags != (MDG_I8_NO_EXACT | MDG_I8_EXACT_ALWAYS),
                "mdg_cov_accum_i8_multi: bad flags 0x%x", flags);
  bool offer_exact = !(flags & MDG_I8_NO_EXACT);
  for (int i = 0; i < count; i++)     // (the event lists address a token's row with a 32-bit byte offset)
    if ((uint64_t)problems[i].n_tokens * (uint64_t)problems[i].ld * 2ull >= (1ull << 32)) offer_exact = false;
  MDG_CHECK_ARG(problems_ok(count, problems),
                "mdg_cov_accum_i8_multi: 1..%d statistics of the same token count; full ones need n_feat %% 128 == 0, per-head ones "
                "head_dim 128 with contiguous [heads][128][128] sigma; leading dimensions at least the widths (use mdg_cov_accum)",
                MAX_PROBLEMS);
  const int64_t n_tokens = problems[0].n_tokens;
  if (n_tokens == 0) return MDG_OK;
  for (int i = 0; i < count; i++) MDG_CHECK_ARG(problems[i].x && problems[i].sigma, "mdg_cov_accum_i8_multi: null pointer");
  ProblemWs pw[MAX_PROBLEMS];
  size_t fb_off = 0;
  const size_t need = layout(count, problems, pw, &fb_off);
  MDG_CHECK_ARG(ws && ws_bytes >= need, "mdg_cov_accum_i8_multi: workspace %zu < required %zu", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const int nk = (int)ceil_div(n_tokens, KS);
  int* flag = (int*)ws;                                               // [2..3]: executed-MFMA counter;  [4..11]: tile-queue counters;
  int* pflag = flag + 12;                                             // [12..15]: route bits per statistic (i8_depth_kernel)
  unsigned long long* mfma_count = (unsigned long long*)(flag + 2);
  double* partial = (double*)((char*)ws + SHARED_BYTES);
  void* fb_ws = (char*)ws + fb_off;
  MDG_HIP(hipMemsetAsync(flag, 0, SHARED_BYTES, st));
  SyrkArgs a;
  a.nprob = count;
  a.nk = nk;
  std::vector<std::pair<int, int>> shapes;
  for (int i = 0; i < count; i++) {
    const mdg_cov_problem& q = problems[i];
    const int n = (int)(q.n_feat * q.batch);    // columns of the activation matrix
    signed char* planes = (signed char*)ws + pw[i].planes;
    int* emax = (int*)((char*)ws + pw[i].ints);
    unsigned long long* stats = (unsigned long long*)(emax + (n + 1) / 2 * 2);
    unsigned char* zmask = (unsigned char*)ws + pw[i].zmask;
    MDG_HIP(hipMemsetAsync(emax, 0, ints_bytes(n), st));
    const bool vec = ((uintptr_t)q.x % 16 == 0) && (q.ld % 8 == 0);
    const int64_t rows_per_block = 2048;
    if (vec) {
      // (the maximum pass of a NARROW statistic: with 2048 tokens per workgroup 1024 columns are 128 workgroups walking 128 dependent
      //  16-byte loads each -- 82 us for 67 MB.  Token slabs sized for ~MDG_COLMAX_WGS workgroups in all, 64 tokens at least)
      const int64_t slabs = std::min(ceil_div(n_tokens, (int64_t)64), std::max((int64_t)1, (int64_t)MDG_COLMAX_WGS / (n / 128)));
      const int64_t rows_vec = ceil_div(ceil_div(n_tokens, slabs), (int64_t)16) * 16;
      hipLaunchKernelGGL(i8_colmax_vec_kernel, dim3((unsigned)(n / 128), (unsigned)ceil_div(n_tokens, rows_vec)), dim3(256), 0,
                         st, (const bf16_t*)q.x, q.ld, n_tokens, rows_vec, emax);
      hipLaunchKernelGGL(i8_split_vec_kernel, dim3((unsigned)(n / 128), (unsigned)ceil_div(nk, 2 * SPLIT_TILES)), dim3(256), 0, st,
                         (const bf16_t*)q.x, q.ld, n_tokens, n, nk, emax, planes, stats, zmask);
    } else {
      hipLaunchKernelGGL(i8_colmax_kernel, dim3((unsigned)ceil_div(n, 64), (unsigned)ceil_div(n_tokens, rows_per_block)), dim3(256),
                         0, st, (const bf16_t*)q.x, q.ld, n_tokens, n, rows_per_block, emax);
      hipLaunchKernelGGL(i8_split_kernel, dim3((unsigned)(n / 32), (unsigned)ceil_div(nk, SPLIT_STEPS)), dim3(256), 0, st,
                         (const bf16_t*)q.x, q.ld, n_tokens, n, nk, emax, planes, stats, zmask);
    }
    // a flag per statistic: the launch takes the deepest route any statistic still on the int8 path asks for (launch_route)
    {
      double* vals = (double*)((char*)ws + pw[i].vals);
      RoutePartial* partial = (RoutePartial*)(vals + (size_t)NVAL * n);
      RouteScratch* scratch = (RouteScratch*)(stats + (size_t)NSTAT * n);      // (inside the region zeroed above)
      hipLaunchKernelGGL(i8_route_kernel, dim3((unsigned)ceil_div(n, ROUTE_THREADS)), dim3(ROUTE_THREADS), 0, st, stats, emax, n,
                         n_tokens, vals, partial, scratch, pflag + i, (RouteOut*)((char*)ws + pw[i].route), route_counts,
                         tolerance);
    }
    hipLaunchKernelGGL(i8_clear_columns_kernel, dim3(ROUTE_JMAX, (unsigned)std::min(64, (nk + 3) / 4)), dim3(256), 0, st,
                       (const RouteOut*)((char*)ws + pw[i].route), pflag + i, emax, planes, zmask, n, nk);
    MDG_LAUNCH_CHECK();
    a.prob[i] = SyrkProblem{planes, emax, zmask, q.sigma, q.ld_sigma, n, q.batch > 1 ? TI : 0};
    shapes.emplace_back(n / TI, q.batch > 1 ? TI : 0);
  }
  for (int i = count; i < MAX_PROBLEMS; i++) a.prob[i] = a.prob[0];
  // All three routes are enqueued; the flag just written decides on the device which one does the work (the other launches'
  // workgroups exit on their first instruction: ~10 us each at the sigma_mlp grid).  No host round trip, graph-capturable.
  a.mfma_count = mfma_count;
  a.route_flag = pflag; a.route_counts = route_counts;
  a.xcd_arrive = flag + 4;
  // the exact route: the remainder lists of every statistic (planes 3 .. 5, after the route's columns were cleared); the product
  // launches below then read the outcome -- {overflow, ran} -- from the shared block
  LoArgs lo;
  a.exact_state = nullptr;
  if (offer_exact) {
    lo.nprob = count;
    lo.nk = nk;
    lo.nch = lo_chunks(n_tokens);
    lo.route_flag = pflag;
    lo.state = flag;
    lo.always = (flags & MDG_I8_EXACT_ALWAYS) ? 1 : 0;
    lo.n_tokens = n_tokens;
    int tiles[3] = {0, 0, 0}, max_groups = 0;
    for (int i = 0; i < count; i++) {
      const mdg_cov_problem& q = problems[i];
      const int n = (int)(q.n_feat * q.batch);
      LoProblem& l = lo.prob[i];
      l.x = (const bf16_t*)q.x; l.ld = q.ld;
      l.planes = a.prob[i].planes; l.zmask = a.prob[i].zmask; l.emax = a.prob[i].emax;
      l.entries = (LoEntry*)((char*)ws + pw[i].lo_entries);
      l.xd = (bf16_t*)((char*)ws + pw[i].lo_xd);
      l.rentries = (LoEntry*)((char*)ws + pw[i].lo_rentries);
      l.rtotals = (int*)((char*)ws + pw[i].lo_rtotals);
      l.pairs = ((uintptr_t)q.x % 4 == 0) && (q.ld % 2 == 0);
      l.counts = (int*)((char*)ws + pw[i].lo_counts);
      l.sigma = q.sigma; l.ld_sigma = q.ld_sigma;
      l.n = n; l.block = q.batch > 1 ? TI : 0;
      const int nG = n / LW_GROUP, per = LW_BLOCK / LW_GROUP;
      for (int tr = 0; tr < 2; tr++) {
        l.tile0[tr] = tiles[tr];
        if (q.batch > 1) {
          tiles[tr] += nG;
        } else {
          for (int P = 0; P * per < nG; P++) tiles[tr] += tr ? std::min(nG, per * (P + 1)) : nG - per * P;
        }
      }
      l.tile0[2] = tiles[2];
      const int rbi = n / TI;
      tiles[2] += q.batch > 1 ? rbi : rbi * (rbi + 1) / 2;
      max_groups = std::max(max_groups, n / 32);
    }
    for (int i = count; i < MAX_PROBLEMS; i++) lo.prob[i] = lo.prob[0];
    for (int k = 0; k < 3; k++) lo.tiles[k] = tiles[k];
    hipLaunchKernelGGL(i8_extract_lo_kernel, dim3((unsigned)max_groups, (unsigned)lo.nch, (unsigned)count), dim3(64), 0, st, lo);
    hipLaunchKernelGGL(i8_compact_lo_kernel, dim3((unsigned)(max_groups * 32), (unsigned)count), dim3(64), 0, st, lo);
    hipLaunchKernelGGL(i8_lo_mode_kernel, dim3(1), dim3(1024), 0, st, lo);
    hipLaunchKernelGGL(i8_residue_lo_kernel, dim3((unsigned)(max_groups * LO_SUB), (unsigned)count), dim3(64), 0, st, lo);
    hipLaunchKernelGGL(i8_copy_xd_kernel, dim3(2048u, (unsigned)count), dim3(256), 0, st, lo);
    hipLaunchKernelGGL(i8_patch_xd_kernel, dim3((unsigned)(max_groups * 32), (unsigned)count), dim3(64), 0, st, lo);
    MDG_LAUNCH_CHECK();
    a.exact_state = flag + EXACT_OVERFLOW;
  }
#ifdef MDG_I8_STAMPS
  static unsigned long long* stamps_dev = nullptr;
  const size_t stamps_n = (size_t)STAMP_WGS * NW * 8;
  if (!stamps_dev) MDG_HIP(hipMalloc(&stamps_dev, stamps_n * 8));
  MDG_HIP(hipMemsetAsync(stamps_dev, 0, stamps_n * 8, st));
  a.stamps = stamps_dev;
#endif
#ifdef MDG_I8_WGTIMES
  static unsigned long long* wg_dev = nullptr;
  if (!wg_dev) MDG_HIP(hipMalloc(&wg_dev, 256 * 64 * 8));
  MDG_HIP(hipMemsetAsync(wg_dev, 0, 256 * 64 * 8, st));
  a.wgtimes = wg_dev;
#endif
  const int n = a.prob[0].n;
  const int rb = n / TI;
  // the persistent launch: one workgroup per CU, tiles of all statistics from one static schedule
  const Schedule* sched_of[2] = {nullptr, nullptr};
  if (rb >= PERSISTENT_MIN_ROWS || count > 1 || problems[0].batch > 1) {
    int dev = 0, n_cu = 0;
    MDG_HIP(hipGetDevice(&dev));
    MDG_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    if (n_cu == 256)   // 8 XCDs x 32 CUs is what the tables are cut for
      for (int i = 0; i < 2; i++) sched_of[i] = schedule_for(shapes, i + 1);
  }
  MDG_CHECK_ARG((count == 1 && problems[0].batch == 1) || (sched_of[0] && sched_of[1]),
                "mdg_cov_accum_i8_multi: several statistics in one launch, and per-head statistics, need the persistent launch (a "
                "256-CU device); use mdg_cov_accum_i8 per full statistic and mdg_cov_accum for the per-head ones");
  {  // partial tiles of the k-split last round (only one of the two product launches runs: they share the region)
    size_t zero_bytes = 0;
    for (int i = 0; i < 2; i++)
      if (sched_of[i]) zero_bytes = std::max(zero_bytes, (size_t)sched_of[i]->pieces * TI * (i == 0 ? 128 : 64) * sizeof(double));
    if (zero_bytes) MDG_HIP(hipMemsetAsync(partial, 0, zero_bytes, st));
  }
  a.partial = partial;
  if (ev_start) MDG_HIP(hipEventRecord((hipEvent_t)ev_start, st));
  for (int planes_used : {3, 5, 6}) {       // 3: the exact route's product (the three top planes, all nine pairs)
    if (planes_used == 3 && !offer_exact) continue;
    const bool wide = wide_tile(planes_used);                                  // 128 x 128 tiles; six planes: 128 x 64
    const int tj = wide ? 128 : 64;
    const size_t lds = (size_t)ring_depth(planes_used) * steps_per_stage(planes_used) * planes_used * (PA + tj * KS);
    const int si = planes_used == 6 ? SB6 : SB5;                               // super-block rows (see the kernel)
    const int sr = (rb + si - 1) / si, nsb = sr * (sr + 1) / 2;                // super-block rows, super-blocks
    const int tps = wide ? si * si : 2 * si * si;                              // tiles per super-block
    dim3 grid((unsigned)((nsb + 7) / 8 * 8 * tps));
    const Schedule* sch = sched_of[wide ? 0 : 1];
    a.sched = nullptr;
    a.tail = nullptr;
    a.ngroups = 0;
    if (sch) {
      a.sched = sch->dev;
      a.tail = sch->tail;
      a.ngroups = sch->ngroups;
      grid = dim3(256);
    }
    if (planes_used == 3) {
      MDG_HIP(hipFuncSetAttribute((const void*)i8_syrk_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((i8_syrk_kernel<3>), grid, dim3(64 * NW), lds, st, a);
      if (sch && sch->n_tail) hipLaunchKernelGGL((i8_tail_combine_kernel<3>), dim3(sch->n_tail * (TI * tj / COMBINE_ELEMS)), dim3(256), 0, st, a, sch->n_tail);
    } else if (planes_used == 6) {
      MDG_HIP(hipFuncSetAttribute((const void*)i8_syrk_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((i8_syrk_kernel<6>), grid, dim3(64 * NW), lds, st, a);
      if (sch && sch->n_tail) hipLaunchKernelGGL((i8_tail_combine_kernel<6>), dim3(sch->n_tail * (TI * tj / COMBINE_ELEMS)), dim3(256), 0, st, a, sch->n_tail);
    } else {
      MDG_HIP(hipFuncSetAttribute((const void*)i8_syrk_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((i8_syrk_kernel<5>), grid, dim3(64 * NW), lds, st, a);
      if (sch && sch->n_tail) hipLaunchKernelGGL((i8_tail_combine_kernel<5>), dim3(sch->n_tail * (TI * tj / COMBINE_ELEMS)), dim3(256), 0, st, a, sch->n_tail);
    }
    MDG_LAUNCH_CHECK();
  }
  if (ev_stop) MDG_HIP(hipEventRecord((hipEvent_t)ev_stop, st));
  if (offer_exact) {   // the remainder products of the exact route (every workgroup exits at once when the truncated product ran instead)
    const size_t lds = (size_t)LO_TILE * LO_PITCH * sizeof(double);
    MDG_HIP(hipFuncSetAttribute((const void*)i8_lo_product_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(i8_lo_product_kernel, dim3((unsigned)lo.tiles[2]), dim3(LO_THREADS), lds, st, lo);          // sparse lists
    hipLaunchKernelGGL(i8_lo_wide_kernel<false>, dim3((unsigned)lo.tiles[0]), dim3(LO_THREADS), 0, st, lo);        // dense lists
    hipLaunchKernelGGL(i8_lo_wide_kernel<true>, dim3((unsigned)lo.tiles[1]), dim3(LO_THREADS), 0, st, lo);
    MDG_LAUNCH_CHECK();
  }
  // the columns the route took off the int8 path: their rows / columns of sigma from the fp64 column kernel (both launches exit at
  // once when there are none)
  for (int i = 0; i < count; i++) {
    const mdg_cov_problem& q = problems[i];
    const int n = (int)(q.n_feat * q.batch);
    ColArgs c;
    c.x = (const bf16_t*)q.x; c.ld = q.ld; c.T = n_tokens; c.n = n;
    c.vec = ((uintptr_t)q.x % 16 == 0) && (q.ld % 8 == 0);
    c.route = (const RouteOut*)((char*)ws + pw[i].route);
    c.flag = pflag + i;
    c.part = (double*)((char*)ws + pw[i].colpart);
    hipLaunchKernelGGL(i8_columns_kernel, dim3((unsigned)ceil_div(n, COLK_WG_COLS), COLK_CHUNKS, ROUTE_JMAX / COLK_GROUP), dim3(64), 0, st, c);
    hipLaunchKernelGGL(i8_columns_reduce_kernel, dim3((unsigned)ceil_div(n, 256), ROUTE_JMAX), dim3(256), 0, st, c, a.prob[i].emax, q.sigma,
                       q.ld_sigma, q.batch > 1 ? TI : 0);
    MDG_LAUNCH_CHECK();
  }
  // a statistic the bound cannot certify on six planes even without its ROUTE_JMAX worst columns (its flag's bit 1) -- that
  // statistic, and only that one -- goes through the fp64 kernel
  for (int i = 0; i < count; i++) {
    const mdg_cov_problem& q = problems[i];
    const int fb = cov_accum_gated(q.x, MDG_BF16, n_tokens, q.n_feat, q.batch, q.ld, 0, q.sigma, q.ld_sigma, q.sigma_batch_stride, fb_ws,
                                   ws_bytes - fb_off, pflag + i, 2, 2, stream);
    if (fb != MDG_OK) return fb;
  }
#ifdef MDG_I8_STAMPS
  {
    static unsigned long long host[STAMP_WGS * NW * 8];
    MDG_HIP(hipMemcpyAsync(host, a.stamps, sizeof(host), hipMemcpyDeviceToHost, st));
    MDG_HIP(hipStreamSynchronize(st));
    double sum[2][6] = {};
    long cnt[2] = {};
    for (int w = 0; w < STAMP_WGS * NW; w++) {
      const unsigned long long* o = host + (size_t)w * 8;
      if (!o[6]) continue;
      const int role = (w % NW) >= NW / 2;
      for (int i = 0; i < 6; i++) sum[role][i] += (double)o[i] / (double)o[6];
      cnt[role]++;
    }
    for (int role = 0; role < 2; role++)
      if (cnt[role])
        fprintf(stderr, "[stamps n=%d] waves %s: per k-step cycles (s_memtime): wait+barrier %.0f  refill-first %.0f  reads+mfma-issue %.0f  "
                        "wait+refill-last %.0f  | whole tile / nk %.0f  mfma/step %.1f  (%ld waves)\n", n, role ? "4-7" : "0-3",
                sum[role][0] / cnt[role], sum[role][1] / cnt[role], sum[role][2] / cnt[role], sum[role][3] / cnt[role],
                sum[role][4] / cnt[role], sum[role][5] / cnt[role], cnt[role]);
  }
#endif
#ifdef MDG_I8_WGTIMES
  {
    static unsigned long long host[256 * 64];
    MDG_HIP(hipMemcpyAsync(host, a.wgtimes, sizeof(host), hipMemcpyDeviceToHost, st));
    MDG_HIP(hipStreamSynchronize(st));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < 256; w++) if (host[w * 64]) { t0 = std::min(t0, host[w * 64]); t1 = std::max(t1, host[w * 64 + 1]); }
    if (t1) {
      std::vector<double> ends;
      double xcd_end[8] = {};
      for (int w = 0; w < 256; w++) { const double e = (host[w * 64 + 1] - t0) * 1e-5; ends.push_back(e); xcd_end[w & 7] = std::max(xcd_end[w & 7], e); }
      std::sort(ends.begin(), ends.end());
      fprintf(stderr, "[wgtimes n=%d] kernel %.3f ms; workgroup end times (ms): min %.3f  p10 %.3f  median %.3f  p90 %.3f  max %.3f; last end per XCD:", n,
              (t1 - t0) * 1e-5, ends[0], ends[25], ends[128], ends[230], ends[255]);
      for (int x = 0; x < 8; x++) fprintf(stderr, " %.3f", xcd_end[x]);
      // time of the last whole round's end and per-round durations of workgroup 0 and of the slowest workgroup
      int slow = 0;
      for (int w = 0; w < 256; w++) if (host[w * 64 + 1] > host[slow * 64 + 1]) slow = w;
      fprintf(stderr, "\n   slowest workgroup %d, its rounds end at (ms):", slow);
      for (int r = 0; r < 60 && host[slow * 64 + 2 + r]; r++) fprintf(stderr, " %.2f", (host[slow * 64 + 2 + r] - t0) * 1e-5);
      int fast = 0;
      for (int w = 0; w < 256; w++) if (host[w * 64 + 1] < host[fast * 64 + 1]) fast = w;
      fprintf(stderr, "\n   fastest workgroup %d, its rounds end at (ms):", fast);
      for (int r = 0; r < 60 && host[fast * 64 + 2 + r]; r++) fprintf(stderr, " %.2f", (host[fast * 64 + 2 + r] - t0) * 1e-5);
      fprintf(stderr, "\n");
    }
  }
#endif
  if (used_i8) {   // measurement / test mode: report the route this call took (costs the host a round trip)
    int pf[MAX_PROBLEMS] = {};
    MDG_HIP(hipMemcpyAsync(pf, pflag, sizeof(pf), hipMemcpyDeviceToHost, st));
    MDG_HIP(hipStreamSynchronize(st));
    int live = 0, six = 0;
    for (int i = 0; i < count; i++)
      if (!(pf[i] & 2)) {
        live++;
        six |= pf[i] & 1;
      }
    *used_i8 = live ? (six ? 6 : 5) : 0;     // the route of the statistics that stayed on the int8 path; 0: all went to the fp64 kernel
  }
  return MDG_OK;
}

static mdg_cov_problem single_problem(const void* x, int64_t n_tokens, int64_t n_feat, int64_t ld, double* sigma, int64_t ld_sigma) {
  mdg_cov_problem q;
  q.x = x; q.n_tokens = n_tokens; q.n_feat = n_feat; q.batch = 1; q.ld = ld;
  q.sigma = sigma; q.ld_sigma = ld_sigma; q.sigma_batch_stride = 0;
  return q;
}

extern "C" size_t mdg_cov_accum_i8_ws_bytes(int64_t n_tokens, int64_t n_feat) {
  if (n_tokens <= 0 || n_feat <= 0) return 0;
  const mdg_cov_problem q = single_problem(nullptr, n_tokens, n_feat, n_feat, nullptr, n_feat);
  return mdg_cov_accum_i8_multi_ws_bytes(1, &q);
}

extern "C" int mdg_cov_accum_i8(const void* x, int64_t n_tokens, int64_t n_feat, int64_t ld, double* sigma, int64_t ld_sigma,
                                void* ws, size_t ws_bytes, double tolerance, int flags, int* used_i8, int* route_counts, void* ev_start,
                                void* ev_stop, void* stream) {
  MDG_CLEAR();
  if (used_i8) *used_i8 = 0;
  MDG_CHECK_ARG(n_tokens >= 0 && n_feat > 0, "mdg_cov_accum_i8: bad sizes (tokens=%lld feat=%lld)", (long long)n_tokens,
                (long long)n_feat);
  MDG_CHECK_ARG(n_feat % TI == 0, "mdg_cov_accum_i8: n_feat=%lld must be a multiple of %d (use mdg_cov_accum)",
                (long long)n_feat, TI);
  const mdg_cov_problem q = single_problem(x, n_tokens, n_feat, ld, sigma, ld_sigma);
  return mdg_cov_accum_i8_multi(1, &q, ws, ws_bytes, tolerance, flags, used_i8, route_counts, ev_start, ev_stop, stream);
}

extern "C" int mdg_cov_accum_i8_route(int count, const mdg_cov_problem* problems, int stat, const void* ws, int* planes, int* n_columns,
                                      int* columns, double* bound, int* exact, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(problems_ok(count, problems) && stat >= 0 && stat < count && ws, "mdg_cov_accum_i8_route: bad arguments");
  ProblemWs pw[MAX_PROBLEMS];
  layout(count, problems, pw, nullptr);
  RouteOut r;
  int state[3] = {0, 0, 0};      // EXACT_OVERFLOW, EXACT_RAN, EXACT_MODE
  hipStream_t st = (hipStream_t)stream;
  MDG_HIP(hipMemcpyAsync(&r, (const char*)ws + pw[stat].route, sizeof(r), hipMemcpyDeviceToHost, st));
  MDG_HIP(hipMemcpyAsync(state, (const int*)ws + EXACT_OVERFLOW, sizeof(state), hipMemcpyDeviceToHost, st));
  MDG_HIP(hipStreamSynchronize(st));
  const bool was_exact = r.planes != 0 && state[1] == 1 && state[0] == 0;
  if (exact) *exact = was_exact ? state[2] : 0;      // 1: the remainder ran on the tile kernel (sparse lists), 2: on the wide kernels
  if (planes) *planes = r.planes;
  if (n_columns) *n_columns = r.n_out;
  if (columns)
    for (int i = 0; i < MDG_I8_MAX_COLUMNS; i++) columns[i] = i < r.n_out ? r.out[i] : -1;
  if (bound) {   // the exact route drops no plane pair: the rounded-element term and fp64 rounding are what is left
    bound[0] = was_exact ? r.rho + MDG_I8_EXACT_ROUNDING : r.sq;
    bound[1] = was_exact ? 0.0 : r.x;
  }
  return MDG_OK;
}

extern "C" int mdg_cov_accum_i8_stats(const void* ws, int64_t n_tokens, int64_t n_feat, unsigned long long* executed_mfma,
                                      void* stream) {
  MDG_CLEAR();
  (void)n_tokens;
  (void)n_feat;
  MDG_CHECK_ARG(ws && executed_mfma, "mdg_cov_accum_i8_stats: bad arguments");
  const void* src = (const char*)ws + 2 * sizeof(int);   // the shared block at the start of every int8 workspace
  hipStream_t st = (hipStream_t)stream;
  MDG_HIP(hipMemcpyAsync(executed_mfma, src, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  MDG_HIP(hipStreamSynchronize(st));
  return MDG_OK;
}
