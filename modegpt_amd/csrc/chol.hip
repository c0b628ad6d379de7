// Blocked fp64 Cholesky, Cholesky solve and diag-of-inverse for the Nystrom MLP path
// (compress_mlp.py:13-25 get_ridge_scores, :52-57 reduced solve).  128-wide panels: the diagonal block is
// factorised AND inverted inside one workgroup's LDS; everything else is GEMM on the fp64 MFMA core
// (panel solve = multiply by the inverted diagonal block, trailing update = lower-only SYRK).
#include "common.hpp"

namespace mdg {

int gemm_f64(int64_t M, int64_t N, int64_t K, double alpha, const void* A, int a_dtype, int64_t sa_i, int64_t sa_k,
             const int64_t* a_rows, const void* B, int b_dtype, int64_t sb_k, int64_t sb_j, double beta, void* C,
             int c_dtype, int64_t ldc, int64_t batch, int64_t a_bs, int64_t b_bs, int64_t c_bs, int flags,
             hipStream_t st);

__device__ __forceinline__ int64_t ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int NB = 128;
constexpr int DP = NB + 1;  // LDS pitch (fp64) of the diagonal block: odd pitch -> conflict-free column walks

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Factorise the nb x nb diagonal block at A (lower), write L back, write inv(L) (identity padded to 128x128) to inv.
// info: first failing global pivot index + 1 (0 = ok).
//
// One workgroup, block resident in LDS.  A column-by-column sweep needs a workgroup barrier per column and again two
// per column for the inverse (384 barriers, ~200 us: the factorisation was barrier-bound).  Instead the 128 columns go
// in 8 steps of 16: the 16x16 diagonal sub-block is factorised AND inverted by one wave entirely in registers (lane i
// owns row i; pivots and multipliers are broadcast with v_readlane, no LDS round trips, no barriers), the 16-wide panel
// below is solved by multiplying with that inverse, and the rank-16 update of the trailing part is spread over all 16
// waves: 3 barriers per step.  inv(L) is then assembled from the 16x16 inverses by three levels of block doubling,
// X21 = -X22 (L21 X11), T = L21 X11 written over L21.  Panel solve, trailing update and both doubling products run as
// 16 x 16 tiles on v_mfma_f64_16x16x4_f64 straight out of the LDS block (round 1 did them on the VALU, two LDS reads per
// FMA, T through global memory: 157 us per block, of which the 8 serial 16 x 16 factorisations are ~36).
// (a, xd, dinv: the workgroup's LDS -- NB * DP, NB and 16 * 17 doubles; 1024 threads)
__device__ __forceinline__ void potrf_diag_block(double* A, int64_t lda, int nb, int64_t k0, double* inv, int* info, double* a, double* xd,
                                                 double* dinv) {
  constexpr int PT = 1024, SB = 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mi = lane & 15, kq = lane >> 4;   // v_mfma_f64_16x16x4_f64: A[mi][kq], B[kq][mi], D[kq + 4 reg][mi]
  for (int e = tid; e < NB * NB; e += PT) {
    const int i = e / NB, j = e % NB;
    double v = 0.;
    if (i < nb && j <= i) v = A[(int64_t)i * lda + j];
    else if (i >= nb && i == j) v = 1.;  // identity padding keeps the arithmetic of a ragged last block uniform
    a[i * DP + j] = v;
  }
  __syncthreads();

  for (int kb = 0; kb < NB; kb += SB) {
    if (wave == 0) {  // ---- 16x16 diagonal sub-block: Cholesky + inverse in registers
      double row[SB], x[SB], rdiag[SB];  // rdiag[j] = 1 / L_jj (wave-uniform)
      const int li = lane & 15;
#pragma unroll
      for (int c = 0; c < SB; c++) row[c] = (lane < SB && c <= li) ? a[(kb + li) * DP + kb + c] : 0.;
      int first_bad = SB;                            // first column of this sub-block whose pivot is not positive (or NaN)
#pragma unroll
      for (int j = 0; j < SB; j++) {
        double d = readlane_f64(row[j], j);
        // (no branch per column: the 16 columns stay one basic block, whose rsqrt chains and rank-1 updates the scheduler may then
        //  interleave; the report happens once, behind the loop)
        const bool bad = !(d > 0.);                  // also catches NaN; uniform across the wave
        first_bad = (bad && first_bad == SB) ? j : first_bad;
        d = bad ? 1. : d;
        const double rs = rsqrt(d);                 // one reciprocal square root instead of a sqrt and a divide
        rdiag[j] = rs;
        row[j] = (li == j) ? d * rs : row[j] * rs;  // l_jj = sqrt(d), l_ij = a_ij / sqrt(d)
#pragma unroll
        for (int c = j + 1; c < SB; c++) row[c] -= row[j] * readlane_f64(row[j], c);
      }
      if (first_bad < SB && lane == 0 && kb + first_bad < nb) atomicCAS(info, 0, (int)(k0 + kb + first_bad + 1));
#pragma unroll
      for (int i = 0; i < SB; i++) {  // lane j solves L x = e_j
        double sacc = (i == li) ? 1. : 0.;
#pragma unroll
        for (int c = 0; c < i; c++) sacc -= readlane_f64(row[c], i) * x[c];
        x[i] = sacc * rdiag[i];
      }
      if (lane < SB) {
#pragma unroll
        for (int c = 0; c < SB; c++) {
          if (c <= li) a[(kb + li) * DP + kb + c] = row[c];          // L (rows of this lane)
          dinv[c * (SB + 1) + li] = (c >= li) ? x[c] : 0.;           // X[c][li], column li of the inverse
          if (c > li) a[(kb + li) * DP + kb + c] = x[c];             // X[c][li] kept transposed in the upper part
        }
        xd[kb + li] = x[li];
      }
    }
    __syncthreads();
    // ---- panel: L21 = A21 inv(L11)^T on the matrix cores, one 16-row tile per wave (the B operand is the zero-padded 16 x 16
    // inverse: B[k][j] = X[j][k]); in place -- a wave has read its tile before it writes it
    const int o = kb + SB, nt = (NB - o) / SB;
    if (wave < nt) {
      const int r0 = o + wave * SB;
      d4 acc = (d4){0., 0., 0., 0.};
#pragma unroll
      for (int k4 = 0; k4 < SB / 4; k4++) {
        const int kk = k4 * 4 + kq;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(r0 + mi) * DP + kb + kk], dinv[mi * (SB + 1) + kk], acc, 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 4; reg++) a[(r0 + kq + 4 * reg) * DP + kb + mi] = acc[reg];
    }
    __syncthreads();
    // ---- rank-16 update of the trailing lower triangle: 16 x 16 tiles dealt to the 16 waves, C -= L21_i L21_j^T
    const int ntile = nt * (nt + 1) / 2;
    for (int t = wave; t < ntile; t += PT / 64) {
      int ti = 0;
      while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
      const int tj = t - ti * (ti + 1) / 2;
      const int ri = o + ti * SB, rj = o + tj * SB;
      d4 acc;
#pragma unroll
      for (int reg = 0; reg < 4; reg++) acc[reg] = a[(ri + kq + 4 * reg) * DP + rj + mi];
#pragma unroll
      for (int k4 = 0; k4 < SB / 4; k4++) {
        const int kk = k4 * 4 + kq;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[(ri + mi) * DP + kb + kk], a[(rj + mi) * DP + kb + kk], acc, 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 4; reg++)
        if (ti != tj || mi <= kq + 4 * reg) a[(ri + kq + 4 * reg) * DP + rj + mi] = acc[reg];   // (the strict upper part holds inv(L))
    }
    __syncthreads();
  }
  for (int e = tid; e < nb * nb; e += PT) {
    const int i = e / nb, j = e % nb;
    if (j <= i) A[(int64_t)i * lda + j] = a[i * DP + j];
  }
  // ---- inv(L) by block doubling: diagonal 16x16 inverses are in place (transposed, upper part + xd).  Per level and pair
  // T = L21 X11 (written over L21: L itself is in global memory by now) and X21 = -X22 T, 16 x 16 tiles on the matrix cores,
  // one tile per wave (4 / 8 / 16 tiles per level); the triangular operands are read with their zero halves filled in.
  for (int s2 = SB; s2 < NB; s2 *= 2) {
    const int tps = s2 / SB, ntile = (NB / (2 * s2)) * tps * tps;
    const int pr = wave / (tps * tps), rt = (wave % (tps * tps)) / tps, ct = wave % tps;
    const int C0 = pr * 2 * s2, R0 = C0 + s2;
    d4 acc = (d4){0., 0., 0., 0.};
    if (wave < ntile) {
      const int cj = ct * SB + mi;
      for (int k4 = ct * (SB / 4); k4 < s2 / 4; k4++) {   // X11 is lower triangular: rows k >= column
        const int kk = k4 * 4 + kq;
        const double bv = kk > cj ? a[(C0 + cj) * DP + C0 + kk] : (kk == cj ? xd[C0 + cj] : 0.);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(R0 + rt * SB + mi) * DP + C0 + kk], bv, acc, 0, 0, 0);
      }
    }
    __syncthreads();   // every tile of T is computed before any of them overwrites L21
    if (wave < ntile) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) a[(R0 + rt * SB + kq + 4 * reg) * DP + C0 + ct * SB + mi] = acc[reg];
    }
    __syncthreads();
    if (wave < ntile) {
      const int ri = rt * SB + mi;
      acc = (d4){0., 0., 0., 0.};
      for (int k4 = 0; k4 < (rt + 1) * (SB / 4); k4++) {   // X22 is lower triangular: columns k <= row
        const int kk = k4 * 4 + kq;
        const double av = ri > kk ? a[(R0 + kk) * DP + R0 + ri] : (ri == kk ? xd[R0 + ri] : 0.);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, a[(R0 + kk) * DP + C0 + ct * SB + mi], acc, 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 4; reg++) a[(C0 + ct * SB + mi) * DP + R0 + rt * SB + kq + 4 * reg] = -acc[reg];   // X21, transposed
    }
    __syncthreads();
  }
  for (int e = tid; e < NB * NB; e += PT) {
    const int i = e / NB, j = e % NB;
    double v;
    if (i >= nb || j >= nb) v = (i == j) ? 1. : 0.;
    else if (j < i) v = a[j * DP + i];
    else if (j == i) v = xd[i];
    else v = 0.;
    inv[e] = v;
  }
}

__global__ __launch_bounds__(1024) void potrf_diag_kernel(double* A, int64_t lda, int nb, int64_t k0, double* inv, int* info) {
  __shared__ double a[NB * DP];        // lower: A then L;  strict upper: inv(L) transposed (X[i][j] at a[j][i])
  __shared__ double xd[NB];            // diagonal of inv(L)
  __shared__ double dinv[16 * 17];     // inverse of the current 16x16 diagonal sub-block
  potrf_diag_block(A, lda, nb, k0, inv, info, a, xd, dinv);
}

// The two short products of an inner step, as LDS-resident 128 x 128 x 128 tiles (K = the 128 columns of block k):
//   MODE 0, panel solve   L(i, k) = A(i, k) inv(L_kk)^T                         one workgroup per 128-row tile i, in place
//   MODE 1, rank-128 update   A(i, c) -= L(i, k) L(c, k)^T   for the panel's remaining columns c    one workgroup per tile (i, c)
// The A operand (128 x 128) is loaded into LDS in one go, the B operand is read from L2 as MFMA fragments, 16 waves x 4 tiles
// of 16 x 16, 32 v_mfma_f64_16x16x4_f64 steps over k in ascending order and v = -acc + C at the end: the arithmetic of the
// general GEMM's tile code, bit for bit (scripts/probes/potrf_bits.py) -- but that kernel walks K in 8 serial 16-deep stages
// behind an offset-table prologue: 38 us for the panel solve and 60 - 100 us for the update whatever their size, 2 x 191
// launches on a layer's critical path.
// MODE 1, look-ahead: the workgroup of the FIRST tile -- (k + 1, k + 1), the next diagonal block, which has all its updates once
// this one is applied -- goes on to factorise and invert it (potrf_diag_block, the same arithmetic as the stand-alone kernel)
// while the other tiles of the step are still being updated: next_nb > 0 says so, next_inv / info are the diagonal kernel's
// outputs.  The 71 us of the 1-workgroup diagonal kernel leave the critical path wherever the update has more than one round of
// tiles to work through (the caller then skips that launch).
template <int MODE>
__global__ __launch_bounds__(1024) void potrf_tile_kernel(double* A, int64_t lda, int64_t r0, int64_t nrows, int64_t k0, int64_t c0,
                                                          int64_t ncols, const double* inv, int next_nb, double* next_inv, int* info) {
  constexpr int PT = 1024, SB = 16;
  __shared__ double a[NB * DP];
  __shared__ double la_xd[MODE == 1 ? NB : 1];
  __shared__ double la_dinv[MODE == 1 ? 16 * 17 : 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mi = lane & 15, kq = lane >> 4;
  const int64_t ctiles = MODE == 1 ? ceil_div_dev(ncols, NB) : 1;
  const int64_t ti = blockIdx.x / ctiles, tc = blockIdx.x % ctiles;
  const int64_t row0 = r0 + ti * NB, col0 = c0 + tc * NB;
  if (MODE == 1 && row0 + NB - 1 < col0) return;             // entirely above the diagonal: nothing there is ever read
  const int nr = (int)(r0 + nrows - row0 < NB ? r0 + nrows - row0 : NB);
  const int nc = MODE == 1 ? (int)(c0 + ncols - col0 < NB ? c0 + ncols - col0 : NB) : NB;
  double* Ai = A + row0 * lda + k0;                           // A operand: rows of tile i, the 128 columns of block k
  const double* Bp = MODE == 1 ? A + col0 * lda + k0 : inv;   // B[k][j] = Bp[j * ldb + k]: rows of L(c, k), or of inv(L_kk)
  const int64_t ldb = MODE == 1 ? lda : NB;
  for (int e = tid; e < NB * NB; e += PT) {
    const int i = e / NB, j = e % NB;
    a[i * DP + j] = i < nr ? Ai[(int64_t)i * lda + j] : 0.;
  }
  const int ct = wave & 7, rt0 = (wave >> 3) * 4;             // wave -> column tile ct, four row tiles
  const int bcol = ct * SB + mi;
  __syncthreads();
  d4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; t++) acc[t] = (d4){0., 0., 0., 0.};
  // the wave's B fragments, 8 k-steps per batch of loads (one round trip to L2 per batch instead of one per step; 16 waves of
  // 1024 threads leave 128 registers per lane, so not all 32 at once)
  // (and the next batch's loads are issued before this batch is multiplied: a tile is then ONE round trip to L2 for its B operand
  //  plus three that run under the matrix cores, instead of four in series -- the arithmetic and its order do not change)
  constexpr int KB = 8, NBATCH = NB / 4 / KB;
  double bv[2][KB];
  auto load_b = [&](int kb, double (&v)[KB]) {
#pragma unroll
    for (int u = 0; u < KB; u++) v[u] = bcol < nc ? Bp[(int64_t)bcol * ldb + (kb + u) * 4 + kq] : 0.;
  };
  // (MODE 1 only: the panel solve, a launch of at most 112 workgroups, measured 2 - 3 us SLOWER with it)
  load_b(0, bv[0]);
#pragma unroll
  for (int q = 0; q < NBATCH; q++) {
    if (MODE == 1 && q + 1 < NBATCH) load_b((q + 1) * KB, bv[(q + 1) & 1]);
    if (MODE == 0 && q > 0) load_b(q * KB, bv[q & 1]);
    if (MODE == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < KB; u++) {
      const int kk = (q * KB + u) * 4 + kq;
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[((rt0 + t) * SB + mi) * DP + kk], bv[q & 1][u], acc[t], 0, 0, 0);
    }
  }
  if (MODE == 0) {
    __syncthreads();   // (in place: every wave has read its rows before any is overwritten -- other waves' rows are in LDS only)
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int row = (rt0 + t) * SB + kq + 4 * reg;
        if (row < nr) Ai[(int64_t)row * lda + bcol] = acc[t][reg];
      }
  } else {
    // (the tile of C is read here, in one batch, rather than ahead of the products: with two batches of B fragments in flight its 32
    //  registers no longer fit beside them, and one exposed round trip replaces three)
    double* C = A + row0 * lda + col0;
    double cold[4][4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int row = (rt0 + t) * SB + kq + 4 * reg;
        cold[t][reg] = (row < nr && bcol < nc) ? C[(int64_t)row * lda + bcol] : 0.;
      }
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int row = (rt0 + t) * SB + kq + 4 * reg;
        if (row < nr && bcol < nc) C[(int64_t)row * lda + bcol] = -1.0 * acc[t][reg] + 1.0 * cold[t][reg];
      }
    if (next_nb > 0 && blockIdx.x == 0) {   // tile (k + 1, k + 1) is final: factorise it here (uniform per workgroup)
      __syncthreads();                      // every thread's part of the tile is in memory (and `a` is free again)
      potrf_diag_block(A + row0 * lda + col0, lda, next_nb, row0, next_inv, info, a, la_xd, la_dinv);
    }
  }
}

// dst lower triangle (incl. diagonal) = src + ridge*I ; optional row/col gather through idx.
__global__ __launch_bounds__(256) void copy_lower_kernel(const double* src, int64_t lds_, const int64_t* idx, double* dst,
                                                         int64_t ldd, int64_t n, double ridge) {
  const int64_t i = blockIdx.y;
  const int64_t si = idx ? idx[i] : i;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j <= i; j += (int64_t)gridDim.x * 256) {
    int64_t sj = idx ? idx[j] : j;
    double v = src[si * lds_ + sj];
    if (j == i) v += ridge;
    dst[i * ldd + j] = v;
  }
}

__global__ __launch_bounds__(256) void place_inv_diag_kernel(const double* inv, double* X, int64_t ldx, int64_t n) {
  const int64_t b = blockIdx.x;
  const double* src = inv + b * NB * NB;
  for (int e = threadIdx.x; e < NB * NB; e += 256) {
    int64_t i = b * NB + e / NB, j = b * NB + e % NB;
    if (i < n && j < n) X[i * ldx + j] = src[e];
  }
}

// out[j] = sum_{i >= j} X[i][j]^2 ; one workgroup per 64 columns, 4 row-strided waves.
// SENS: the same pass also takes t_j = sum_{i >= j} |X[i][j]| u[i] and writes sens[j] = t_j^2 (see lower_abs_rowsum_kernel); the sums
// behind out[j] are formed in the same order either way (the scores do not change by a bit when the sensitivity is asked for).
template <bool SENS>
__global__ __launch_bounds__(256) void lower_colnorm2_kernel(const double* X, int64_t ldx, int64_t n, double* out, const double* u,
                                                             double* sens) {
  __shared__ double red[4][64], red_t[SENS ? 4 : 1][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 64 + lane;
  const int64_t i0 = (int64_t)blockIdx.x * 64;
  double s = 0., t = 0.;
  if (j < n)
    for (int64_t i = i0 + wave; i < n; i += 4)
      if (i >= j) {
        double v = X[i * ldx + j];
        s += v * v;
        if (SENS) t += fabs(v) * u[i];
      }
  red[wave][lane] = s;
  if (SENS) red_t[wave][lane] = t;
  __syncthreads();
  if (wave == 0 && j < n) {
    out[j] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    if (SENS) {
      const double tt = red_t[0][lane] + red_t[1][lane] + red_t[2][lane] + red_t[3][lane];
      sens[j] = tt * tt;
    }
  }
}

// The sensitivity of the ridge scores to an ENTRY-WISE RELATIVE perturbation of C (what the int8 covariance route guarantees:
// |E_ab| <= eps sqrt(c_aa c_bb)).  With A = C + ridge I = L L^T, X = inv(L), z_j = inv(A) e_j = X^T X e_j and s_j = (inv(A))_jj,
// to first order  delta s_j = -z_j^T E z_j,  so
//     |delta s_j| <= eps (sum_a sqrt(c_aa) |z_aj|)^2 <= eps (sum_b |X_bj| u_b)^2 =: eps sens_j,    u_b = sum_a |X_ba| sqrt(c_aa)
// (the triangle inequality inside z_aj = sum_b X_ba X_bj).  Two passes over the triangle of X that mdg_ridge_scores has in its
// workspace anyway (n^2 / 2 * 8 B each: 0.8 GB at n = 14336), no extra GEMM: d = sqrt(diag C), u = |X| d (this kernel, one wave per
// row), t = |X|^T u (fused into the column-norm pass above).  mdg_select_margin turns sens into the certificate of the selection.
__global__ __launch_bounds__(256) void sqrt_diag_kernel(const double* C, int64_t ldc, int64_t n, double* d) {
  const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (a < n) {
    const double c = C[a * ldc + a];
    d[a] = c > 0. ? sqrt(c) : 0.;
  }
}
__global__ __launch_bounds__(256) void lower_abs_rowsum_kernel(const double* X, int64_t ldx, int64_t n, const double* d, double* u) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= n) return;
  double acc = 0.;
  for (int64_t a = lane; a <= b; a += 64) acc += fabs(X[b * ldx + a]) * d[a];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) u[b] = acc;
}


// Two-level right-looking Cholesky.  Inner level: 128-wide panels (diagonal block factorised + inverted in LDS,
// panel solve = GEMM with the inverse), whose rank-128 updates touch only the rest of the current NBO-wide OUTER
// panel.  Outer level: one rank-NBO lower-only update of everything behind the outer panel.  A rank-128 update of a
// 128x128 fp64 tile moves 512 KB for 4.2 MFLOP (8 flop/B: HBM-bound at ~30 TF); rank-512 already quadruples the intensity
// and puts the bulk of the n^3/3 flops back under the MFMA roof.
#ifndef MDG_CHOL_TILE_KERNELS
#define MDG_CHOL_TILE_KERNELS 1   // panel solve and rank-128 update as LDS-resident 128^3 tiles (potrf_tile_kernel) instead of the general GEMM
#endif
#ifndef MDG_CHOL_LOOKAHEAD
#define MDG_CHOL_LOOKAHEAD 1      // the update launch of step k factorises diagonal block k + 1 in its first workgroup (see potrf_tile_kernel)
#endif
#ifndef MDG_CHOL_NBO
#define MDG_CHOL_NBO 2048   // (round 2, ridge scores + Nystrom solve of a Llama-3-8B layer: 256: 66.4 + 76.3 ms; 512: 63.0 + 72.2; 1024: 62.5 + 69.7.
                            //  Round 4, with the next diagonal block factorised inside the update launch the inner steps got cheaper
                            //  and the optimum moved: potrf_lower at n = 14336 + 10035: 512: 31.7 + 17.4 ms; 1024: 29.6 + 16.3;
                            //  1536: 29.3 + 15.7; 2048: 28.8 + 15.2; 2560: 29.2 + 15.3; 3072: 29.6 + 15.6; 4096: 30.4 + 16.1 --
                            //  profiles/r04_cholesky_outer_block.log)
#endif
constexpr int NBO = MDG_CHOL_NBO;

int potrf_lower(double* A, int64_t n, int64_t lda, double* inv_diag, hipStream_t st) {
  const int64_t nblk = ceil_div(n, NB);
  int* dflag = (int*)(inv_diag + nblk * NB * NB);
  MDG_HIP(hipMemsetAsync(dflag, 0, sizeof(int), st));
  for (int64_t J0 = 0; J0 < n; J0 += NBO) {
    const int64_t Jend = J0 + NBO < n ? J0 + NBO : n;
    bool diag_done = false;     // block k0 was factorised by the previous step's update launch (look-ahead)
    for (int64_t k0 = J0; k0 < Jend; k0 += NB) {
      const int nb = (int)(n - k0 < NB ? n - k0 : NB);
      double* inv = inv_diag + (k0 / NB) * NB * NB;
      if (!diag_done) hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(1024), 0, st, A + k0 * lda + k0, lda, nb, k0, inv, dflag);
      diag_done = false;
      MDG_LAUNCH_CHECK();
      const int64_t rest = n - k0 - nb;
      if (rest <= 0) break;
      double* A21 = A + (k0 + nb) * lda + k0;
      const int64_t w = Jend - (k0 + nb);
#if MDG_CHOL_TILE_KERNELS
      // L21 = A21 * inv(L11)^T, then the rank-128 update of the remaining columns of this outer panel only (nb == 128 here: a
      // ragged block is the last one and has nothing below it)
      hipLaunchKernelGGL(potrf_tile_kernel<0>, dim3((unsigned)ceil_div(rest, NB)), dim3(1024), 0, st, A, lda, k0 + nb, rest, k0,
                         (int64_t)0, (int64_t)0, inv, 0, (double*)nullptr, (int*)nullptr);
      if (w > 0) {   // (then k0 + nb < Jend: the next block belongs to this outer panel and is final after this update)
        const int next_nb = (int)(n - (k0 + nb) < NB ? n - (k0 + nb) : NB);
        hipLaunchKernelGGL(potrf_tile_kernel<1>, dim3((unsigned)(ceil_div(rest, NB) * ceil_div(w, NB))), dim3(1024), 0, st, A, lda,
                           k0 + nb, rest, k0, k0 + nb, w, inv, MDG_CHOL_LOOKAHEAD ? next_nb : 0, inv + NB * NB, dflag);
        diag_done = MDG_CHOL_LOOKAHEAD != 0;
      }
      MDG_LAUNCH_CHECK();
#else
      // L21 = A21 * inv(L11)^T  (in place: one tile column, each workgroup reads only its own rows)
      MDG_TRY(gemm_f64(rest, nb, nb, 1.0, A21, MDG_F64, lda, 1, nullptr, inv, MDG_F64, 1, NB, 0.0, A21, MDG_F64, lda, 1,
                       0, 0, 0, 0, st));
      // rank-128 update of the remaining columns of this outer panel only
      if (w > 0)
        MDG_TRY(gemm_f64(rest, w, nb, -1.0, A21, MDG_F64, lda, 1, nullptr, A21, MDG_F64, 1, lda, 1.0,
                         A + (k0 + nb) * lda + (k0 + nb), MDG_F64, lda, 1, 0, 0, 0, 0, st));
#endif
    }
    const int64_t rest = n - Jend;
    if (rest > 0) {  // rank-(Jend-J0) update of the trailing matrix, lower tiles only
      const double* Lp = A + Jend * lda + J0;
      MDG_TRY(gemm_f64(rest, rest, Jend - J0, -1.0, Lp, MDG_F64, lda, 1, nullptr, Lp, MDG_F64, 1, lda, 1.0,
                       A + Jend * lda + Jend, MDG_F64, lda, 1, 0, 0, 0, MDG_GEMM_LOWER_ONLY, st));
    }
  }
  return finish_flag(dflag, st, STATUS_NOT_PD, "Cholesky");   // (one host round trip, or none in deferred-status mode)
}

// Triangular inverse by recursive doubling on the block size, starting from the inverted 128-wide diagonal blocks the
// factorisation left behind: after the level of pair size s, X holds inv of every aligned 2s-wide diagonal block of L.
// Levels run while s < s_limit: s_limit = n gives X = inv(L); s_limit = NBO the inverses of the NBO-wide diagonal blocks only
// (potrs_lower).  X is addressed as X[i * ldx + j] for (i, j) inside one diagonal block; with ldx < n the blocks are simply
// skewed in memory (row i starts at i * ldx, its block's columns at + j): n * ldx + n elements hold them all.
int tri_inverse_doubling(const double* L, int64_t n, int64_t ldl, const double* inv_diag, double* X, int64_t ldx, double* T,
                         int64_t s_limit, hipStream_t st) {
  const int64_t nblk = ceil_div(n, NB);
  hipLaunchKernelGGL(place_inv_diag_kernel, dim3((unsigned)nblk), dim3(256), 0, st, inv_diag, X, ldx, n);
  MDG_LAUNCH_CHECK();
  for (int64_t s = NB; s < n && s < s_limit; s *= 2) {
    // pair p: Cb = [2ps, 2ps+s), R = [2ps+s, min(2ps+2s, n))
    const int64_t full = n / (2 * s);                 // pairs with a complete R
    const int64_t r0_last = full * 2 * s + s;         // a trailing ragged pair exists if r0_last < n
    const int64_t pair_stride_L = 2 * s * ldl + 2 * s, pair_stride_X = 2 * s * ldx + 2 * s;
    for (int pass = 0; pass < 2; pass++) {
      int64_t p0, cnt, m;
      if (pass == 0) { p0 = 0; cnt = full; m = s; }
      else { p0 = full; cnt = r0_last < n ? 1 : 0; m = n - r0_last; }
      if (cnt <= 0) continue;
      const int64_t c0 = p0 * 2 * s, r0 = c0 + s;
      // T_p = L[R, Cb] * X[Cb, Cb]      (X[Cb,Cb] lower triangular)
      MDG_TRY(gemm_f64(m, s, s, 1.0, L + r0 * ldl + c0, MDG_F64, ldl, 1, nullptr, X + c0 * ldx + c0, MDG_F64, ldx, 1,
                       0.0, T, MDG_F64, s, cnt, pair_stride_L, pair_stride_X, s * s, MDG_GEMM_B_LOWER_TRI, st));
      // X[R, Cb] = -X[R, R] * T_p       (X[R,R] lower triangular)
      MDG_TRY(gemm_f64(m, s, m, -1.0, X + r0 * ldx + r0, MDG_F64, ldx, 1, nullptr, T, MDG_F64, s, 1, 0.0,
                       X + r0 * ldx + c0, MDG_F64, ldx, cnt, pair_stride_X, s * s, pair_stride_X,
                       MDG_GEMM_A_LOWER_TRI, st));
    }
  }
  return MDG_OK;
}

// X = inv(L), then column norms.  sens != nullptr: also the first-order sensitivities of out[j] to an entry-wise relative
// perturbation of C (C, ldc: the matrix whose diagonal scales the perturbation; T, free again after the doubling, holds d and u).
int chol_inverse_diag(const double* L, int64_t n, int64_t ldl, const double* inv_diag, double* out, double* X,
                      double* T, hipStream_t st, const double* C = nullptr, int64_t ldc = 0, double* sens = nullptr) {
  const int64_t ldx = n;
  MDG_TRY(tri_inverse_doubling(L, n, ldl, inv_diag, X, ldx, T, n, st));
  if (sens) {
    double* d = T;          // (T has n^2 / 4 + 128^2 elements >= 2 n)
    double* u = T + n;
    hipLaunchKernelGGL(sqrt_diag_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, C, ldc, n, d);
    hipLaunchKernelGGL(lower_abs_rowsum_kernel, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, st, X, ldx, n, d, u);
    hipLaunchKernelGGL(lower_colnorm2_kernel<true>, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, st, X, ldx, n, out, u, sens);
  } else {
    hipLaunchKernelGGL(lower_colnorm2_kernel<false>, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, st, X, ldx, n, out,
                       (const double*)nullptr, (double*)nullptr);
  }
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}

#ifndef MDG_CHOL_NBS
#define MDG_CHOL_NBS 2048   // (n = 10035, nrhs = 4096: 512 -> 21.9 ms, 1024 -> 18.7, 2048 -> 17.9)
#endif
constexpr int NBS = MDG_CHOL_NBS;

// (L L^T) X = B by blocks of NBS rows.  The substitution inside a block is ONE triangular-aware GEMM with the block's explicit
// inverse (tri_inverse_doubling up to NBS: ~n NBS^2 / 3 flops, log2(NBS / 128) batched levels), and one rank-NBS GEMM carries the
// block's solution to everything behind (forward) / before (backward) it.  Round 1 walked the 128-wide diagonal blocks
// inside every outer block -- 2 x n / 128 steps of a 128-row multiply (32 workgroups) and a thin update, ~85 us a step
// whatever the flops: 24.5 ms for n = 10035, nrhs = 4096, of which the carries are 12.  Out of place between X and a
// second right-hand-side buffer: forward X -> W, backward W -> X.
size_t potrs_ws_elems(int64_t n, int64_t nrhs) {
  return (size_t)n * NBS + (size_t)n + ((size_t)n * NBS / 4 + NB * NB) + (size_t)n * nrhs;
}

int potrs_lower(const double* L, int64_t n, int64_t ldl, const double* inv_diag, double* X, int64_t nrhs, int64_t ldx,
                double* ws, hipStream_t st) {
  double* Linv = ws;                                   // skewed [n][NBS]: inv of every NBS-wide diagonal block
  double* T = Linv + (size_t)n * NBS + n;
  double* W = T + ((size_t)n * NBS / 4 + NB * NB);     // [n][nrhs]
  MDG_TRY(tri_inverse_doubling(L, n, ldl, inv_diag, Linv, NBS, T, NBS, st));
  for (int64_t J0 = 0; J0 < n; J0 += NBS) {  // L Y = B:  W_J = inv(L_JJ) X_J;  X[Jend:] -= L[Jend:, J] W_J
    const int64_t Jend = J0 + NBS < n ? J0 + NBS : n, nbj = Jend - J0;
    MDG_TRY(gemm_f64(nbj, nrhs, nbj, 1.0, Linv + J0 * NBS + J0, MDG_F64, NBS, 1, nullptr, X + J0 * ldx, MDG_F64, ldx, 1, 0.0,
                     W + J0 * nrhs, MDG_F64, nrhs, 1, 0, 0, 0, MDG_GEMM_A_LOWER_TRI, st));
    if (n - Jend > 0)
      MDG_TRY(gemm_f64(n - Jend, nrhs, nbj, -1.0, L + Jend * ldl + J0, MDG_F64, ldl, 1, nullptr, W + J0 * nrhs, MDG_F64, nrhs,
                       1, 1.0, X + Jend * ldx, MDG_F64, ldx, 1, 0, 0, 0, 0, st));
  }
  const int64_t last_J0 = ((n - 1) / NBS) * NBS;
  for (int64_t J0 = last_J0; J0 >= 0; J0 -= NBS) {  // L^T X = Y:  X_J = inv(L_JJ)^T W_J;  W[0:J0] -= L[J, 0:J0]^T X_J
    const int64_t Jend = J0 + NBS < n ? J0 + NBS : n, nbj = Jend - J0;
    MDG_TRY(gemm_f64(nbj, nrhs, nbj, 1.0, Linv + J0 * NBS + J0, MDG_F64, 1, NBS, nullptr, W + J0 * nrhs, MDG_F64, nrhs, 1, 0.0,
                     X + J0 * ldx, MDG_F64, ldx, 1, 0, 0, 0, MDG_GEMM_A_UPPER_TRI, st));
    if (J0 > 0)
      MDG_TRY(gemm_f64(J0, nrhs, nbj, -1.0, L + J0 * ldl, MDG_F64, 1, ldl, nullptr, X + J0 * ldx, MDG_F64, ldx, 1, 1.0, W,
                       MDG_F64, nrhs, 1, 0, 0, 0, 0, st));
  }
  return MDG_OK;
}

int copy_lower(const double* src, int64_t ld_src, const int64_t* idx, double* dst, int64_t ldd, int64_t n,
               double ridge, hipStream_t st) {
  MDG_CHECK_ARG(n < 65536 * 4, "copy_lower: n too large");
  unsigned gx = (unsigned)(ceil_div(n, 256) < 64 ? ceil_div(n, 256) : 64);
  hipLaunchKernelGGL(copy_lower_kernel, dim3(gx, (unsigned)n), dim3(256), 0, st, src, ld_src, idx, dst, ldd, n, ridge);
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}

}  // namespace mdg

using namespace mdg;

extern "C" size_t mdg_potrf_inv_diag_elems(int64_t n) { return (size_t)ceil_div(n, NB) * NB * NB + 16; }

extern "C" int mdg_potrf_lower(double* A, int64_t n, int64_t lda, double* inv_diag, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(A && inv_diag && n > 0 && lda >= n, "mdg_potrf_lower: bad arguments");
  return potrf_lower(A, n, lda, inv_diag, (hipStream_t)stream);
}

extern "C" size_t mdg_potrs_lower_ws_bytes(int64_t n, int64_t nrhs) {
  if (n <= 0 || nrhs <= 0) return 0;
  return potrs_ws_elems(n, nrhs) * sizeof(double);
}

extern "C" int mdg_potrs_lower(const double* L, int64_t n, int64_t ldl, const double* inv_diag, double* X,
                               int64_t nrhs, int64_t ldx, void* ws, size_t ws_bytes, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(L && inv_diag && X && n > 0 && nrhs > 0 && ldl >= n && ldx >= nrhs, "mdg_potrs_lower: bad arguments");
  MDG_CHECK_ARG(ws && ws_bytes >= mdg_potrs_lower_ws_bytes(n, nrhs), "mdg_potrs_lower: workspace %zu < required %zu", ws_bytes,
                mdg_potrs_lower_ws_bytes(n, nrhs));
  return potrs_lower(L, n, ldl, inv_diag, X, nrhs, ldx, (double*)ws, (hipStream_t)stream);
}

extern "C" size_t mdg_chol_inverse_diag_ws_bytes(int64_t n) {
  return ((size_t)n * n + (size_t)n * n / 4 + NB * NB) * sizeof(double);
}

extern "C" int mdg_chol_inverse_diag(const double* L, int64_t n, int64_t ldl, const double* inv_diag, double* out,
                                     void* ws, size_t ws_bytes, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(L && inv_diag && out && n > 0 && ldl >= n, "mdg_chol_inverse_diag: bad arguments");
  MDG_CHECK_ARG(ws && ws_bytes >= mdg_chol_inverse_diag_ws_bytes(n), "mdg_chol_inverse_diag: workspace too small");
  double* X = (double*)ws;
  double* T = X + (size_t)n * n;
  return chol_inverse_diag(L, n, ldl, inv_diag, out, X, T, (hipStream_t)stream);
}

extern "C" size_t mdg_ridge_scores_ws_bytes(int64_t n) {
  return (size_t)n * n * sizeof(double) + mdg_potrf_inv_diag_elems(n) * sizeof(double) +
         mdg_chol_inverse_diag_ws_bytes(n);
}

extern "C" int mdg_ridge_scores(const double* C, int64_t n, int64_t ldc, double ridge, double* scores, double* sens, void* ws,
                                size_t ws_bytes, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(C && scores && n > 0 && ldc >= n, "mdg_ridge_scores: bad arguments");
  MDG_CHECK_ARG(ws && ws_bytes >= mdg_ridge_scores_ws_bytes(n), "mdg_ridge_scores: workspace %zu < required %zu",
                ws_bytes, mdg_ridge_scores_ws_bytes(n));
  hipStream_t st = (hipStream_t)stream;
  double* Lb = (double*)ws;
  double* inv = Lb + (size_t)n * n;
  double* rest = inv + mdg_potrf_inv_diag_elems(n);
  MDG_TRY(copy_lower(C, ldc, nullptr, Lb, n, n, ridge, st));
  MDG_TRY(potrf_lower(Lb, n, n, inv, st));
  return chol_inverse_diag(Lb, n, n, inv, scores, rest, rest + (size_t)n * n, st, C, ldc, sens);
}
