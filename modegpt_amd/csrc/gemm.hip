// General strided fp64 GEMM on v_mfma_f64_16x16x4_f64, the building block of the decomposition kernels
// (Cholesky trailing updates, TRSM by inverted diagonal blocks, triangular inverse, Nystrom cross term
// C[idx,:] @ W_d^T with fused row gather, VO Gram / factor products with fused bf16 load and store).
// Same 128x128 workgroup tile / 4-wave / LDS-fp64-panel core as cov.hip.
#include "common.hpp"

namespace mdg {

struct GemmArgs {
  int64_t M, N, K;
  double alpha, beta;
  const void* A;
  int64_t sa_i, sa_k;
  const int64_t* a_rows;
  const void* B;
  int64_t sb_k, sb_j;
  void* C;
  int64_t ldc;
  int64_t a_bs, b_bs, c_bs;
  int flags, tiles_m, tiles_n, c_dtype;
};

// Stage one BKx128 panel: element (x, k) of the operand lives at base + off(x) + k*sk.
// kcontig: consecutive threads walk k (operand contiguous along k), else they walk x.
template <int DT, bool kcontig>
__device__ __forceinline__ void gemm_load_panel(const void* base, const int64_t* xoff, int64_t sk, int64_t k0,
                                                int64_t k_end, int tid, double* regs) {
  // re-read the offset table from LDS every stage: hoisted into registers it costs 32 VGPRs and spills
  asm volatile("" ::: "memory");
#pragma unroll
  for (int p = 0; p < 8; p++) {
    int e = tid + 256 * p;
    int k = kcontig ? (e % BK) : (e / TILE);
    int64_t off = xoff[kcontig ? (e / BK) : (e % TILE)];
    double v = 0.;
    if (off >= 0 && k0 + k < k_end) v = load_f64<DT>(base, off + (k0 + k) * sk);
    regs[p] = v;
  }
}

template <bool kcontig>
__device__ __forceinline__ void gemm_store_panel(double* panel, int tid, const double* regs) {
#pragma unroll
  for (int p = 0; p < 8; p++) {
    int e = tid + 256 * p;
    int k = kcontig ? (e % BK) : (e / TILE);
    int x = kcontig ? (e / BK) : (e % TILE);
    panel[k * PITCH + x] = regs[p];
  }
}

template <int ADT, int BDT, bool AKC, bool BKC>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
  __shared__ double lds[4 * PANEL + 2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int bi, bj;
  if (g.flags & MDG_GEMM_LOWER_ONLY) {
    tri_decode(blockIdx.x, bi, bj);
  } else {
    bi = blockIdx.x / g.tiles_n;
    bj = blockIdx.x % g.tiles_n;
  }
  const int64_t batch = blockIdx.y;
  const int64_t i0 = (int64_t)bi * TILE, j0 = (int64_t)bj * TILE;
  const char* Ab = (const char*)g.A + batch * g.a_bs * (int64_t)sizeof(typename ElemOf<ADT>::type);
  const char* Bb = (const char*)g.B + batch * g.b_bs * (int64_t)sizeof(typename ElemOf<BDT>::type);

  int64_t k_begin = 0, k_end = g.K;
  if (g.flags & MDG_GEMM_A_LOWER_TRI) k_end = min(k_end, i0 + TILE);
  if (g.flags & MDG_GEMM_B_LOWER_TRI) k_begin = max(k_begin, j0);
  if (g.flags & MDG_GEMM_A_UPPER_TRI) k_begin = max(k_begin, i0);

  // element offsets of the 128 rows of op(A) / columns of op(B) this tile touches (-1 = out of range -> zero fill)
  int64_t* aoff = (int64_t*)(lds + 4 * PANEL);
  int64_t* boff = aoff + TILE;
  if (tid < TILE) {
    int64_t gi = i0 + tid;
    aoff[tid] = gi < g.M ? (g.a_rows ? g.a_rows[gi] : gi) * g.sa_i : -1;
  } else {
    int64_t gj = j0 + (tid - TILE);
    boff[tid - TILE] = gj < g.N ? gj * g.sb_j : -1;
  }
  __syncthreads();

  Acc acc;
  acc_zero(acc);
  double rg[8];  // one panel's prefetch at a time: A rides under the first half of a stage, B under the second
  const int64_t n_stage = k_end > k_begin ? (k_end - k_begin + BK - 1) / BK : 0;
  if (n_stage > 0) {
    gemm_load_panel<ADT, AKC>(Ab, aoff, g.sa_k, k_begin, k_end, tid, rg);
    gemm_store_panel<AKC>(lds, tid, rg);
    gemm_load_panel<BDT, BKC>(Bb, boff, g.sb_k, k_begin, k_end, tid, rg);
    gemm_store_panel<BKC>(lds + 2 * PANEL, tid, rg);
  }
  __syncthreads();
  for (int64_t s = 0; s < n_stage; s++) {
    const int cur = (int)(s & 1);
    const bool more = s + 1 < n_stage;
    const int64_t k0 = k_begin + (s + 1) * BK;
    const double* As = lds + cur * PANEL;
    const double* Bs = lds + (2 + cur) * PANEL;
    if (more) gemm_load_panel<ADT, AKC>(Ab, aoff, g.sa_k, k0, k_end, tid, rg);
    mma_steps<0, BK / 8>(As, Bs, wr, wc, lane, acc);
    if (more) {
      gemm_store_panel<AKC>(lds + (cur ^ 1) * PANEL, tid, rg);
      gemm_load_panel<BDT, BKC>(Bb, boff, g.sb_k, k0, k_end, tid, rg);
    }
    mma_steps<BK / 8, BK / 4>(As, Bs, wr, wc, lane, acc);
    if (more) gemm_store_panel<BKC>(lds + (2 + (cur ^ 1)) * PANEL, tid, rg);
    __syncthreads();
  }

  const bool c_bf16 = (g.c_dtype == MDG_BF16);
  char* Cb = (char*)g.C + batch * g.c_bs * (c_bf16 ? 2 : 8);
#pragma unroll
  for (int sa = 0; sa < 4; sa++)
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      int64_t gr = i0 + acc_row(wr, lane, sa, reg);
      if (gr >= g.M) continue;
      int64_t gc0 = j0 + acc_col(wc, lane, 0);
      int64_t e0 = gr * g.ldc + gc0;
#pragma unroll
      for (int sb = 0; sb < 4; sb++) {
        if (gc0 + sb >= g.N) continue;
        double v = g.alpha * acc.v[sa][sb][reg];
        if (!c_bf16) {
          double* dst = (double*)Cb + e0 + sb;
          if (g.beta != 0.) v += g.beta * *dst;
          *dst = v;
        } else {
          ((bf16_t*)Cb)[e0 + sb] = f64_to_bf16(v);
        }
      }
    }
}

template <int ADT, int BDT>
static void launch_gemm(const GemmArgs& g, dim3 grid, hipStream_t st) {
  const bool a_kc = (g.sa_k == 1 && g.sa_i != 1), b_kc = (g.sb_k == 1 && g.sb_j != 1);
  if (a_kc && b_kc) hipLaunchKernelGGL((gemm_f64_kernel<ADT, BDT, true, true>), grid, dim3(256), 0, st, g);
  else if (a_kc) hipLaunchKernelGGL((gemm_f64_kernel<ADT, BDT, true, false>), grid, dim3(256), 0, st, g);
  else if (b_kc) hipLaunchKernelGGL((gemm_f64_kernel<ADT, BDT, false, true>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((gemm_f64_kernel<ADT, BDT, false, false>), grid, dim3(256), 0, st, g);
}

int gemm_f64(int64_t M, int64_t N, int64_t K, double alpha, const void* A, int a_dtype, int64_t sa_i, int64_t sa_k,
             const int64_t* a_rows, const void* B, int b_dtype, int64_t sb_k, int64_t sb_j, double beta, void* C,
             int c_dtype, int64_t ldc, int64_t batch, int64_t a_bs, int64_t b_bs, int64_t c_bs, int flags,
             hipStream_t st) {
  MDG_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && batch >= 0, "mdg_gemm_f64: negative size");
  if (M == 0 || N == 0 || batch == 0) return MDG_OK;
  MDG_CHECK_ARG(A && B && C, "mdg_gemm_f64: null pointer");
  MDG_CHECK_ARG((a_dtype == MDG_F64 || a_dtype == MDG_BF16) && (b_dtype == MDG_F64 || b_dtype == MDG_BF16) &&
                    (c_dtype == MDG_F64 || c_dtype == MDG_BF16),
                "mdg_gemm_f64: operand dtypes must be f64 or bf16");
  MDG_CHECK_ARG(c_dtype == MDG_F64 || beta == 0., "mdg_gemm_f64: bf16 output needs beta == 0");
  MDG_CHECK_ARG(ldc >= N, "mdg_gemm_f64: ldc < N");
  MDG_CHECK_ARG(!(flags & MDG_GEMM_LOWER_ONLY) || M == N, "mdg_gemm_f64: LOWER_ONLY needs M == N");
  GemmArgs g;
  g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
  g.A = A; g.sa_i = sa_i; g.sa_k = sa_k; g.a_rows = a_rows;
  g.B = B; g.sb_k = sb_k; g.sb_j = sb_j;
  g.C = C; g.ldc = ldc; g.a_bs = a_bs; g.b_bs = b_bs; g.c_bs = c_bs;
  g.flags = flags;
  g.tiles_m = (int)ceil_div(M, TILE);
  g.tiles_n = (int)ceil_div(N, TILE);
  int64_t nt = (flags & MDG_GEMM_LOWER_ONLY) ? (int64_t)g.tiles_m * (g.tiles_m + 1) / 2 : (int64_t)g.tiles_m * g.tiles_n;
  MDG_CHECK_ARG(nt < (1ll << 31) && batch < 65536, "mdg_gemm_f64: grid too large");
  dim3 grid((unsigned)nt, (unsigned)batch);
  g.c_dtype = c_dtype;
  if (a_dtype == MDG_F64 && b_dtype == MDG_F64) launch_gemm<MDG_F64, MDG_F64>(g, grid, st);
  else if (a_dtype == MDG_F64) launch_gemm<MDG_F64, MDG_BF16>(g, grid, st);
  else if (b_dtype == MDG_F64) launch_gemm<MDG_BF16, MDG_F64>(g, grid, st);
  else launch_gemm<MDG_BF16, MDG_BF16>(g, grid, st);
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}

}  // namespace mdg

extern "C" int mdg_gemm_f64(int64_t M, int64_t N, int64_t K, double alpha, const void* A, int a_dtype, int64_t sa_i,
                            int64_t sa_k, const int64_t* a_rows, const void* B, int b_dtype, int64_t sb_k,
                            int64_t sb_j, double beta, void* C, int c_dtype, int64_t ldc, int64_t batch,
                            int64_t a_bs, int64_t b_bs, int64_t c_bs, int flags, void* stream) {
  MDG_CLEAR();
  return mdg::gemm_f64(M, N, K, alpha, A, a_dtype, sa_i, sa_k, a_rows, B, b_dtype, sb_k, sb_j, beta, C, c_dtype, ldc,
                       batch, a_bs, b_bs, c_bs, flags, (hipStream_t)stream);
}
