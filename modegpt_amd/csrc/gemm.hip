// General strided fp64 GEMM on v_mfma_f64_16x16x4_f64, the building block of the decomposition kernels
// (Cholesky trailing updates, TRSM by inverted diagonal blocks, triangular inverse, Nystrom cross term
// C[idx,:] @ W_d^T with fused row gather, VO Gram / factor products with fused bf16 load and store).
// Same 128x128 workgroup tile / 4-wave / LDS-fp64-panel core as cov.hip.
#include <type_traits>

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
  int fast_ok;  // operands satisfy the 16-byte alignment rules of the vector staging path (host-checked)
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

__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const Acc& acc, int64_t batch, int64_t i0, int64_t j0,
                                              int wr, int wc, int lane) {
  const bool c_bf16 = (g.c_dtype == MDG_BF16);
  char* Cb = (char*)g.C + batch * g.c_bs * (c_bf16 ? 2 : 8);
  const int64_t gc0 = j0 + acc_col(wc, lane, 0);
#pragma unroll
  for (int sa = 0; sa < 4; sa++) {
    // beta != 0: the 16 values of C this lane updates in this quarter are loaded FIRST, all of them, and added and stored afterwards.
    // Written as `v += beta * *dst; *dst = v` per element the compiler has to keep every store in front of the next row's load (it
    // cannot tell the rows apart): 16 dependent memory round trips per tile, ~25 us -- a tenth of a rank-1024 update's tile, which is
    // what kept those GEMMs (Cholesky trailing updates, substitution carries) at 55 - 60 TF where a beta = 0 product runs at 61 - 70.
    double old[4][4];
    if (!c_bf16 && g.beta != 0.) {
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int64_t gr = i0 + acc_row(wr, lane, sa, reg);
#pragma unroll
        for (int sb = 0; sb < 4; sb++) old[reg][sb] = (gr < g.M && gc0 + sb < g.N) ? ((const double*)Cb)[gr * g.ldc + gc0 + sb] : 0.;
      }
    }
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
      const int64_t gr = i0 + acc_row(wr, lane, sa, reg);
      if (gr >= g.M) continue;
      const int64_t e0 = gr * g.ldc + gc0;
#pragma unroll
      for (int sb = 0; sb < 4; sb++) {
        if (gc0 + sb >= g.N) continue;
        double v = g.alpha * acc.v[sa][sb][reg];
        if (!c_bf16) {
          if (g.beta != 0.) v += g.beta * old[reg][sb];
          ((double*)Cb)[e0 + sb] = v;
        } else {
          ((bf16_t*)Cb)[e0 + sb] = f64_to_bf16(v);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- vector staging for interior tiles
// Same lesson as the covariance kernel: every staging instruction costs MFMA issue slots.  Interior tiles whose k-range
// is a whole number of stages take this path: 16-byte global loads at (uniform base + constant per-lane 32-bit
// offset) -- the base lives in SGPRs and is bumped by scalar adds, so a load costs no VALU -- no bounds tests, no
// offset-table reads, static buffer parity.
template <int DT, bool KC> struct FastPanel;

template <bool KC> struct FastPanel<MDG_F64, KC> {
  d2 r[4];
  unsigned voff[4];  // byte offsets from the stage base
  int lo[4];         // LDS element index of the first value
  // xmax: rows / columns of the operand this tile really has (< 128 in the last, ragged tile row or column).  With k contiguous a
  // thread's loads belong to ONE row, so a row beyond the end simply re-reads the last valid one (its products are never stored);
  // with x contiguous a 16-byte load spans two rows' worth of x and the caller keeps ragged tiles off this path.
  __device__ __forceinline__ void init(int64_t sx, int64_t sk, int64_t x0, const int64_t* rows, int tid, int xmax = 128) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int e = tid + 256 * p;
      if (KC) {
        const int kp = e & 7, x = min(e >> 3, xmax - 1);
        const int64_t row = rows ? rows[x0 + x] : x0 + x;
        voff[p] = (unsigned)((row * sx + 2 * kp) * 8);
        lo[p] = (2 * kp) * PITCH + (e >> 3);
      } else {
        const int xp = e & 63, k = e >> 6;
        voff[p] = (unsigned)((k * sk + x0 + 2 * xp) * 8);
        lo[p] = k * PITCH + 2 * xp;
      }
    }
  }
  __device__ __forceinline__ void load(const char* base) {
#pragma unroll
    for (int p = 0; p < 4; p++) r[p] = *(const d2*)(base + voff[p]);
  }
  __device__ __forceinline__ void store(double* panel) const {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      if (KC) {
        panel[lo[p]] = r[p].x;
        panel[lo[p] + PITCH] = r[p].y;
      } else {
        *(d2*)(panel + lo[p]) = r[p];
      }
    }
  }
  static __device__ __forceinline__ int64_t stage_bytes(int64_t sk) { return (KC ? BK : BK * sk) * 8; }
};

template <bool KC> struct FastPanel<MDG_BF16, KC> {
  uint4 r;
  unsigned voff;
  int lo;
  __device__ __forceinline__ void init(int64_t sx, int64_t sk, int64_t x0, const int64_t* rows, int tid, int xmax = 128) {
    if (KC) {
      const int kp = tid & 1, x = min(tid >> 1, xmax - 1);
      const int64_t row = rows ? rows[x0 + x] : x0 + x;
      voff = (unsigned)((row * sx + 8 * kp) * 2);
      lo = (8 * kp) * PITCH + (tid >> 1);
    } else {
      const int xp = tid & 15, k = tid >> 4;
      voff = (unsigned)((k * sk + x0 + 8 * xp) * 2);
      lo = k * PITCH + 8 * xp;
    }
  }
  __device__ __forceinline__ void load(const char* base) { r = *(const uint4*)(base + voff); }
  __device__ __forceinline__ void store(double* panel) const {
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const double v0 = (double)__uint_as_float(w[q] << 16), v1 = (double)__uint_as_float(w[q] & 0xffff0000u);
      if (KC) {
        panel[lo + (2 * q) * PITCH] = v0;
        panel[lo + (2 * q + 1) * PITCH] = v1;
      } else {
        *(d2*)(panel + lo + 2 * q) = (d2){v0, v1};
      }
    }
  }
  static __device__ __forceinline__ int64_t stage_bytes(int64_t sk) { return (KC ? BK : BK * sk) * 2; }
};

template <int ADT, int BDT, bool AKC, bool BKC>
__device__ __forceinline__ void gemm_tile_fast(const GemmArgs& g, double* lds, const char* Ab, const char* Bb, int64_t i0,
                                               int64_t j0, int64_t k_begin, int64_t n_stage, Acc& acc, int tid, int lane,
                                               int wr, int wc) {
  FastPanel<ADT, AKC> fa;
  FastPanel<BDT, BKC> fb;
  // uniform stage bases: tile origin + first k of the range; per-lane offsets cover the row/col and in-stage k
  const int64_t esa = (int64_t)sizeof(typename ElemOf<ADT>::type), esb = (int64_t)sizeof(typename ElemOf<BDT>::type);
  const bool gather = g.a_rows != nullptr;
  fa.init(g.sa_i, g.sa_k, gather ? i0 : 0, g.a_rows, tid, (int)min((int64_t)TILE, g.M - i0));   // with a gather the row offset is absolute
  fb.init(g.sb_j, g.sb_k, 0, nullptr, tid, (int)min((int64_t)TILE, g.N - j0));
  const char* pa = Ab + ((gather || !AKC ? 0 : i0 * g.sa_i) + (AKC ? k_begin : k_begin * g.sa_k + i0)) * esa;
  const char* pb = Bb + ((BKC ? j0 * g.sb_j + k_begin : k_begin * g.sb_k + j0)) * esb;
  const int64_t da = FastPanel<ADT, AKC>::stage_bytes(g.sa_k), db = FastPanel<BDT, BKC>::stage_bytes(g.sb_k);
  fa.load(pa);
  fb.load(pb);
  fa.store(lds);
  fb.store(lds + 2 * PANEL);
  __syncthreads();
  auto stage = [&](int64_t s, auto cur_c) {
    constexpr int CUR = decltype(cur_c)::value;
    const bool more = s + 1 < n_stage;
    if (more) {
      pa += da;
      pb += db;
      fa.load(pa);
    }
    mma_steps<0, BK / 8>(lds + CUR * PANEL, lds + (2 + CUR) * PANEL, wr, wc, lane, acc);
    if (more) {
      fa.store(lds + (CUR ^ 1) * PANEL);
      fb.load(pb);
    }
    mma_steps<BK / 8, BK / 4>(lds + CUR * PANEL, lds + (2 + CUR) * PANEL, wr, wc, lane, acc);
    if (more) fb.store(lds + (2 + (CUR ^ 1)) * PANEL);
    __syncthreads();
  };
  int64_t s = 0;
  for (; s + 1 < n_stage; s += 2) {
    stage(s, std::integral_constant<int, 0>{});
    stage(s + 1, std::integral_constant<int, 1>{});
  }
  if (s < n_stage) stage(s, std::integral_constant<int, 0>{});
}

template <int ADT, int BDT, bool AKC, bool BKC>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
  __shared__ double lds[4 * PANEL + 2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int bi, bj;
  if (g.flags & MDG_GEMM_LOWER_ONLY) {
    tri_decode(blockIdx.x, bi, bj);
  } else if (g.flags & MDG_GEMM_B_LOWER_TRI) {
    // a tile's k range starts at its first column: the tiles of tile column 0 are the longest, those of the last column 1 / tiles_n of
    // that.  Longest first (the dispatcher hands workgroups out in index order): in row-major order the last tile ROW's long tiles
    // start when the launch is nearly over and run on alone -- the triangular inverse's T = L21 X11 at s = 8192 took 9.2 ms for
    // 6.5 ms of work at the rate of its untriangular twin.
    bj = blockIdx.x / g.tiles_m;
    bi = blockIdx.x % g.tiles_m;
  } else if (g.flags & MDG_GEMM_A_LOWER_TRI) {
    bi = g.tiles_m - 1 - blockIdx.x / g.tiles_n;     // (k range ends behind the tile's last row: the last tile row is the longest)
    bj = blockIdx.x % g.tiles_n;
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

  const int64_t n_stage = k_end > k_begin ? (k_end - k_begin + BK - 1) / BK : 0;
  // (a ragged last tile row / column stays on the vector path where that operand is k-contiguous: FastPanel::init)
  const bool interior = (AKC || i0 + TILE <= g.M) && (BKC || j0 + TILE <= g.N);
  if (g.fast_ok && interior && n_stage > 0 && (k_end - k_begin) % BK == 0 && (AKC || !g.a_rows)) {
    Acc facc;  // separate accumulator set: the two paths never share live registers
    acc_zero(facc);
    gemm_tile_fast<ADT, BDT, AKC, BKC>(g, lds, Ab, Bb, i0, j0, k_begin, n_stage, facc, tid, lane, wr, wc);
    gemm_epilogue(g, facc, batch, i0, j0, wr, wc, lane);
    return;
  }
  Acc acc;
  acc_zero(acc);
  double rg[8];  // one panel's prefetch at a time: A rides under the first half of a stage, B under the second
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

  gemm_epilogue(g, acc, batch, i0, j0, wr, wc, lane);
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
  {  // 16-byte alignment rules of FastPanel: per-operand unit (2 doubles / 8 bf16) along the vector-load axis
    const int64_t ua = a_dtype == MDG_F64 ? 2 : 8, ub = b_dtype == MDG_F64 ? 2 : 8;
    const bool a_kc = (sa_k == 1 && sa_i != 1), b_kc = (sb_k == 1 && sb_j != 1);
    bool ok = ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && (a_bs % ua == 0) && (b_bs % ub == 0);
    ok = ok && (a_kc ? sa_i % ua == 0 : (sa_i == 1 && sa_k % ua == 0));
    ok = ok && (b_kc ? sb_j % ub == 0 : (sb_j == 1 && sb_k % ub == 0));
    // per-lane offsets are 32-bit: the furthest element a stage touches must be < 4 GiB from the stage base
    const int64_t span_a = a_rows ? (int64_t)1 << 62 : (a_kc ? TILE * sa_i : BK * sa_k + TILE) * (int64_t)(16 / ua);
    const int64_t span_b = (b_kc ? TILE * sb_j : BK * sb_k + TILE) * (int64_t)(16 / ub);
    ok = ok && span_b < ((int64_t)1 << 32) && (a_rows || span_a < ((int64_t)1 << 32));
    g.fast_ok = ok ? 1 : 0;
  }
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
