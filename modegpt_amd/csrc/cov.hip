// Activation-covariance accumulation: sigma (lower) += X^T X in fp64 on v_mfma_f64_16x16x4_f64.
//
// Replaces the calibration hooks' `H.T @ H`, `sum(X.mT @ X, 0)` and per-head bmm
// (LlamaAdapter.py:115-147, model_adapter.py:546-567) -- >98 % of the path's flops.
//
// Launch shape: one 256-thread workgroup per 128x128 output tile of the LOWER triangle (x batch x k-split);
// the token dimension is streamed through LDS in 16-token stages, converted to fp64 once while staging, so
// the inner loop is LDS reads + 16 MFMAs per 4 tokens.  The kernel is fp64-MFMA bound: at 128x128 tiles the
// operand traffic is 2 B/cycle/CU (DESIGN.md "cov_accum").
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace mdg {

struct CovArgs {
  const void* x;
  int64_t ld, n_tokens;
  int n_feat, batch, tiles, ntri;
  double* sigma;
  int64_t ld_sigma, sigma_bs;
  int ksplit;
  int64_t tokens_per_split;
  double* partial;
  int vec_ok;
  // device-side route gate (mdg_cov_accum_i8's fp64 fallback): when set, every workgroup exits at once unless
  // (*gate & gate_mask) == gate_want -- the launch is enqueued unconditionally and the DEVICE decides whether it runs
  const int* gate;
  int gate_mask, gate_want;
};

template <int DT> struct Stage {
  typedef typename ElemOf<DT>::type T;
  static constexpr int VEC = 16 / (int)sizeof(T);        // elements per 16-byte chunk
  static constexpr int CPR = TILE / VEC;                 // chunks per token row
  static constexpr int CPT = BK * CPR / 256;             // chunks per thread
  union Chunk {
    uint4 q;
    T e[VEC];
  };
  // Which (token row, first column) staging chunk c of a BK x 128 panel is.  NOT the obvious c/CPR, c%CPR: a chunk
  // widens to VEC*8 bytes of fp64, so neighbouring chunks of one row sit 64 B (bf16) apart and the 8 lanes of a
  // ds_write_b128 group would share 2 of the 8 bank quads (4-way conflict, measured: it starves the fragment
  // reads).  Rows are 1040 B apart = 1 bank quad (mod 8), so a group of RG rows x 8/RG chunks covers all 8 quads.
  static constexpr int RG = VEC >= 8 ? 4 : (VEC == 4 ? 2 : 1);
  static constexpr int CG = 8 / RG;
  __device__ static __forceinline__ void chunk_rc(int c, int& row, int& col) {
    const int g = c >> 3, i = c & 7;
    row = (g % (BK / RG)) * RG + (i % RG);
    col = ((g / (BK / RG)) * CG + i / RG) * VEC;
  }
};

// FAST: every tile is full and 16-byte loads are legal, so the stage is one unconditional vector load per
// chunk and its s_waitcnt can sink below the stage's MFMAs; the generic path (ragged n_feat, unaligned
// views) goes element by element.
template <int DT, bool FAST, bool INB = false>
__device__ __forceinline__ void load_panel(const CovArgs& a, int64_t tok0, int64_t tok_end, int64_t col0,
                                           int col_lim, int tid, typename Stage<DT>::Chunk* regs) {
  typedef Stage<DT> S;
  typedef typename S::T T;
  const T* x = (const T*)a.x;
#pragma unroll
  for (int p = 0; p < S::CPT; p++) {
    int row, col;
    S::chunk_rc(tid + 256 * p, row, col);
    int64_t tok = tok0 + row;
    typename S::Chunk ch;
    ch.q = make_uint4(0, 0, 0, 0);
    if (FAST && INB) {
      ch.q = *(const uint4*)(x + tok * a.ld + col0 + col);  // whole stage in range: no exec masking, no branch
    } else if (FAST) {
      if (tok < tok_end) ch.q = *(const uint4*)(x + tok * a.ld + col0 + col);
    } else if (tok < tok_end) {
      const T* src = x + tok * a.ld + col0 + col;
      if (a.vec_ok && col + S::VEC <= col_lim) {
        ch.q = *(const uint4*)src;
      } else {
#pragma unroll
        for (int e = 0; e < S::VEC; e++)
          if (col + e < col_lim) ch.e[e] = src[e];
      }
    }
    regs[p] = ch;
  }
}

template <int DT, bool RELU>
__device__ __forceinline__ void store_panel(double* panel, int tid, const typename Stage<DT>::Chunk* regs) {
  typedef Stage<DT> S;
#pragma unroll
  for (int p = 0; p < S::CPT; p++) {
    int row, col;
    S::chunk_rc(tid + 256 * p, row, col);
    double* dst = panel + row * PITCH + col;
#pragma unroll
    for (int e = 0; e < S::VEC; e += 2) {
      double v0 = load_f64<DT>(regs[p].e, e), v1 = load_f64<DT>(regs[p].e, e + 1);
      if (RELU) {
        v0 = v0 > 0. ? v0 : 0.;
        v1 = v1 > 0. ? v1 : 0.;
      }
      *(d2*)(dst + e) = (d2){v0, v1};
    }
  }
}

// One quarter of store_panel: elements [2q .. 2q+2) x (VEC/8 ... ) -- every dtype stages a panel in 4 pieces of
// 2*CPT*VEC/8 fp64 values, so the conversion + LDS write can be dealt out between MFMAs.
template <int DT, bool RELU>
__device__ __forceinline__ void store_piece(double* panel, int tid, const typename Stage<DT>::Chunk* regs, int piece) {
  typedef Stage<DT> S;
  constexpr int PAIRS = S::CPT * S::VEC / 2;  // d2 writes per thread per panel
  constexpr int PER = PAIRS / 4;
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int w = piece * PER + u;          // which d2 of this thread
    const int p = w / (S::VEC / 2), e = (w % (S::VEC / 2)) * 2;
    int row, col;
    S::chunk_rc(tid + 256 * p, row, col);
    double v0 = load_f64<DT>(regs[p].e, e), v1 = load_f64<DT>(regs[p].e, e + 1);
    if (RELU) {
      v0 = v0 > 0. ? v0 : 0.;
      v1 = v1 > 0. ? v1 : 0.;
    }
    *(d2*)(panel + row * PITCH + col + e) = (d2){v0, v1};
  }
}

// Second half of a stage (k4-steps 2 and 3, 32 MFMAs) with the next stage's staging dealt out in 8 pieces, one
// after every 4 MFMAs; sched_barrier pins the interleave so the conversions and ds_writes retire in the MFMAs'
// shadow instead of in a block in front of the s_barrier.
template <int DT, bool RELU, bool DIAG>
__device__ __forceinline__ void mma_half_with_staging(const double* __restrict__ As, const double* __restrict__ Bs,
                                                      int wr, int wc, int lane, Acc& acc, bool stage_next,
                                                      double* nextA, double* nextB, int tid,
                                                      const typename Stage<DT>::Chunk* ra,
                                                      const typename Stage<DT>::Chunk* rb) {
  const int m = lane & 15, kq = lane >> 4;
  double a[2][4], b[2][4];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int k4 = BK / 8 + h;
    const double* ap = As + (k4 * 4 + kq) * PITCH + wr * WTILE + 4 * m;
    const double* bp = Bs + (k4 * 4 + kq) * PITCH + wc * WTILE + 4 * m;
    d2 a01 = *(const d2*)ap, a23 = *(const d2*)(ap + 2);
    d2 b01 = *(const d2*)bp, b23 = *(const d2*)(bp + 2);
    a[h][0] = a01.x; a[h][1] = a01.y; a[h][2] = a23.x; a[h][3] = a23.y;
    b[h][0] = b01.x; b[h][1] = b01.y; b[h][2] = b23.x; b[h][3] = b23.y;
  }
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int sa = 0; sa < 4; sa++) {
#pragma unroll
      for (int sb = 0; sb < 4; sb++)
        acc.v[sa][sb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[h][sa], b[h][sb], acc.v[sa][sb], 0, 0, 0);
      if (stage_next) {
        const int piece = h * 4 + sa;
        if (piece < 4) store_piece<DT, RELU>(nextA, tid, ra, piece);
        else if (!DIAG) store_piece<DT, RELU>(nextB, tid, rb, piece - 4);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
}

// `split` = which token range, `slot` = index of this (batch, tile) among the a.batch * a.ntri tiles of the problem.
template <int DT, bool RELU, bool FAST, bool DIAG>
__device__ __forceinline__ void cov_tile(const CovArgs& a, double* lds, int b, int bi, int bj, int split, int slot) {
  typedef Stage<DT> S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t tok_begin = (int64_t)split * a.tokens_per_split;
  int64_t tok_end = tok_begin + a.tokens_per_split;
  if (tok_end > a.n_tokens) tok_end = a.n_tokens;
  const int64_t colA = (int64_t)b * a.n_feat + (int64_t)bi * TILE;
  const int64_t colB = (int64_t)b * a.n_feat + (int64_t)bj * TILE;
  const int limA = min(TILE, a.n_feat - bi * TILE), limB = min(TILE, a.n_feat - bj * TILE);

  Acc acc;
  acc_zero(acc);
  // Software pipeline, prefetch distance 2: while stage s is multiplied, stage s+1 sits in registers (loaded
  // during stage s-1) and is converted + written to the other LDS buffer in the SHADOW of this stage's MFMAs
  // (between its two halves), and the global loads of stage s+2 are in flight.  What is left between the last
  // MFMA of a stage and the first of the next is one s_barrier and one LDS read latency.
  typename S::Chunk ra1[S::CPT], rb1[S::CPT], ra2[S::CPT], rb2[S::CPT];
  const int64_t n_stage = tok_end > tok_begin ? (tok_end - tok_begin + BK - 1) / BK : 0;

  if (n_stage > 0) {
    load_panel<DT, FAST>(a, tok_begin, tok_end, colA, limA, tid, ra1);
    if (!DIAG) load_panel<DT, FAST>(a, tok_begin, tok_end, colB, limB, tid, rb1);
    store_panel<DT, RELU>(lds, tid, ra1);
    if (!DIAG) store_panel<DT, RELU>(lds + 2 * PANEL, tid, rb1);
    if (n_stage > 1) {
      load_panel<DT, FAST>(a, tok_begin + BK, tok_end, colA, limA, tid, ra1);
      if (!DIAG) load_panel<DT, FAST>(a, tok_begin + BK, tok_end, colB, limB, tid, rb1);
    }
  }
  __syncthreads();
  // One stage: issue the loads of stage s+2 into `ld`, multiply LDS buffer CUR, deal stage s+1 (registers `st`) out to
  // buffer CUR^1 between the MFMAs, barrier.  Unrolled by two so buffer parity and the two register sets are
  // compile-time; STEADY = stages s+1 and s+2 exist and lie wholly inside the token range, so the steady-state loop
  // carries no branches and no exec masking (every non-MFMA instruction costs MFMA issue slots on the SIMD).
  auto stage = [&](int64_t s, auto cur_c, auto steady_c, typename S::Chunk* st_a, typename S::Chunk* st_b,
                   typename S::Chunk* ld_a, typename S::Chunk* ld_b) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    if (STEADY || s + 2 < n_stage) {
      const int64_t tk = tok_begin + (s + 2) * BK;
      load_panel<DT, FAST, STEADY>(a, tk, tok_end, colA, limA, tid, ld_a);
      if (!DIAG) load_panel<DT, FAST, STEADY>(a, tk, tok_end, colB, limB, tid, ld_b);
    }
    const double* As = lds + CUR * PANEL;
    const double* Bs = DIAG ? As : lds + (2 + CUR) * PANEL;
    mma_steps<0, BK / 8>(As, Bs, wr, wc, lane, acc);
    mma_half_with_staging<DT, RELU, DIAG>(As, Bs, wr, wc, lane, acc, STEADY || s + 1 < n_stage, lds + (CUR ^ 1) * PANEL,
                                          lds + (2 + (CUR ^ 1)) * PANEL, tid, st_a, st_b);
    __syncthreads();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  int64_t s = 0;
  // steady state: stage s+3 must be complete too, since the unrolled pair loads stages s+2 and s+3
  const int64_t n_full = (tok_end - tok_begin) / BK;  // stages wholly inside the token range
  for (; s + 3 < n_full; s += 2) {
    stage(s, I0{}, std::true_type{}, ra1, rb1, ra2, rb2);
    stage(s + 1, I1{}, std::true_type{}, ra2, rb2, ra1, rb1);
  }
  for (; s + 1 < n_stage; s += 2) {
    stage(s, I0{}, std::false_type{}, ra1, rb1, ra2, rb2);
    stage(s + 1, I1{}, std::false_type{}, ra2, rb2, ra1, rb1);
  }
  if (s < n_stage) stage(s, I0{}, std::false_type{}, ra1, rb1, ra2, rb2);

  // epilogue
  if (a.ksplit == 1) {
    double* sg = a.sigma + (int64_t)b * a.sigma_bs;
#pragma unroll
    for (int sa = 0; sa < 4; sa++)
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        int r = acc_row(wr, lane, sa, reg);
        int64_t gr = (int64_t)bi * TILE + r;
        if (gr >= a.n_feat) continue;
        int c0 = acc_col(wc, lane, 0);
        int64_t gc0 = (int64_t)bj * TILE + c0;
        double* row = sg + gr * a.ld_sigma + gc0;
#pragma unroll
        for (int sb = 0; sb < 4; sb++)
          if (gc0 + sb < a.n_feat) row[sb] += acc.v[sa][sb][reg];
      }
  } else {
    double* pt = a.partial + ((int64_t)split * (a.batch * a.ntri) + slot) * (TILE * TILE);
#pragma unroll
    for (int sa = 0; sa < 4; sa++)
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        int r = acc_row(wr, lane, sa, reg);
        int c0 = acc_col(wc, lane, 0);
        d4 v = {acc.v[sa][0][reg], acc.v[sa][1][reg], acc.v[sa][2][reg], acc.v[sa][3][reg]};
        *(d4*)(pt + r * TILE + c0) = v;
      }
  }
}

template <int DT, bool RELU, bool FAST>
__global__ __launch_bounds__(256, 2) void cov_accum_kernel(CovArgs a) {
  __shared__ double lds[4 * PANEL];  // As[2], Bs[2]
  if (a.gate && (*a.gate & a.gate_mask) != a.gate_want) return;
  const int b = blockIdx.x / a.ntri, t = blockIdx.x % a.ntri;
  int bi, bj;
  tri_decode(t, bi, bj);
  if (bi == bj) cov_tile<DT, RELU, FAST, true>(a, lds, b, bi, bj, blockIdx.y, blockIdx.x);
  else cov_tile<DT, RELU, FAST, false>(a, lds, b, bi, bj, blockIdx.y, blockIdx.x);
}

// Several covariance problems of one calibration batch in ONE launch (the four hooks of a layer: sigma_mlp, sigma_x,
// sigma_q, sigma_k).  Launched one after the other, each pays its own under-filled last round (6328 tiles on 512
// slots leave 328 slots idle for a whole tile time) and the small problems cannot fill the GPU at all; in one grid the
// workgroups of the small problems (split finely over tokens) are dispatched into the slots the big one leaves free.
// Units are ordered big problem first.  FAST path only (full tiles, 16-byte aligned rows), no ReLU.
constexpr int MULTI_MAX = 4;
struct CovMulti {
  int n;
  int unit_start[MULTI_MAX + 1];
  CovArgs p[MULTI_MAX];
};

template <int DT>
__global__ __launch_bounds__(256, 2) void cov_accum_multi_kernel(CovMulti m) {
  __shared__ double lds[4 * PANEL];
  int u = blockIdx.x, pi = 0;
  while (pi + 1 < m.n && u >= m.unit_start[pi + 1]) pi++;
  u -= m.unit_start[pi];
  const CovArgs& a = m.p[pi];
  const int tiles = a.batch * a.ntri;
  const int split = u / tiles, slot = u % tiles;
  const int b = slot / a.ntri, t = slot % a.ntri;
  int bi, bj;
  tri_decode(t, bi, bj);
  if (bi == bj) cov_tile<DT, false, true, true>(a, lds, b, bi, bj, split, slot);
  else cov_tile<DT, false, true, false>(a, lds, b, bi, bj, split, slot);
}

// sigma tile += sum over splits (fixed order) of the partial tiles.  grid = (tiles, 64): 256 elements per block.
__global__ __launch_bounds__(256) void cov_reduce_kernel(CovArgs a) {
  if (a.gate && (*a.gate & a.gate_mask) != a.gate_want) return;
  const int b = blockIdx.x / a.ntri, t = blockIdx.x % a.ntri;
  int bi, bj;
  tri_decode(t, bi, bj);
  double* sg = a.sigma + (int64_t)b * a.sigma_bs;
  const int64_t tile_stride = (int64_t)gridDim.x * (TILE * TILE);
  const double* pt = a.partial + (int64_t)blockIdx.x * (TILE * TILE);
  const int e = blockIdx.y * 256 + threadIdx.x;
  const int r = e / TILE, c = e % TILE;
  const int64_t gr = (int64_t)bi * TILE + r, gc = (int64_t)bj * TILE + c;
  if (gr >= a.n_feat || gc >= a.n_feat) return;
  double s = 0.;
  for (int k = 0; k < a.ksplit; k++) s += pt[k * tile_stride + e];
  sg[gr * a.ld_sigma + gc] += s;
}

// lower -> scaled lower + mirrored upper, 32x32 tiles through LDS.
__global__ __launch_bounds__(256) void cov_finalize_kernel(double* sigma, int n, int64_t ld, int64_t bs, int tiles,
                                                           int ntri, double scale) {
  __shared__ double tl[32][33];
  const int b = blockIdx.x / ntri, t = blockIdx.x % ntri;
  int ti, tj;
  tri_decode(t, ti, tj);
  double* sg = sigma + (int64_t)b * bs;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    int gr = ti * 32 + r, gc = tj * 32 + tx;
    tl[r][tx] = (gr < n && gc < n) ? sg[(int64_t)gr * ld + gc] * scale : 0.;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int gr = ti * 32 + r, gc = tj * 32 + tx;
    if (gr < n && gc < n) {
      double v = (ti == tj && tx > r) ? tl[tx][r] : tl[r][tx];
      sg[(int64_t)gr * ld + gc] = v;
    }
    if (ti != tj) {
      int ur = tj * 32 + r, uc = ti * 32 + tx;  // transposed tile
      if (ur < n && uc < n) sg[(int64_t)ur * ld + uc] = tl[tx][r];
    }
  }
}

// ---------------------------------------------------------------- BI score
template <int DT>
__global__ __launch_bounds__(256) void bi_partial_kernel(const void* xin, const void* xout, int64_t n_tokens,
                                                         int64_t d, int64_t ld, double* partial) {
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double local = 0.;
  for (int64_t tok = (int64_t)blockIdx.x * 4 + wave; tok < n_tokens; tok += (int64_t)gridDim.x * 4) {
    double dot = 0., na = 0., nb = 0.;
    for (int64_t j = lane; j < d; j += 64) {
      double u = load_f64<DT>(xin, tok * ld + j), v = load_f64<DT>(xout, tok * ld + j);
      dot += u * v;
      na += u * u;
      nb += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) {
      dot += __shfl_xor(dot, o);
      na += __shfl_xor(na, o);
      nb += __shfl_xor(nb, o);
    }
    const double eps = 1e-8;  // torch.cosine_similarity default
    double den = fmax(sqrt(na), eps) * fmax(sqrt(nb), eps);
    local += 1.0 - dot / den;
  }
  if (lane == 0) red[wave] = local;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void bi_final_kernel(const double* partial, int n, double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.;
    for (int i = 0; i < n; i++) s += partial[i];
    *out += s;
  }
}

// ---------------------------------------------------------------- host side
// Split-K factor from a two-term cost model: MFMA time of the rounded-up number of workgroup rounds (2 workgroups
// per CU) plus the HBM round trip of the fp64 partial tiles.  Deterministic in its arguments (the _ws_bytes twin
// must agree), so the CU count is fixed to the MI355X's 256.
static int cov_ksplit(int64_t n_tokens, int64_t n_feat, int64_t batch, int64_t* tokens_per_split) {
  const int64_t T = ceil_div(n_feat, TILE);
  const int64_t blocks = batch * T * (T + 1) / 2;
  const double slots = 512.0;
  const double us_per_token = 2.0 * TILE * TILE / (78.6e12 / slots) * 1e6;  // one workgroup, one token
  const double us_per_tile_rt = 2.0 * TILE * TILE * 8 / 4.0e12 * 1e6;       // write + read one partial tile
  int64_t max_split = n_tokens / (8 * BK);                                  // >= 128 tokens per split
  if (max_split < 1) max_split = 1;
  if (max_split > 256) max_split = 256;
  int64_t best = 1;
  double best_cost = 0.;
#ifdef MDG_EXPERIMENT   // knob of scripts/bench_kernels.py; compiled out of the product library (no env var changes its arithmetic)
  if (const char* ev = getenv("MDG_COV_KSPLIT")) {
    const int64_t f = atoll(ev);
    if (f >= 1 && f <= max_split) {
      const int64_t tps = (int64_t)align_up((size_t)ceil_div(n_tokens, f), BK);
      *tokens_per_split = tps;
      return (int)ceil_div(n_tokens, tps);
    }
  }
#endif
  for (int64_t ks = 1; ks <= max_split; ks++) {
    const int64_t tps = (int64_t)align_up((size_t)ceil_div(n_tokens, ks), BK);
    const int64_t real = ceil_div(n_tokens, tps);
    if (real != ks) continue;
    double rounds = (double)ceil_div(blocks * ks, (int64_t)slots);
    double cost = rounds * (double)tps * us_per_token + (ks > 1 ? (double)(blocks * ks) * us_per_tile_rt + 3.0 : 0.);
    // hysteresis towards fewer splits: leaving ks = 1 must pay 5 % (it adds a reduce kernel and a workspace; the tail
    // round of an unsplit launch runs one workgroup per CU and is faster than this model says), going deeper 2 %
    if (ks == 1 || cost < best_cost * (best == 1 ? 0.95 : 0.98)) {
      best = ks;
      best_cost = cost;
    }
  }
  int64_t tps = (int64_t)align_up((size_t)ceil_div(n_tokens > 0 ? n_tokens : 1, best), BK);
  *tokens_per_split = tps;
  return (int)ceil_div(n_tokens > 0 ? n_tokens : 1, tps);
}

template <int DT>
static void launch_cov(const CovArgs& a, int relu, bool fast, hipStream_t st) {
  dim3 grid(a.batch * a.ntri, a.ksplit);
  if (fast) {
    if (relu) hipLaunchKernelGGL((cov_accum_kernel<DT, true, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((cov_accum_kernel<DT, false, true>), grid, dim3(256), 0, st, a);
  } else {
    if (relu) hipLaunchKernelGGL((cov_accum_kernel<DT, true, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((cov_accum_kernel<DT, false, false>), grid, dim3(256), 0, st, a);
  }
}

}  // namespace mdg

using namespace mdg;

extern "C" size_t mdg_cov_accum_ws_bytes(int64_t n_tokens, int64_t n_feat, int64_t batch) {
  if (n_tokens <= 0 || n_feat <= 0 || batch <= 0) return 0;
  int64_t tps;
  int ks = cov_ksplit(n_tokens, n_feat, batch, &tps);
  if (ks == 1) return 0;
  int64_t T = ceil_div(n_feat, TILE);
  return (size_t)ks * batch * (T * (T + 1) / 2) * TILE * TILE * sizeof(double);
}

extern "C" int mdg_cov_accum(const void* x, int dtype, int64_t n_tokens, int64_t n_feat, int64_t batch,
                             int64_t ld, int relu, double* sigma, int64_t ld_sigma, int64_t sigma_bs, void* ws,
                             size_t ws_bytes, void* stream) {
  return mdg::cov_accum_gated(x, dtype, n_tokens, n_feat, batch, ld, relu, sigma, ld_sigma, sigma_bs, ws, ws_bytes, nullptr, 0, 0,
                              stream);
}

int mdg::cov_accum_gated(const void* x, int dtype, int64_t n_tokens, int64_t n_feat, int64_t batch, int64_t ld, int relu,
                         double* sigma, int64_t ld_sigma, int64_t sigma_bs, void* ws, size_t ws_bytes, const int* gate,
                         int gate_mask, int gate_want, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(n_tokens >= 0 && n_feat > 0 && batch > 0, "mdg_cov_accum: bad sizes (tokens=%lld feat=%lld batch=%lld)",
                (long long)n_tokens, (long long)n_feat, (long long)batch);
  MDG_CHECK_ARG(dtype >= MDG_BF16 && dtype <= MDG_F64, "mdg_cov_accum: unknown dtype %d", dtype);
  MDG_CHECK_ARG(ld >= n_feat * batch, "mdg_cov_accum: ld %lld < batch*n_feat %lld", (long long)ld,
                (long long)(n_feat * batch));
  MDG_CHECK_ARG(ld_sigma >= n_feat, "mdg_cov_accum: ld_sigma %lld < n_feat", (long long)ld_sigma);
  MDG_CHECK_ARG(batch == 1 || sigma_bs >= n_feat * ld_sigma - (ld_sigma - n_feat),
                "mdg_cov_accum: sigma batch stride too small");
  MDG_CHECK_ARG(n_feat < (1 << 30) && batch * n_feat < (1ll << 31), "mdg_cov_accum: n_feat too large");
  if (n_tokens == 0) return MDG_OK;
  MDG_CHECK_ARG(x && sigma, "mdg_cov_accum: null pointer");
  hipStream_t st = (hipStream_t)stream;
  CovArgs a;
  a.x = x;
  a.ld = ld;
  a.n_tokens = n_tokens;
  a.n_feat = (int)n_feat;
  a.batch = (int)batch;
  a.tiles = (int)ceil_div(n_feat, TILE);
  a.ntri = a.tiles * (a.tiles + 1) / 2;
  a.sigma = sigma;
  a.ld_sigma = ld_sigma;
  a.sigma_bs = sigma_bs;
  a.ksplit = cov_ksplit(n_tokens, n_feat, batch, &a.tokens_per_split);
  a.partial = (double*)ws;
  a.gate = gate;
  a.gate_mask = gate_mask;
  a.gate_want = gate_want;
  size_t esz = dtype_size(dtype);
  a.vec_ok = ((uintptr_t)x % 16 == 0) && ((ld * esz) % 16 == 0) && ((n_feat * esz) % 16 == 0 || batch == 1);
  if (a.ksplit > 1) {
    size_t need = mdg_cov_accum_ws_bytes(n_tokens, n_feat, batch);
    MDG_CHECK_ARG(ws && ws_bytes >= need, "mdg_cov_accum: workspace %zu < required %zu", ws_bytes, need);
  }
  MDG_CHECK_ARG((int64_t)a.batch * a.ntri < (1ll << 31) && a.ksplit < 65536, "mdg_cov_accum: grid too large");
  const bool fast = a.vec_ok && (n_feat % TILE == 0);
  switch (dtype) {
    case MDG_BF16: launch_cov<MDG_BF16>(a, relu, fast, st); break;
    case MDG_F16: launch_cov<MDG_F16>(a, relu, fast, st); break;
    case MDG_F32: launch_cov<MDG_F32>(a, relu, fast, st); break;
    default: launch_cov<MDG_F64>(a, relu, fast, st); break;
  }
  MDG_LAUNCH_CHECK();
  if (a.ksplit > 1) {
    hipLaunchKernelGGL(cov_reduce_kernel, dim3(a.batch * a.ntri, TILE * TILE / 256), dim3(256), 0, st, a);
    MDG_LAUNCH_CHECK();
  }
  return MDG_OK;
}

// split factor of a secondary problem inside a fused launch: units of about 4096 tokens pack the free slots finely
static int multi_ksplit(int64_t n_tokens, int64_t* tokens_per_split) {
  int64_t ks = n_tokens / 4096;
  if (ks < 1) ks = 1;
  if (ks > 64) ks = 64;
  int64_t tps = (int64_t)align_up((size_t)ceil_div(n_tokens, ks), BK);
  *tokens_per_split = tps;
  return (int)ceil_div(n_tokens, tps);
}

static int multi_plan(int n, const mdg_cov_problem* pr, int dtype, CovMulti* m, size_t* ws_need) {
  MDG_CHECK_ARG(n >= 1 && n <= MULTI_MAX && pr, "mdg_cov_accum_multi: 1..%d problems", MULTI_MAX);
  MDG_CHECK_ARG(dtype >= MDG_BF16 && dtype <= MDG_F64, "mdg_cov_accum_multi: unknown dtype %d", dtype);
  const size_t esz = dtype_size(dtype);
  size_t off = 0;
  int64_t units = 0;
  m->n = n;
  for (int i = 0; i < n; i++) {
    const mdg_cov_problem& q = pr[i];
    MDG_CHECK_ARG(q.n_tokens > 0 && q.n_feat > 0 && q.batch > 0 && q.x && q.sigma, "mdg_cov_accum_multi: problem %d empty", i);
    MDG_CHECK_ARG(q.n_feat % TILE == 0 && q.ld >= q.n_feat * q.batch && q.ld_sigma >= q.n_feat,
                  "mdg_cov_accum_multi: problem %d needs n_feat %% 128 == 0 (got %lld) and consistent strides", i,
                  (long long)q.n_feat);
    MDG_CHECK_ARG((uintptr_t)q.x % 16 == 0 && (q.ld * esz) % 16 == 0,
                  "mdg_cov_accum_multi: problem %d is not 16-byte aligned (use mdg_cov_accum)", i);
    CovArgs& a = m->p[i];
    a.x = q.x; a.ld = q.ld; a.n_tokens = q.n_tokens; a.n_feat = (int)q.n_feat; a.batch = (int)q.batch;
    a.tiles = (int)(q.n_feat / TILE); a.ntri = a.tiles * (a.tiles + 1) / 2;
    a.sigma = q.sigma; a.ld_sigma = q.ld_sigma; a.sigma_bs = q.sigma_batch_stride; a.vec_ok = 1;
    a.gate = nullptr; a.gate_mask = a.gate_want = 0;
    const int64_t tiles = (int64_t)a.batch * a.ntri;
    if (i == 0 && tiles >= 768) { a.ksplit = 1; a.tokens_per_split = (int64_t)align_up((size_t)q.n_tokens, BK); }
    else a.ksplit = multi_ksplit(q.n_tokens, &a.tokens_per_split);
    a.partial = nullptr;
    m->unit_start[i] = (int)units;
    units += tiles * a.ksplit;
    MDG_CHECK_ARG(units < (1ll << 30), "mdg_cov_accum_multi: grid too large");
    if (a.ksplit > 1) {
      a.partial = (double*)off;  // offset for now, rebased on the workspace by the caller
      off += (size_t)a.ksplit * tiles * TILE * TILE * sizeof(double);
    }
  }
  m->unit_start[n] = (int)units;
  *ws_need = off;
  return MDG_OK;
}

extern "C" size_t mdg_cov_accum_multi_ws_bytes(int n, const mdg_cov_problem* problems, int dtype) {
  CovMulti m;
  size_t need = 0;
  if (multi_plan(n, problems, dtype, &m, &need) != MDG_OK) return 0;
  return need;
}

extern "C" int mdg_cov_accum_multi(int n, const mdg_cov_problem* problems, int dtype, void* ws, size_t ws_bytes,
                                   void* stream) {
  MDG_CLEAR();
  CovMulti m;
  size_t need = 0;
  MDG_TRY(multi_plan(n, problems, dtype, &m, &need));
  MDG_CHECK_ARG(need == 0 || (ws && ws_bytes >= need), "mdg_cov_accum_multi: workspace %zu < required %zu", ws_bytes, need);
  for (int i = 0; i < n; i++)
    if (m.p[i].ksplit > 1) m.p[i].partial = (double*)((char*)ws + (size_t)m.p[i].partial);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)m.unit_start[n]);
  switch (dtype) {
    case MDG_BF16: hipLaunchKernelGGL(cov_accum_multi_kernel<MDG_BF16>, grid, dim3(256), 0, st, m); break;
    case MDG_F16: hipLaunchKernelGGL(cov_accum_multi_kernel<MDG_F16>, grid, dim3(256), 0, st, m); break;
    case MDG_F32: hipLaunchKernelGGL(cov_accum_multi_kernel<MDG_F32>, grid, dim3(256), 0, st, m); break;
    default: hipLaunchKernelGGL(cov_accum_multi_kernel<MDG_F64>, grid, dim3(256), 0, st, m); break;
  }
  MDG_LAUNCH_CHECK();
  for (int i = 0; i < n; i++)
    if (m.p[i].ksplit > 1) {
      hipLaunchKernelGGL(cov_reduce_kernel, dim3(m.p[i].batch * m.p[i].ntri, TILE * TILE / 256), dim3(256), 0, st, m.p[i]);
      MDG_LAUNCH_CHECK();
    }
  return MDG_OK;
}

extern "C" int mdg_cov_finalize(double* sigma, int64_t n, int64_t batch, int64_t ld_sigma, int64_t sigma_bs,
                                double scale, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(sigma && n > 0 && batch > 0 && ld_sigma >= n, "mdg_cov_finalize: bad arguments");
  int tiles = (int)ceil_div(n, 32);
  int64_t ntri = (int64_t)tiles * (tiles + 1) / 2;
  MDG_CHECK_ARG(batch * ntri < (1ll << 31), "mdg_cov_finalize: grid too large");
  hipLaunchKernelGGL(cov_finalize_kernel, dim3((unsigned)(batch * ntri)), dim3(256), 0, (hipStream_t)stream, sigma,
                     (int)n, ld_sigma, sigma_bs, tiles, (int)ntri, scale);
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}

static const int BI_MAX_BLOCKS = 2048;
extern "C" size_t mdg_bi_ws_bytes(int64_t n_tokens) { return (size_t)BI_MAX_BLOCKS * sizeof(double); }

extern "C" int mdg_bi_accum(const void* x_in, const void* x_out, int dtype, int64_t n_tokens, int64_t d, int64_t ld,
                            double* out, void* ws, size_t ws_bytes, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(x_in && x_out && out && n_tokens >= 0 && d > 0 && ld >= d, "mdg_bi_accum: bad arguments");
  MDG_CHECK_ARG(ws && ws_bytes >= mdg_bi_ws_bytes(n_tokens), "mdg_bi_accum: workspace too small");
  if (n_tokens == 0) return MDG_OK;
  int blocks = (int)(ceil_div(n_tokens, 4) < BI_MAX_BLOCKS ? ceil_div(n_tokens, 4) : BI_MAX_BLOCKS);
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)ws;
  switch (dtype) {
    case MDG_BF16: hipLaunchKernelGGL(bi_partial_kernel<MDG_BF16>, dim3(blocks), dim3(256), 0, st, x_in, x_out, n_tokens, d, ld, partial); break;
    case MDG_F16: hipLaunchKernelGGL(bi_partial_kernel<MDG_F16>, dim3(blocks), dim3(256), 0, st, x_in, x_out, n_tokens, d, ld, partial); break;
    case MDG_F32: hipLaunchKernelGGL(bi_partial_kernel<MDG_F32>, dim3(blocks), dim3(256), 0, st, x_in, x_out, n_tokens, d, ld, partial); break;
    case MDG_F64: hipLaunchKernelGGL(bi_partial_kernel<MDG_F64>, dim3(blocks), dim3(256), 0, st, x_in, x_out, n_tokens, d, ld, partial); break;
    default: MDG_CHECK_ARG(false, "mdg_bi_accum: unknown dtype %d", dtype);
  }
  MDG_LAUNCH_CHECK();
  hipLaunchKernelGGL(bi_final_kernel, dim3(1), dim3(64), 0, st, partial, blocks, out);
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}
