// Index work of the path: top-k selection with torch's ordering contracts, row gathers, the QK CR scores,
// and the fp64 -> bf16 transpose-cast.  All HBM-bound byte/index kernels: coalesced rows, no MFMA.
#include "common.hpp"

namespace mdg {

// total order used for ranking: a before b  <=>  a < b, or equal and lower index; NaN is largest.
__device__ __forceinline__ bool before_asc(double a, int64_t ia, double b, int64_t ib) {
  const bool an = a != a, bn = b != b;
  if (an || bn) return (!an && bn) || (an && bn && ia < ib);
  return a < b || (a == b && ia < ib);
}

// Thread i ranks score i among all n (n^2 compares, trivially parallel); the unique element of rank k-1 is the
// selection threshold and its index is parked in idx[0] for compact_kernel.
__global__ __launch_bounds__(256) void rank_threshold_kernel(const double* s, int64_t n, int64_t k, int64_t* idx) {
  __shared__ double tile[1024];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const double mine = i < n ? s[i] : 0.;
  int64_t rank = 0;
  for (int64_t j0 = 0; j0 < n; j0 += 1024) {
    for (int e = threadIdx.x; e < 1024; e += 256) tile[e] = (j0 + e < n) ? s[j0 + e] : 0.;
    __syncthreads();
    const int lim = (int)(n - j0 < 1024 ? n - j0 : 1024);
    if (i < n)
      for (int e = 0; e < lim; e++) rank += before_asc(tile[e], j0 + e, mine, i) ? 1 : 0;
    __syncthreads();
  }
  if (i < n && rank == k - 1) idx[0] = i;
}

// single workgroup: ordered compaction (ascending index) of everything not after the threshold.
__global__ __launch_bounds__(1024) void compact_kernel(const double* s, int64_t n, int64_t* idx) {
  __shared__ int sums[1024];
  const int tid = threadIdx.x;
  const int64_t it = idx[0];
  const double t = s[it];
  __syncthreads();  // everyone holds the threshold before idx[0] is overwritten
  const int64_t per = (n + 1023) / 1024;
  const int64_t b = tid * per, e = (b + per < n) ? b + per : n;
  int c = 0;
  for (int64_t i = b; i < e; i++) c += (i == it || before_asc(s[i], i, t, it)) ? 1 : 0;
  sums[tid] = c;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
    int v = tid >= o ? sums[tid - o] : 0;
    __syncthreads();
    sums[tid] += v;
    __syncthreads();
  }
  int64_t pos = sums[tid] - c;
  for (int64_t i = b; i < e; i++)
    if (i == it || before_asc(s[i], i, t, it)) idx[pos++] = i;
}

// The certificate of a selection of the k smallest scores (idx: the selected indices, ascending) against a perturbation of every
// score by at most eps * sens[j]: one workgroup.  out[0] = largest selected score, out[1] = smallest unselected score,
// out[2] = max over selected (s + eps sens), out[3] = min over unselected (s - eps sens), out[4] / out[5] = largest sens among the
// selected / the unselected, out[6] = number of scores whose interval [s - eps sens, s + eps sens] reaches across the midpoint of
// out[0] and out[1], out[7] = 1 if out[2] < out[3] (no perturbation within the bound can change the selected SET) else 0.
// NaN scores rank last (mdg_select_smallest_sorted) and take no part.  k == 0 or k == n: nothing to separate, certified.
__global__ __launch_bounds__(1024) void select_margin_kernel(const double* s, const double* sens, const int64_t* idx, int64_t n, int64_t k,
                                                             double eps, double* out) {
  __shared__ double red[6][1024];
  __shared__ int cnt[1024];
  const int tid = threadIdx.x;
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  double sel_max = -inf, unsel_min = inf, hi = -inf, lo = inf, b_sel = 0., b_unsel = 0.;
  for (int64_t j = tid; j < n; j += 1024) {
    const double v = s[j];
    if (v != v) continue;
    int64_t a = 0, b = k;                       // is j among idx[0 .. k)?  (sorted ascending: binary search)
    while (a < b) {
      const int64_t m = (a + b) >> 1;
      if (idx[m] < j) a = m + 1; else b = m;
    }
    const bool selected = a < k && idx[a] == j;
    const double w = eps * sens[j];
    if (selected) { sel_max = fmax(sel_max, v); hi = fmax(hi, v + w); b_sel = fmax(b_sel, sens[j]); }
    else { unsel_min = fmin(unsel_min, v); lo = fmin(lo, v - w); b_unsel = fmax(b_unsel, sens[j]); }
  }
  red[0][tid] = sel_max; red[1][tid] = unsel_min; red[2][tid] = hi; red[3][tid] = lo; red[4][tid] = b_sel; red[5][tid] = b_unsel;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) {
      red[0][tid] = fmax(red[0][tid], red[0][tid + o]); red[1][tid] = fmin(red[1][tid], red[1][tid + o]);
      red[2][tid] = fmax(red[2][tid], red[2][tid + o]); red[3][tid] = fmin(red[3][tid], red[3][tid + o]);
      red[4][tid] = fmax(red[4][tid], red[4][tid + o]); red[5][tid] = fmax(red[5][tid], red[5][tid + o]);
    }
    __syncthreads();
  }
  const double mid = 0.5 * (red[0][0] + red[1][0]);
  int c = 0;
  if (red[0][0] > -inf && red[1][0] < inf)
    for (int64_t j = tid; j < n; j += 1024) {
      const double v = s[j], w = eps * sens[j];
      if (v == v && v - w <= mid && mid <= v + w) c++;
    }
  cnt[tid] = c;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) cnt[tid] += cnt[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    for (int i = 0; i < 6; i++) out[i] = red[i][0];
    out[6] = (double)cnt[0];
    out[7] = red[2][0] < red[3][0] ? 1. : 0.;
  }
}

__global__ __launch_bounds__(256) void gather_rows16_kernel(const unsigned short* src, int64_t ld_src,
                                                            const int64_t* rows, int64_t n_cols, unsigned short* out,
                                                            int64_t ld_out, int vec_ok) {
  const int64_t r = blockIdx.x;
  const unsigned short* s = src + rows[r] * ld_src;
  unsigned short* d = out + r * ld_out;
  if (vec_ok) {
    const int64_t nv = n_cols / 8;
    for (int64_t c = threadIdx.x; c < nv; c += 256) ((uint4*)d)[c] = ((const uint4*)s)[c];
  } else {
    for (int64_t c = threadIdx.x; c < n_cols; c += 256) d[c] = s[c];
  }
}

// One workgroup per kv head.  Scores from the diagonal identity, then rank-by-count for a score-descending,
// lower-index-first order (torch.topk).
__global__ __launch_bounds__(128) void qk_select_kernel(const double* cov_q, const double* cov_k, int n_heads, int n_kv,
                                                        int hd, double ridge_q, double ridge_k, int rank, int mode,
                                                        int64_t* mask, int64_t* q_rows, int64_t* k_rows) {
  __shared__ double score[128];
  __shared__ int sel[128];  // sel[pos] = index at descending position pos
  const int h = blockIdx.x, tid = threadIdx.x;
  const int g = n_heads / n_kv;
  const int ns = (mode == MDG_QK_OPT) ? hd : hd / 2;   // number of scored units
  const int take = (mode == MDG_QK_OPT) ? rank : rank / 2;
  const double* Ck = cov_k + (int64_t)h * hd * hd;
  if (tid < ns) {
    double sc;
    if (mode == MDG_QK_OPT) {
      const double* Cq = cov_q + (int64_t)h * hd * hd;
      double nq2 = fmax(Cq[tid * hd + tid] + ridge_q, 0.), nk2 = fmax(Ck[tid * hd + tid] + ridge_k, 0.);
      sc = sqrt(nq2) * sqrt(nk2);  // compress_qk.py:458-461
    } else {
      const int j1 = tid, j2 = tid + hd / 2;
      double nk1 = fmax(Ck[j1 * hd + j1] + ridge_k, 0.), nk2 = fmax(Ck[j2 * hd + j2] + ridge_k, 0.);
      double acc = 0.;
      for (int q = 0; q < g; q++) {
        const double* Cq = cov_q + (int64_t)(h * g + q) * hd * hd;
        double nq1 = fmax(Cq[j1 * hd + j1] + ridge_q, 0.), nq2 = fmax(Cq[j2 * hd + j2] + ridge_q, 0.);
        acc += nq1 * nk1 + nq2 * nk2;  // compress_qk.py:360-362 / :414-416
      }
      sc = (mode == MDG_QK_ROPE_GROUPED) ? sqrt(acc) : acc;  // :364
    }
    score[tid] = sc;
  }
  __syncthreads();
  if (tid < ns) {
    const double mine = score[tid];
    int pos = 0;
    for (int j = 0; j < ns; j++) {
      const double o = score[j];
      // descending; NaN first (torch treats NaN as the largest); ties -> lower index first
      const bool on = o != o, mn = mine != mine;
      bool ahead;
      if (on || mn) ahead = (on && !mn) || (on && mn && j < tid);
      else ahead = o > mine || (o == mine && j < tid);
      pos += ahead ? 1 : 0;
    }
    sel[pos] = tid;
  }
  __syncthreads();
  if (tid < take) {
    const int j = sel[tid];
    int64_t* m = mask + (int64_t)h * rank;
    if (mode == MDG_QK_OPT) {
      m[tid] = j;
    } else {
      m[tid] = j;
      m[tid + take] = j + hd / 2;  // torch.cat((topk, topk + hd/2)), compress_qk.py:367
    }
  }
  __syncthreads();
  // absolute row lists for the gathers (compress_qk.py:375-380: all query heads of the group, then K)
  for (int e = tid; e < rank; e += 128) {
    const int64_t j = mask[(int64_t)h * rank + e];
    k_rows[(int64_t)h * rank + e] = (int64_t)h * hd + j;
    for (int q = 0; q < g; q++) q_rows[(int64_t)(h * g + q) * rank + e] = (int64_t)(h * g + q) * hd + j;
  }
}

// out[j][i] = bf16(in[i][j]) through a 32x32 LDS tile.
__global__ __launch_bounds__(256) void cast_transpose_kernel(const double* in, int64_t rows, int64_t cols,
                                                             int64_t ld_in, bf16_t* out, int64_t ld_out) {
  __shared__ double t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  for (int r = ty; r < 32; r += 8)
    t[r][tx] = (r0 + r < rows && c0 + tx < cols) ? in[(r0 + r) * ld_in + c0 + tx] : 0.;
  __syncthreads();
  for (int c = ty; c < 32; c += 8)
    if (c0 + c < cols && r0 + tx < rows) out[(c0 + c) * ld_out + r0 + tx] = f64_to_bf16(t[tx][c]);
}

}  // namespace mdg

using namespace mdg;

extern "C" int mdg_select_smallest_sorted(const double* scores, int64_t n, int64_t k, int64_t* idx, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(scores && idx && n > 0 && k >= 0 && k <= n, "mdg_select_smallest_sorted: bad arguments (n=%lld k=%lld)",
                (long long)n, (long long)k);
  if (k == 0) return MDG_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rank_threshold_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, scores, n, k, idx);
  MDG_LAUNCH_CHECK();
  hipLaunchKernelGGL(compact_kernel, dim3(1), dim3(1024), 0, st, scores, n, idx);
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}

extern "C" int mdg_select_margin(const double* scores, const double* sens, const int64_t* idx, int64_t n, int64_t k, double eps,
                                 double* out8, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(scores && sens && out8 && n > 0 && k >= 0 && k <= n && (idx || k == 0) && eps >= 0., "mdg_select_margin: bad arguments");
  hipLaunchKernelGGL(select_margin_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, scores, sens, idx, n, k, eps, out8);
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}

extern "C" int mdg_gather_rows_16(const void* src, int64_t ld_src, const int64_t* rows, int64_t n_rows, int64_t n_cols,
                                  void* out, int64_t ld_out, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(n_rows >= 0 && n_cols > 0 && ld_src >= n_cols && ld_out >= n_cols, "mdg_gather_rows_16: bad sizes");
  if (n_rows == 0) return MDG_OK;
  MDG_CHECK_ARG(src && rows && out, "mdg_gather_rows_16: null pointer");
  MDG_CHECK_ARG(n_rows < (1ll << 31), "mdg_gather_rows_16: too many rows");
  int vec_ok = ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0) && (ld_src % 8 == 0) && (ld_out % 8 == 0) &&
               (n_cols % 8 == 0);
  hipLaunchKernelGGL(gather_rows16_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)src, ld_src, rows, n_cols, (unsigned short*)out, ld_out, vec_ok);
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}

extern "C" int mdg_qk_select(const double* cov_q, const double* cov_k, int n_heads, int n_kv, int hd, double ridge_q,
                             double ridge_k, int rank, int mode, int64_t* mask, int64_t* q_rows, int64_t* k_rows,
                             void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(cov_q && cov_k && mask && q_rows && k_rows, "mdg_qk_select: null pointer");
  MDG_CHECK_ARG(n_kv > 0 && n_heads >= n_kv && n_heads % n_kv == 0, "mdg_qk_select: n_heads %d not a multiple of n_kv %d",
                n_heads, n_kv);
  MDG_CHECK_ARG(hd > 0 && hd <= 256 && rank >= 1 && rank <= hd, "mdg_qk_select: bad head_dim %d / rank %d", hd, rank);
  MDG_CHECK_ARG(mode >= MDG_QK_ROPE_GROUPED && mode <= MDG_QK_OPT, "mdg_qk_select: unknown mode %d", mode);
  if (mode != MDG_QK_OPT) {
    MDG_CHECK_ARG(hd % 2 == 0 && rank % 2 == 0, "mdg_qk_select: RoPE modes need even head_dim and rank");
    MDG_CHECK_ARG(hd / 2 <= 128, "mdg_qk_select: head_dim %d too large", hd);
  } else {
    MDG_CHECK_ARG(hd <= 128 && n_heads == n_kv, "mdg_qk_select: OPT mode needs head_dim <= 128 and n_heads == n_kv");
  }
  if (mode == MDG_QK_ROPE_MHA) MDG_CHECK_ARG(n_heads == n_kv, "mdg_qk_select: ROPE_MHA needs n_heads == n_kv");
  hipLaunchKernelGGL(qk_select_kernel, dim3(n_kv), dim3(128), 0, (hipStream_t)stream, cov_q, cov_k, n_heads, n_kv, hd,
                     ridge_q, ridge_k, rank, mode, mask, q_rows, k_rows);
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}

extern "C" int mdg_cast_transpose_f64_bf16(const double* in, int64_t rows, int64_t cols, int64_t ld_in, void* out,
                                           int64_t ld_out, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(in && out && rows > 0 && cols > 0 && ld_in >= cols && ld_out >= rows, "mdg_cast_transpose: bad arguments");
  dim3 grid((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32));
  MDG_CHECK_ARG(grid.y < 65536, "mdg_cast_transpose: too many rows");
  hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, rows, cols, ld_in,
                     (bf16_t*)out, ld_out);
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}
