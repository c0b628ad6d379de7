// Nystrom refit of down_proj (compress_mlp.py:52-62,97):
//   W_d' = (C[idx,idx] + eps I)^-1  C[idx,:]  W_d^T      -> stored transposed as [d, r] bf16
// gather -> blocked Cholesky -> gather-GEMM cross term -> blocked substitution -> transpose-cast.
#include "common.hpp"

namespace mdg {
int gemm_f64(int64_t M, int64_t N, int64_t K, double alpha, const void* A, int a_dtype, int64_t sa_i, int64_t sa_k,
             const int64_t* a_rows, const void* B, int b_dtype, int64_t sb_k, int64_t sb_j, double beta, void* C,
             int c_dtype, int64_t ldc, int64_t batch, int64_t a_bs, int64_t b_bs, int64_t c_bs, int flags,
             hipStream_t st);
int potrf_lower(double* A, int64_t n, int64_t lda, double* inv_diag, hipStream_t st);
int potrs_lower(const double* L, int64_t n, int64_t ldl, const double* inv_diag, double* X, int64_t nrhs, int64_t ldx,
                double* ws, hipStream_t st);
size_t potrs_ws_elems(int64_t n, int64_t nrhs);
int copy_lower(const double* src, int64_t ld_src, const int64_t* idx, double* dst, int64_t ldd, int64_t n,
               double ridge, hipStream_t st);
}  // namespace mdg

using namespace mdg;

// Row pitch of C_kk in the workspace: r rounded up to 16 doubles.  With pitch r an odd rank (10035 of 14336 at 30 %) leaves every
// second row off a 16-byte boundary, and every GEMM of the factorisation and the substitution on the element-wise staging path --
// potrf_lower 15.1 -> 14.3 ms, potrs_lower 17.8 -> 16.9 ms at r = 10035 (scripts/probes/decomp_phases.py).
static int64_t ckk_pitch(int64_t r) { return (r + 15) / 16 * 16; }
extern "C" size_t mdg_nystrom_down_ws_bytes(int64_t n, int64_t r, int64_t d) {
  return ((size_t)r * ckk_pitch(r) + mdg_potrf_inv_diag_elems(r) + (size_t)r * d + potrs_ws_elems(r, d)) * sizeof(double);
}

static int nystrom_down(const double* C, int64_t n, int64_t ldc, const int64_t* idx, int64_t r, const void* Wd, int64_t d, int64_t ld_wd,
                        int w_dtype, double eps, void* down_out, int64_t ld_out, double* down_f64, void* ws, size_t ws_bytes,
                        void* side_stream, void* ev_fork, void* ev_join, void* stream);

extern "C" int mdg_nystrom_down(const double* C, int64_t n, int64_t ldc, const int64_t* idx, int64_t r, const void* Wd,
                                int64_t d, int64_t ld_wd, int w_dtype, double eps, void* down_out, int64_t ld_out,
                                double* down_f64, void* ws, size_t ws_bytes, void* stream) {
  return nystrom_down(C, n, ldc, idx, r, Wd, d, ld_wd, w_dtype, eps, down_out, ld_out, down_f64, ws, ws_bytes, nullptr, nullptr, nullptr, stream);
}

extern "C" int mdg_nystrom_down_overlapped(const double* C, int64_t n, int64_t ldc, const int64_t* idx, int64_t r, const void* Wd,
                                           int64_t d, int64_t ld_wd, int w_dtype, double eps, void* down_out, int64_t ld_out,
                                           double* down_f64, void* ws, size_t ws_bytes, void* side_stream, void* ev_fork, void* ev_join,
                                           void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(side_stream && ev_fork && ev_join && side_stream != stream,
                "mdg_nystrom_down_overlapped: needs a second stream and two events of the caller's");
  return nystrom_down(C, n, ldc, idx, r, Wd, d, ld_wd, w_dtype, eps, down_out, ld_out, down_f64, ws, ws_bytes, side_stream, ev_fork, ev_join, stream);
}

static int nystrom_down(const double* C, int64_t n, int64_t ldc, const int64_t* idx, int64_t r, const void* Wd, int64_t d, int64_t ld_wd,
                        int w_dtype, double eps, void* down_out, int64_t ld_out, double* down_f64, void* ws, size_t ws_bytes,
                        void* side_stream, void* ev_fork, void* ev_join, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(C && idx && Wd && down_out, "mdg_nystrom_down: null pointer");
  MDG_CHECK_ARG(w_dtype == MDG_BF16 || w_dtype == MDG_F64, "mdg_nystrom_down: W_d must be bf16 or f64 (got %d)", w_dtype);
  MDG_CHECK_ARG(n > 0 && r > 0 && r <= n && d > 0 && ldc >= n && ld_wd >= n && ld_out >= r,
                "mdg_nystrom_down: bad sizes (n=%lld r=%lld d=%lld)", (long long)n, (long long)r, (long long)d);
  MDG_CHECK_ARG(ws && ws_bytes >= mdg_nystrom_down_ws_bytes(n, r, d), "mdg_nystrom_down: workspace %zu < required %zu",
                ws_bytes, mdg_nystrom_down_ws_bytes(n, r, d));
  hipStream_t st = (hipStream_t)stream;
  double* Ckk = (double*)ws;
  const int64_t ldk = ckk_pitch(r);
  double* inv = Ckk + (size_t)r * ldk;
  double* X = inv + mdg_potrf_inv_diag_elems(r);
  double* solve_ws = X + (size_t)r * d;
  // C_kk + eps I  (lower)                                           compress_mlp.py:52,56
  MDG_TRY(copy_lower(C, ldc, idx, Ckk, ldk, r, eps, st));
  // cross = C[idx,:] @ W_d^T  -> [r, d]                             compress_mlp.py:54
  // (with a second stream: beside the factorisation of C_kk, which does not need it -- the chain of 79 diagonal-block steps
  // leaves most of the chip idle between its GEMMs, the 1.2 TFLOP product fills it: 36 -> 27 ms for the two)
  hipStream_t cross_st = st;
  if (side_stream) {
    cross_st = (hipStream_t)side_stream;
    MDG_HIP(hipEventRecord((hipEvent_t)ev_fork, st));
    MDG_HIP(hipStreamWaitEvent(cross_st, (hipEvent_t)ev_fork, 0));
  }
  MDG_TRY(gemm_f64(r, d, n, 1.0, C, MDG_F64, ldc, 1, idx, Wd, w_dtype, 1, ld_wd, 0.0, X, MDG_F64, d, 1, 0, 0, 0, 0, cross_st));
  if (side_stream) MDG_HIP(hipEventRecord((hipEvent_t)ev_join, cross_st));
  const int rc_potrf = potrf_lower(Ckk, r, ldk, inv, st);         // compress_mlp.py:56
  if (side_stream) MDG_HIP(hipStreamWaitEvent(st, (hipEvent_t)ev_join, 0));   // (also on failure: the workspace is the caller's to free)
  if (rc_potrf != MDG_OK) return rc_potrf;
  MDG_TRY(potrs_lower(Ckk, r, ldk, inv, X, d, d, solve_ws, st));  // compress_mlp.py:57
  if (down_f64) MDG_HIP(hipMemcpyAsync(down_f64, X, (size_t)r * d * sizeof(double), hipMemcpyDeviceToDevice, st));
  // [r, d] fp64 -> [d, r] bf16                                      compress_mlp.py:61,97
  return mdg_cast_transpose_f64_bf16(X, r, d, d, down_out, ld_out, stream);
}
