// VO compression (compress_vo.py:14-223) on the Gram matrix.
//
// Reference: A_h = sqrt(C) W_v,h^T = U S V^T (thin SVD), W_v' = (C^-1/2 U_r)^T, W_o'[j] = (S_r V_r^T W_o,j^T)^T.
// Identity used here (C := Sigma_x + rho I, symmetric PD so sqrt(C)^2 = C):
//   G_h := A_h^T A_h = W_v,h C W_v,h^T = V S^2 V^T          (128 x 128, batched Jacobi)
//   C^-1/2 U_r = C^-1/2 A_h V_r S_r^-1 = W_v,h^T V_r S_r^-1  => W_v' = S_r^-1 V_r^T W_v,h
//   W_o'[j]   = W_o,j V_r S_r
// so the d x d eigensolve and LU inverse of compress_vo.py:43-45 never happen.  The MHA variant (two SVDs,
// :162-223) reduces the same way: B = S V^T W_o,h^T, B B^T = (S V^T)(W_o,h^T W_o,h)(V S) = U_p S_p^2 U_p^T,
// W_v' = U_p,r^T S^-1 V^T W_v,h,  W_o' = W_o,h V S U_p,r.
#include "common.hpp"

namespace mdg {
int gemm_f64(int64_t M, int64_t N, int64_t K, double alpha, const void* A, int a_dtype, int64_t sa_i, int64_t sa_k,
             const int64_t* a_rows, const void* B, int b_dtype, int64_t sb_k, int64_t sb_j, double beta, void* C,
             int c_dtype, int64_t ldc, int64_t batch, int64_t a_bs, int64_t b_bs, int64_t c_bs, int flags,
             hipStream_t st);
int syevj_batched(double* A, int64_t n, int64_t batch, double* evals, double* evecs, int* dflag, hipStream_t st,
                  int max_sweeps = 40, int sorted = 1);
int check_flag(int* dflag, hipStream_t st, const char* what);

// dst (f64, ld) = scale * src (bf16 or f64, ld_src)
template <int DT>
__global__ __launch_bounds__(256) void scale_to_f64_kernel(const void* src, int64_t ld_src, int64_t cols, double scale,
                                                           double* dst, int64_t ld) {
  const int64_t r = blockIdx.x;
  for (int64_t c = threadIdx.x; c < cols; c += 256) dst[r * ld + c] = scale * load_f64<DT>(src, r * ld_src + c);
}

// grouped: P[h][a][k] = V[k][a] / S_a, Q[h][k][a] = V[k][a] * S_a  (a < r)
__global__ __launch_bounds__(256) void vo_factors_grouped_kernel(const double* evals, const double* evecs, int hd, int r,
                                                                 double* P, double* Q) {
  const int h = blockIdx.x;
  const double* lam = evals + (int64_t)h * hd;
  const double* V = evecs + (int64_t)h * hd * hd;
  for (int e = threadIdx.x; e < r * hd; e += 256) {
    const int a = e / hd, k = e % hd;
    const double l = lam[a];
    const double s = sqrt(l > 0. ? l : 0.);
    const double v = V[k * hd + a];
    P[(int64_t)h * r * hd + a * hd + k] = s > 0. ? v / s : 0.;
    Q[(int64_t)h * hd * r + k * r + a] = v * s;
  }
}

// Y[h][a][k] = S_a V[k][a]   (all hd components)
__global__ __launch_bounds__(256) void vo_sv_kernel(const double* evals, const double* evecs, int hd, double* Y) {
  const int h = blockIdx.x;
  const double* lam = evals + (int64_t)h * hd;
  const double* V = evecs + (int64_t)h * hd * hd;
  for (int e = threadIdx.x; e < hd * hd; e += 256) {
    const int a = e / hd, k = e % hd;
    const double l = lam[a];
    Y[(int64_t)h * hd * hd + e] = sqrt(l > 0. ? l : 0.) * V[k * hd + a];
  }
}

// MHA: P[h][a][k] = sum_b Up[b][a] / S_b * V[k][b],  Q[h][k][a] = sum_b V[k][b] * S_b * Up[b][a]   (a < r)
__global__ __launch_bounds__(256) void vo_factors_mha_kernel(const double* evals, const double* evecs, const double* up,
                                                             int hd, int r, double* P, double* Q) {
  __shared__ double s_[128], is_[128];
  const int h = blockIdx.x;
  const double* lam = evals + (int64_t)h * hd;
  const double* V = evecs + (int64_t)h * hd * hd;
  const double* U = up + (int64_t)h * hd * hd;
  if (threadIdx.x < hd) {
    const double l = lam[threadIdx.x];
    const double s = sqrt(l > 0. ? l : 0.);
    s_[threadIdx.x] = s;
    is_[threadIdx.x] = s > 0. ? 1. / s : 0.;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < r * hd; e += 256) {
    const int a = e / hd, k = e % hd;
    double p = 0., q = 0.;
    for (int b = 0; b < hd; b++) {
      const double u = U[b * hd + a], v = V[k * hd + b];
      p += u * is_[b] * v;
      q += v * s_[b] * u;
    }
    P[(int64_t)h * r * hd + a * hd + k] = p;
    Q[(int64_t)h * hd * r + k * r + a] = q;
  }
}

struct VoWs {
  double *T, *G, *evals, *evecs, *P, *Q, *Mh, *tmp, *Y, *evals2, *evecs2;
  int* flag;
  size_t bytes;
};

static VoWs vo_layout(void* ws, int64_t d, int n_heads, int n_kv, int hd) {
  VoWs w;
  double* p = (double*)ws;
  const size_t hh = (size_t)hd * hd;
  w.T = p;      p += (size_t)n_kv * hd * d;
  w.G = p;      p += n_kv * hh;
  w.evals = p;  p += (size_t)n_kv * hd;
  w.evecs = p;  p += n_kv * hh;
  w.P = p;      p += n_kv * hh;
  w.Q = p;      p += n_kv * hh;
  w.Mh = p;     p += n_kv * hh;
  w.tmp = p;    p += n_kv * hh;
  w.Y = p;      p += n_kv * hh;
  w.evals2 = p; p += (size_t)n_kv * hd;
  w.evecs2 = p; p += n_kv * hh;
  w.flag = (int*)p; p += 8;
  w.bytes = (size_t)((char*)p - (char*)ws);
  return w;
}

}  // namespace mdg

using namespace mdg;

extern "C" size_t mdg_vo_compress_ws_bytes(int64_t d, int n_heads, int n_kv, int hd) {
  return vo_layout(nullptr, d, n_heads, n_kv, hd).bytes;
}

extern "C" int mdg_vo_compress(const double* cov_x, int64_t d, int64_t ldc, const void* Wv, int64_t ld_wv, const void* Wo,
                               int64_t ld_wo, int w_dtype, int n_heads, int n_kv, int hd, int rank, double ridge, void* v_out,
                               int64_t ld_v, void* o_out, int64_t ld_o, double* v_f64, double* o_f64, void* ws,
                               size_t ws_bytes, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(cov_x && Wv && Wo && v_out && o_out, "mdg_vo_compress: null pointer");
  MDG_CHECK_ARG(w_dtype == MDG_BF16 || w_dtype == MDG_F64, "mdg_vo_compress: weights must be bf16 or f64 (got %d)", w_dtype);
  const int64_t wsz = w_dtype == MDG_BF16 ? 2 : 8;
  MDG_CHECK_ARG(n_kv > 0 && n_heads % n_kv == 0 && hd >= 2 && hd <= 128 && hd % 2 == 0,
                "mdg_vo_compress: unsupported head layout (n_heads=%d n_kv=%d hd=%d)", n_heads, n_kv, hd);
  MDG_CHECK_ARG(rank >= 1 && rank <= hd, "mdg_vo_compress: rank %d outside [1, %d]", rank, hd);
  MDG_CHECK_ARG(d > 0 && ldc >= d && ld_wv >= d && ld_wo >= (int64_t)n_heads * hd && ld_v >= d &&
                    ld_o >= (int64_t)n_heads * rank, "mdg_vo_compress: bad leading dimensions");
  MDG_CHECK_ARG(ws && ws_bytes >= mdg_vo_compress_ws_bytes(d, n_heads, n_kv, hd), "mdg_vo_compress: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  VoWs w = vo_layout(ws, d, n_heads, n_kv, hd);
  const int g = n_heads / n_kv;
  const bool mha = (g == 1);
  const int64_t rows = (int64_t)n_kv * hd, hh = (int64_t)hd * hd;
  // T = W_v (Sigma_x + rho I) = rho W_v + W_v Sigma_x                       [n_kv*hd, d]
  if (w_dtype == MDG_BF16)
    hipLaunchKernelGGL(scale_to_f64_kernel<MDG_BF16>, dim3((unsigned)rows), dim3(256), 0, st, Wv, ld_wv, d, ridge, w.T, d);
  else
    hipLaunchKernelGGL(scale_to_f64_kernel<MDG_F64>, dim3((unsigned)rows), dim3(256), 0, st, Wv, ld_wv, d, ridge, w.T, d);
  MDG_LAUNCH_CHECK();
  MDG_TRY(gemm_f64(rows, d, d, 1.0, Wv, w_dtype, ld_wv, 1, nullptr, cov_x, MDG_F64, ldc, 1, 1.0, w.T, MDG_F64, d, 1, 0, 0,
                   0, 0, st));
  // G_h = T_h W_v,h^T                                                        [hd, hd] per kv head
  MDG_TRY(gemm_f64(hd, hd, d, 1.0, w.T, MDG_F64, d, 1, nullptr, Wv, w_dtype, 1, ld_wv, 0.0, w.G, MDG_F64, hd, n_kv,
                   (int64_t)hd * d, (int64_t)hd * ld_wv, hh, 0, st));
  MDG_TRY(syevj_batched(w.G, hd, n_kv, w.evals, w.evecs, w.flag, st));
  if (!mha) {
    hipLaunchKernelGGL(vo_factors_grouped_kernel, dim3(n_kv), dim3(256), 0, st, w.evals, w.evecs, hd, rank, w.P, w.Q);
    MDG_LAUNCH_CHECK();
  } else {
    MDG_TRY(check_flag(w.flag, st, "mdg_vo_compress (first SVD)"));
    // M_h = W_o,h^T W_o,h
    MDG_TRY(gemm_f64(hd, hd, d, 1.0, Wo, w_dtype, 1, ld_wo, nullptr, Wo, w_dtype, ld_wo, 1, 0.0, w.Mh, MDG_F64, hd,
                     n_heads, hd, hd, hh, 0, st));
    hipLaunchKernelGGL(vo_sv_kernel, dim3(n_kv), dim3(256), 0, st, w.evals, w.evecs, hd, w.Y);
    MDG_LAUNCH_CHECK();
    // B B^T = Y M Y^T
    MDG_TRY(gemm_f64(hd, hd, hd, 1.0, w.Y, MDG_F64, hd, 1, nullptr, w.Mh, MDG_F64, hd, 1, 0.0, w.tmp, MDG_F64, hd, n_kv, hh,
                     hh, hh, 0, st));
    MDG_TRY(gemm_f64(hd, hd, hd, 1.0, w.tmp, MDG_F64, hd, 1, nullptr, w.Y, MDG_F64, 1, hd, 0.0, w.Mh, MDG_F64, hd, n_kv, hh,
                     hh, hh, 0, st));
    MDG_TRY(syevj_batched(w.Mh, hd, n_kv, w.evals2, w.evecs2, w.flag, st));
    hipLaunchKernelGGL(vo_factors_mha_kernel, dim3(n_kv), dim3(256), 0, st, w.evals, w.evecs, w.evecs2, hd, rank, w.P, w.Q);
    MDG_LAUNCH_CHECK();
  }
  // W_v'[h] = P_h W_v,h   -> rows [h*rank, (h+1)*rank) of v_out
  for (int pass = 0; pass < (v_f64 ? 2 : 1); pass++) {
    void* out = pass ? (void*)v_f64 : v_out;
    int64_t ld = pass ? d : ld_v;
    MDG_TRY(gemm_f64(rank, d, hd, 1.0, w.P, MDG_F64, hd, 1, nullptr, Wv, w_dtype, ld_wv, 1, 0.0, out,
                     pass ? MDG_F64 : MDG_BF16, ld, n_kv, (int64_t)rank * hd, (int64_t)hd * ld_wv, (int64_t)rank * ld, 0,
                     st));
  }
  // W_o'[h*g + j] = W_o[:, head] Q_h  -> columns [(h*g+j)*rank, +rank) of o_out
  for (int pass = 0; pass < (o_f64 ? 2 : 1); pass++) {
    for (int j = 0; j < g; j++) {
      char* out = pass ? (char*)(o_f64 + (int64_t)j * rank) : (char*)o_out + (int64_t)j * rank * 2;
      int64_t ld = pass ? (int64_t)n_heads * rank : ld_o;
      MDG_TRY(gemm_f64(d, rank, hd, 1.0, (const char*)Wo + (int64_t)j * hd * wsz, w_dtype, ld_wo, 1, nullptr, w.Q, MDG_F64,
                       rank, 1, 0.0, out, pass ? MDG_F64 : MDG_BF16, ld, n_kv, (int64_t)g * hd, (int64_t)hd * rank,
                       (int64_t)g * rank, 0, st));
    }
  }
  return check_flag(w.flag, st, "mdg_vo_compress");
}
