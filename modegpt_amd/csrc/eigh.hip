// Batched symmetric eigensolver for head-sized matrices (n <= 128): two-sided cyclic Jacobi with the
// round-robin parallel ordering, the matrix resident in one workgroup's LDS.  It serves every eigen/SVD need
// of the path after the Gram reformulation (DESIGN.md "Identities"): VO's thin SVD (compress_vo.py:130,187,194)
// and sqrt_M at head size (compression_utils.py:21).  The north_star keeps the eigensolve off the MFMA; this is
// a latency/LDS kernel.
#include "common.hpp"

namespace mdg {

constexpr int JN = 128;      // max n
constexpr int JP = JN + 1;   // LDS pitch
constexpr int JT = 512;      // threads

// A [batch][n][n] (lower triangle read, buffer then reused as V^T scratch), evals desc, evecs columns.
__global__ __launch_bounds__(JT) void syevj_kernel(double* Ag, int n, double* evals, double* evecs, int* info,
                                                   int max_sweeps) {
  __shared__ double a[JN * JP];
  __shared__ double cs_c[JN / 2], cs_s[JN / 2];
  __shared__ int pr[JN / 2], qr[JN / 2];
  __shared__ int rotated;
  __shared__ int order[JN];
  const int tid = threadIdx.x;
  const int m = n / 2;
  double* Vt = Ag + (int64_t)blockIdx.x * n * n;
  double* ev = evals + (int64_t)blockIdx.x * n;
  double* ec = evecs + (int64_t)blockIdx.x * n * n;

  for (int e = tid; e < n * n; e += JT) {
    int i = e / n, j = e % n;
    a[i * JP + j] = (j <= i) ? Vt[i * n + j] : Vt[j * n + i];
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += JT) Vt[e] = (e / n == e % n) ? 1. : 0.;
  __syncthreads();

  const double eps = 2.220446049250313e-16;
  int p = 0, q = 0;
  bool mine_rot = false;
  int sweep = 0;
  bool converged = false;
  for (; sweep < max_sweeps && !converged; sweep++) {
    if (tid == 0) rotated = 0;
    __syncthreads();
    for (int r = 0; r < n - 1; r++) {
      if (tid < m) {
        if (mine_rot) {  // finish the previous round's pair
          a[p * JP + q] = 0.;
          a[q * JP + p] = 0.;
        }
        if (tid == 0) {
          p = n - 1;
          q = r;
        } else {
          p = (r + tid) % (n - 1);
          q = (r - tid + (n - 1)) % (n - 1);
        }
        const double app = a[p * JP + p], aqq = a[q * JP + q], apq = a[p * JP + q];
        double c = 1., s = 0.;
        mine_rot = fabs(apq) > eps * sqrt(fabs(app * aqq));
        if (mine_rot) {
          const double tau = (aqq - app) / (2. * apq);
          const double t = (tau >= 0. ? 1. : -1.) / (fabs(tau) + sqrt(1. + tau * tau));
          c = 1. / sqrt(1. + t * t);
          s = t * c;
          rotated = 1;
        }
        pr[tid] = p;
        qr[tid] = q;
        cs_c[tid] = c;
        cs_s[tid] = s;
      }
      __syncthreads();
      // A <- A J (columns) and V <- V J (rows of V^T)
      for (int e = tid; e < m * n; e += JT) {
        const int k = e / n, i = e % n;
        const double s = cs_s[k];
        if (s == 0.) continue;
        const double c = cs_c[k];
        const int pp = pr[k], qq = qr[k];
        const double x = a[i * JP + pp], y = a[i * JP + qq];
        a[i * JP + pp] = c * x - s * y;
        a[i * JP + qq] = s * x + c * y;
        const double vx = Vt[pp * n + i], vy = Vt[qq * n + i];
        Vt[pp * n + i] = c * vx - s * vy;
        Vt[qq * n + i] = s * vx + c * vy;
      }
      __syncthreads();
      // A <- J^T A (rows)
      for (int e = tid; e < m * n; e += JT) {
        const int k = e / n, j = e % n;
        const double s = cs_s[k];
        if (s == 0.) continue;
        const double c = cs_c[k];
        const int pp = pr[k], qq = qr[k];
        const double x = a[pp * JP + j], y = a[qq * JP + j];
        a[pp * JP + j] = c * x - s * y;
        a[qq * JP + j] = s * x + c * y;
      }
      __syncthreads();
    }
    converged = (rotated == 0);
    __syncthreads();
  }
  if (tid < m && mine_rot) {
    a[p * JP + q] = 0.;
    a[q * JP + p] = 0.;
  }
  if (tid == 0 && !converged) atomicExch(info, 1);
  __syncthreads();
  // descending order, ties by index
  if (tid < n) {
    const double mine = a[tid * JP + tid];
    int pos = 0;
    for (int k = 0; k < n; k++) {
      const double o = a[k * JP + k];
      pos += (o > mine || (o == mine && k < tid)) ? 1 : 0;
    }
    order[pos] = tid;
    ev[pos] = mine;
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += JT) {
    const int i = e / n, pos = e % n;
    ec[i * n + pos] = Vt[order[pos] * n + i];
  }
}

// root = V f(lambda) V^T, inv_root = V g(lambda) V^T  (compression_utils.py:35-36,47-55)
__global__ __launch_bounds__(256) void sqrt_rebuild_kernel(const double* evals, const double* evecs, int n, double ridge,
                                                           int scaled, double* root, double* inv_root) {
  __shared__ double f[JN], g[JN];
  const int b = blockIdx.x, tid = threadIdx.x;
  const double* lam = evals + (int64_t)b * n;
  const double* V = evecs + (int64_t)b * n * n;
  if (tid < n) {
    const double scale = scaled ? lam[0] : 1.0;  // evals are descending: lam[0] is the max
    const double l = lam[tid] + ridge * scale;
    const double rt = sqrt(l > 0. ? l : 0.);
    f[tid] = rt;
    g[tid] = 1.0 / (rt > 1e-12 ? rt : 1e-12);
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e % n;
    double s = 0., si = 0.;
    for (int k = 0; k < n; k++) {
      const double vv = V[i * n + k] * V[j * n + k];
      s += vv * f[k];
      si += vv * g[k];
    }
    root[(int64_t)b * n * n + e] = s;
    if (inv_root) inv_root[(int64_t)b * n * n + e] = si;
  }
}

int syevj_batched(double* A, int64_t n, int64_t batch, double* evals, double* evecs, int* dflag, hipStream_t st) {
  MDG_CHECK_ARG(n >= 2 && n <= JN && n % 2 == 0, "syevj: n=%lld must be even and in [2, 128]", (long long)n);
  MDG_CHECK_ARG(batch > 0 && batch < (1ll << 31), "syevj: bad batch");
  MDG_HIP(hipMemsetAsync(dflag, 0, sizeof(int), st));
  hipLaunchKernelGGL(syevj_kernel, dim3((unsigned)batch), dim3(JT), 0, st, A, (int)n, evals, evecs, dflag, 40);
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}

int check_flag(int* dflag, hipStream_t st, const char* what) {
  int flag = 0;
  MDG_HIP(hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, st));
  MDG_HIP(hipStreamSynchronize(st));
  if (flag != 0) {
    set_error("%s: Jacobi eigensolver did not converge in 40 sweeps", what);
    return MDG_ERR_NO_CONVERGE;
  }
  return MDG_OK;
}

}  // namespace mdg

using namespace mdg;

extern "C" int mdg_syevj_batched(double* A, int64_t n, int64_t batch, double* evals, double* evecs, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(A && evals && evecs, "mdg_syevj_batched: null pointer");
  hipStream_t st = (hipStream_t)stream;
  // standalone entry: a 4-byte stream-ordered scratch for the convergence flag (internal callers pass workspace)
  int* dflag = nullptr;
  MDG_HIP(hipMallocAsync((void**)&dflag, sizeof(int), st));
  int rc = syevj_batched(A, n, batch, evals, evecs, dflag, st);
  if (rc == MDG_OK) rc = check_flag(dflag, st, "mdg_syevj_batched");
  (void)hipFreeAsync(dflag, st);
  return rc;
}

extern "C" size_t mdg_sqrt_psd_small_ws_bytes(int64_t n, int64_t batch) {
  return (size_t)batch * (2 * n * n + n) * sizeof(double) + 64;
}

extern "C" int mdg_sqrt_psd_small(const double* M, int64_t n, int64_t batch, double ridge, int scaled, double* root,
                                  double* inv_root, double* evals_out, void* ws, size_t ws_bytes, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(M && root && n > 0 && batch > 0, "mdg_sqrt_psd_small: bad arguments");
  MDG_CHECK_ARG(ws && ws_bytes >= mdg_sqrt_psd_small_ws_bytes(n, batch), "mdg_sqrt_psd_small: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* Acopy = (double*)ws;
  double* evecs = Acopy + batch * n * n;
  double* evals = evecs + batch * n * n;
  int* dflag = (int*)(evals + batch * n);
  MDG_HIP(hipMemcpyAsync(Acopy, M, (size_t)batch * n * n * sizeof(double), hipMemcpyDeviceToDevice, st));
  MDG_TRY(syevj_batched(Acopy, n, batch, evals, evecs, dflag, st));
  hipLaunchKernelGGL(sqrt_rebuild_kernel, dim3((unsigned)batch), dim3(256), 0, st, evals, evecs, (int)n, ridge, scaled,
                     root, inv_root);
  MDG_LAUNCH_CHECK();
  if (evals_out)
    MDG_HIP(hipMemcpyAsync(evals_out, evals, (size_t)batch * n * sizeof(double), hipMemcpyDeviceToDevice, st));
  return check_flag(dflag, st, "mdg_sqrt_psd_small");
}
