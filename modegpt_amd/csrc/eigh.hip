// Batched symmetric eigensolver for head-sized matrices (n <= 128): two-sided cyclic Jacobi with the
// round-robin parallel ordering, the matrix resident in one workgroup's LDS.  It serves every eigen/SVD need
// of the path after the Gram reformulation (DESIGN.md "Identities"): VO's thin SVD (compress_vo.py:130,187,194)
// and sqrt_M at head size (compression_utils.py:21).  The north_star keeps the eigensolve off the MFMA; this is
// a latency/LDS kernel.
#include <stdlib.h>

#include "common.hpp"

namespace mdg {

constexpr int JN = 128;      // max n
constexpr int JP = JN + 1;   // LDS pitch
constexpr int JT = 1024;     // threads (the passes are latency-bound per thread: 512 -> 1024 threads, 8.2 -> 7.0 ms for 8 x 128 x 128)

// A [batch][n][n] (lower triangle read, buffer then reused as V^T scratch), evals desc, evecs columns.
// sorted != 0: eigenvalues descending (ties by index); sorted == 0: eigenvalue j stays in position j, so for a
// nearly diagonal input the eigenvector matrix is a small rotation -- what block Jacobi needs (a sorting solver acts
// as a permutation there and shuffles off-diagonal mass around the schedule instead of annihilating it).
// NT: the size as a compile-time constant (128, 64: the head sizes) so that the element loops' e / n, e % n are shifts -- they are
// what a round costs (the passes are instruction-bound: 16 iterations x ~40 instructions per thread and pass, two integer
// divisions among them) -- or 0 for any even n <= 128 at run time.
template <int NT>
__global__ __launch_bounds__(JT) void syevj_kernel(double* Ag, int n_rt, double* evals, double* evecs, int* info,
                                                   int max_sweeps, int sorted) {
  const int n = NT ? NT : n_rt;
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
    int pos = sorted ? 0 : tid;
    if (sorted)
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

int syevj_batched(double* A, int64_t n, int64_t batch, double* evals, double* evecs, int* dflag, hipStream_t st,
                  int max_sweeps = 40, int sorted = 1) {
  MDG_CHECK_ARG(n >= 2 && n <= JN && n % 2 == 0, "syevj: n=%lld must be even and in [2, 128]", (long long)n);
  MDG_CHECK_ARG(batch > 0 && batch < (1ll << 31), "syevj: bad batch");
  MDG_HIP(hipMemsetAsync(dflag, 0, sizeof(int), st));
  if (n == 128)
    hipLaunchKernelGGL(syevj_kernel<128>, dim3((unsigned)batch), dim3(JT), 0, st, A, (int)n, evals, evecs, dflag, max_sweeps, sorted);
  else if (n == 64)
    hipLaunchKernelGGL(syevj_kernel<64>, dim3((unsigned)batch), dim3(JT), 0, st, A, (int)n, evals, evecs, dflag, max_sweeps, sorted);
  else
    hipLaunchKernelGGL(syevj_kernel<0>, dim3((unsigned)batch), dim3(JT), 0, st, A, (int)n, evals, evecs, dflag, max_sweeps, sorted);
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}

int check_flag(int* dflag, hipStream_t st, const char* what) { return finish_flag(dflag, st, STATUS_NO_CONVERGE, what); }


// ---------------------------------------------------------------- large n: block Jacobi out of the pieces above
// sqrt_M at d_model size (the reference calls it from compress_vo.py:44; this engine's VO stage does not need it, the
// entry exists so the function keeps its full domain).  Two-sided block Jacobi on 64-wide blocks: neighbouring blocks
// (2i, 2i+1) form 128x128 sub-problems solved by syevj_kernel, their rotations are applied to block rows / columns
// and to V by three batched GEMMs, then the blocks are physically rotated round-robin so that after nb-1 rounds every
// pair has met.  Everything is uniform-stride batched work; no per-pair launches.
int gemm_f64(int64_t M, int64_t N, int64_t K, double alpha, const void* A, int a_dtype, int64_t sa_i, int64_t sa_k,
             const int64_t* a_rows, const void* B, int b_dtype, int64_t sb_k, int64_t sb_j, double beta, void* C,
             int c_dtype, int64_t ldc, int64_t batch, int64_t a_bs, int64_t b_bs, int64_t c_bs, int flags,
             hipStream_t st);

__global__ __launch_bounds__(256) void bj_init_kernel(const double* M, int64_t n, int64_t ld, double* A, double* V,
                                                      int64_t np) {
  const int64_t i = blockIdx.x;
  for (int64_t j = threadIdx.x; j < np; j += 256) {
    double v = 0.;
    if (i < n && j < n) v = (j <= i) ? M[i * ld + j] : M[j * ld + i];  // lower triangle is authoritative
    A[i * np + j] = v;
    V[i * np + j] = (i == j) ? 1. : 0.;
  }
}

__global__ __launch_bounds__(256) void bj_extract_diag_kernel(const double* A, int64_t np, double* D) {
  const int64_t b = blockIdx.x;
  const double* src = A + b * 128 * np + b * 128;
  for (int e = threadIdx.x; e < 128 * 128; e += 256) D[b * 128 * 128 + e] = src[(int64_t)(e / 128) * np + e % 128];
}

// dst[I][J] = src[rowmap(I)][colmap(J)] with the maps acting on 64-wide blocks (perm == nullptr -> identity on rows)
__global__ __launch_bounds__(256) void bj_permute_kernel(const double* src, double* dst, int64_t np, const int* row_perm,
                                                         const int* col_perm) {
  const int64_t i = blockIdx.x;
  const int64_t si = row_perm ? (int64_t)row_perm[i >> 6] * 64 + (i & 63) : i;
  for (int64_t j = threadIdx.x; j < np; j += 256) {
    const int64_t sj = (int64_t)col_perm[j >> 6] * 64 + (j & 63);
    dst[i * np + j] = src[si * np + sj];
  }
}

// red[0] += sum of squares of the off-diagonal, red[1] += of everything (one workgroup per row)
__global__ __launch_bounds__(256) void bj_offnorm_kernel(const double* A, int64_t np, double* red) {
  __shared__ double s0[256], s1[256];
  const int64_t i = blockIdx.x;
  double off = 0., tot = 0.;
  for (int64_t j = threadIdx.x; j < np; j += 256) {
    const double v = A[i * np + j];
    tot += v * v;
    if (j != i) off += v * v;
  }
  s0[threadIdx.x] = off;
  s1[threadIdx.x] = tot;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      s0[threadIdx.x] += s0[threadIdx.x + o];
      s1[threadIdx.x] += s1[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(&red[0], s0[0]);
    atomicAdd(&red[1], s1[0]);
  }
}

// W[:, j] = V[:, j] * f(lambda_j), W2 likewise with g; lambda = diag(A); single workgroup computes max first
__global__ __launch_bounds__(256) void bj_scale_kernel(const double* A, const double* V, int64_t np, int64_t n, double ridge,
                                                       int scaled, double* W, double* W2, double* evals_out) {
  __shared__ double red[256];
  double mx = -1e300;
  for (int64_t j = threadIdx.x; j < np; j += 256) mx = fmax(mx, A[j * np + j]);
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  const double scale = scaled ? red[0] : 1.0;
  const int64_t i = blockIdx.x;
  for (int64_t j = threadIdx.x; j < np; j += 256) {
    const double lam = A[j * np + j];
    const double l = lam + ridge * scale;
    const double rt = sqrt(l > 0. ? l : 0.);
    const double v = V[i * np + j];
    W[i * np + j] = v * rt;
    if (W2) W2[i * np + j] = v / (rt > 1e-12 ? rt : 1e-12);
    if (i == 0 && evals_out && j < np) evals_out[j] = lam;
  }
}

}  // namespace mdg

// ---------------------------------------------------------------- sqrt(M + ridge I) without an eigensolve
// For the plain call sqrt_M(M, ridge) (no eigenvalues wanted, ridge not scaled by lambda_max) the clamps of the reference
// never bind on a numerically PSD input, and the function is the principal square root of A = M + ridge I.  The coupled
// Newton-Schulz iteration (Higham, Functions of Matrices, eq. 6.35)
//     Y <- Y (3 I - Z Y) / 2,   Z <- (3 I - Z Y) Z / 2,   Y0 = A / c,  Z0 = I,  c = ||A||_F
// converges to (A/c)^(1/2) and (A/c)^(-1/2) and is nothing but fp64 GEMMs: three n x n x n products per step on the
// matrix cores, ~log_2.25(c / ridge) + 5 steps.  n = 4096: about 25 steps, against 30 sweeps x 63 rounds of block
// Jacobi.  Anything the iteration cannot certify (a genuinely indefinite input) goes back to the eigen route.
namespace mdg {
namespace {

// lower triangle of M mirrored (torch.linalg.eigh reads uplo = 'L'), ridge on the diagonal; red[0] += sum of squares
__global__ __launch_bounds__(256) void ns_load_kernel(const double* M, int64_t n, int64_t ld, double ridge, double* Y, double* red) {
  const int64_t i = blockIdx.x;
  double ss = 0.;
  for (int64_t j = threadIdx.x; j < n; j += 256) {
    double v = (j <= i) ? M[i * ld + j] : M[j * ld + i];
    if (i == j) v += ridge;
    Y[i * n + j] = v;
    ss += v * v;
  }
  __shared__ double part[256];
  part[threadIdx.x] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(red, part[0]);
}

// Y /= sqrt(red[0]);  Z = I
__global__ __launch_bounds__(256) void ns_start_kernel(double* Y, double* Z, int64_t n, const double* red) {
  const int64_t i = blockIdx.x;
  const double s = 1. / sqrt(red[0]);
  for (int64_t j = threadIdx.x; j < n; j += 256) {
    Y[i * n + j] *= s;
    Z[i * n + j] = (i == j) ? 1. : 0.;
  }
}

// T holds -Z Y: T += 3 I, and res[0] += ||I - Z Y||_F^2 = ||T - 2 I||_F^2
__global__ __launch_bounds__(256) void ns_shift_kernel(double* T, int64_t n, double* res) {
  const int64_t i = blockIdx.x;
  double ss = 0.;
  for (int64_t j = threadIdx.x; j < n; j += 256) {
    double v = T[i * n + j];
    if (i == j) {
      v += 3.;
      T[i * n + j] = v;
      v -= 2.;
    }
    ss += v * v;
  }
  __shared__ double part[256];
  part[threadIdx.x] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(res, part[0]);
}

// out = scale(red) * (S + S^T) / 2 with scale = c^(+-1/2)
__global__ __launch_bounds__(256) void ns_finish_kernel(const double* S, int64_t n, const double* red, int inverse, double* out,
                                                        int64_t ld_out) {
  const int64_t i = blockIdx.x;
  const double c = sqrt(sqrt(red[0]));  // sqrt(||A||_F)
  const double s = inverse ? 0.5 / c : 0.5 * c;
  for (int64_t j = threadIdx.x; j < n; j += 256) out[i * ld_out + j] = s * (S[i * n + j] + S[j * n + i]);
}

// returns MDG_OK with *ok = 1 when the iteration reached round-off, *ok = 0 when it did not (caller falls back)
int sqrt_psd_newton(const double* M, int64_t n, int64_t ld, double ridge, double* root, double* inv_root, double* ws,
                    hipStream_t st, int* ok) {
  *ok = 0;
  const int64_t nn = n * n;
  double* Y = ws;
  double* Z = Y + nn;
  double* T = Z + nn;
  double* Y2 = T + nn;
  double* Z2 = Y2 + nn;
  double* red = Z2 + nn;  // [0] ||A||_F^2, [1 + k] residual^2 of step k
  constexpr int MAX_STEPS = 64;
  MDG_HIP(hipMemsetAsync(red, 0, (MAX_STEPS + 2) * sizeof(double), st));
  hipLaunchKernelGGL(ns_load_kernel, dim3((unsigned)n), dim3(256), 0, st, M, n, ld, ridge, Y, red);
  hipLaunchKernelGGL(ns_start_kernel, dim3((unsigned)n), dim3(256), 0, st, Y, Z, n, red);
  MDG_LAUNCH_CHECK();
  double prev = 1e300;
  for (int k = 0; k < MAX_STEPS; k++) {
    MDG_TRY(gemm_f64(n, n, n, -1.0, Z, MDG_F64, n, 1, nullptr, Y, MDG_F64, n, 1, 0.0, T, MDG_F64, n, 1, 0, 0, 0, 0, st));
    hipLaunchKernelGGL(ns_shift_kernel, dim3((unsigned)n), dim3(256), 0, st, T, n, red + 1 + k);
    MDG_LAUNCH_CHECK();
    double r2;
    MDG_HIP(hipMemcpyAsync(&r2, red + 1 + k, sizeof(double), hipMemcpyDeviceToHost, st));
    MDG_HIP(hipStreamSynchronize(st));
    const double res = sqrt(r2);
    if (MDG_KNOB("MDG_DEBUG_NEWTON")) fprintf(stderr, "newton-schulz step %d: ||I - ZY||_F = %.3e\n", k, res);
    if (!(res == res) || res > 1e6) return MDG_OK;                       // diverging: not positive definite
    // round-off floor: the residual has stopped contracting (it squares per step once below ~0.5)
    if (res < 1e-7 && res > 0.25 * prev) { *ok = 1; break; }
    if (res < 1e-14 * (double)n) { *ok = 1; break; }
    if (k > 8 && res > 0.999 * prev && res > 0.9 * sqrt((double)n)) return MDG_OK;  // stuck at ||I||: singular / indefinite
    prev = res;
    MDG_TRY(gemm_f64(n, n, n, 0.5, Y, MDG_F64, n, 1, nullptr, T, MDG_F64, n, 1, 0.0, Y2, MDG_F64, n, 1, 0, 0, 0, 0, st));
    MDG_TRY(gemm_f64(n, n, n, 0.5, T, MDG_F64, n, 1, nullptr, Z, MDG_F64, n, 1, 0.0, Z2, MDG_F64, n, 1, 0, 0, 0, 0, st));
    double* t = Y; Y = Y2; Y2 = t;
    t = Z; Z = Z2; Z2 = t;
  }
  if (!*ok) return MDG_OK;
  hipLaunchKernelGGL(ns_finish_kernel, dim3((unsigned)n), dim3(256), 0, st, Y, n, red, 0, root, n);
  if (inv_root) hipLaunchKernelGGL(ns_finish_kernel, dim3((unsigned)n), dim3(256), 0, st, Z, n, red, 1, inv_root, n);
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}

}  // namespace
}  // namespace mdg

using namespace mdg;

extern "C" size_t mdg_sqrt_psd_large_ws_bytes(int64_t n) {
  const size_t np = (size_t)ceil_div(n, 128) * 128;
  const size_t jacobi = (4 * np * np + 2 * (np / 128) * 128 * 128 + np + 64) * sizeof(double) + 2 * (np / 64) * sizeof(int) + 256;
  const size_t newton = (5 * (size_t)n * n + 80) * sizeof(double);
  return jacobi > newton ? jacobi : newton;
}

extern "C" int mdg_sqrt_psd_large(const double* M, int64_t n, int64_t ld, double ridge, int scaled, double* root,
                                  double* inv_root, double* evals_out, void* ws, size_t ws_bytes, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(M && root && n > 0 && ld >= n, "mdg_sqrt_psd_large: bad arguments");
  MDG_CHECK_ARG(ws && ws_bytes >= mdg_sqrt_psd_large_ws_bytes(n), "mdg_sqrt_psd_large: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (!scaled && !evals_out && ridge >= 1e-12 && !MDG_KNOB("MDG_SQRT_JACOBI")) {
    int ok = 0;
    MDG_TRY(sqrt_psd_newton(M, n, ld, ridge, root, inv_root, (double*)ws, st, &ok));
    if (ok) return MDG_OK;  // otherwise: the eigen route below, which implements the reference's clamps
  }
  const int64_t np = ceil_div(n, 128) * 128, m = np / 128, nb = 2 * m;
  double* A = (double*)ws;
  double* A2 = A + np * np;
  double* V = A2 + np * np;
  double* V2 = V + np * np;
  double* D = V2 + np * np;
  double* J = D + m * 128 * 128;
  double* ev = J + m * 128 * 128;
  double* red = ev + np;          // [0] off^2, [1] total^2
  int* dflag = (int*)(red + 8);
  int* perm = (int*)(red + 16);   // round-robin source block of every destination block
  // round-robin rotation of the 64-blocks: pairs are (2i, 2i+1); block 0 stays, the others move one seat
  {
    int host_perm[2 * 1024];
    MDG_CHECK_ARG(nb <= 2 * 1024, "mdg_sqrt_psd_large: n too large");
    for (int64_t i = 0; i < m; i++) {
      host_perm[2 * i] = (i == 0) ? 0 : (i == 1 ? 1 : (int)(2 * (i - 1)));
      host_perm[2 * i + 1] = (i < m - 1) ? (int)(2 * (i + 1) + 1) : (int)(2 * (m - 1));
    }
    if (m == 1) { host_perm[0] = 0; host_perm[1] = 1; }
    MDG_HIP(hipMemcpyAsync(perm, host_perm, nb * sizeof(int), hipMemcpyHostToDevice, st));
    MDG_HIP(hipStreamSynchronize(st));  // host_perm is a stack array
  }
  hipLaunchKernelGGL(bj_init_kernel, dim3((unsigned)np), dim3(256), 0, st, M, n, ld, A, V, np);
  MDG_LAUNCH_CHECK();
  const int64_t bs_rows = 128 * np, bs_cols = 128, bs_blk = 128 * 128;
  const int rounds = nb > 2 ? (int)(nb - 1) : 1;
  bool done = false;
  for (int sweep = 0; sweep < 30 && !done; sweep++) {
    for (int r = 0; r < rounds; r++) {
      hipLaunchKernelGGL(bj_extract_diag_kernel, dim3((unsigned)m), dim3(256), 0, st, A, np, D);
      MDG_LAUNCH_CHECK();
      MDG_TRY(syevj_batched(D, 128, m, ev, J, dflag, st, 6, 0));  // partial, UNSORTED inner solves
      // A2 = blockdiag(J)^T A     (block rows)
      MDG_TRY(gemm_f64(128, np, 128, 1.0, J, MDG_F64, 1, 128, nullptr, A, MDG_F64, np, 1, 0.0, A2, MDG_F64, np, m, bs_blk,
                       bs_rows, bs_rows, 0, st));
      // A = A2 blockdiag(J)       (block columns)
      MDG_TRY(gemm_f64(np, 128, 128, 1.0, A2, MDG_F64, np, 1, nullptr, J, MDG_F64, 128, 1, 0.0, A, MDG_F64, np, m, bs_cols,
                       bs_blk, bs_cols, 0, st));
      // V2 = V blockdiag(J)
      MDG_TRY(gemm_f64(np, 128, 128, 1.0, V, MDG_F64, np, 1, nullptr, J, MDG_F64, 128, 1, 0.0, V2, MDG_F64, np, m, bs_cols,
                       bs_blk, bs_cols, 0, st));
      if (nb > 2) {
        hipLaunchKernelGGL(bj_permute_kernel, dim3((unsigned)np), dim3(256), 0, st, A, A2, np, perm, perm);
        hipLaunchKernelGGL(bj_permute_kernel, dim3((unsigned)np), dim3(256), 0, st, V2, V, np, (const int*)nullptr, perm);
        MDG_LAUNCH_CHECK();
        double* t = A; A = A2; A2 = t;
      } else {
        double* t = V; V = V2; V2 = t;
      }
    }
    MDG_HIP(hipMemsetAsync(red, 0, 2 * sizeof(double), st));
    hipLaunchKernelGGL(bj_offnorm_kernel, dim3((unsigned)np), dim3(256), 0, st, A, np, red);
    MDG_LAUNCH_CHECK();
    double h[2];
    MDG_HIP(hipMemcpyAsync(h, red, sizeof(h), hipMemcpyDeviceToHost, st));
    MDG_HIP(hipStreamSynchronize(st));
    if (MDG_KNOB("MDG_DEBUG_JACOBI")) fprintf(stderr, "block-jacobi sweep %d: off^2 %.3e tot^2 %.3e\n", sweep, h[0], h[1]);
    done = h[0] <= 1e-26 * h[1] || h[1] == 0.;  // ||off||_F <= 1e-13 ||A||_F
  }
  if (!done) {
    set_error("mdg_sqrt_psd_large: block Jacobi did not converge in 30 sweeps");
    return MDG_ERR_NO_CONVERGE;
  }
  // root = (V f(lambda)) V^T ; inverse root likewise.  W lives in A2 / V2 (both free now).
  hipLaunchKernelGGL(bj_scale_kernel, dim3((unsigned)np), dim3(256), 0, st, A, V, np, n, ridge, scaled, A2,
                     inv_root ? V2 : (double*)nullptr, evals_out ? ev : (double*)nullptr);
  MDG_LAUNCH_CHECK();
  MDG_TRY(gemm_f64(n, n, np, 1.0, A2, MDG_F64, np, 1, nullptr, V, MDG_F64, 1, np, 0.0, root, MDG_F64, n, 1, 0, 0, 0, 0, st));
  if (inv_root)
    MDG_TRY(gemm_f64(n, n, np, 1.0, V2, MDG_F64, np, 1, nullptr, V, MDG_F64, 1, np, 0.0, inv_root, MDG_F64, n, 1, 0, 0, 0, 0,
                     st));
  if (evals_out) MDG_HIP(hipMemcpyAsync(evals_out, ev, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
  return MDG_OK;
}

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
