/* modegpt_hip.h -- C ABI of libmodegpt_hip.so, the MI355X (gfx950) engine for MoDeGPT's per-layer
 * compression path: activation-covariance accumulation, Nystrom / CR / SVD factorisation of the MLP and
 * attention weights, and the compressed-weight build.
 *
 * The reference (cbacary/MoDeGPT) has no FFI: its seam is Python.  This ABI sits directly
 * beneath the reference's five hot-path functions; each entry point below names the reference lines it
 * replaces (paths relative to the reference root).  INTEGRATION.md shows the ctypes stubs a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the library borrows it for the call
 *     and owns no memory across calls (workspaces are passed in; query their size with the *_ws_bytes twin)
 *   - matrices are row-major with an explicit leading dimension in ELEMENTS
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work unless documented as
 *     synchronising
 *   - return value: MDG_OK or a negative MDG_ERR_* code; mdg_last_error() gives the message (thread-local)
 *   - re-entrant per device, no global lock, callable from any host thread
 */
#ifndef MODEGPT_HIP_H
#define MODEGPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDG_ABI_VERSION 9 /* 2: w_dtype on mdg_nystrom_down / mdg_vo_compress, mdg_rope_gather added; 3: mdg_cov_accum_i8_stats added;
                             4: mdg_cov_accum_i8 chooses its route on the device (route_counts argument, no host synchronisation);
                                mdg_comm_* / mdg_allgather_layers added;
                             5: mdg_potrs_lower takes a workspace (mdg_potrs_lower_ws_bytes);
                             6: mdg_cov_accum_i8_multi added (several statistics in one int8 launch); the int8 workspace layout changed;
                             7: the int8 route is derived from a per-call error bound, single columns can leave the int8 path for an fp64
                                column kernel (route_counts has 4 entries, mdg_cov_accum_i8_route added, workspace layout changed);
                                mdg_shutdown and mdg_deferred_status_* added;
                             8: mdg_cov_i8_set_tolerance / mdg_cov_i8_tolerance, mdg_nystrom_down_overlapped added;
                             9: the int8 route's tolerance factor is an ARGUMENT of mdg_cov_accum_i8 / mdg_cov_accum_i8_multi (the
                                process-wide setter / getter of ABI 8 are gone: no accuracy state in the library);
                                mdg_ridge_scores takes `sens`, mdg_select_margin added (certificate of the MLP rank selection);
                                the exact route of the int8 covariance (`flags`, route_counts[4], mdg_cov_accum_i8_route's `exact`) */

enum mdg_status {
  MDG_OK = 0,
  MDG_ERR_BAD_ARG = -1,     /* shape / pointer / alignment / workspace-size problem */
  MDG_ERR_HIP = -2,         /* a HIP runtime call failed */
  MDG_ERR_NOT_PD = -3,      /* Cholesky met a non-positive pivot (torch.linalg.cholesky would raise) */
  MDG_ERR_NO_CONVERGE = -4, /* Jacobi eigensolver hit its sweep limit */
  MDG_ERR_NO_DEVICE = -5    /* no gfx950 device visible */
};

enum mdg_dtype { MDG_BF16 = 0, MDG_F16 = 1, MDG_F32 = 2, MDG_F64 = 3 };

/* QK scoring variants (compress_qk.py:244-285) */
enum mdg_qk_mode {
  MDG_QK_ROPE_GROUPED = 0, /* compress_head_llama_grouped, compress_qk.py:320-382 */
  MDG_QK_ROPE_MHA = 1,     /* compress_head_llama,         compress_qk.py:387-436 */
  MDG_QK_OPT = 2           /* compress_head_opt,           compress_qk.py:439-476 */
};

int mdg_abi_version(void);
const char* mdg_last_error(void);
/* Fills name (<= cap bytes) with the gcnArchName of `device`; MDG_ERR_NO_DEVICE if HIP sees no GPU. */
int mdg_device_info(int device, char* name, int cap, int* n_cu, int64_t* hbm_bytes);
/* Deferred status.  The decomposition entry points below that say "SYNCHRONISES" do so for one reason: a 4-byte device status
 * word (the Cholesky's not-positive-definite pivot, the Jacobi solver's convergence flag) has to reach the host before they can
 * return MDG_ERR_NOT_PD / MDG_ERR_NO_CONVERGE.  Between mdg_deferred_status_begin and mdg_deferred_status_end, on the calling
 * thread, they merge that word into `status_dev` (DEVICE int[2], zeroed by _begin on `stream`: {kind of the FIRST failure, its
 * detail}) with a one-thread kernel and return MDG_OK at once -- a whole layer's chain (ridge scores, Nystrom refit, QK selection,
 * VO factors) then enqueues without a host round trip, and the caller reads the two ints whenever it next has to wait for the
 * stream anyway.  mdg_deferred_status_decode turns the two ints (copied to the host by the caller) into the status code and
 * mdg_last_error() text the synchronising call would have produced: MDG_ERR_NOT_PD -> torch.linalg.LinAlgError upstream
 * (torch.linalg.cholesky, compress_mlp.py:20,56), MDG_ERR_NO_CONVERGE -> RuntimeError.  After a failure the remaining kernels of
 * the chain still run (on garbage, within their buffers); their outputs must be discarded. */
int mdg_deferred_status_begin(int* status_dev, void* stream);
int mdg_deferred_status_end(void);
int mdg_deferred_status_decode(const int* status_host);
/* Releases what the library keeps across calls: the device copies of the int8 product's tile schedules (a few KB per shape,
 * built at first use).  The library owns no HIP streams or events -- those are the caller's -- and makes no HIP call from a static
 * destructor; call this before the process tears the HIP runtime down (modegpt_amd/_lib.py registers it with atexit), with no
 * library call in flight.  Safe to call repeatedly; the next covariance call rebuilds what it needs. */
int mdg_shutdown(void);

/* ------------------------------------------------------------------ covariance (calibration hooks)
 * sigma[b] (lower triangle incl. diagonal tiles) += X_b^T X_b, products and sums in fp64 of the exactly
 * converted inputs.  X is [n_tokens, >= batch*n_feat] row-major (ld elements per token); problem b reads
 * columns [b*n_feat, (b+1)*n_feat).  batch=1: replaces `H.T @ H` (LlamaAdapter.py:127-136, model_adapter.py:546-554)
 * and `sum(X.mT @ X, 0)` (LlamaAdapter.py:138-147); batch=n_heads, n_feat=head_dim: replaces the permute + bmm of
 * LlamaAdapter.py:115-125 / model_adapter.py:556-567 without the permute copy.
 * Only the lower triangle of sigma is valid until mdg_cov_finalize mirrors it.
 * relu != 0 applies max(x, 0) on load (OPT fc1 hook).  ws may be NULL when mdg_cov_accum_ws_bytes says 0;
 * it holds split-K partials (reduced in fixed order -> run-to-run deterministic). */
size_t mdg_cov_accum_ws_bytes(int64_t n_tokens, int64_t n_feat, int64_t batch);
int mdg_cov_accum(const void* x, int dtype, int64_t n_tokens, int64_t n_feat, int64_t batch, int64_t ld,
                  int relu, double* sigma, int64_t ld_sigma, int64_t sigma_batch_stride, void* ws,
                  size_t ws_bytes, void* stream);
/* The same accumulate for up to 4 problems of one calibration batch in ONE launch (the four hooks of a layer:
 * sigma_mlp, sigma_x, sigma_q, sigma_k): the small problems' workgroups fill the slots the large one's last,
 * under-filled round leaves idle.  Put the largest problem first.  Every problem must have n_feat % 128 == 0 and
 * 16-byte aligned rows (otherwise call mdg_cov_accum per problem); no ReLU.  `problems` is a HOST array. */
typedef struct {
  const void* x;   /* [n_tokens, >= batch*n_feat], ld elements per token */
  int64_t n_tokens, n_feat, batch, ld;
  double* sigma;   /* [batch][n_feat][ld_sigma] */
  int64_t ld_sigma, sigma_batch_stride;
} mdg_cov_problem;
size_t mdg_cov_accum_multi_ws_bytes(int n, const mdg_cov_problem* problems, int dtype);
int mdg_cov_accum_multi(int n, const mdg_cov_problem* problems, int dtype, void* ws, size_t ws_bytes, void* stream);
/* The same accumulation for ONE bf16 matrix through the int8 matrix cores (csrc/cov_i8.hip): an ERROR-FREE SPLIT of every bf16 value
 * into six balanced base-256 digits against a per-column power-of-two scale, and a TRUNCATED PRODUCT -- the digit-plane products with
 * s + t < P are formed by v_mfma_i32_32x32x32_i8 with exact int32 accumulation and folded into sigma in fp64 every 65504 tokens (the
 * exact int32 bound); the pairs with s + t >= P are dropped.  What the dropped pairs can amount to is bounded per call from integer
 * plane energies the split pass accumulates (Cauchy-Schwarz over the tokens; derivation in csrc/cov_i8.hip at i8_route_kernel and
 * DESIGN.md section 7, host model tests/i8_model.py):
 *     |sigma_ij - exact| <= (SQ_P + X_P) sqrt(sigma_ii sigma_jj)   entry-wise, for any input,
 * and the route is the smallest P in {5, 6} with SQ_P <= 1e-12 (the part of the bound that is attained) and X_P <= 1e-11 (cross
 * terms; 20-50x above what uncorrelated columns produce).  GUARANTEED for any input: <= 1.1e-11 (times `tolerance`).  TYPICAL:
 * 1e-13 (an empirical figure, not a promise: <= 1e-12 on every distribution family of scripts/probes/i8_fuzz.py and at the
 * product's widths; data whose columns have proportional digit sequences can come arbitrarily close to the bound).  Columns that alone break the bound -- a bulk 10-15 binades under a
 * few massive activations, columns holding an Inf / NaN -- leave the int8 path one by one (at most MDG_I8_MAX_COLUMNS per statistic
 * and call): the fold skips their rows and columns of sigma and an fp64 column kernel (plain fp64 sums of products, the reference's
 * arithmetic, LlamaAdapter.py:127-147) computes them.  Only when that is not enough does the whole statistic run through
 * mdg_cov_accum.
 * The route is chosen ON THE DEVICE: the call enqueues the five-plane product, the six-plane product, the column kernel and the fp64
 * kernel back to back, and the launches the route does not select exit at once -- the call only enqueues and never waits for the host
 * (it can be captured in a hipGraph) when used_i8 is NULL.  Every decision is a function of integer sums: run-to-run bit-identical.
 * route_counts (DEVICE pointer to 5 ints, optional): [0] += 1 when five planes ran, [1] six planes, [2] the fp64 kernel for the whole
 *   statistic, [3] += the number of columns handed to the fp64 column kernel, [4] += 1 when the exact route ran (see below; such a
 *   call is also booked under [0] or [1], the class the route kernel gave it); the caller keeps it across calls and reads it
 *   whenever it likes (calibration reads it once, at the end).
 * used_i8 (HOST pointer, optional; measurement and tests): receives the route of THIS call -- 5, 6, or 0 for the fp64 kernel --
 *   at the price of one stream synchronisation.
 * tolerance: the accuracy / speed dial of THIS call, one factor f in [1, 1e6] on both thresholds of the route (SQ_P <= f 1e-12,
 *   X_P <= f tau_x).  f = 1 is the guarantee stated above.  A caller who accepts f times that bound gets five planes where the default
 *   takes six -- SiLU-gated MLP activations have X_5 = 3.7e-10, so f >= 37 moves them to five planes: measured 3.8e-12 instead of
 *   8e-14, the sigma_mlp launch 29 instead of 37 ms -- and fewer columns on the fp64 column kernel.  The call still computes and
 *   reports its own bound (mdg_cov_accum_i8_route), so what was guaranteed for a given input is known, whatever f.  An argument,
 *   not process state: concurrent callers with different factors do not see each other.  Not in the reference (plain fp64 there).
 * n_feat must be a multiple of 128, n_tokens < 2^28.  ws: mdg_cov_accum_i8_ws_bytes (about 10 bytes per element of x: six digit planes, the
 * exact route's event lists and its bf16 copy of x).
 * ev_start / ev_stop: optional hipEvent_t recorded on `stream` right before / after the two product launches (bench.py times
 * the dominant kernel alone with them); NULL otherwise. */
/* THE EXACT ROUTE (since ABI 9.  flags = 0: taken where it is the faster product -- launches the route kernel classes as six planes,
 * i.e. SiLU- / GELU-gated MLP activations, and five-plane launches whose first statistic has >= 4096 features; MDG_I8_EXACT_ALWAYS:
 * wherever the remainder lists fit; MDG_I8_NO_EXACT: never).  Planes 3 .. 5 are reached only by elements 17
 * binades and more below their column's maximum -- 3e-5 of the elements of a Gaussian column, 0.5 % of a SiLU-gated one -- so the
 * call lists those elements (token, column, low 24 bits) and, when every list fits (at most 6.2 % of any column x 2048 tokens),
 * replaces the truncated product by an exact one:  X^T X = X_d^T X_d + X_lo^T X + X_d^T X_lo  with X_d the top three digit planes --
 * all NINE of their plane pairs on the int8 matrix cores (a product launch of its own: three planes, no piece masks) -- and the two
 * remainder products as fp64 sums over the listed elements (i8_lo_product_kernel for sparse lists, i8_lo_wide_kernel for dense
 * ones; the device picks).  No plane pair is dropped: the error is fp64
 * rounding (<= MDG_I8_EXACT_ROUNDING of sqrt(sigma_ii sigma_jj)) plus the bound's rho term for elements more than 38 binades under
 * their column maximum, whatever `tolerance` says; 9 executed plane pairs instead of 9.4 (Gaussian) / 15.1 (SiLU-gated).  The route
 * kernel's decisions are unchanged -- which columns leave for the fp64 column kernel, whether the whole statistic goes to
 * mdg_cov_accum, five or six planes when a list does not fit -- and so is everything the call reports, plus: route_counts[4] += the
 * statistics that took the exact route (they are also booked under the five- / six-plane class the route kernel gave them), and
 * mdg_cov_accum_i8_route's `exact`. */
#define MDG_I8_MAX_COLUMNS 32
#define MDG_I8_NO_EXACT 1             /* flags: never the exact route (the truncated five- / six-plane product with its bound) */
#define MDG_I8_EXACT_ALWAYS 2         /* flags: the exact route wherever the remainder lists fit, also for launches of the five-plane class */
#define MDG_I8_EXACT_ROUNDING 5e-15   /* what mdg_cov_accum_i8_route reports beside the rho term for a call on the exact route */
size_t mdg_cov_accum_i8_ws_bytes(int64_t n_tokens, int64_t n_feat);
int mdg_cov_accum_i8(const void* x, int64_t n_tokens, int64_t n_feat, int64_t ld, double* sigma, int64_t ld_sigma, void* ws,
                     size_t ws_bytes, double tolerance, int flags, int* used_i8, int* route_counts, void* ev_start, void* ev_stop,
                     void* stream);
/* v_mfma instructions the product kernel of the LAST mdg_cov_accum_i8 call on workspace `ws` executed (0 after a call that
 * fell back to mdg_cov_accum).  The split pass records, per k-step and 32-row group, which digit planes hold a nonzero
 * there; the product kernel neither loads nor multiplies planes that are all-zero over a tile panel, so the count is at
 * most -- and on real activations well below -- the dense (tiles) x (k-steps) x (waves) x (MFMAs per step).  Copies 8 bytes
 * device -> host on `stream` and synchronises it.  bench.py prices the kernel with this count. */
int mdg_cov_accum_i8_stats(const void* ws, int64_t n_tokens, int64_t n_feat, unsigned long long* executed_mfma, void* stream);
/* The route the LAST mdg_cov_accum_i8 / mdg_cov_accum_i8_multi call on workspace `ws` took for statistic `stat` of `problems` (the
 * array that call was given): *planes = 5, 6, or 0 (whole statistic through mdg_cov_accum); *n_columns and columns[MDG_I8_MAX_COLUMNS]
 * (-1 padded) = the columns the fp64 column kernel computed, in the order the route took them; bound[0] = SQ_P, bound[1] = X_P of
 * the columns that stayed (their sum bounds the entry-wise error relative to sqrt(sigma_ii sigma_jj) of this call's tokens); *exact != 0
 * when the call ran the exact route -- 1: its remainder products on the tile kernel (sparse event lists), 2: on the wide kernels --
 * and then bound[0] = the rho term + MDG_I8_EXACT_ROUNDING, bound[1] = 0.  Any output pointer may be NULL.  Copies device -> host on `stream` and synchronises it: tests and measurements only. */
int mdg_cov_accum_i8_route(int count, const mdg_cov_problem* problems, int stat, const void* ws, int* planes, int* n_columns,
                           int* columns, double* bound, int* exact, void* stream);
/* Up to 4 statistics of ONE calibration batch (the four hooks of a layer) through the int8 digit-plane kernels with ONE
 * persistent product launch: the tiles of all statistics share one static tile schedule, so the small ones fill what the large
 * one's last round leaves idle instead of ending launches of their own, and one route -- the deepest any statistic on the int8 path
 * asks for (more planes never loosen a bound); every statistic has its own bound, its own columns for the fp64 column kernel, and
 * leaves the launch alone for mdg_cov_accum when its bound cannot be met (its tiles are skipped on the device).  `problems` is a HOST
 * array, largest statistic first, all with the same n_tokens, bf16.  batch == 1: sigma [n_feat][ld_sigma], n_feat a multiple of 128.
 * batch > 1: per-head Grams of an activation [n_tokens][batch * 128] -- n_feat must be 128, sigma contiguous
 * [batch][128][128] (ld_sigma 128, sigma_batch_stride 16384); only the diagonal tiles are computed.  Several statistics need a
 * 256-CU device (the schedule is cut for 8 XCDs x 32 CUs); otherwise call mdg_cov_accum_i8 per statistic.
 * tolerance / flags / used_i8 / route_counts / ev_start / ev_stop as in mdg_cov_accum_i8 (route_counts += the number of statistics per route;
 * used_i8 = 5 or 6, the planes of the statistics that stayed on the int8 path, 0 when all of them went to the fp64 kernel);
 * mdg_cov_accum_i8_stats(ws, 0, 0, ...) reads the executed-MFMA count of the whole launch.  mdg_cov_accum_i8 is this call with
 * one statistic. */
size_t mdg_cov_accum_i8_multi_ws_bytes(int count, const mdg_cov_problem* problems);
int mdg_cov_accum_i8_multi(int count, const mdg_cov_problem* problems, void* ws, size_t ws_bytes, double tolerance, int flags,
                           int* used_i8, int* route_counts, void* ev_start, void* ev_stop, void* stream);
/* sigma[b] <- scale * sigma[b] on the lower triangle, mirrored into the upper.  scale = 1/(n_texts*2048)
 * reproduces calibration.py:141-146. */
int mdg_cov_finalize(double* sigma, int64_t n, int64_t batch, int64_t ld_sigma, int64_t sigma_batch_stride,
                     double scale, void* stream);
/* Block-Influence partial: *out += sum over tokens of (1 - cos(x_in[t], x_out[t])) in fp64
 * (calibration.py:118-124; the caller divides by T and n_texts).  ws: mdg_bi_ws_bytes(n_tokens). */
size_t mdg_bi_ws_bytes(int64_t n_tokens);
int mdg_bi_accum(const void* x_in, const void* x_out, int dtype, int64_t n_tokens, int64_t d, int64_t ld,
                 double* out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ dense fp64 building blocks
 * C = alpha * op(A) * op(B) + beta * C on v_mfma_f64_16x16x4_f64.  Element (i,k) of op(A) is
 * A[i*sa_i + k*sa_k] (or A[a_rows[i]*sa_i + k*sa_k] when a_rows != NULL), element (k,j) of op(B) is
 * B[k*sb_k + j*sb_j]; C is row-major [M, N] with ldc, dtype f64 or bf16 (bf16: beta must be 0, rounding as
 * torch's .to(bfloat16)).  flags: MDG_GEMM_*.  Batched with element strides. */
#define MDG_GEMM_LOWER_ONLY 1   /* M==N: only tiles with row-block >= col-block are computed (SYRK update) */
#define MDG_GEMM_A_LOWER_TRI 2  /* op(A)[i,k] = 0 for k > i (k range clipped per row tile) */
#define MDG_GEMM_B_LOWER_TRI 4  /* op(B)[k,j] = 0 for k < j (k range clipped per col tile) */
#define MDG_GEMM_A_UPPER_TRI 8  /* op(A)[i,k] = 0 for k < i */
int mdg_gemm_f64(int64_t M, int64_t N, int64_t K, double alpha, const void* A, int a_dtype, int64_t sa_i,
                 int64_t sa_k, const int64_t* a_rows, const void* B, int b_dtype, int64_t sb_k,
                 int64_t sb_j, double beta, void* C, int c_dtype, int64_t ldc, int64_t batch,
                 int64_t a_batch_stride, int64_t b_batch_stride, int64_t c_batch_stride, int flags,
                 void* stream);

/* In-place lower Cholesky A = L L^T (upper triangle left untouched), blocked right-looking with 128-wide
 * panels; inv_diag receives the inverses of the 128x128 diagonal blocks of L ([ceil(n/128)][128][128],
 * identity-padded).  SYNCHRONISES the stream once at the end to read the pivot flag.
 * Replaces torch.linalg.cholesky at compress_mlp.py:20,56. */
size_t mdg_potrf_inv_diag_elems(int64_t n);
int mdg_potrf_lower(double* A, int64_t n, int64_t lda, double* inv_diag, void* stream);
/* X (n x nrhs, ldx, in place) <- (L L^T)^-1 X by blocks of 2048 rows: the diagonal blocks of L are inverted explicitly
 * (recursive doubling from inv_diag), a block's substitution is one triangular-aware GEMM with that inverse, one rank-2048
 * GEMM carries its solution on.  ws: mdg_potrs_lower_ws_bytes(n, nrhs) (the block inverses + a second right-hand-side buffer).
 * Replaces torch.cholesky_solve at compress_mlp.py:57. */
size_t mdg_potrs_lower_ws_bytes(int64_t n, int64_t nrhs);
int mdg_potrs_lower(const double* L, int64_t n, int64_t ldl, const double* inv_diag, double* X, int64_t nrhs,
                    int64_t ldx, void* ws, size_t ws_bytes, void* stream);
/* out[j] = ((L L^T)^-1)_jj = || L^-1 e_j ||^2.  ws: mdg_inv_diag_of_spd_ws_bytes(n).
 * Replaces torch.cholesky_inverse + torch.diag at compress_mlp.py:21-23. */
size_t mdg_chol_inverse_diag_ws_bytes(int64_t n);
int mdg_chol_inverse_diag(const double* L, int64_t n, int64_t ldl, const double* inv_diag, double* out,
                          void* ws, size_t ws_bytes, void* stream);

/* Batched symmetric eigensolver, n <= 128 and even: cyclic Jacobi in LDS, one workgroup per matrix.
 * A [batch][n][n] (symmetric, destroyed), evals [batch][n] DESCENDING, evecs [batch][n][n] with eigenvector j
 * in COLUMN j.  SYNCHRONISES to read the convergence flag.  (The reference reaches torch.linalg.eigh / svd
 * through sqrt_M, compression_utils.py:21, and compress_vo.py:130,187,194.) */
int mdg_syevj_batched(double* A, int64_t n, int64_t batch, double* evals, double* evecs, void* stream);

/* ------------------------------------------------------------------ MLP: ridge-leverage + Nystrom
 * scores = diag((C + ridge I)^-1).  `ridge` is added as given: pass the fp32-rounded lambda to reproduce the
 * float32 eye of compress_mlp.py:18.  C is not modified.  ws: mdg_ridge_scores_ws_bytes(n).  SYNCHRONISES.
 * sens (optional, [n]; NULL = not wanted): the first-order sensitivity of every score to an ENTRY-WISE RELATIVE perturbation of C,
 * |E_ab| <= eps sqrt(c_aa c_bb) -- the form of the covariance routes' error bounds (mdg_cov_accum_i8: eps <= 1.1e-11 guaranteed;
 * fp64 accumulation: eps <= tokens 2^-53):  |delta scores[j]| <= eps sens[j] + O(eps^2),
 *     sens[j] = (sum_b |X_bj| sum_a |X_ba| sqrt(c_aa))^2 >= (sum_a sqrt(c_aa) |((C + ridge I)^-1)_aj|)^2,   X = inv(chol(C + ridge I)).
 * Two extra passes over the triangle of X the call holds anyway (1.6 GB of reads at n = 14336, < 1 ms); the scores are the same
 * bits with or without it. */
size_t mdg_ridge_scores_ws_bytes(int64_t n);
int mdg_ridge_scores(const double* C, int64_t n, int64_t ldc, double ridge, double* scores, double* sens, void* ws,
                     size_t ws_bytes, void* stream);
/* idx[0..k) = indices of the k smallest scores, ascending index order (topk(largest=False) + sort,
 * compress_mlp.py:45-47).  Ties: lower index first.  NaN ranks as largest. */
int mdg_select_smallest_sorted(const double* scores, int64_t n, int64_t k, int64_t* idx, void* stream);
/* The certificate of that selection ("rank selections bit-identical" made checkable): with every score known only up to
 * eps * sens[j] (mdg_ridge_scores), can the selected SET differ?  idx [k]: the selection (ascending); out8 (DEVICE, 8 doubles):
 *   [0] largest selected score s_(k)      [1] smallest unselected score s_(k+1)     -> margin = ([1] - [0]) / [0]
 *   [2] max over selected of s + eps sens [3] min over unselected of s - eps sens   -> certified iff [2] < [3]
 *   [4] / [5] largest sens among the selected / unselected   ([1] - [0]) / ([4] + [5]) <= the largest eps that still certifies
 *   [6] how many scores' intervals reach across the midpoint of [0] and [1]        [7] 1.0 if certified else 0.0
 * Only enqueues; the caller reads the 8 doubles when it next waits for the stream (modegpt_amd reads them with the chain's
 * status).  Not in the reference: there the selection is whatever torch.topk makes of the fp64 scores (compress_mlp.py:45-47). */
int mdg_select_margin(const double* scores, const double* sens, const int64_t* idx, int64_t n, int64_t k, double eps,
                      double* out8, void* stream);
/* out[i,:] = src[rows[i],:] for 2-byte elements (W_u[topk,:], W_g[topk,:], compress_mlp.py:49-50;
 * Q/K row gathers, compress_qk.py:375-376). */
int mdg_gather_rows_16(const void* src, int64_t ld_src, const int64_t* rows, int64_t n_rows, int64_t n_cols,
                       void* out, int64_t ld_out, void* stream);
/* down_out [d, r] (bf16, ld_out) = ((C[idx,idx] + eps I)^-1 C[idx,:] W_d^T)^T, W_d [d, n] (ld_wd) of dtype w_dtype:
 * MDG_BF16, or MDG_F64 for checkpoints in another precision (fp16 OPT: the caller widens exactly, as the reference's
 * .to(float64) does).
 * compress_mlp.py:52-62,97.  down_f64 (optional, [r, d] row-major) receives the fp64 solution before the
 * cast.  ws: mdg_nystrom_down_ws_bytes(n, r, d).  SYNCHRONISES. */
size_t mdg_nystrom_down_ws_bytes(int64_t n, int64_t r, int64_t d);
int mdg_nystrom_down(const double* C, int64_t n, int64_t ldc, const int64_t* idx, int64_t r, const void* Wd,
                     int64_t d, int64_t ld_wd, int w_dtype, double eps, void* down_out, int64_t ld_out, double* down_f64,
                     void* ws, size_t ws_bytes, void* stream);
/* The same with the gathered cross product C[idx,:] W_d^T on `side_stream`, beside the factorisation of C_kk on `stream` (they do
 * not depend on each other; the factorisation's 128-column steps leave most of the chip idle between their GEMMs).  Streams and the
 * two events are the CALLER'S (any two hipEvent_t, timing disabled is fine): ev_fork is recorded on `stream` and waited for by
 * `side_stream` before the product, ev_join is recorded behind the product and waited for by `stream` before the solve; when the
 * call returns everything later on `stream` is ordered behind both.  Results are bit-identical to mdg_nystrom_down. */
int mdg_nystrom_down_overlapped(const double* C, int64_t n, int64_t ldc, const int64_t* idx, int64_t r, const void* Wd,
                                int64_t d, int64_t ld_wd, int w_dtype, double eps, void* down_out, int64_t ld_out,
                                double* down_f64, void* ws, size_t ws_bytes, void* side_stream, void* ev_fork, void* ev_join,
                                void* stream);

/* ------------------------------------------------------------------ QK: CR selection
 * mask [n_kv, rank] (int64, score-descending, NOT sorted: compress_qk.py:366-367,418-419,464),
 * q_rows [n_heads*rank] / k_rows [n_kv*rank]: absolute row indices into W_q / W_k for mdg_gather_rows_16.
 * cov_q [n_heads, hd, hd], cov_k [n_kv, hd, hd] fp64.  Scores use ||col_j(sqrt(C + rho I))||^2 = C_jj + rho
 * (exact for symmetric PSD C; DESIGN.md section "Identities").  ROPE modes need even rank. */
int mdg_qk_select(const double* cov_q, const double* cov_k, int n_heads, int n_kv, int hd, double ridge_q,
                  double ridge_k, int rank, int mode, int64_t* mask, int64_t* q_rows, int64_t* k_rows,
                  void* stream);

/* ------------------------------------------------------------------ VO: SVD of sqrt(C) W_v^T
 * v_out [n_kv*rank, d] bf16, o_out [d, n_heads*rank] bf16 (compress_vo.py:89-90).  W_v [n_kv*hd, d],
 * W_o [d, n_heads*hd], both of dtype w_dtype (MDG_BF16 or MDG_F64).  n_kv == n_heads selects the two-SVD MHA variant (compress_vo.py:162-223), else the
 * grouped one (:112-159).  Works on the Gram matrix W_v (C + rho I) W_v^T (DESIGN.md "Identities"), so the
 * d x d eigensolve and inverse of compress_vo.py:43-45 never happen.  v_f64 / o_f64 optional fp64 copies of
 * the factors.  ws: mdg_vo_compress_ws_bytes.  SYNCHRONISES. */
size_t mdg_vo_compress_ws_bytes(int64_t d, int n_heads, int n_kv, int hd);
int mdg_vo_compress(const double* cov_x, int64_t d, int64_t ldc, const void* Wv, int64_t ld_wv, const void* Wo,
                    int64_t ld_wo, int w_dtype, int n_heads, int n_kv, int hd, int rank, double ridge, void* v_out,
                    int64_t ld_v, void* o_out, int64_t ld_o, double* v_f64, double* o_f64, void* ws,
                    size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ sqrt_M (compression_utils.py:15-55)
 * root = V diag(sqrt(max(lambda + ridge*scale, 0))) V^T, inv_root (optional) with the 1e-12 clamp;
 * scale = max eigenvalue if scaled else 1.  n <= 128 and even, batched.  evals_out optional [batch][n]
 * (pre-ridge, descending) for the caller's diagnostics.  SYNCHRONISES. */
int mdg_sqrt_psd_small(const double* M, int64_t n, int64_t batch, double ridge, int scaled, double* root,
                       double* inv_root, double* evals_out, void* ws, size_t ws_bytes, void* stream);
size_t mdg_sqrt_psd_small_ws_bytes(int64_t n, int64_t batch);
/* The same function for one n x n matrix of any size (the d_model-sized call of compress_vo.py:44).
 * evals_out == NULL and scaled == 0 (the plain sqrt_M(M, ridge) call): sqrt(M + ridge I) and its inverse by the coupled
 * Newton-Schulz iteration -- three n^3 fp64-MFMA GEMMs per step, ~25 steps at n = 4096 -- valid because the reference's
 * clamps never bind on a PSD input; an input it cannot certify (indefinite) falls through to the eigen route.
 * Otherwise: two-sided block Jacobi on 64-wide blocks -- 128x128 pair sub-problems through the LDS Jacobi kernel,
 * rotations applied by batched fp64-MFMA GEMMs, blocks rotated round-robin; evals_out [n], UNSORTED (pre-ridge).
 * Not used by this engine's own VO stage (DESIGN.md "Identities").  SYNCHRONISES once per step / sweep. */
size_t mdg_sqrt_psd_large_ws_bytes(int64_t n);
int mdg_sqrt_psd_large(const double* M, int64_t n, int64_t ld, double ridge, int scaled, double* root,
                       double* inv_root, double* evals_out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ compressed-model attention
 * Rotary embedding of a compressed layer's q or k projection, fused with the gather of cos/sin by the layer's rotary
 * mask, the optional Qwen3 masked RMSNorm, and the transpose into the layout attention consumes.  Replaces the eager
 * chains apply_rotary_pos_emb(..., rotary_mask) (src/patchers/LlamaRebuild.py:153-176) and _masked_rms_norm
 * (src/patchers/DenseQwenRebuild.py:262-286).
 *   x    [B*T, n_heads*r] rows of pitch ld_x elements (the projection output, token-major), dtype bf16 / f16 / f32
 *   out  [B, n_heads, T, r] contiguous, same dtype
 *   cos, sin  [Bc, T, hd] same dtype; cs_batch_stride elements between batches, 0 = one table shared by every batch
 *   mask int64 [n_kv, r] -- kept column j of (kv) head g is original column mask[g, j] in [0, hd); query heads use the
 *        row of their kv head (h / (n_heads / n_kv)); the two halves [0, r/2) and [r/2, r) are the rotate_half partners.
 *        NULL = identity (then r must equal hd).  Out-of-range entries are clamped (memory safety only).
 *   norm_w  [hd] same dtype or NULL: y = dtype(norm_w[mask] * (float(x) * rsqrt(mean(float(x)^2 over the r kept columns)
 *        + eps))) is applied before the rotation.
 * Every product and the sum are rounded to the element dtype one op at a time, as torch's eager expression does: the
 * rotation is bit-identical to it; the fp32 sum of squares of the norm is taken in a fixed (16-lane tree) order. */
int mdg_rope_gather(const void* x, int dtype, int64_t ld_x, int64_t B, int64_t T, int n_heads, int n_kv, int r, int hd,
                    const void* cos, const void* sin, int64_t cs_batch_stride, const int64_t* mask,
                    const void* norm_w, double eps, void* out, void* stream);

/* ------------------------------------------------------------------ multi-GPU (the reference is single-process,
 * its unit of independent work is the layer loop of src/run_modegpt.py:107-156)
 * Layers shard over the GPUs of a node, one process per GPU, no data-path exchange until the end: ONE all-gather of the
 * ranks' packed per-layer records (modegpt_amd/sharding.py documents the record: header with shapes, rotary mask, bf16
 * payload; every rank pads to the same bytes_per_rank).  This engine's driver runs that step through torch.distributed
 * (backend "nccl" = RCCL over xGMI); the entry points below are the same step for a host without torch.  RCCL is resolved
 * with dlopen("librccl.so.1") on first use -- no link-time dependency.
 *   mdg_comm_unique_id   rank 0 fills id128 (128 bytes) and hands it to the other ranks by the host's own means (file, env, MPI)
 *   mdg_comm_init        every rank, after hipSetDevice(its GPU): *comm receives the communicator
 *   mdg_allgather_layers recv[r * bytes_per_rank ...] = rank r's send buffer, for every r; device pointers; enqueued on `stream`
 *   mdg_comm_destroy     releases the communicator (NULL is accepted) */
int mdg_comm_unique_id(void* id128);
int mdg_comm_init(void** comm, int world, int rank, const void* id128);
int mdg_allgather_layers(const void* send, void* recv, size_t bytes_per_rank, void* comm, void* stream);
int mdg_comm_destroy(void* comm);

/* ------------------------------------------------------------------ utilities
 * out_bf16[j, i] = bf16(in_f64[i, j])  (transpose + cast with torch's double->float->bf16 rounding) */
int mdg_cast_transpose_f64_bf16(const double* in, int64_t rows, int64_t cols, int64_t ld_in, void* out,
                                int64_t ld_out, void* stream);
/* Raw fp64 MFMA issue-rate probe used by bench.py to state the measured peak next to the spec:
 * returns TFLOP/s over `iters` back-to-back v_mfma_f64_16x16x4_f64 per wave on every CU.  SYNCHRONISES. */
int mdg_probe_mfma_f64(int iters, double* tflops, void* stream);
/* The same for the int8 pipe: TOP/s of back-to-back v_mfma_i32_32x32x32_i8 from registers, two waves per SIMD on every CU, the
 * operands changing from one MFMA to the next -- all zero (random_operands = 0: nothing toggles, the pipe runs at full clock,
 * ~0.97 of the 5 POP/s nominal peak) or random bytes (random_operands = 1: the board sits at its power cap and the clock
 * gives way, ~0.68 of nominal on an MI355X -- the ceiling of ANY int8 kernel on random data, before a single byte is loaded).
 * bench.py states both next to the kernel's achieved rate.  SYNCHRONISES. */
int mdg_probe_mfma_i8(int iters, int random_operands, double* tops, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MODEGPT_HIP_H */
