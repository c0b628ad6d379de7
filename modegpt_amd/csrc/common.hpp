// Shared device/host helpers for the MoDeGPT gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/modegpt_hip.h"

namespace mdg {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned short bf16_t;  // raw bits
typedef unsigned short f16_t;   // raw bits

// Experiment / debug knobs read from the environment exist only in builds with -DMDG_EXPERIMENT (scripts/probes/*.sh,
// scripts/bench_kernels.py); the product library never lets an environment variable change what it computes or prints.
#ifdef MDG_EXPERIMENT
#define MDG_KNOB(name) (getenv(name) != nullptr)
#else
#define MDG_KNOB(name) false
#endif

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);
// A device status word (Cholesky pivot / Jacobi convergence) at the end of a chain of launches.  Normally: one 4-byte copy to
// the host + a stream synchronisation, then `fail` (MDG_ERR_NOT_PD / MDG_ERR_NO_CONVERGE) with the message `what` when it is
// nonzero.  Between mdg_deferred_status_begin / _end the word is merged into the caller's device status instead and MDG_OK is
// returned at once: the chain of a whole layer then enqueues without a single host round trip (api.hip).
enum { STATUS_NOT_PD = 1, STATUS_NO_CONVERGE = 2 };
int finish_flag(int* dflag, hipStream_t st, int kind, const char* what);

// mdg_cov_accum with a device-side gate (cov.hip): the launches are enqueued unconditionally, and every workgroup exits at once
// unless (*gate & gate_mask) == gate_want when it runs (gate == nullptr: always runs).  mdg_cov_accum_i8 enqueues its fp64
// fallback this way, so that the route is chosen on the device and the call never waits for the host.
int release_i8_schedules();   // cov_i8.hip: frees the cached tile schedules (mdg_shutdown)
int cov_accum_gated(const void* x, int dtype, int64_t n_tokens, int64_t n_feat, int64_t batch, int64_t ld, int relu, double* sigma,
                    int64_t ld_sigma, int64_t sigma_bs, void* ws, size_t ws_bytes, const int* gate, int gate_mask, int gate_want,
                    void* stream);
#define MDG_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      mdg::set_error(__VA_ARGS__);          \
      return MDG_ERR_BAD_ARG;               \
    }                                       \
  } while (0)
#define MDG_HIP(call)                                                        \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      mdg::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),  \
                     __FILE__, __LINE__);                                    \
      return MDG_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)
#define MDG_LAUNCH_CHECK() MDG_HIP(hipGetLastError())
// hipGetLastError is sticky per thread: drop whatever an earlier, unrelated HIP call left behind
#define MDG_CLEAR() (void)hipGetLastError()
#define MDG_TRY(call)              \
  do {                             \
    int s_ = (call);               \
    if (s_ != MDG_OK) return s_;   \
  } while (0)

// ---------------------------------------------------------------- conversions
__device__ __forceinline__ double bf16_to_f64(bf16_t b) {
  return (double)__uint_as_float(((unsigned)b) << 16);
}
__device__ __forceinline__ double f16_to_f64(f16_t h) {
  __half_raw r;
  r.x = h;
  return (double)__half2float(__half(r));
}
// fp64 -> bf16 the way torch's .to(bfloat16) does it: double -> float (RNE),
// then float -> bf16 (RNE); NaN stays NaN.
__device__ __forceinline__ bf16_t f64_to_bf16(double v) {
  float f = (float)v;
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

template <int DT> struct ElemOf;
template <> struct ElemOf<MDG_BF16> { typedef bf16_t type; };
template <> struct ElemOf<MDG_F16> { typedef f16_t type; };
template <> struct ElemOf<MDG_F32> { typedef float type; };
template <> struct ElemOf<MDG_F64> { typedef double type; };

template <int DT> __device__ __forceinline__ double load_f64(const void* p, int64_t i);
template <> __device__ __forceinline__ double load_f64<MDG_BF16>(const void* p, int64_t i) {
  return bf16_to_f64(((const bf16_t*)p)[i]);
}
template <> __device__ __forceinline__ double load_f64<MDG_F16>(const void* p, int64_t i) {
  return f16_to_f64(((const f16_t*)p)[i]);
}
template <> __device__ __forceinline__ double load_f64<MDG_F32>(const void* p, int64_t i) {
  return (double)((const float*)p)[i];
}
template <> __device__ __forceinline__ double load_f64<MDG_F64>(const void* p, int64_t i) {
  return ((const double*)p)[i];
}

static inline size_t dtype_size(int dt) {
  switch (dt) {
    case MDG_BF16: case MDG_F16: return 2;
    case MDG_F32: return 4;
    case MDG_F64: return 8;
  }
  return 0;
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------- fp64 MFMA tile core
// One workgroup = 256 threads = 4 waves arranged 2x2; workgroup tile 128x128, wave tile 64x64 built from
// 4x4 v_mfma_f64_16x16x4_f64 sub-tiles.  Operand panels live in LDS as fp64, [k][128 + PAD] (k-major).
//
// Feature interleave: MFMA row m (0..15) of sub-tile s (0..3) is panel column 4*m + s inside the wave's
// 64-wide half, so a lane fetches the operands of all four sub-tiles with one 32-byte LDS read.
// v_mfma_f64_16x16x4_f64 maps (cdna_hip_programming.md section 3): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// D[row = (lane>>4) + 4*reg][col = lane&15].
constexpr int TILE = 128;       // workgroup tile edge
constexpr int WTILE = 64;       // wave tile edge
constexpr int BK = 16;          // k depth of one LDS stage
constexpr int PITCH = TILE + 2; // fp64 elements per LDS panel row (pad keeps 16-B alignment, spreads banks)
constexpr int PANEL = BK * PITCH;

struct Acc {
  d4 v[4][4];  // [sub-tile row sa][sub-tile col sb]
};

__device__ __forceinline__ void acc_zero(Acc& a) {
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) a.v[i][j] = (d4){0., 0., 0., 0.};
}

// Consume k4-steps [K4_LO, K4_HI) of one LDS stage (a stage is BK/4 steps of depth 4).  As/Bs point at the
// stage's panels; wr/wc are the wave's row/col half (0/1).
template <int K4_LO, int K4_HI>
__device__ __forceinline__ void mma_steps(const double* __restrict__ As, const double* __restrict__ Bs, int wr,
                                          int wc, int lane, Acc& acc) {
  const int m = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int k4 = K4_LO; k4 < K4_HI; k4++) {
    const double* ap = As + (k4 * 4 + kq) * PITCH + wr * WTILE + 4 * m;
    const double* bp = Bs + (k4 * 4 + kq) * PITCH + wc * WTILE + 4 * m;
    d2 a01 = *(const d2*)ap, a23 = *(const d2*)(ap + 2);
    d2 b01 = *(const d2*)bp, b23 = *(const d2*)(bp + 2);
    double a[4] = {a01.x, a01.y, a23.x, a23.y};
    double b[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
    for (int sa = 0; sa < 4; sa++)
#pragma unroll
      for (int sb = 0; sb < 4; sb++)
        acc.v[sa][sb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[sa], b[sb], acc.v[sa][sb], 0, 0, 0);
  }
}

__device__ __forceinline__ void mma_stage(const double* __restrict__ As, const double* __restrict__ Bs, int wr,
                                          int wc, int lane, Acc& acc) {
  mma_steps<0, BK / 4>(As, Bs, wr, wc, lane, acc);
}

// Where accumulator element (sa, sb, reg) of this lane sits inside the 128x128 workgroup tile.
__device__ __forceinline__ int acc_row(int wr, int lane, int sa, int reg) {
  return wr * WTILE + 4 * ((lane >> 4) + 4 * reg) + sa;
}
__device__ __forceinline__ int acc_col(int wc, int lane, int sb) { return wc * WTILE + 4 * (lane & 15) + sb; }

// Lower-triangle tile enumeration: t -> (bi, bj), bj <= bi.
__device__ __forceinline__ void tri_decode(int t, int& bi, int& bj) {
  int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= t) i++;
  while (i * (i + 1) / 2 > t) i--;
  bi = i;
  bj = t - i * (i + 1) / 2;
}

}  // namespace mdg
