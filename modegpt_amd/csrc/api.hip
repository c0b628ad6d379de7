// Error plumbing, device query and the fp64-MFMA issue-rate probe of libmodegpt_hip.so.
#include <stdarg.h>

#include "common.hpp"

namespace mdg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- deferred status (see common.hpp finish_flag)
static thread_local int* g_deferred_status = nullptr;   // DEVICE int[2]: {kind of the first failure (STATUS_*), its detail}

__global__ void status_merge_kernel(const int* dflag, int* status, int kind) {
  if (*dflag != 0 && status[0] == 0) {   // the first failure of the chain is the one reported
    status[0] = kind;
    status[1] = *dflag;
  }
}

static int status_to_error(int kind, int detail, const char* what) {
  if (kind == STATUS_NOT_PD) {
    set_error("Cholesky: the factorization could not be completed because the input is not positive-definite "
              "(the leading minor of order %d is not positive-definite)", detail);
    return MDG_ERR_NOT_PD;
  }
  if (kind == STATUS_NO_CONVERGE) {
    set_error("%s: Jacobi eigensolver did not converge in 40 sweeps", what ? what : "eigensolver");
    return MDG_ERR_NO_CONVERGE;
  }
  return MDG_OK;
}

int finish_flag(int* dflag, hipStream_t st, int kind, const char* what) {
  if (g_deferred_status) {
    hipLaunchKernelGGL(status_merge_kernel, dim3(1), dim3(1), 0, st, dflag, g_deferred_status, kind);
    MDG_LAUNCH_CHECK();
    return MDG_OK;
  }
  int flag = 0;
  MDG_HIP(hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, st));
  MDG_HIP(hipStreamSynchronize(st));
  return flag ? status_to_error(kind, flag, what) : MDG_OK;
}

// Each wave issues `iters` x 16 independent-accumulator v_mfma_f64_16x16x4_f64; operands never leave registers.
__global__ __launch_bounds__(256, 2) void probe_mfma_f64_kernel(int iters, double* sink) {
  d4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = (d4){0., 0., 0., 0.};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-3;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.;
#pragma unroll
  for (int i = 0; i < 16; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 123.456) sink[0] = s;  // keep the chain alive without a store on the normal path
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// v_mfma_i32_32x32x32_i8 back to back from registers, the two operands CHANGING from one MFMA to the next (four A and four B
// registers per lane in rotation; random bytes or all zero).  No LDS, no memory: what the matrix pipes and the register file
// alone sustain -- with changing random operands well under the nominal rate, because the board is at its power cap.
__global__ __launch_bounds__(256, 2) void probe_mfma_i8_kernel(int iters, int random, int* sink) {
  i32x4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      unsigned s = (blockIdx.x * 256 + threadIdx.x) * 64 + i * 8 + j, t;
      s ^= s >> 16; s *= 0x7feb352du; s ^= s >> 15; s *= 0x846ca68bu; s ^= s >> 16;
      t = s * 0x9e3779b1u + 0x85ebca6bu; t ^= t >> 13;
      a[i][j] = random ? (int)s : 0;
      b[i][j] = random ? (int)t : 0;
    }
  i32x16 c[8];
#pragma unroll
  for (int i = 0; i < 8; i++) c[i] = (i32x16)0;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 3], b[(i + (i >> 2)) & 3], c[i], 0, 0, 0);
  int x = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) x ^= c[i][r];
  if (x == 0x12345678) sink[0] = x;
}

}  // namespace mdg

using namespace mdg;

extern "C" int mdg_abi_version(void) { return MDG_ABI_VERSION; }
extern "C" const char* mdg_last_error(void) { return g_err; }

extern "C" int mdg_device_info(int device, char* name, int cap, int* n_cu, int64_t* hbm_bytes) {
  MDG_CLEAR();
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device >= count) {
    set_error("mdg_device_info: no HIP device %d (count %d)", device, count);
    return MDG_ERR_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  MDG_HIP(hipGetDeviceProperties(&prop, device));
  if (name && cap > 0) {
    strncpy(name, prop.gcnArchName, cap - 1);
    name[cap - 1] = 0;
  }
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return MDG_OK;
}

extern "C" int mdg_deferred_status_begin(int* status_dev, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(status_dev, "mdg_deferred_status_begin: null status pointer");
  MDG_CHECK_ARG(!mdg::g_deferred_status, "mdg_deferred_status_begin: already in deferred mode on this thread");
  MDG_HIP(hipMemsetAsync(status_dev, 0, 2 * sizeof(int), (hipStream_t)stream));
  mdg::g_deferred_status = status_dev;
  return MDG_OK;
}

extern "C" int mdg_deferred_status_end(void) {
  MDG_CLEAR();
  mdg::g_deferred_status = nullptr;
  return MDG_OK;
}

extern "C" int mdg_deferred_status_decode(const int* status_host) {
  MDG_CLEAR();
  MDG_CHECK_ARG(status_host, "mdg_deferred_status_decode: null pointer");
  return mdg::status_to_error(status_host[0], status_host[1], "decomposition chain");
}

extern "C" int mdg_shutdown(void) {
  MDG_CLEAR();
  return mdg::release_i8_schedules();
}

extern "C" int mdg_probe_mfma_f64(int iters, double* tflops, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(iters > 0 && tflops, "mdg_probe_mfma_f64: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int dev = 0, n_cu = 0;
  MDG_HIP(hipGetDevice(&dev));
  MDG_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  double* sink = nullptr;
  MDG_HIP(hipMalloc((void**)&sink, 64));
  hipEvent_t e0, e1;
  MDG_HIP(hipEventCreate(&e0));
  MDG_HIP(hipEventCreate(&e1));
  const int blocks = n_cu * 2;  // 8 waves per CU = 2 per SIMD
  hipLaunchKernelGGL(probe_mfma_f64_kernel, dim3(blocks), dim3(256), 0, st, iters, sink);  // warm
  float ms = 1e30f;
  for (int rep = 0; rep < 3; rep++) {  // best of three: the clock ramps under load
    MDG_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(probe_mfma_f64_kernel, dim3(blocks), dim3(256), 0, st, iters, sink);
    MDG_HIP(hipEventRecord(e1, st));
    MDG_HIP(hipEventSynchronize(e1));
    float t = 0.f;
    MDG_HIP(hipEventElapsedTime(&t, e0, e1));
    if (t < ms) ms = t;
  }
  const double flop = (double)blocks * 4 /*waves*/ * (double)iters * 16 * (2.0 * 16 * 16 * 4);
  *tflops = flop / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  return MDG_OK;
}

extern "C" int mdg_probe_mfma_i8(int iters, int random_operands, double* tops, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(iters > 0 && tops, "mdg_probe_mfma_i8: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int dev = 0, n_cu = 0;
  MDG_HIP(hipGetDevice(&dev));
  MDG_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  int* sink = nullptr;
  MDG_HIP(hipMalloc((void**)&sink, 64));
  hipEvent_t e0, e1;
  MDG_HIP(hipEventCreate(&e0));
  MDG_HIP(hipEventCreate(&e1));
  const int blocks = n_cu * 2;  // 8 waves per CU = 2 per SIMD
  hipLaunchKernelGGL(probe_mfma_i8_kernel, dim3(blocks), dim3(256), 0, st, iters / 4 + 1, random_operands, sink);  // warm: lets the clock settle
  MDG_HIP(hipEventRecord(e0, st));
  hipLaunchKernelGGL(probe_mfma_i8_kernel, dim3(blocks), dim3(256), 0, st, iters, random_operands, sink);
  MDG_HIP(hipEventRecord(e1, st));
  MDG_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MDG_HIP(hipEventElapsedTime(&ms, e0, e1));
  const double op = (double)blocks * 4 /*waves*/ * (double)iters * 8 * (2.0 * 32 * 32 * 32);
  *tops = op / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  return MDG_OK;
}
