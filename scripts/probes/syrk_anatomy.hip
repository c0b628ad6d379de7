// Which ingredient of the cov_accum stage loop costs MFMA issue slots?  Same 4-wave / 64x64-per-wave / 16 MFMAs per
// k4-step skeleton, ingredients switched on one at a time.  FLAGS: 1 = operands re-read from LDS every step,
// 2 = one s_barrier per 32-MFMA stage, 4 = 4 x (2 cvt + ds_write_b128) per stage, 8 = one 16-B global load per stage.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int PITCH = 130;

template <int FLAGS>
__global__ __launch_bounds__(256, 2) void k(int stages, const uint4* __restrict__ g, double* sink) {
  __shared__ double lds[3 * 2 * 8 * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int m = lane & 15, kq = lane >> 4;
  for (int i = tid; i < 3 * 2 * 8 * PITCH; i += 256) lds[i] = 1.0 + i * 1e-6;
  __syncthreads();
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (d4){0., 0., 0., 0.};
  double a[4] = {1., 2., 3., 4.}, b[4] = {.5, .25, .125, 2.};
  uint4 r = make_uint4(0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80);
  const uint4* gp = g + (size_t)blockIdx.x * 256 + tid;
  int cur = 0;
  for (int s = 0; s < stages; s++) {
    uint4 rn = r;
    if (FLAGS & 8) rn = gp[(size_t)(s & 1023) * 256 * 1024];
#pragma unroll
    for (int k4 = 0; k4 < 2; k4++) {
      if (FLAGS & 1) {
        const double* ap = lds + cur * 2 * 8 * PITCH + (k4 * 4 + kq) * PITCH + wr * 64 + 4 * m;
        const double* bp = lds + cur * 2 * 8 * PITCH + 8 * PITCH + (k4 * 4 + kq) * PITCH + wc * 64 + 4 * m;
        d2 a01 = *(const d2*)ap, a23 = *(const d2*)(ap + 2), b01 = *(const d2*)bp, b23 = *(const d2*)(bp + 2);
        a[0] = a01.x; a[1] = a01.y; a[2] = a23.x; a[3] = a23.y;
        b[0] = b01.x; b[1] = b01.y; b[2] = b23.x; b[3] = b23.y;
      }
      if (k4 == 1 && (FLAGS & 2)) {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int sa = 0; sa < 4; sa++) {
#pragma unroll
        for (int sb = 0; sb < 4; sb++) acc[sa][sb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[sa], b[sb], acc[sa][sb], 0, 0, 0);
        if (k4 == 1 && (FLAGS & 4)) {
          unsigned w = sa == 0 ? r.x : sa == 1 ? r.y : sa == 2 ? r.z : r.w;
          double v0 = (double)__uint_as_float(w << 16), v1 = (double)__uint_as_float(w & 0xffff0000u);
          int wrt = (cur + 2) % 3;
          *(d2*)(lds + wrt * 2 * 8 * PITCH + (tid >> 4) * PITCH / 2 * 0 + ((tid * 8 + sa * 2) % (2 * 8 * PITCH - 2))) = (d2){v0, v1};
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    r = rn;
    cur = cur == 2 ? 0 : cur + 1;
  }
  double t = 0.;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) t += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
  if (t == 123.456) sink[0] = t;
}

template <int F> void run(int blocks, int stages, const uint4* g, double* sink) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(256), 0, 0, stages, g, sink);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double fl = (double)blocks * 4 * stages * 32 * 2048.0;
  printf("flags %2d blocks %4d: %.3f ms  %.1f TF\n", F, blocks, best, fl / best / 1e9);
}

int main() {
  uint4* g; double* sink;
  (void)hipMalloc((void**)&g, (size_t)1024 * 256 * 1024 * 16 + (size_t)512 * 256 * 16 + 4096);  // 4 GB stream
  (void)hipMemset(g, 0x3f, (size_t)1024 * 256 * 1024 * 16);
  (void)hipMalloc((void**)&sink, 64);
  const int stages = 4096;
  for (int blocks : {512, 256}) {
    run<0>(blocks, stages, g, sink);
    run<1>(blocks, stages, g, sink);
    run<3>(blocks, stages, g, sink);
    run<5>(blocks, stages, g, sink);
    run<7>(blocks, stages, g, sink);
    run<15>(blocks, stages, g, sink);
  }
  return 0;
}
