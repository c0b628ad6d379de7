// What can an elementwise read-once / write-once kernel reach on this part, and what does the rope-gather access
// pattern (176-byte row pieces, [B,T,H*r] -> [B,H,T,r]) cost against a flat copy?   hipcc --offload-arch=gfx950 -O3
//   mode 0: flat copy, 16 B per lane, grid-stride
//   mode 1: flat copy, 8 B per lane, one pack per thread (no loop)            -- latency / occupancy bound?
//   mode 2: rope pattern, 16-lane groups, 4 heads per group, 8-B packs, 11/16 lanes live, no gathers, no math
//   mode 3: mode 2 + transposing store only (reads flat)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short u16;
struct alignas(8) P4 { u16 v[4]; };
__global__ __launch_bounds__(256) void flat16(const uint4* a, uint4* o, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i];
}
__global__ __launch_bounds__(256) void flat8(const P4* a, P4* o, size_t n) {
  size_t i = blockIdx.x * 256ull + threadIdx.x;
  if (i < n) o[i] = a[i];
}
// x [B*T, H*r] -> out [B, H, T, r];  block = 16 tokens x (kv head, 4 q heads)
//   mode 4: mode 2 + 8 mask dword loads per lane (feeding the stored value)
//   mode 5: mode 2 + 64 VGPRs held live across the loads (occupancy 8 -> 5 waves per SIMD)
//   mode 6: mode 2 + cos/sin rows staged through LDS + __syncthreads + 16 ds_read_u16 gathers
template <bool RD_PATTERN, int EXTRA = 0>
__global__ __launch_bounds__(256) void ropepat(const u16* x, u16* out, int B, int T, int H, int r, const int64_t* mask = nullptr,
                                               const u16* cs = nullptr) {
  const int half = r / 2, chunks = H / 4;
  const int64_t id = blockIdx.x, seq = id / 8;
  const int64_t tile = (seq / chunks) * 8 + id % 8;
  const int h0 = (int)(seq % chunks) * 4;
  const int ttiles = T / 16;
  if (tile >= (int64_t)B * ttiles) return;
  const int64_t b = tile / ttiles, t = (tile % ttiles) * 16 + threadIdx.x / 16;
  const int l = threadIdx.x % 16, j0 = l * 4;
  __shared__ u16 lds[16 * 256];
  if (EXTRA == 6) {
    *(uint4*)(lds + (threadIdx.x / 16) * 256 + l * 8) = *(const uint4*)(cs + t * 128 + l * 8);
    *(uint4*)(lds + (threadIdx.x / 16) * 256 + 128 + l * 8) = *(const uint4*)(cs + T * 128 + t * 128 + l * 8);
  }
  const bool on = j0 < half;
  if (EXTRA != 6 && !on) return;
  const int jc = on ? j0 : 0;
  float pad[64];
  if (EXTRA == 5) {
#pragma unroll
    for (int i = 0; i < 64; i++) { pad[i] = (float)(threadIdx.x + i); asm volatile("" : "+v"(pad[i])); }
  }
  int m[8];
  if (EXTRA == 4 || EXTRA == 6) {
    const int64_t* mrow = mask + (h0 / 4) * r;
#pragma unroll
    for (int v = 0; v < 4; v++) { m[v] = (int)mrow[jc + v] & 127; m[4 + v] = (int)mrow[half + jc + v] & 127; }
  }
  P4 p1[4], p2[4];
  if (on) {
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const u16* row = RD_PATTERN ? x + ((b * T + t) * H + h0 + u) * r : x + (((b * H + h0 + u) * T + t)) * r;
    p1[u] = *(const P4*)(row + j0);
    p2[u] = *(const P4*)(row + half + j0);
  }
  }
  if (EXTRA == 6) {
    __syncthreads();
    const u16* cr = lds + (threadIdx.x / 16) * 256;
    u16 acc = 0;
#pragma unroll
    for (int v = 0; v < 8; v++) acc ^= cr[m[v]] ^ cr[128 + m[v]];
    if (acc == 0x1234) p1[0].v[0] = 7;
    if (!on) return;
  }
  if (EXTRA == 4) {
#pragma unroll
    for (int v = 0; v < 8; v++) if (m[v] == 999) p1[0].v[0] = 7;
  }
  if (EXTRA == 5) {
#pragma unroll
    for (int i = 0; i < 64; i++) asm volatile("" ::"v"(pad[i]));
  }
#pragma unroll
  for (int u = 0; u < 4; u++) {
    u16* row = out + ((b * H + h0 + u) * T + t) * r;
    *(P4*)(row + j0) = p1[u];
    *(P4*)(row + half + j0) = p2[u];
  }
}
int main() {
  const int B = 16, T = 2048, H = 32, r = 88;
  const size_t elems = (size_t)B * T * H * r, bytes = elems * 2;
  u16 *a, *o;
  hipMalloc(&a, bytes); hipMalloc(&o, bytes);
  hipMemset(a, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int64_t* mask; hipMalloc(&mask, 8 * r * 8); hipMemset(mask, 0, 8 * r * 8);
  u16* cs; hipMalloc(&cs, 2 * T * 128 * 2); hipMemset(cs, 1, 2 * T * 128 * 2);
  for (int mode = 0; mode < 7; mode++) {
    float best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(flat16, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)a, (uint4*)o, bytes / 16);
      if (mode == 1) hipLaunchKernelGGL(flat8, dim3((unsigned)((bytes / 8 + 255) / 256)), dim3(256), 0, 0, (const P4*)a, (P4*)o, bytes / 8);
      if (mode == 2) hipLaunchKernelGGL(ropepat<true>, dim3(B * (T / 16) * (H / 4)), dim3(256), 0, 0, a, o, B, T, H, r);
      if (mode == 3) hipLaunchKernelGGL(ropepat<false>, dim3(B * (T / 16) * (H / 4)), dim3(256), 0, 0, a, o, B, T, H, r);
      if (mode == 4) hipLaunchKernelGGL((ropepat<true, 4>), dim3(B * (T / 16) * (H / 4)), dim3(256), 0, 0, a, o, B, T, H, r, mask, cs);
      if (mode == 5) hipLaunchKernelGGL((ropepat<true, 5>), dim3(B * (T / 16) * (H / 4)), dim3(256), 0, 0, a, o, B, T, H, r, mask, cs);
      if (mode == 6) hipLaunchKernelGGL((ropepat<true, 6>), dim3(B * (T / 16) * (H / 4)), dim3(256), 0, 0, a, o, B, T, H, r, mask, cs);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("mode %d: %.1f us  %.0f GB/s (read + write)\n", mode, best * 1e3, 2.0 * bytes / best / 1e6);
  }
  return 0;
}
