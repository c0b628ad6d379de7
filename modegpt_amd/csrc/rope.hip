// Compressed-head rotary embedding (+ Qwen3 masked RMSNorm) -- the inference-side semantics of a compressed
// checkpoint, replacing the eager chain of the reference's patched modeling files:
//   src/patchers/LlamaRebuild.py:153-176      cos/sin gathered by the layer's rotary mask, rotate_half over the KEPT columns
//   src/patchers/DenseQwenRebuild.py:262-286  RMSNorm over the kept columns with the norm weight gathered by the same mask
// The eager chain materialises cos/sin per head ([B, n_heads, T, r]), a rotated copy, two products and a sum: about ten
// passes over HBM.  Here each element is read once and written once, already in the [B, n_heads, T, r] layout attention
// consumes.  HBM-bound: algorithmic bytes = 2 * B*T*n_heads*r * sizeof(elt); cos/sin/mask/weights stay in L2.
//
// Arithmetic follows torch's eager semantics for the tensor dtype, op by op: every product and the final sum is computed in
// fp32 and rounded to the element type before the next op uses it (no FMA contraction), so the RoPE path is bit-identical
// to the reference expression; only the fp32 sum of squares of the norm has an order torch does not define.
#include "common.hpp"

// hipcc contracts a*b+c into an FMA by default, and HIP's __fmul_rn / __fadd_rn are plain operators compiled under that
// default: with them the fp32 route would skip the rounding of the products that torch's separate mul / add kernels
// perform.  The arithmetic below therefore uses its own operators, defined after contraction is switched off.
#pragma STDC FP_CONTRACT OFF

namespace mdg {
namespace {

__device__ __forceinline__ float mul_r(float a, float b) { return a * b; }
__device__ __forceinline__ float add_r(float a, float b) { return a + b; }

template <int DT> struct Lp;
template <> struct Lp<MDG_BF16> {
  typedef bf16_t T;
  static __device__ __forceinline__ float up(T v) { return __uint_as_float(((unsigned)v) << 16); }
  static __device__ __forceinline__ T down(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (T)0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (T)(u >> 16);
  }
};
template <> struct Lp<MDG_F16> {
  typedef f16_t T;
  static __device__ __forceinline__ float up(T v) {
    __half_raw r;
    r.x = v;
    return __half2float(__half(r));
  }
  static __device__ __forceinline__ T down(float f) { return __half_raw(__float2half_rn(f)).x; }
};
template <> struct Lp<MDG_F32> {
  typedef float T;
  static __device__ __forceinline__ float up(T v) { return v; }
  static __device__ __forceinline__ T down(float f) { return f; }
};

constexpr int GROUP = 16;      // lanes that share one (token, head) row
constexpr int ROWS = 16;       // rows (= consecutive tokens of one head) per 256-thread workgroup
constexpr int MAX_HALF = 128;  // r / 2 <= 128  (head_dim <= 256)
constexpr int N_XCD = 8;

struct RopeArgs {
  const void* x;
  int64_t ld_x;
  int64_t B, T;
  int n_heads, group, r, hd;
  const void* cos;
  const void* sin;
  int64_t cs_bstride;
  const int64_t* mask;
  const void* norm_w;
  float eps;
  void* out;
  int64_t n_tiles;  // B * ceil(T / ROWS)
  int t_tiles;
};

template <typename E, int VEC> struct alignas(sizeof(E) * VEC) Pack { E v[VEC]; };

// blockIdx -> (token tile, head).  Workgroups are dealt round-robin to the 8 XCDs; all heads of one token tile are given
// to the same XCD, back to back, so the pieces of a token's row that share a 128-B line meet in one L2.
template <int DT, int VEC, bool NORM>
__global__ __launch_bounds__(256) void rope_gather_kernel(RopeArgs a) {
  typedef typename Lp<DT>::T E;
  typedef Pack<E, VEC> P;
  constexpr int ITMAX = MAX_HALF / (GROUP * VEC);
  const int64_t id = blockIdx.x;
  const int64_t seq = id / N_XCD;
  const int64_t tile = (seq / a.n_heads) * N_XCD + (id % N_XCD);
  const int h = (int)(seq % a.n_heads);
  if (tile >= a.n_tiles) return;
  const int64_t b = tile / a.t_tiles;
  const int64_t t = (tile % a.t_tiles) * ROWS + (threadIdx.x / GROUP);
  if (t >= a.T) return;  // whole 16-lane groups leave together
  const int l = threadIdx.x % GROUP;
  const int half = a.r >> 1;
  const int hk = h / a.group;

  const E* xrow = (const E*)a.x + (b * a.T + t) * a.ld_x + (int64_t)h * a.r;
  const E* crow = (const E*)a.cos + b * a.cs_bstride + t * a.hd;
  const E* srow = (const E*)a.sin + b * a.cs_bstride + t * a.hd;
  const int64_t* mrow = a.mask ? a.mask + (int64_t)hk * a.r : nullptr;
  E* orow = (E*)a.out + ((b * a.n_heads + h) * a.T + t) * a.r;

  float y1[ITMAX][VEC], y2[ITMAX][VEC];
  int m1[ITMAX][VEC], m2[ITMAX][VEC];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < ITMAX; it++) {
    const int j0 = (it * GROUP + l) * VEC;
    if (j0 < half) {
      const P p1 = *(const P*)(xrow + j0), p2 = *(const P*)(xrow + half + j0);
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        y1[it][v] = Lp<DT>::up(p1.v[v]);
        y2[it][v] = Lp<DT>::up(p2.v[v]);
        int64_t i1 = mrow ? mrow[j0 + v] : (int64_t)(j0 + v);
        int64_t i2 = mrow ? mrow[half + j0 + v] : (int64_t)(half + j0 + v);
        m1[it][v] = (int)min(max(i1, (int64_t)0), (int64_t)a.hd - 1);  // memory safety; the binding validates masks
        m2[it][v] = (int)min(max(i2, (int64_t)0), (int64_t)a.hd - 1);
        if (NORM) ss = add_r(ss, add_r(mul_r(y1[it][v], y1[it][v]), mul_r(y2[it][v], y2[it][v])));
      }
    }
  }
  float inv = 1.f;
  if (NORM) {
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1) ss = add_r(ss, __shfl_xor(ss, o, GROUP));
    inv = __fdiv_rn(1.f, __fsqrt_rn(add_r(__fdiv_rn(ss, (float)a.r), a.eps)));
  }
#pragma unroll
  for (int it = 0; it < ITMAX; it++) {
    const int j0 = (it * GROUP + l) * VEC;
    if (j0 < half) {
      P o1, o2;
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        float a1 = y1[it][v], a2 = y2[it][v];
        if (NORM) {  // (gathered_weight * (x_float * rsqrt(var + eps))).to(dtype)
          const E* w = (const E*)a.norm_w;
          a1 = Lp<DT>::up(Lp<DT>::down(mul_r(Lp<DT>::up(w[m1[it][v]]), mul_r(a1, inv))));
          a2 = Lp<DT>::up(Lp<DT>::down(mul_r(Lp<DT>::up(w[m2[it][v]]), mul_r(a2, inv))));
        }
        const float c1 = Lp<DT>::up(crow[m1[it][v]]), s1 = Lp<DT>::up(srow[m1[it][v]]);
        const float c2 = Lp<DT>::up(crow[m2[it][v]]), s2 = Lp<DT>::up(srow[m2[it][v]]);
        // q * cos + rotate_half(q) * sin, rotate_half(q) = cat(-q[half:], q[:half])
        const float u1 = Lp<DT>::up(Lp<DT>::down(mul_r(a1, c1)));
        const float w1 = Lp<DT>::up(Lp<DT>::down(mul_r(-a2, s1)));
        const float u2 = Lp<DT>::up(Lp<DT>::down(mul_r(a2, c2)));
        const float w2 = Lp<DT>::up(Lp<DT>::down(mul_r(a1, s2)));
        o1.v[v] = Lp<DT>::down(add_r(u1, w1));
        o2.v[v] = Lp<DT>::down(add_r(u2, w2));
      }
      *(P*)(orow + j0) = o1;
      *(P*)(orow + half + j0) = o2;
    }
  }
}

template <int DT, int VEC> void launch_rope(const RopeArgs& a, dim3 grid, hipStream_t st) {
  if (a.norm_w)
    hipLaunchKernelGGL((rope_gather_kernel<DT, VEC, true>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((rope_gather_kernel<DT, VEC, false>), grid, dim3(256), 0, st, a);
}

template <int DT> void launch_rope_vec(const RopeArgs& a, int vec, dim3 grid, hipStream_t st) {
  if (vec == 4) launch_rope<DT, 4>(a, grid, st);
  else if (vec == 2) launch_rope<DT, 2>(a, grid, st);
  else launch_rope<DT, 1>(a, grid, st);
}

}  // namespace
}  // namespace mdg

using namespace mdg;

extern "C" int mdg_rope_gather(const void* x, int dtype, int64_t ld_x, int64_t B, int64_t T, int n_heads, int n_kv, int r,
                               int hd, const void* cos, const void* sin, int64_t cs_batch_stride, const int64_t* mask,
                               const void* norm_w, double eps, void* out, void* stream) {
  MDG_CLEAR();
  MDG_CHECK_ARG(dtype == MDG_BF16 || dtype == MDG_F16 || dtype == MDG_F32, "mdg_rope_gather: dtype %d (bf16, f16 or f32)",
                dtype);
  MDG_CHECK_ARG(B >= 0 && T >= 0, "mdg_rope_gather: negative extent");
  MDG_CHECK_ARG(n_heads > 0 && n_kv > 0 && n_heads % n_kv == 0, "mdg_rope_gather: n_heads %d must be a multiple of n_kv %d",
                n_heads, n_kv);
  MDG_CHECK_ARG(r >= 2 && r % 2 == 0 && r <= hd && r <= 2 * MAX_HALF,
                "mdg_rope_gather: kept width r=%d must be even, >= 2, <= head_dim=%d and <= %d", r, hd, 2 * MAX_HALF);
  MDG_CHECK_ARG(mask || r == hd, "mdg_rope_gather: no mask given but r=%d != head_dim=%d", r, hd);
  MDG_CHECK_ARG(ld_x >= (int64_t)n_heads * r, "mdg_rope_gather: ld_x=%lld < n_heads*r=%lld", (long long)ld_x,
                (long long)n_heads * r);
  MDG_CHECK_ARG(cs_batch_stride == 0 || cs_batch_stride >= T * hd, "mdg_rope_gather: cos/sin batch stride %lld",
                (long long)cs_batch_stride);
  if (B == 0 || T == 0) return MDG_OK;
  MDG_CHECK_ARG(x && cos && sin && out, "mdg_rope_gather: null pointer");
  RopeArgs a;
  a.x = x; a.ld_x = ld_x; a.B = B; a.T = T; a.n_heads = n_heads; a.group = n_heads / n_kv; a.r = r; a.hd = hd;
  a.cos = cos; a.sin = sin; a.cs_bstride = cs_batch_stride; a.mask = mask; a.norm_w = norm_w; a.eps = (float)eps; a.out = out;
  a.t_tiles = (int)ceil_div(T, ROWS);
  a.n_tiles = B * a.t_tiles;
  const int64_t blocks = ceil_div(a.n_tiles, N_XCD) * N_XCD * n_heads;
  MDG_CHECK_ARG(blocks < (int64_t)1 << 31, "mdg_rope_gather: %lld workgroups exceed the grid limit", (long long)blocks);
  const size_t es = dtype_size(dtype);
  const int half = r / 2;
  int vec = 4;
  while (vec > 1 && (half % vec || ld_x % vec || ((uintptr_t)x | (uintptr_t)out) % (es * vec))) vec >>= 1;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)blocks);
  if (dtype == MDG_BF16) launch_rope_vec<MDG_BF16>(a, vec, grid, st);
  else if (dtype == MDG_F16) launch_rope_vec<MDG_F16>(a, vec, grid, st);
  else launch_rope_vec<MDG_F32>(a, vec, grid, st);
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}
