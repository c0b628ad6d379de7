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

// Sum over the 16 lanes of a DPP row, every lane ending with the same value: two quad permutes, then the row's half mirror and
// full mirror (fp32 addition commutes, so the two sides of every exchange compute identical sums).
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float s) {
  s = s + dpp_f32<0xB1>(s);   // quad_perm [1,0,3,2]
  s = s + dpp_f32<0x4E>(s);   // quad_perm [2,3,0,1]
  s = s + dpp_f32<0x141>(s);  // row_half_mirror
  s = s + dpp_f32<0x140>(s);  // row_mirror
  return s;
}

__device__ __forceinline__ float mul_r(float a, float b) { return a * b; }
__device__ __forceinline__ float add_r(float a, float b) { return a + b; }

template <int DT> struct Lp;
template <> struct Lp<MDG_BF16> {
  typedef bf16_t T;
  static __device__ __forceinline__ float up(T v) { return __uint_as_float(((unsigned)v) << 16); }
  // v_cvt_pk_bf16_f32: round to nearest even in one instruction
  static __device__ __forceinline__ T down(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
};
template <> struct Lp<MDG_F16> {
  typedef f16_t T;
  static __device__ __forceinline__ float up(T v) { return (float)__builtin_bit_cast(_Float16, v); }
  static __device__ __forceinline__ T down(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
};
template <> struct Lp<MDG_F32> {
  typedef float T;
  static __device__ __forceinline__ float up(T v) { return v; }
  static __device__ __forceinline__ T down(float f) { return f; }
};

constexpr int GROUP = 16;      // lanes that share one (token, head) row
constexpr int ROWS = 16;       // thread groups per 256-thread workgroup
constexpr int UNITS = 4;       // rows in flight per thread group: HPT heads x TPT tokens
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
  unsigned n_tiles; // B * t_tiles
  int t_tiles;      // ceil(T / (ROWS * TPT))
  int n_kv, chunks; // kv heads; head chunks per kv head = group / HPT
  int cs_vec16;     // cos / sin rows are 16-byte aligned and a multiple of 16 bytes long
  int nw_vec16;     // so is the norm weight
};

template <typename E, int VEC> struct alignas(sizeof(E) * VEC) Pack { E v[VEC]; };
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // 16-byte register quad (HIP's uint4 is a union: it ends up in scratch)

// One workgroup = 16 thread groups of 16 lanes.  A thread group owns UNITS = HPT x TPT rows: HPT query heads of one kv
// head (they share the kv head's mask row, hence the gathered cos / sin) for each of TPT tokens, and has all of their
// loads in flight before it computes -- an elementwise kernel at 8 TB/s needs ~16 MB outstanding, a single 176-byte row
// per 16 lanes does not get there.  blockIdx -> (token tile, kv head, head chunk): workgroups are dealt round-robin to
// the 8 XCDs, and every head of one token tile goes to the same XCD, back to back, so the pieces of a token's row that
// share a 128-B line meet in one L2.
template <int DT, int VEC, bool NORM, int HPT>
__global__ __launch_bounds__(256) void rope_gather_kernel(RopeArgs a) {
  typedef typename Lp<DT>::T E;
  typedef Pack<E, VEC> P;
  constexpr int TPT = UNITS / HPT;
  // grid = (8 * n_kv, ceil(n_tiles / 8), chunks): the low 3 bits of blockIdx.x are the XCD the workgroup lands on
  const unsigned tile = blockIdx.y * N_XCD + (blockIdx.x & (N_XCD - 1));
  if (tile >= a.n_tiles) return;
  const int hk = blockIdx.x >> 3;
  const int h0 = hk * a.group + blockIdx.z * HPT;
  const unsigned bu = tile / (unsigned)a.t_tiles;
  const int64_t b = bu;
  const int64_t t0 = (int64_t)(tile - bu * (unsigned)a.t_tiles) * (ROWS * TPT) + (threadIdx.x / GROUP);
  const int l = threadIdx.x % GROUP;
  const int half = a.r >> 1;
  const int iters = (half + GROUP * VEC - 1) / (GROUP * VEC);
  const int64_t* mrow = a.mask ? a.mask + (int64_t)hk * a.r : nullptr;
  const E* nw = (const E*)a.norm_w;

  // rows of this thread group: unit u = tt * HPT + hh
  const E* xrow[UNITS];
  E* orow[UNITS];
  bool live[TPT];
#pragma unroll
  for (int tt = 0; tt < TPT; tt++) {
    const int64_t t = t0 + tt * ROWS;
    live[tt] = t < a.T;
    const int64_t tc = live[tt] ? t : 0;
#pragma unroll
    for (int hh = 0; hh < HPT; hh++) {
      xrow[tt * HPT + hh] = (const E*)a.x + (b * a.T + tc) * a.ld_x + (int64_t)(h0 + hh) * a.r;
      orow[tt * HPT + hh] = (E*)a.out + ((b * a.n_heads + h0 + hh) * a.T + tc) * a.r;
    }
  }

  float inv[UNITS];
#pragma unroll
  for (int u = 0; u < UNITS; u++) inv[u] = 1.f;
  if (NORM && iters > 1) {  // wide heads / narrow packs: the row does not fit one pass, take the sums first
#pragma unroll
    for (int u = 0; u < UNITS; u++) {
      float ss = 0.f;
      if (live[u / HPT])
        for (int j = l; j < a.r; j += GROUP) {
          const float v = Lp<DT>::up(xrow[u][j]);
          ss = add_r(ss, mul_r(v, v));
        }
      ss = row16_sum(ss);
      inv[u] = 1.f / sqrtf(add_r(ss / (float)a.r, a.eps));
    }
  }

  // cos / sin rows of this thread group's tokens (and the norm weight) -> LDS, so that the gather of the kept columns is
  // an LDS read and costs no second trip to memory.  Common case (a row is 256 bytes: head_dim 128, half types): one
  // 16-byte load per lane and row, issued here, written to LDS only after the first packs and the mask entries are in
  // flight too.  Everything the kernel loads is addressed unconditionally (out-of-range tokens / columns are clamped, only
  // the stores are predicated): a predicated load makes the compiler drain the memory queue at every branch.
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  E* cs_lds = (E*)lds_raw;
  E* nw_lds = cs_lds + (size_t)ROWS * TPT * 2 * a.hd;
  constexpr int PER16 = 16 / (int)sizeof(E);
  const bool one_shot = a.cs_vec16 && a.hd == GROUP * PER16;
  u32x4 creg[TPT], sreg[TPT], wreg;
  if (one_shot) {
#pragma unroll
    for (int tt = 0; tt < TPT; tt++) {
      const int64_t t = live[tt] ? t0 + tt * ROWS : 0;
      creg[tt] = *(const u32x4*)((const E*)a.cos + b * a.cs_bstride + t * a.hd + l * PER16);
      sreg[tt] = *(const u32x4*)((const E*)a.sin + b * a.cs_bstride + t * a.hd + l * PER16);
    }
    if (NORM && a.nw_vec16) wreg = *(const u32x4*)(nw + l * PER16);
  } else {
#pragma unroll
    for (int tt = 0; tt < TPT; tt++) {
      const int64_t t = live[tt] ? t0 + tt * ROWS : 0;
      const E* crow = (const E*)a.cos + b * a.cs_bstride + t * a.hd;
      const E* srow = (const E*)a.sin + b * a.cs_bstride + t * a.hd;
      E* dst = cs_lds + (size_t)((threadIdx.x / GROUP) * TPT + tt) * 2 * a.hd;
      for (int c = l; c < a.hd; c += GROUP) {
        dst[c] = crow[c];
        dst[a.hd + c] = srow[c];
      }
    }
  }
  if (NORM && !(one_shot && a.nw_vec16))
    for (int c = threadIdx.x; c < a.hd; c += 256) nw_lds[c] = nw[c];

  for (int it = 0; it < iters; it++) {
    const int j0 = (it * GROUP + l) * VEC;
    const bool on = j0 < half;
    const int jc = on ? j0 : 0;
    // 1. every row's two packs, all in flight together
    P p1[UNITS], p2[UNITS];
#pragma unroll
    for (int u = 0; u < UNITS; u++) {
      p1[u] = *(const P*)(xrow[u] + jc);
      p2[u] = *(const P*)(xrow[u] + half + jc);
    }
    // 2. mask entries of these columns (shared by all rows)
    int m1[VEC], m2[VEC];
    if (mrow) {
      const Pack<int64_t, VEC> q1 = *(const Pack<int64_t, VEC>*)(mrow + jc), q2 = *(const Pack<int64_t, VEC>*)(mrow + half + jc);
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        m1[v] = min(max((int)q1.v[v], 0), a.hd - 1);  // memory safety only; the binding validates masks
        m2[v] = min(max((int)q2.v[v], 0), a.hd - 1);
      }
    } else {
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        m1[v] = jc + v;
        m2[v] = half + jc + v;
      }
    }
    if (it == 0) {
      if (one_shot) {
#pragma unroll
        for (int tt = 0; tt < TPT; tt++) {
          E* dst = cs_lds + (size_t)((threadIdx.x / GROUP) * TPT + tt) * 2 * a.hd;
          *(u32x4*)(dst + l * PER16) = creg[tt];
          *(u32x4*)(dst + a.hd + l * PER16) = sreg[tt];
        }
        if (NORM && a.nw_vec16 && threadIdx.x < GROUP) *(u32x4*)(nw_lds + l * PER16) = wreg;
      }
      __syncthreads();  // staged rows visible
    }
    float w1[VEC], w2[VEC];
    if (NORM) {
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        w1[v] = Lp<DT>::up(nw_lds[m1[v]]);
        w2[v] = Lp<DT>::up(nw_lds[m2[v]]);
      }
    }
    // 3. the sum of squares of a row that fits one pass comes from the registers.  The correctly rounded 1 / sqrt costs
    //    as much as the whole rotation: lane u of the thread group does it for unit u and the others read its result.
    if (NORM && iters == 1) {
      float mine = 0.f;
#pragma unroll
      for (int u = 0; u < UNITS; u++) {
        float ss = 0.f;
        if (on) {
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            const float y1 = Lp<DT>::up(p1[u].v[v]), y2 = Lp<DT>::up(p2[u].v[v]);
            ss = add_r(ss, add_r(mul_r(y1, y1), mul_r(y2, y2)));
          }
        }
        ss = row16_sum(ss);
        if ((l & (UNITS - 1)) == u) mine = ss;
      }
      mine = 1.f / sqrtf(add_r(mine / (float)a.r, a.eps));
      inv[0] = dpp_f32<0x00>(mine);  // quad_perm broadcasts of lane 0..3 of each quad
      inv[1] = dpp_f32<0x55>(mine);
      inv[2] = dpp_f32<0xAA>(mine);
      inv[3] = dpp_f32<0xFF>(mine);
    }
    // 4. token by token: cos / sin of the kept columns from the staged rows (shared by the token's HPT heads), rotate, store
#pragma unroll
    for (int tt = 0; tt < TPT; tt++) {
      const E* crow = cs_lds + (size_t)((threadIdx.x / GROUP) * TPT + tt) * 2 * a.hd;
      const E* srow = crow + a.hd;
      float c1[VEC], s1[VEC], c2[VEC], s2[VEC];
#pragma unroll
      for (int v = 0; v < VEC; v++) {
        c1[v] = Lp<DT>::up(crow[m1[v]]);
        s1[v] = Lp<DT>::up(srow[m1[v]]);
        c2[v] = Lp<DT>::up(crow[m2[v]]);
        s2[v] = Lp<DT>::up(srow[m2[v]]);
      }
#pragma unroll
      for (int hh = 0; hh < HPT; hh++) {
        const int u = tt * HPT + hh;
        P o1, o2;
#pragma unroll
        for (int v = 0; v < VEC; v++) {
          float a1 = Lp<DT>::up(p1[u].v[v]), a2 = Lp<DT>::up(p2[u].v[v]);
          if (NORM) {  // (gathered_weight * (x_float * rsqrt(var + eps))).to(dtype)
            a1 = Lp<DT>::up(Lp<DT>::down(mul_r(w1[v], mul_r(a1, inv[u]))));
            a2 = Lp<DT>::up(Lp<DT>::down(mul_r(w2[v], mul_r(a2, inv[u]))));
          }
          // q * cos + rotate_half(q) * sin, rotate_half(q) = cat(-q[half:], q[:half])
          const float u1 = Lp<DT>::up(Lp<DT>::down(mul_r(a1, c1[v])));
          const float v1 = Lp<DT>::up(Lp<DT>::down(mul_r(-a2, s1[v])));
          const float u2 = Lp<DT>::up(Lp<DT>::down(mul_r(a2, c2[v])));
          const float v2 = Lp<DT>::up(Lp<DT>::down(mul_r(a1, s2[v])));
          o1.v[v] = Lp<DT>::down(add_r(u1, v1));
          o2.v[v] = Lp<DT>::down(add_r(u2, v2));
        }
        if (on && live[tt]) {
          *(P*)(orow[u] + jc) = o1;
          *(P*)(orow[u] + half + jc) = o2;
        }
      }
    }
  }
}


// ---------------------------------------------------------------- widths whose half is odd (or operands the packs cannot address)
// The kernel above moves 8-byte packs of the two rotate_half partners; when r/2 is not a multiple of 4 its packs shrink to 4
// or 2 bytes and partial-line stores are amplified (WRITE_SIZE 1.3x / 2.3x the payload at r = 76 / 102).  For 2-byte packs
// this one goes through LDS instead: a workgroup copies the [tokens][CH heads * r] slab of its tile in with the widest chunks the
// addresses allow (the CH heads of a token are one contiguous run), rotates pair by pair out of LDS with every lane busy
// (flat pair index, no 16-lane padding), writes the result transposed to [head][token][r] in LDS and copies each head's
// run of tokens out, again in wide chunks.  More LDS instructions per element, but memory sees full lines.
struct TileArgs {
  RopeArgs a;
  int ch, tt;              // heads per workgroup (1, 2, 4: query heads of one kv head), tokens per workgroup
  int wi, wo;              // chunk bytes of the copies in / out
  unsigned half_magic;     // floor(2^32 / hp) + 1, hp = ceil(half / 2): item / hp == umulhi(item, half_magic) for item < 2^16
};

__host__ __device__ __forceinline__ size_t al16(size_t n) { return (n + 15) & ~(size_t)15; }
static size_t tile_lds_bytes(int tt, int ch, int r, int hd, size_t es) {
  return 2 * al16((size_t)tt * ch * r * es) + al16((size_t)tt * 2 * hd * es) + al16((size_t)hd * es) +
         al16((size_t)tt * ch * sizeof(float)) + al16((size_t)r * sizeof(short));
}

template <int W> struct ChunkT;
template <> struct ChunkT<16> { typedef u32x4 type; };
template <> struct ChunkT<8> { typedef unsigned long long type; };
template <> struct ChunkT<4> { typedef unsigned type; };
template <> struct ChunkT<2> { typedef unsigned short type; };

// `segs` runs of `seg_bytes` bytes: run s starts at src + s * src_stride (bytes) and lands at dst + s * dst_stride
template <int W>
__device__ __forceinline__ void copy_runs(unsigned char* dst, size_t dst_stride, const unsigned char* src, size_t src_stride,
                                          int segs, int seg_bytes) {
  typedef typename ChunkT<W>::type C;
  const int cps = seg_bytes / W, total = segs * cps;
  // four chunks per lane in flight: all loads of a batch are issued before the first store waits for one
  for (int c0 = threadIdx.x; c0 < total; c0 += 4 * 256) {
    C buf[4];
    size_t off[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int c = min(c0 + u * 256, total - 1);
      const int sidx = c / cps, k = c - sidx * cps;
      buf[u] = *(const C*)(src + sidx * src_stride + (size_t)k * W);
      off[u] = sidx * dst_stride + (size_t)k * W;
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (c0 + u * 256 < total) *(C*)(dst + off[u]) = buf[u];
  }
}
__device__ __forceinline__ void copy_runs_w(int w, unsigned char* dst, size_t dst_stride, const unsigned char* src,
                                            size_t src_stride, int segs, int seg_bytes) {
  if (w == 16) copy_runs<16>(dst, dst_stride, src, src_stride, segs, seg_bytes);
  else if (w == 8) copy_runs<8>(dst, dst_stride, src, src_stride, segs, seg_bytes);
  else if (w == 4) copy_runs<4>(dst, dst_stride, src, src_stride, segs, seg_bytes);
  else copy_runs<2>(dst, dst_stride, src, src_stride, segs, seg_bytes);
}

template <int DT, bool NORM, bool HALF_EVEN>
__global__ __launch_bounds__(256) void rope_tile_kernel(TileArgs ta) {
  typedef typename Lp<DT>::T E;
  typedef Pack<E, 2> P2;
  typedef Pack<short, 2> M2;
  const RopeArgs& a = ta.a;
  const unsigned tile = blockIdx.y * N_XCD + (blockIdx.x & (N_XCD - 1));
  if (tile >= a.n_tiles) return;
  const int hk = blockIdx.x >> 3;
  const int CH = ta.ch, TT = ta.tt;
  const int h0 = hk * a.group + blockIdx.z * CH;
  const unsigned bu = tile / (unsigned)a.t_tiles;
  const int64_t b = bu;
  const int64_t t0 = (int64_t)(tile - bu * (unsigned)a.t_tiles) * TT;
  const int tv = (int)min((int64_t)TT, a.T - t0);  // tokens of this tile that exist
  const int r = a.r, half = r >> 1, hd = a.hd;
  const int rows = tv * CH;

  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const size_t slab = al16((size_t)TT * CH * r * sizeof(E));
  E* xin = (E*)lds_raw;                                                      // [TT][CH * r]
  E* xout = (E*)(lds_raw + slab);                                            // [CH][TT][r]
  P2* cs = (P2*)(lds_raw + 2 * slab);                                        // [TT][hd] of (cos, sin): one read per gather
  E* nw = (E*)((unsigned char*)cs + al16((size_t)TT * 2 * hd * sizeof(E)));  // [hd]
  float* inv = (float*)((unsigned char*)nw + al16((size_t)hd * sizeof(E)));  // [TT * CH]
  short* msk = (short*)((unsigned char*)inv + al16((size_t)TT * CH * sizeof(float)));  // [r]

  // ---- in: the tile's slab, its tokens' cos / sin rows (interleaved), the mask row, the norm weight; one barrier
  copy_runs_w(ta.wi, (unsigned char*)xin, (size_t)CH * r * sizeof(E),
              (const unsigned char*)((const E*)a.x + (b * a.T + t0) * a.ld_x + (int64_t)h0 * r), (size_t)a.ld_x * sizeof(E), tv,
              CH * r * (int)sizeof(E));
  {
    const E* cg = (const E*)a.cos + b * a.cs_bstride + t0 * hd;
    const E* sg = (const E*)a.sin + b * a.cs_bstride + t0 * hd;
    if (a.cs_vec16) {  // 16 bytes of cos + 16 bytes of sin per lane and step, interleaved in registers
      constexpr int PER16 = 16 / (int)sizeof(E);
      typedef Pack<E, PER16> Q;
      for (int c = threadIdx.x * PER16; c < tv * hd; c += 256 * PER16) {
        const Q cv = *(const Q*)(cg + c), sv = *(const Q*)(sg + c);
        P2 il[PER16];
#pragma unroll
        for (int k = 0; k < PER16; k++) {
          il[k].v[0] = cv.v[k];
          il[k].v[1] = sv.v[k];
        }
        *(u32x4*)(cs + c) = *(const u32x4*)&il[0];
        *(u32x4*)(cs + c + PER16 / 2) = *(const u32x4*)&il[PER16 / 2];
      }
    } else {
      for (int c = threadIdx.x; c < tv * hd; c += 256) {
        P2 v;
        v.v[0] = cg[c];
        v.v[1] = sg[c];
        cs[c] = v;
      }
    }
  }
  for (int j = threadIdx.x; j < r; j += 256) {
    const int m = a.mask ? (int)a.mask[(int64_t)hk * r + j] : j;
    msk[j] = (short)min(max(m, 0), hd - 1);  // memory safety only; the binding validates masks
  }
  if (NORM)
    for (int c = threadIdx.x; c < hd; c += 256) nw[c] = ((const E*)a.norm_w)[c];
  __syncthreads();

  if (NORM) {  // four lanes per row
    for (int row = threadIdx.x >> 2; row < rows; row += 64) {
      const E* xr = xin + (size_t)row * r;
      float ss = 0.f;
      for (int j = threadIdx.x & 3; j < r; j += 4) {
        const float v = Lp<DT>::up(xr[j]);
        ss = add_r(ss, mul_r(v, v));
      }
      ss = ss + dpp_f32<0xB1>(ss);
      ss = ss + dpp_f32<0x4E>(ss);
      if ((threadIdx.x & 3) == 0) inv[row] = 1.f / sqrtf(add_r(ss / (float)r, a.eps));
    }
    __syncthreads();
  }

  // ---- rotate: one (row, two adjacent pairs) item per lane and step; row = token * CH + head
  const int hp = (half + 1) >> 1;
  const int items = rows * hp;
  const int ch_shift = CH == 4 ? 2 : CH == 2 ? 1 : 0;
  // IT items per lane in flight: the rotate step is a chain of dependent LDS reads (mask -> table, row -> math), so the
  // reads of all IT items are issued phase by phase before anything waits on them
  constexpr int IT = 2;
  for (int base = threadIdx.x; base < items; base += 256 * IT) {
    int row[IT], j[IT];
    bool live[IT], two[IT];
    E e1[IT][2], e2[IT][2];
    short m1[IT][2], m2[IT][2];
#pragma unroll
    for (int u = 0; u < IT; u++) {
      const int item = base + u * 256;
      live[u] = item < items;
      const int ic = live[u] ? item : base;
      row[u] = hp == 1 ? ic : (int)__umulhi((unsigned)ic, ta.half_magic);  // 2^32 / 1 does not fit the magic
      j[u] = (ic - row[u] * hp) * 2;
      two[u] = HALF_EVEN || j[u] + 1 < half;
      const E* xr = xin + (size_t)row[u] * r;
      {
        const P2 p = *(const P2*)(xr + j[u]);  // j is even and rows are even: aligned; the 2nd element may be the partner half
        e1[u][0] = p.v[0];
        e1[u][1] = p.v[1];
        const M2 q = *(const M2*)(msk + j[u]);
        m1[u][0] = q.v[0];
        m1[u][1] = q.v[1];
      }
      if (HALF_EVEN) {
        const P2 p = *(const P2*)(xr + half + j[u]);
        e2[u][0] = p.v[0];
        e2[u][1] = p.v[1];
        const M2 q = *(const M2*)(msk + half + j[u]);
        m2[u][0] = q.v[0];
        m2[u][1] = q.v[1];
      } else {
        e2[u][0] = xr[half + j[u]];
        m2[u][0] = msk[half + j[u]];
        e2[u][1] = two[u] ? xr[half + j[u] + 1] : e2[u][0];
        m2[u][1] = two[u] ? msk[half + j[u] + 1] : m2[u][0];
      }
      if (!two[u]) m1[u][1] = m1[u][0];
    }
    P2 t1[IT][2], t2[IT][2];
    float w1[IT][2], w2[IT][2], iv[IT];
#pragma unroll
    for (int u = 0; u < IT; u++) {
      const P2* cr = cs + (size_t)(row[u] >> ch_shift) * hd;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        t1[u][k] = cr[m1[u][k]];
        t2[u][k] = cr[m2[u][k]];
        if (NORM) {
          w1[u][k] = Lp<DT>::up(nw[m1[u][k]]);
          w2[u][k] = Lp<DT>::up(nw[m2[u][k]]);
        }
      }
      if (NORM) iv[u] = inv[row[u]];
    }
#pragma unroll
    for (int u = 0; u < IT; u++) {
      E o1[2], o2[2];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        float a1 = Lp<DT>::up(e1[u][k]), a2 = Lp<DT>::up(e2[u][k]);
        if (NORM) {
          a1 = Lp<DT>::up(Lp<DT>::down(mul_r(w1[u][k], mul_r(a1, iv[u]))));
          a2 = Lp<DT>::up(Lp<DT>::down(mul_r(w2[u][k], mul_r(a2, iv[u]))));
        }
        const float u1 = Lp<DT>::up(Lp<DT>::down(mul_r(a1, Lp<DT>::up(t1[u][k].v[0]))));
        const float v1 = Lp<DT>::up(Lp<DT>::down(mul_r(-a2, Lp<DT>::up(t1[u][k].v[1]))));
        const float u2 = Lp<DT>::up(Lp<DT>::down(mul_r(a2, Lp<DT>::up(t2[u][k].v[0]))));
        const float v2 = Lp<DT>::up(Lp<DT>::down(mul_r(a1, Lp<DT>::up(t2[u][k].v[1]))));
        o1[k] = Lp<DT>::down(add_r(u1, v1));
        o2[k] = Lp<DT>::down(add_r(u2, v2));
      }
      if (!live[u]) continue;
      const int tok = row[u] >> ch_shift, hh = row[u] & (CH - 1);
      E* orow = xout + ((size_t)hh * TT + tok) * r;
      if (two[u]) {
        P2 w;
        w.v[0] = o1[0];
        w.v[1] = o1[1];
        *(P2*)(orow + j[u]) = w;
      } else {
        orow[j[u]] = o1[0];
      }
      if (HALF_EVEN) {
        P2 w;
        w.v[0] = o2[0];
        w.v[1] = o2[1];
        *(P2*)(orow + half + j[u]) = w;
      } else {
        orow[half + j[u]] = o2[0];
        if (two[u]) orow[half + j[u] + 1] = o2[1];
      }
    }
  }
  __syncthreads();

  // ---- out: per head one contiguous run of tv * r elements
  copy_runs_w(ta.wo, (unsigned char*)((E*)a.out + ((b * a.n_heads + h0) * a.T + t0) * r), (size_t)a.T * r * sizeof(E),
              (const unsigned char*)xout, (size_t)TT * r * sizeof(E), CH, tv * r * (int)sizeof(E));
}

template <int DT> hipError_t launch_rope_tile(const TileArgs& ta, dim3 grid, size_t lds, hipStream_t st) {
  auto launch = [&](auto kernel) -> hipError_t {
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, ta);
    return hipSuccess;
  };
  const bool even = (ta.a.r / 2) % 2 == 0;
  if (ta.a.norm_w) return even ? launch(rope_tile_kernel<DT, true, true>) : launch(rope_tile_kernel<DT, true, false>);
  return even ? launch(rope_tile_kernel<DT, false, true>) : launch(rope_tile_kernel<DT, false, false>);
}

template <int DT, int VEC, int HPT> hipError_t launch_rope(const RopeArgs& a, dim3 grid, hipStream_t st) {
  const size_t lds = ((size_t)ROWS * (UNITS / HPT) * 2 + 1) * a.hd * sizeof(typename Lp<DT>::T);
  auto launch = [&](auto kernel) -> hipError_t {
    if (lds > 64 * 1024) {  // fp32 tables of a 256-wide head with one head per thread group: above the default window
      hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, a);
    return hipSuccess;
  };
  return a.norm_w ? launch(rope_gather_kernel<DT, VEC, true, HPT>) : launch(rope_gather_kernel<DT, VEC, false, HPT>);
}

template <int DT, int VEC> hipError_t launch_rope_hpt(const RopeArgs& a, int hpt, dim3 grid, hipStream_t st) {
  if (hpt == 4) return launch_rope<DT, VEC, 4>(a, grid, st);
  if (hpt == 2) return launch_rope<DT, VEC, 2>(a, grid, st);
  return launch_rope<DT, VEC, 1>(a, grid, st);
}


template <int DT> hipError_t launch_rope_vec(const RopeArgs& a, int vec, int hpt, dim3 grid, hipStream_t st) {
  if (vec == 4) return launch_rope_hpt<DT, 4>(a, hpt, grid, st);
  return launch_rope_hpt<DT, 2>(a, hpt, grid, st);
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
  const int hpt = a.group % 4 == 0 ? 4 : a.group % 2 == 0 ? 2 : 1;  // query heads of one kv head per thread group
  a.n_kv = n_kv;
  a.chunks = a.group / hpt;
  a.t_tiles = (int)ceil_div(T, ROWS * (UNITS / hpt));
  a.n_tiles = (unsigned)(B * a.t_tiles);
  const int64_t tiles8 = ceil_div(B * a.t_tiles, N_XCD);
  MDG_CHECK_ARG(tiles8 <= 65535 && (int64_t)n_kv * N_XCD <= 0x7fffffff && a.chunks <= 65535,
                "mdg_rope_gather: B*T = %lld tokens exceed one launch (%d tokens per workgroup row, 65535 * 8 rows)",
                (long long)(B * T), ROWS * (UNITS / hpt));
  const size_t es = dtype_size(dtype);
  a.cs_vec16 = ((uintptr_t)cos % 16 == 0 && (uintptr_t)sin % 16 == 0 && (hd * es) % 16 == 0 &&
                (cs_batch_stride * es) % 16 == 0)
                   ? 1
                   : 0;
  a.nw_vec16 = norm_w && (uintptr_t)norm_w % 16 == 0;
  const int half = r / 2;
  hipStream_t st = (hipStream_t)stream;
  int vec = 4;  // elements per pack of the direct kernel
  while (vec > 1 && (half % vec || ld_x % vec || ((uintptr_t)x | (uintptr_t)out) % (es * vec) ||
                     (mask && (uintptr_t)mask % (8 * vec))))
    vec >>= 1;
  // Packs of the rotate_half partners straight from / to memory when they are at least 4 bytes.  Measured on
  // [16, 2048, 32 x r] bf16, direct vs LDS route: r = 88 (8-byte packs) 84 vs 113 us; r = 76 (4-byte) 105 vs 99 us, a tie;
  // r = 102 (2-byte) 199 vs 132 us.  MDG_ROPE_TILE=1 forces the LDS route (experiment knob of scripts/bench_kernels.py).
  if (vec > 1 && !MDG_KNOB("MDG_ROPE_TILE")) {
    const dim3 grid((unsigned)(N_XCD * n_kv), (unsigned)tiles8, (unsigned)a.chunks);
    if (dtype == MDG_BF16) MDG_HIP(launch_rope_vec<MDG_BF16>(a, vec, hpt, grid, st));
    else if (dtype == MDG_F16) MDG_HIP(launch_rope_vec<MDG_F16>(a, vec, hpt, grid, st));
    else MDG_HIP(launch_rope_vec<MDG_F32>(a, vec, hpt, grid, st));
  } else {       // through LDS (see rope_tile_kernel)
    TileArgs ta;
    int tt = 64 / hpt;
    while (tt > 1 && tile_lds_bytes(tt, hpt, r, hd, es) > 64 * 1024) tt >>= 1;
    const size_t lds = tile_lds_bytes(tt, hpt, r, hd, es);
    a.t_tiles = (int)ceil_div(T, tt);
    a.n_tiles = (unsigned)(B * a.t_tiles);
    const int64_t t8 = ceil_div(B * a.t_tiles, N_XCD);
    MDG_CHECK_ARG(t8 <= 65535, "mdg_rope_gather: B*T = %lld tokens exceed one launch (%d tokens per workgroup, 65535 * 8)",
                  (long long)(B * T), tt);
    auto widest = [&](uintptr_t base, size_t stride_a, size_t stride_b) {
      int w = 16;
      while (w > (int)es && (base % w || stride_a % w || stride_b % w)) w >>= 1;
      return w;
    };
    ta.a = a;
    ta.ch = hpt;
    ta.tt = tt;
    ta.wi = widest((uintptr_t)x, (size_t)ld_x * es, (size_t)hpt * r * es);
    ta.wo = widest((uintptr_t)out, (size_t)T * r * es, (size_t)tt * r * es);
    ta.half_magic = (unsigned)((1ull << 32) / (unsigned)((half + 1) / 2)) + 1u;   // items index (row, pair of pairs)
    const dim3 grid((unsigned)(N_XCD * n_kv), (unsigned)t8, (unsigned)a.chunks);
    if (dtype == MDG_BF16) MDG_HIP(launch_rope_tile<MDG_BF16>(ta, grid, lds, st));
    else if (dtype == MDG_F16) MDG_HIP(launch_rope_tile<MDG_F16>(ta, grid, lds, st));
    else MDG_HIP(launch_rope_tile<MDG_F32>(ta, grid, lds, st));
  }
  MDG_LAUNCH_CHECK();
  return MDG_OK;
}
