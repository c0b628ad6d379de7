// Timing harness for mdg_rope_gather outside Python (compiles the product kernel source as is).
//   0 full kernel   1 no math (packs copied through)   2 no gathers (cos = 1, sin = 0 constants; staging kept)
//   3 no staging, no gathers   hipcc --offload-arch=gfx950 -O3 -std=c++17 rope_anatomy.hip
#include <stdarg.h>
#include "../../modegpt_amd/csrc/common.hpp"
namespace mdg { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }
#include "../../modegpt_amd/csrc/rope.hip"
#include <vector>
int main() {
  const int B = 16, T = 2048, H = 32, KV = 8, r = 88, hd = 128;
  const size_t elems = (size_t)B * T * H * r;
  unsigned short *x, *o, *c, *s; int64_t* m;
  hipMalloc(&x, elems * 2); hipMalloc(&o, elems * 2); hipMalloc(&c, T * hd * 2); hipMalloc(&s, T * hd * 2); hipMalloc(&m, KV * r * 8);
  hipMemset(x, 0x3c, elems * 2); hipMemset(c, 0x3c, T * hd * 2); hipMemset(s, 0x3c, T * hd * 2);
  std::vector<int64_t> hm(KV * r);
  for (int k = 0; k < KV; k++) for (int j = 0; j < r / 2; j++) { hm[k * r + j] = (j * 7 + k) % (hd / 2); hm[k * r + r / 2 + j] = hm[k * r + j] + hd / 2; }
  hipMemcpy(m, hm.data(), hm.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int heads : {H, KV}) {
    float best = 1e9;
    for (int rep = 0; rep < 8; rep++) {
      hipEventRecord(e0);
      int rc = mdg_rope_gather(x, MDG_BF16, (int64_t)heads * r, B, T, heads, heads == H ? KV : heads, r, hd, c, s, 0, m, nullptr, 1e-6, o, nullptr);
      hipEventRecord(e1); hipEventSynchronize(e1);
      if (rc) { printf("rc %d\n", rc); return 1; }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 1 && ms < best) best = ms;
    }
    printf("EXP %d heads %2d: %.1f us  %.0f GB/s\n", MDG_ROPE_EXP, heads, best * 1e3, 4.0 * B * T * heads * r / best / 1e6);
  }
  return 0;
}
