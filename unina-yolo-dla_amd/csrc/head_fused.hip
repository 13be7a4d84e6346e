// head_fused.hip -- one launch for a whole DetectionHead (reference: unina_yolo_dla/model.py:274-303).
//
//     h0 = ReLU(BN(cls.0(x))) | ReLU(BN(reg.0(x)))        3x3, C -> C | C    (step 0: one GEMM, N = 2C)
//     h1 = ReLU(BN(cls.1(h0[:C]))) | ReLU(BN(reg.1(h0[C:])))   3x3, C -> C per branch   (step 1: grouped, N = 2C)
//     cls = cls.2(h1[:C]) + b,  reg = reg.2(h1[C:]) + b   1x1, C -> nc | 4, no BN / activation, fp32 planar (step 2)
//
// Same structure as c3k2_fused.hip (block_pipeline.h): a workgroup owns a TH x TW tile of the head's output; the input
// patch (tile + 2-pixel halo) is DMA'd into LDS once, h0 (tile + 1-pixel halo, zero outside the image = the second
// 3x3's zero padding) and h1 live in LDS as fp16 images, the weights of all three layers stream L2 -> registers through
// the per-wave prefetch queue, and only the eight raw output planes go to HBM. The unfused table runs the head as
// three launches with two 6.5 MB round trips through h0 / h1 (P2 at 640^2).
// Only heads whose whole weight set a workgroup can afford to stream are fused (C = 64: 296 KB; the P3 / P4 heads
// carry 1.2 / 4.7 MB and stay split over output channels, conv_igemm.hip).
// Arithmetic is identical to the unfused kernels (same MFMA, K order, fp32 epilogue, fp16 rounding points).
#include "block_kernels.h"

#include <cstdlib>
#include <cstring>

namespace unina {

using namespace dev;

extern __shared__ __align__(16) unsigned char head_smem[];

template <int C, int TH, int TW, int NW, int D>
__global__ __launch_bounds__(NW * 64) void head_fused_kernel(const HeadParams p) {
  head_fused_body<C, TH, TW, NW, D>(p, (int)blockIdx.x, head_smem);
}

template <int C, int TH, int NW>
__global__ __launch_bounds__(NW * 64) void head_ws_kernel(const HeadParams p) {
  head_ws_body<C, TH, NW>(p, (int)blockIdx.x, head_smem);
}

// ------------------------------------------------------------------------------------------------- host side
namespace {

struct HClass {
  int c, th, tw, nw;
  const char* name;
  void (*fn)(const HeadParams);
  bool ws = false;     // row-streaming / weights-stationary body (head_ws_body): th x 14 output pixels per workgroup
};
const HClass kHeadClasses[] = {
    {64, 8, 16, 8, "head_fused<64,8x16,8w>", head_fused_kernel<64, 8, 16, 8, 16>},   // UNINA_HEAD_ALT=0: the tile form (round 1's), kept as the
                                                                                     // independent implementation the row-streaming form is tested against
    {64, kHeadWsTH, 14, 8, "head_ws<64,13x14,8w>", head_ws_kernel<64, kHeadWsTH, 8>, true},   // default
};
const HClass* find_hclass(int c) {
  // default: the row-streaming class (same-box A/B against the 8x16 tile class: the block dual 30.5 -> 23.5 us, -7 us serial
  // latency, frames/s equal); UNINA_HEAD_ALT=0 selects the tile class
  static const int alt = (getenv("UNINA_HEAD_ALT") && getenv("UNINA_HEAD_ALT")[0] == '0') ? 0 : 1;
  return kHeadClasses[alt].c == c ? &kHeadClasses[alt] : nullptr;
}
constexpr int kMaxLds = 160 * 1024;
int align_up(int v, int a) { return (v + a - 1) / a * a; }

}  // namespace

hipError_t head_init() {
  for (const HClass& c : kHeadClasses) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

bool head_supported(int c) { return find_hclass(c) != nullptr; }

bool head_layout(HeadParams* p) {
  const HClass* c = find_hclass(p->C);
  if (!c || p->n_cls < 1 || p->n_cls > 16 || p->n_reg < 1 || p->n_reg > 16) return false;
  const int C = p->C;
  p->tiles_x = (p->W + c->tw - 1) / c->tw;
  p->tiles_y = (p->H + c->th - 1) / c->th;
  p->tiles_x_magic = div_magic((unsigned)p->tiles_x);
  p->n_bias = 4 * C + 32;
  int off = 0;
  p->off_bias = off; off += align_up(p->n_bias * 4, 1024);
  if (c->ws) {   // padded images (head_ws_body): x patch (th+4) x 18, two h0 rows of 18, h1 th x 16 pixels
    const int px = C * 2 + 32, ph = 2 * C * 2 + 32;
    p->off_x = off; off += align_up((c->th + 4) * 18 * px, 1024);
    p->off_h0 = off; off += align_up(2 * 18 * ph, 1024);
    p->off_h1 = off; off += align_up(c->th * 16 * ph, 1024);
    p->smem_bytes = off;
    return off <= kMaxLds;
  }
  const int p0 = (c->th + 4) * (c->tw + 4), p1 = (c->th + 2) * (c->tw + 2), pt = c->th * c->tw;
  const int x_bytes = align_up(p0 * C * 2, 1024) + 1024, h1_bytes = pt * 2 * C * 2;
  p->off_x = off;
  p->off_h1 = off;                                            // h1 replaces the patch once step 0 has consumed it
  off += align_up(x_bytes > h1_bytes ? x_bytes : h1_bytes, 1024);
  p->off_h0 = off; off += align_up(p1 * 2 * C * 2, 1024);
  p->smem_bytes = off;
  return off <= kMaxLds;
}

hipError_t head_launch(const HeadParams& p, hipStream_t stream) {
  const HClass* c = find_hclass(p.C);
  if (!c) return hipErrorInvalidValue;
  hipLaunchKernelGGL(c->fn, dim3(p.tiles_x * p.tiles_y, 1, 1), dim3(c->nw * 64, 1, 1), p.smem_bytes, stream, p);
  return hipGetLastError();
}

bool head_tile_is(const HeadParams& p, int th, int tw) {
  const HClass* k = find_hclass(p.C);
  return k && k->th == th && k->tw == tw;
}

bool head_is_ws(int c) {
  const HClass* k = find_hclass(c);
  return k && k->ws;
}
const char* head_kernel_name(int c) {
  const HClass* k = find_hclass(c);
  return k ? k->name : "head_fused<?>";
}
int head_block_threads(int c) {
  const HClass* k = find_hclass(c);
  return k ? k->nw * 64 : 0;
}

}  // namespace unina
