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
#include "kernels.h"
#include "block_pipeline.h"

#include <cstdlib>
#include <cstring>

namespace unina {

using namespace dev;

namespace {

template <int C>
struct HeadPlan {
  static constexpr int N = 3;
  static constexpr int kb(int s) { return s == 2 ? C / 32 : 9 * C / 32; }
  static constexpr int ns(int s) { return s == 2 ? 2 : 2 * C / 16; }
  static constexpr int wnt(int s) { return s == 2 ? 1 : 2; }   // wave tile 2 channel subtiles x up to 6 pixel subtiles
};

}  // namespace

extern __shared__ __align__(16) unsigned char head_smem[];

template <int C, int TH, int TW, int NW, int D>
__global__ __launch_bounds__(NW * 64) void head_fused_kernel(const HeadParams p) {
  typedef StepTable<HeadPlan<C>, NW> ST;
  static_assert(ST::valid(), "wave roles");
  constexpr int R0W = TW + 4, R0H = TH + 4;                 // input region (tile + 2-pixel halo)
  constexpr int R1W = TW + 2, P1 = (TH + 2) * R1W;          // h0 region (tile + 1-pixel halo)
  constexpr int PT = TH * TW, CB = C / 32, NT = NW * 64;
  constexpr int G1 = 16 * ST::wnt(1);                       // channels per wave in step 1 (all inside one branch)
  static_assert(C % G1 == 0, "a wave's channel subtiles must belong to one branch");

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  unsigned char* smem = head_smem;
  const int tyi = fast_div((int)blockIdx.x, p.tiles_x_magic), txi = (int)blockIdx.x - tyi * p.tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TW;

  const unsigned char* wbase = p.wstream + (4 * l15 + (lq ^ swz_g(l15))) * 16;
  half8 q[D];

  float* bias_lds = reinterpret_cast<float*>(smem + p.off_bias);
  for (int i = threadIdx.x; i < p.n_bias; i += NT) bias_lds[i] = p.bias[i];
  constexpr Img X = make_img(0, C / 8);
  load_patch<R0H, R0W, C, NT>(smem + p.off_x, p.src, p.src_ld, p.H, p.W, ty0 - 2, tx0 - 2, p.zeros, wid, lane);
  static_for<0, D>([&](auto gc) { wq_fetch<ST, D, decltype(gc)::value>(q, wbase, wid); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the patch has landed
  lds_barrier();

  const Img Xi = Img{p.off_x, X.nch, X.sh, X.mask};
  const Img H0 = make_img(p.off_h0, 2 * C / 8);
  const Img H1 = make_img(p.off_h1, 2 * C / 8);
  auto run_step = [&](auto sc, auto pc, auto baddr, auto epi) {
    dev::run_step<ST, D, decltype(sc)::value, decltype(pc)::value>(q, wbase, smem, wid, lane, baddr, epi);
  };
#define STEP(S, P) std::integral_constant<int, (S)>{}, std::integral_constant<int, (P)>{}

  // ---- step 0: h0 = ReLU(3x3(x) + b) for both branches on the tile + 1-pixel halo, 0 outside the image ----
  run_step(STEP(0, P1),
      [&](int sub, auto kc) {
        constexpr int kb = decltype(kc)::value, tap = kb / CB, cb = kb - tap * CB, th3 = tap / 3;
        int pp = sub * 16 + l15;
        pp = pp < P1 ? pp : P1 - 1;
        const int py = pp / R1W, px = pp - py * R1W;
        return Xi.addr((py + th3) * R0W + px + (tap - th3 * 3), cb * 4 + lq);
      },
      [&](int sub, int n, const floatx4& acc) {
        const int pp = sub * 16 + l15;
        if (pp >= P1) return;
        const int py = pp / R1W, px = pp - py * R1W;
        floatx4 v = bias_relu(acc, bias_lds, n);
        if (!((unsigned)(ty0 - 1 + py) < (unsigned)p.H && (unsigned)(tx0 - 1 + px) < (unsigned)p.W)) v = floatx4{0.f, 0.f, 0.f, 0.f};
        store_h4(smem, H0, pp, n, v);
      });

  // ---- step 1: h1 = ReLU(3x3(h0[branch]) + b) on the tile; a wave's channels all belong to one branch ----
  const int br1 = (wid % ST::waves_n(1)) * G1 / C;   // 0 = cls, 1 = reg
  run_step(STEP(1, PT),
      [&](int sub, auto kc) {
        constexpr int kb = decltype(kc)::value, tap = kb / CB, cb = kb - tap * CB, th3 = tap / 3;
        int pp = sub * 16 + l15;
        pp = pp < PT ? pp : PT - 1;
        const int py = pp / TW, px = pp - py * TW;
        return H0.addr((py + th3) * R1W + px + (tap - th3 * 3), br1 * (C / 8) + cb * 4 + lq);
      },
      [&](int sub, int n, const floatx4& acc) {
        const int pp = sub * 16 + l15;
        if (pp < PT) store_h4(smem, H1, pp, n, bias_relu(acc, bias_lds + 2 * C, n));
      });

  // ---- step 2: raw outputs = W2 h1[branch] + b2 (no activation), planar fp32 straight from the accumulators ----
  const int br2 = wid % ST::waves_n(2);
  const float* bias_2 = bias_lds + 4 * C;
  const int M = p.H * p.W;
  run_step(STEP(2, PT),
      [&](int sub, auto kc) {
        int pp = sub * 16 + l15;
        pp = pp < PT ? pp : PT - 1;
        return H1.addr(pp, br2 * (C / 8) + decltype(kc)::value * 4 + lq);
      },
      [&](int sub, int n, const floatx4& acc) {
        const int pp = sub * 16 + l15;
        const int oy = ty0 + pp / TW, ox = tx0 + pp % TW;
        if (pp >= PT || oy >= p.H || ox >= p.W) return;
        const floatx4 v = acc + *reinterpret_cast<const floatx4*>(bias_2 + n);
        const int c = n & 15;                       // channel inside the branch's (zero-padded) 16-row weight block
        float* dst = (n < 16 ? p.out_cls : p.out_reg) + (size_t)oy * p.W + ox;
        const int cnt = n < 16 ? p.n_cls : p.n_reg;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (c + r < cnt) dst[(size_t)(c + r) * M] = v[r];
      });
#undef STEP
}

// ------------------------------------------------------------------------------------------------- host side
namespace {

struct HClass {
  int c, th, tw, nw;
  const char* name;
  void (*fn)(const HeadParams);
};
const HClass kHeadClasses[] = {
    {64, 8, 16, 8, "head_fused<64,8x16,8w>", head_fused_kernel<64, 8, 16, 8, 16>},
    {64, 8, 8, 8, "head_fused<64,8x8,8w>", head_fused_kernel<64, 8, 8, 8, 16>},     // UNINA_HEAD_ALT=1 (A/B experiments)
};
const HClass* find_hclass(int c) {
  static const int alt = getenv("UNINA_HEAD_ALT") ? atoi(getenv("UNINA_HEAD_ALT")) : 0;
  const int n = (int)(sizeof(kHeadClasses) / sizeof(kHeadClasses[0]));
  for (int i = alt ? 1 : 0; i < (alt ? n : 1); ++i)
    if (kHeadClasses[i].c == c) return &kHeadClasses[i];
  return nullptr;
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
  const int p0 = (c->th + 4) * (c->tw + 4), p1 = (c->th + 2) * (c->tw + 2), pt = c->th * c->tw;
  p->tiles_x = (p->W + c->tw - 1) / c->tw;
  p->tiles_y = (p->H + c->th - 1) / c->th;
  p->tiles_x_magic = div_magic((unsigned)p->tiles_x);
  p->n_bias = 4 * C + 32;
  const int x_bytes = align_up(p0 * C * 2, 1024) + 1024, h1_bytes = pt * 2 * C * 2;
  int off = 0;
  p->off_bias = off; off += align_up(p->n_bias * 4, 1024);
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

const char* head_kernel_name(int c) {
  const HClass* k = find_hclass(c);
  return k ? k->name : "head_fused<?>";
}
int head_block_threads(int c) {
  const HClass* k = find_hclass(c);
  return k ? k->nw * 64 : 0;
}

}  // namespace unina
