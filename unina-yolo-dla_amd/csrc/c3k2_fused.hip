// c3k2_fused.hip -- one launch for a whole C3k2 block (reference: unina_yolo_dla/model.py:76-110, Bottleneck :53-73).
//
//     a | b = ReLU(BN(cv1(x))) | ReLU(BN(cv2(x)))                 1x1, Cin -> h | h          (step 0, one GEMM, N = 2h)
//     for each bottleneck:  t = ReLU(BN(b.cv1(a)))                1x1, h -> h                (odd steps)
//                           a = ReLU(BN(b.cv2(t))) + a            3x3, h -> h, pad 1         (even steps)
//     y = ReLU(BN(cv3(cat[a, b])))                                1x1, 2h -> 2h              (last step)
//
// The unfused op table runs this as 2 + 2n launches that each sit on the ~4 us launch / latency floor. Here a
// workgroup owns a TH x TW tile of the block's output and keeps every intermediate tensor of its tile IN LDS:
//   * the input patch (tile + n-pixel halo, all Cin channels) is DMA'd into LDS once (global_load_lds);
//   * every step is an MFMA GEMM whose B operand (activations, pixels as columns) is read from an LDS image;
//   * the A operand (weights) never touches LDS: a wave owns ONE 16-channel subtile of a step's output, so each
//     1-KiB weight fragment block is needed by exactly that wave (x the few waves that split the pixels) and is
//     loaded straight from L2 into its registers, 16 bytes per lane. All steps' blocks form one flat per-wave
//     sequence (packed by c3k2_pack in consumption order) that is prefetched D blocks ahead through a circular
//     register queue ACROSS step boundaries -- weights depend on nothing, so the L2 latency is paid once per launch
//     and the bytes in flight per CU are D KiB x waves, not an LDS ring (the first version streamed weights through
//     a 48 KiB LDS-DMA ring and was bound by DMA issue rate and ring depth: 33 us for the 917 KB of stage3's block);
//   * epilogues (bias, ReLU, zero outside the image = the 3x3's zero padding, residual) write fp16 LDS images;
//     only cv3's output goes to HBM (staged through LDS, full 16-byte row segments).
// The 1x1 convs in front of a 3x3 are recomputed on the halo (1.27-1.9x of their small cost). Every loop is
// unrolled at compile time (shapes, including Cin, are template parameters), so queue slots are plain registers and
// the compiler's own counted s_waitcnt vmcnt tracks the prefetches.
// Arithmetic is identical to the unfused kernels: same MFMA (v_mfma_f32_16x16x32_f16), same K order (tap-major, 32
// channels per block), same fp32 epilogue and the same fp16 rounding points -> results are bit-identical (tested).
//
#include "kernels.h"
#include "block_pipeline.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace unina {

using namespace dev;

namespace {

// step 0: cv1|cv2 (K = CIN, N = 2h); odd steps: bottleneck 1x1 (K = h, N = h); even steps: bottleneck 3x3 (K = 9h,
// N = h); last step: cv3 (K = 2h, N = 2h). A wave owns one channel subtile, or ns / NW of them when ns > NW.
// TAIL = 1 appends the lateral 1x1 conv (2h -> h, + nearest x2 upsample in its store) that follows an FPN block.
template <int H_, int NB, int CIN, int NW, int TAIL>
struct C3k2Plan {
  static constexpr int CV3 = 1 + 2 * NB;        // index of the cv3 step
  static constexpr int N = 2 + 2 * NB + TAIL;
  static constexpr int kb(int s) { return s == 0 ? CIN / 32 : (s >= CV3 ? 2 * H_ / 32 : ((s & 1) ? H_ / 32 : 9 * H_ / 32)); }
  static constexpr int ns(int s) { return (s == 0 || s == CV3) ? 2 * H_ / 16 : H_ / 16; }
  static constexpr int wnt(int s) { return ns(s) <= NW ? 1 : ns(s) / NW; }
};

}  // namespace

extern __shared__ __align__(16) unsigned char c3_smem[];

// NW waves per workgroup (roles per step: Steps::waves_n / wnt / waves_m); D = weight prefetch depth in 1-KiB blocks
// per wave. There is no wave-uniform branch around any global load: the kernel is straight-line code, so the
// compiler's counted s_waitcnt vmcnt keeps D loads in flight across every step boundary.
template <int H_, int TH, int TW, int NB, int CIN, int NW, int D, int TAIL = 0>
__global__ __launch_bounds__(NW * 64) void c3k2_fused_kernel(const C3k2Params p) {
  static_assert(NB == 1 || NB == 2, "bottleneck count");
  typedef StepTable<C3k2Plan<H_, NB, CIN, NW, TAIL>, NW> ST;
  static_assert(ST::valid(), "wave roles");
  static_assert((NW & (NW - 1)) == 0 && NW >= 2 && NW <= 16, "waves per workgroup");
  constexpr int R0W = TW + 2 * NB, P0 = (TH + 2 * NB) * R0W;   // input / first-level region (tile + NB-pixel halo)
  constexpr int R1W = TW + 2, P1 = (TH + 2) * R1W;              // NB == 2: region of the first bottleneck's output
  constexpr int PT = TH * TW;
  constexpr int HB = H_ / 32;                                   // k-blocks per tap of the hidden width
  constexpr int NT = NW * 64;

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  unsigned char* smem = c3_smem;
  const int tyi = fast_div((int)blockIdx.x, p.tiles_x_magic), txi = (int)blockIdx.x - tyi * p.tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TW;

  // ---- weight prefetch queue: element g of this wave's flat sequence lives in slot g % D ----
  const unsigned char* wbase = p.wstream + (4 * l15 + (lq ^ swz_g(l15))) * 16;   // this lane's 16 bytes of any block
  half8 q[D];
  // biases of every step -> LDS
  float* bias_lds = reinterpret_cast<float*>(smem + p.off_bias);
  for (int i = threadIdx.x; i < p.n_bias; i += NT) bias_lds[i] = p.bias[i];

  // input patch (tile + NB-pixel halo, all CIN channels) -> LDS image
  constexpr Img X = make_img(0, CIN / 8);
  load_patch<TH + 2 * NB, R0W, CIN, NT>(smem + p.off_x, p.src, p.src_ld, p.H, p.W, ty0 - NB, tx0 - NB, p.zeros, wid, lane);
  static_for<0, D>([&](auto gc) { wq_fetch<ST, D, decltype(gc)::value>(q, wbase, wid); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the patch has landed (LDS-DMA is not tracked by the compiler)
  lds_barrier();

  const Img Xi = Img{p.off_x, X.nch, X.sh, X.mask};
  const Img Y = make_img(p.off_y, 2 * H_ / 8);   // a | b on R0
  const Img T = make_img(p.off_t, H_ / 8);       // t of the current bottleneck (R0, then R1)
  const Img U1 = make_img(p.off_u1, H_ / 8);     // NB == 2: first bottleneck's output on R1
  const Img U2 = make_img(p.off_u2, H_ / 8);     // last bottleneck's output on the tile
  auto in_image = [&](int iy, int ix) { return (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W; };

  auto run_step = [&](auto sc, auto pc, auto baddr, auto epi) {
    dev::run_step<ST, D, decltype(sc)::value, decltype(pc)::value>(q, wbase, smem, wid, lane, baddr, epi);
  };
#define STEP(S, P) std::integral_constant<int, (S)>{}, std::integral_constant<int, (P)>{}

  // ---- step 0: a | b = ReLU(W12 x + b12) on R0 --------------------------------------------------------------------
  run_step(STEP(0, P0),
      [&](int sub, auto kc) {
        const int r = sub * 16 + l15;
        return Xi.addr(r < P0 ? r : P0 - 1, decltype(kc)::value * 4 + lq);
      },
      [&](int sub, int n, const floatx4& acc) {
        const int r = sub * 16 + l15;
        if (r < P0) store_h4(smem, Y, r, n, bias_relu(acc, bias_lds, n));
      });
  const float* bias_b = bias_lds + 2 * H_;

  // ---- bottleneck 0 -----------------------------------------------------------------------------------------------
  // t = ReLU(Wb1 a + b) on R0, forced to 0 outside the image (zero padding of the 3x3 that follows)
  run_step(STEP(1, P0),
      [&](int sub, auto kc) {
        const int r = sub * 16 + l15;
        return Y.addr(r < P0 ? r : P0 - 1, decltype(kc)::value * 4 + lq);
      },
      [&](int sub, int n, const floatx4& acc) {
        const int r = sub * 16 + l15;
        if (r >= P0) return;
        const int ry = r / R0W, rx = r - ry * R0W;
        floatx4 v = bias_relu(acc, bias_b, n);
        if (!in_image(ty0 - NB + ry, tx0 - NB + rx)) v = floatx4{0.f, 0.f, 0.f, 0.f};
        store_h4(smem, T, r, n, v);
      });
  if constexpr (NB == 1) {
    // u = ReLU(3x3(t) + b) + a on the tile
    run_step(STEP(2, PT),
        [&](int sub, auto kc) {
          constexpr int kb = decltype(kc)::value, tap = kb / HB, cb = kb - tap * HB, th3 = tap / 3;
          int pp = sub * 16 + l15;
          pp = pp < PT ? pp : PT - 1;
          const int py = pp / TW, px = pp - py * TW;
          return T.addr((py + th3) * R0W + px + (tap - th3 * 3), cb * 4 + lq);
        },
        [&](int sub, int n, const floatx4& acc) {
          const int pp = sub * 16 + l15;
          if (pp >= PT) return;
          const int py = pp / TW, px = pp - py * TW;
          const floatx4 v = bias_relu(acc, bias_b + H_, n) + load_h4(smem, Y, (py + 1) * R0W + px + 1, n);
          store_h4(smem, U2, pp, n, v);
        });
  } else {
    // u1 = ReLU(3x3(t1) + b) + a on R1
    run_step(STEP(2, P1),
        [&](int sub, auto kc) {
          constexpr int kb = decltype(kc)::value, tap = kb / HB, cb = kb - tap * HB, th3 = tap / 3;
          int pp = sub * 16 + l15;
          pp = pp < P1 ? pp : P1 - 1;
          const int py = pp / R1W, px = pp - py * R1W;
          return T.addr((py + th3) * R0W + px + (tap - th3 * 3), cb * 4 + lq);
        },
        [&](int sub, int n, const floatx4& acc) {
          const int pp = sub * 16 + l15;
          if (pp >= P1) return;
          const int py = pp / R1W, px = pp - py * R1W;
          const floatx4 v = bias_relu(acc, bias_b + H_, n) + load_h4(smem, Y, (py + 1) * R0W + px + 1, n);
          store_h4(smem, U1, pp, n, v);
        });
    // ---- bottleneck 1 ---------------------------------------------------------------------------------------------
    const float* bias_c = bias_b + 2 * H_;
    // t2 = ReLU(Wb1' u1 + b) on R1, 0 outside the image
    run_step(STEP(3, P1),
        [&](int sub, auto kc) {
          const int r = sub * 16 + l15;
          return U1.addr(r < P1 ? r : P1 - 1, decltype(kc)::value * 4 + lq);
        },
        [&](int sub, int n, const floatx4& acc) {
          const int r = sub * 16 + l15;
          if (r >= P1) return;
          const int ry = r / R1W, rx = r - ry * R1W;
          floatx4 v = bias_relu(acc, bias_c, n);
          if (!in_image(ty0 - 1 + ry, tx0 - 1 + rx)) v = floatx4{0.f, 0.f, 0.f, 0.f};
          store_h4(smem, T, r, n, v);
        });
    // u2 = ReLU(3x3(t2) + b) + u1 on the tile
    run_step(STEP(4, PT),
        [&](int sub, auto kc) {
          constexpr int kb = decltype(kc)::value, tap = kb / HB, cb = kb - tap * HB, th3 = tap / 3;
          int pp = sub * 16 + l15;
          pp = pp < PT ? pp : PT - 1;
          const int py = pp / TW, px = pp - py * TW;
          return T.addr((py + th3) * R1W + px + (tap - th3 * 3), cb * 4 + lq);
        },
        [&](int sub, int n, const floatx4& acc) {
          const int pp = sub * 16 + l15;
          if (pp >= PT) return;
          const int py = pp / TW, px = pp - py * TW;
          const floatx4 v = bias_relu(acc, bias_c + H_, n) + load_h4(smem, U1, (py + 1) * R1W + px + 1, n);
          store_h4(smem, U2, pp, n, v);
        });
  }

  // ---- last step: y = ReLU(W3 [u | b] + b3) on the tile -> staging image (linear rows) -> HBM ----------------------
  constexpr int ROWB = 2 * H_ * 2 + 16;  // staged output row: 2h halfs + 16 bytes of padding (bank spread)
  unsigned char* stage = smem + p.off_stage;
  const float* bias_3 = bias_lds + 2 * H_ * (1 + NB);
  run_step(STEP(1 + 2 * NB, PT),
      [&](int sub, auto kc) {
        constexpr int kb = decltype(kc)::value;
        int pp = sub * 16 + l15;
        pp = pp < PT ? pp : PT - 1;
        if constexpr (kb < HB) {                                  // k-blocks [0, h/32): u
          return U2.addr(pp, kb * 4 + lq);
        } else {                                                  // k-blocks [h/32, 2h/32): b = channels [h, 2h) of Y
          const int py = pp / TW, px = pp - py * TW;
          return Y.addr((py + NB) * R0W + px + NB, H_ / 8 + (kb - HB) * 4 + lq);
        }
      },
      [&](int sub, int n, const floatx4& acc) {
        const int pp = sub * 16 + l15;
        if (pp >= PT) return;
        const floatx4 v = bias_relu(acc, bias_3, n);
        half4 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = (half_t)v[r];
        *reinterpret_cast<half4*>(stage + pp * ROWB + n * 2) = hv;
      });
  typedef float vec16 __attribute__((ext_vector_type(4)));  // 16 opaque bytes
  constexpr int CPR = 2 * H_ * 2 / 16;                      // 16-byte chunks per output pixel
  for (int c = threadIdx.x; c < PT * CPR; c += NT) {
    const int pp = c / CPR, ch = c - pp * CPR;
    const int oy = ty0 + pp / TW, ox = tx0 + pp % TW;
    if (oy < p.H && ox < p.W)
      *reinterpret_cast<vec16*>(p.dst + (size_t)(oy * p.W + ox) * p.dst_ld + ch * 8) =
          *reinterpret_cast<const vec16*>(stage + pp * ROWB + ch * 16);
  }

  if constexpr (TAIL) {
    // ---- tail: lateral 1x1 (model.py:256,259: ConvBlock 2h -> h) on the block's output, still in the staging image,
    //      then nearest x2 upsample (model.py:145-147) in the store: each pixel's h channels go to its 2x2 block ----
    const Img YS = Img{p.off_stage, ROWB / 16, 0, 0};        // the staging image is linear: pitch ROWB, no swizzle
    constexpr int ROWT = H_ * 2 + 16;
    unsigned char* tout = smem + p.off_tail;
    const float* bias_t = bias_lds + 2 * H_ * (2 + NB);
    run_step(STEP(2 + 2 * NB, PT),
        [&](int sub, auto kc) {
          int pp = sub * 16 + l15;
          pp = pp < PT ? pp : PT - 1;
          return YS.addr(pp, decltype(kc)::value * 4 + lq);
        },
        [&](int sub, int n, const floatx4& acc) {
          const int pp = sub * 16 + l15;
          if (pp >= PT) return;
          const floatx4 v = bias_relu(acc, bias_t, n);
          half4 hv;
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = (half_t)v[r];
          *reinterpret_cast<half4*>(tout + pp * ROWT + n * 2) = hv;
        });
    constexpr int CPT = H_ * 2 / 16;
    const size_t px = (size_t)p.dst2_ld, row = (size_t)(2 * p.W) * px;
    for (int c = threadIdx.x; c < PT * CPT; c += NT) {
      const int pp = c / CPT, ch = c - pp * CPT;
      const int oy = ty0 + pp / TW, ox = tx0 + pp % TW;
      if (oy < p.H && ox < p.W) {
        const vec16 v = *reinterpret_cast<const vec16*>(tout + pp * ROWT + ch * 16);
        half_t* d = p.dst2 + ((size_t)(2 * oy) * (2 * p.W) + 2 * ox) * px + ch * 8;
        *reinterpret_cast<vec16*>(d) = v;
        *reinterpret_cast<vec16*>(d + px) = v;
        *reinterpret_cast<vec16*>(d + row) = v;
        *reinterpret_cast<vec16*>(d + row + px) = v;
      }
    }
  }
#undef STEP
}

// ------------------------------------------------------------------------------------------------- host side
namespace {

struct Class {
  int hid, nb, cin, tail, th, tw, nw;
  const char* name;
  void (*fn)(const C3k2Params);
};
#define C3K2(H_, TH, TW, NB, CIN, NW, D) \
  {H_, NB, CIN, 0, TH, TW, NW, "c3k2_fused<" #H_ "," #TH "x" #TW "," #NB "," #CIN "," #NW "w>", c3k2_fused_kernel<H_, TH, TW, NB, CIN, NW, D, 0>}
#define C3K2T(H_, TH, TW, NB, CIN, NW, D) \
  {H_, NB, CIN, 1, TH, TW, NW, "c3k2_fused<" #H_ "," #TH "x" #TW "," #NB "," #CIN "," #NW "w,lat>", c3k2_fused_kernel<H_, TH, TW, NB, CIN, NW, D, 1>}
const Class kClasses[] = {
    // Tiles chosen by end-to-end A/B at 640^2: twice the workgroups of the first choice (8x16 / 8x8 / 4x8) cost more
    // halo recompute but cut the serial frame by ~10 us at equal throughput; one step smaller still (4x8 / 4x4 / 4x4)
    // lost 4-6 % throughput; deeper weight queues (D x2, x1.5) changed nothing.
    C3K2(32, 8, 8, 1, 64, 8, 4),       // backbone.stage1_block        160^2 at 640: 400 workgroups
    C3K2(32, 8, 8, 1, 128, 8, 4),      // neck.fpn_c3k2_2
    C3K2(64, 4, 8, 2, 128, 8, 8),      // backbone.stage2_c3k2          80^2: 200
    C3K2(64, 4, 8, 1, 256, 8, 8),      // neck.fpn_c3k2_1
    C3K2T(64, 4, 8, 1, 256, 8, 8),     // neck.fpn_c3k2_1 + neck.lateral_p2 (+ x2 upsample)
    C3K2(64, 4, 8, 1, 192, 8, 8),      // neck.pan_c3k2_1
    C3K2(128, 4, 4, 2, 256, 8, 16),    // backbone.stage3_c3k2          40^2: 100
    C3K2(128, 4, 4, 1, 384, 8, 16),    // neck.pan_c3k2_2
    C3K2(128, 4, 4, 1, 512, 8, 16),    // graph (B) fpn_c3k2_1 (qat.py:397)
    C3K2T(128, 4, 4, 1, 512, 8, 16),   // graph (B) fpn_c3k2_1 + lateral_p3
};
#undef C3K2
#undef C3K2T
const Class* find_class(int hid, int nb, int cin, int tail) {
  for (const Class& c : kClasses)
    if (c.hid == hid && c.nb == nb && c.cin == cin && c.tail == tail) return &c;
  return nullptr;
}
constexpr int kMaxLds = 160 * 1024;
int align_up(int v, int a) { return (v + a - 1) / a * a; }

}  // namespace

hipError_t c3k2_init() {
  for (const Class& c : kClasses) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

bool c3k2_supported(int hid, int nb, int cin, int tail) {
  C3k2Params p;
  memset(&p, 0, sizeof p);
  p.hid = hid; p.nb = nb; p.Cin = cin; p.tail = tail; p.H = p.W = 64;
  return c3k2_layout(&p);
}

// Fills tile geometry and the LDS layout of `p` (needs hid, nb, Cin, H, W). False = no such class / no fit.
bool c3k2_layout(C3k2Params* p) {
  const Class* c = find_class(p->hid, p->nb, p->Cin, p->tail);
  if (!c) return false;
  const int h = p->hid, nb = p->nb;
  const int p0 = (c->th + 2 * nb) * (c->tw + 2 * nb), p1 = (c->th + 2) * (c->tw + 2), pt = c->th * c->tw;
  p->tiles_x = (p->W + c->tw - 1) / c->tw;
  p->tiles_y = (p->H + c->th - 1) / c->th;
  p->tiles_x_magic = div_magic((unsigned)p->tiles_x);
  p->n_bias = 2 * h * (2 + nb) + (p->tail ? h : 0);
  const int x_bytes = align_up(p0 * p->Cin * 2, 1024) + 1024;  // the last patch DMA instruction may overrun by < 1 KiB
  const int t_bytes = p0 * h * 2, u1_bytes = nb == 2 ? p1 * h * 2 : 0, u2_bytes = pt * h * 2;
  const int stage_bytes = pt * (2 * h * 2 + 16);
  // region A: the input patch; once step 0 has consumed it, t (and later the output staging tile), u1 and u2 live there
  const int head = t_bytes > stage_bytes ? t_bytes : stage_bytes;
  const int a_need = align_up(head, 16) + align_up(u1_bytes, 16) + align_up(u2_bytes, 16);
  const int a_bytes = x_bytes > a_need ? x_bytes : a_need;
  int off = 0;
  p->off_bias = off; off += align_up(p->n_bias * 4, 1024);
  p->off_x = off;
  p->off_t = off;
  p->off_stage = off;
  p->off_u1 = off + align_up(head, 16);
  p->off_u2 = p->off_u1 + align_up(u1_bytes, 16);
  off += align_up(a_bytes, 1024);
  p->off_y = off;
  p->off_tail = off;                                          // the tail's output tile replaces a | b (dead after cv3)
  const int y_bytes = p0 * 2 * h * 2, tail_bytes = p->tail ? pt * (h * 2 + 16) : 0;
  off += align_up(y_bytes > tail_bytes ? y_bytes : tail_bytes, 1024);
  p->smem_bytes = off;
  return off <= kMaxLds;
}

// Packs the weights of a block's convs (each given as the exporter's [n/16][K/32] 1-KiB fragment blocks, up to two
// output slices) into the stream the block kernels read: per conv, k-block-major [K/32][N/16] blocks (the wave that
// owns subtile j of step s reads blocks blk(s) + kb*ns + j, kb = 0..), and concatenates the biases.
void block_pack(const C3k2Conv* convs, int nconv, std::vector<unsigned char>* stream, std::vector<float>* bias) {
  stream->clear();
  bias->clear();
  for (int ci = 0; ci < nconv; ++ci) {
    const C3k2Conv& cv = convs[ci];
    const int ns = (cv.n[0] + cv.n[1]) / 16, kbn = cv.K / 32;
    const size_t base = stream->size();
    stream->resize(base + (size_t)kbn * ns * 1024, 0);
    for (int kb = 0; kb < kbn; ++kb)
      for (int s = 0; s < ns; ++s) {
        const int seg = s * 16 < cv.n[0] ? 0 : 1;
        const int ls = seg ? s - cv.n[0] / 16 : s;
        memcpy(stream->data() + base + ((size_t)kb * ns + s) * 1024, cv.w[seg] + ((size_t)ls * kbn + kb) * 1024, 1024);
      }
    for (int seg = 0; seg < 2; ++seg)
      for (int i = 0; i < cv.n[seg]; ++i) bias->push_back(cv.bias[seg][i]);
  }
}

// C3k2 convs in execution order: cv1|cv2, {b.cv1, b.cv2} x nb, cv3.
bool c3k2_pack(int hid, int nb, int cin, int tail, const C3k2Conv* convs, std::vector<unsigned char>* stream, std::vector<float>* bias) {
  if (!find_class(hid, nb, cin, tail)) return false;
  const int ncv3 = 1 + 2 * nb, nconv = 2 + 2 * nb + (tail ? 1 : 0);
  for (int ci = 0; ci < nconv; ++ci) {
    const C3k2Conv& cv = convs[ci];
    const int want_n = (ci == 0 || ci == ncv3) ? 2 * hid : hid;
    const int want_k = ci == 0 ? cin : (ci >= ncv3 ? 2 * hid : ((ci & 1) ? hid : 9 * hid));
    if (cv.n[0] + cv.n[1] != want_n || cv.K != want_k || cv.n[0] % 16 || cv.n[1] % 16) return false;
  }
  block_pack(convs, nconv, stream, bias);
  return true;
}

hipError_t c3k2_launch(const C3k2Params& p, hipStream_t stream) {
  const Class* c = find_class(p.hid, p.nb, p.Cin, p.tail);
  if (!c) return hipErrorInvalidValue;
  hipLaunchKernelGGL(c->fn, dim3(p.tiles_x * p.tiles_y, 1, 1), dim3(c->nw * 64, 1, 1), p.smem_bytes, stream, p);
  return hipGetLastError();
}

const char* c3k2_kernel_name(int hid, int nb, int cin, int tail) {
  const Class* c = find_class(hid, nb, cin, tail);
  return c ? c->name : "c3k2_fused<?>";
}

int c3k2_block_threads(int hid, int nb, int cin, int tail) {
  const Class* c = find_class(hid, nb, cin, tail);
  return c ? c->nw * 64 : 0;
}

}  // namespace unina
