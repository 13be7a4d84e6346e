// block_kernels.h -- device bodies of the block-fused kernels (included by c3k2_fused.hip, head_fused.hip and
// block_dual.hip, which runs one body of each side by side in one grid). `bid` = workgroup index inside its own
// kernel's grid, `smem` = the launch's dynamic LDS.
#pragma once
#include "kernels.h"
#include "block_pipeline.h"

namespace unina {
namespace dev {

// step 0: cv1|cv2 (K = CIN, N = 2h); odd steps: bottleneck 1x1 (K = h, N = h); even steps: bottleneck 3x3 (K = 9h,
// N = h); last step: cv3 (K = 2h, N = 2h). A wave owns one channel subtile, or ns / NW of them when ns > NW.
// TAIL = 1 appends the lateral 1x1 conv (2h -> h, + nearest x2 upsample in its store) that follows an FPN block;
// TAIL = 2 appends a plain 1x1 ConvBlock 2h -> h on the block's output (stage3_c3k2 -> sppf.cv1, model.py:215-216);
// TAIL = 3 is TAIL = 1 for an int8 block whose lateral writes an fp16 tensor (INT8 engines: fpn_c3k2_1 -> lateral_p2
// into the fp16 concat buffer of the narrow fpn_c3k2_2 block).
// KBLK = k per weight block (32 fp16 / 64 int8).
// CPRE != 0 prepends the 3x3 / stride-2 ConvBlock (CPRE -> CX channels) that produces the first CX channels of the block's
// input: all of it (stage1_conv -> stage1_block, model.py:177-190) or the down-sampling half of a PAN concat (down1 ->
// pan_c3k2_1 over [p2_down | p3_fused], model.py:263-268; the other CIN - CX channels come from HBM as before). Its output
// on the block's input region is computed in place instead of being written to HBM by one launch and DMA'd back by the next.
template <int H_, int NB, int CIN, int NW, int TAIL, int KBLK = 32, int CPRE = 0, int CX = CIN>
struct C3k2Plan {
  static constexpr int PRE = CPRE ? 1 : 0;
  static constexpr int CV3 = PRE + 1 + 2 * NB;  // index of the cv3 step
  static constexpr int N = PRE + 2 + 2 * NB + (TAIL ? 1 : 0);
  static constexpr int kb(int s) {
    if (PRE && s == 0) return 9 * CPRE / KBLK;
    const int t = s - PRE;
    return t == 0 ? CIN / KBLK : (s >= CV3 ? 2 * H_ / KBLK : ((t & 1) ? H_ / KBLK : 9 * H_ / KBLK));
  }
  static constexpr int ns(int s) {
    if (PRE && s == 0) return CX / 16;
    return (s == PRE || s == CV3) ? 2 * H_ / 16 : H_ / 16;
  }
  static constexpr int wnt(int s) { return ns(s) <= NW ? 1 : ns(s) / NW; }
  static constexpr int nch(int s) { return 16 * ns(s); }              // output channels of step s
  static constexpr int cfirst(int s) {                                 // channels of all earlier steps
    int t = 0;
    for (int i = 0; i < s; ++i) t += nch(i);
    return t;
  }
};


// NW waves per workgroup (roles per step: Steps::waves_n / wnt / waves_m); D = weight prefetch depth in 1-KiB blocks
// per wave. There is no wave-uniform branch around any global load: the kernel is straight-line code, so the
// compiler's counted s_waitcnt vmcnt keeps D loads in flight across every step boundary.
// E = EltH (fp16 engines and carve-outs) or EltI8 (INT8 engines: every tensor of the block is an int8 code image with
// its per-tensor scale; the epilogues re-quantise exactly as the per-op kernels do, conv_igemm.hip conv_epilogue).
template <int H_, int TH, int TW, int NB, int CIN, int NW, int D, int TAIL = 0, typename E = EltH, int CPRE = 0, int CX = CIN, bool STAMPS = false>
__device__ __forceinline__ void c3k2_fused_body(const C3k2Params& p, int bid, unsigned char* smem) {
  // debug twin: shader-clock stamp k of the mid workgroup (0 entry, 1 patch + first weights landed, then one per step, the
  // output store and the tail), slot 15 = the 100 MHz wall clock at entry, 14 at the end
  auto stamp = [&](int k) {
    if constexpr (STAMPS) {
      if (p.stamps && bid == ((p.tiles_x * p.tiles_y) >> 1) && threadIdx.x == 0) {
        p.stamps[k] = __builtin_amdgcn_s_memtime();
        if (k == 0) p.stamps[15] = wall_clock64();
      }
    }
  };
  stamp(0);
  static_assert(NB == 1 || NB == 2, "bottleneck count");
  typedef C3k2Plan<H_, NB, CIN, NW, TAIL, E::KBLK, CPRE, CX> PL;
  constexpr int PRE = PL::PRE;
  static_assert(CPRE % E::KBLK == 0 && CX % E::KBLK == 0 && CX <= CIN && (PRE || CX == CIN), "pre-conv channel split");
  constexpr int CREST = CIN - CX;   // input channels that still come from HBM when a pre-conv produces the first CX
  typedef StepTable<PL, NW> ST;
  static_assert(ST::valid(), "wave roles");
  static_assert((NW & (NW - 1)) == 0 && NW >= 2 && NW <= 16, "waves per workgroup");
  static_assert(H_ % E::KBLK == 0 && CIN % E::KBLK == 0, "a weight block must not straddle a tap");
  constexpr int R0W = TW + 2 * NB, P0 = (TH + 2 * NB) * R0W;   // input / first-level region (tile + NB-pixel halo)
  constexpr int R1W = TW + 2, P1 = (TH + 2) * R1W;              // NB == 2: region of the first bottleneck's output
  constexpr int PT = TH * TW;
  constexpr int HB = H_ / E::KBLK;                              // k-blocks per tap of the hidden width
  constexpr int NT = NW * 64;
  constexpr int ESZ = E::ESZ;
  constexpr int NPL = E::SPLIT ? 2 : 1;     // split fp16: every tensor is a (hi, lo) pair of images / planes
  static_assert(!E::SPLIT || !UNINA_BLOCK_PATCH_REGS, "the split type loads its patches by LDS-DMA");

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int tyi = fast_div(bid, p.tiles_x_magic), txi = bid - tyi * p.tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int lds_lo = E::SPLIT ? p.lds_lo : 0;   // LDS distance image -> lo twin

  // ---- weight prefetch queue: element g of this wave's flat sequence lives in slot g % D ----
  const unsigned char* wbase = p.wstream + lane * 16;   // this lane's 16 bytes of any block
  typename E::frag q[D];
  // per-channel constants of every step -> LDS
  float* bias_lds = reinterpret_cast<float*>(smem + p.off_bias);
  floatx4 cregs[kConstVecs];
#define CST(S) (bias_lds + E::CM * PL::cfirst((S) + PRE))

  // input patch (tile + NB-pixel halo, all CIN channels) -> LDS image; with a pre-conv: ITS input footprint instead
  constexpr Img X = make_img(0, CX / E::CH);                               // (CX == CIN without a pre-conv)
  constexpr Img XR = make_img(0, (CREST ? CREST : E::KBLK) / E::CH);       // the HBM part next to a pre-conv's output
  constexpr int PH = 2 * (TH + 2 * NB - 1) + 3, PW = 2 * (R0W - 1) + 3;   // footprint of R0 under a 3x3 / stride-2 conv
#if UNINA_BLOCK_PATCH_REGS
  // patch (and constants) through registers, requested BEFORE the weight queue: the commits below then wait only for
  // them (counted vmcnt) and the D weight blocks stay in flight across the barrier
  PatchRegs<PH, PW, (CPRE ? CPRE : E::KBLK), NT, E> pr_pre;
  PatchRegs<TH + 2 * NB, R0W, (CREST ? CREST : E::KBLK), NT, E> pr_rest;
  PatchRegs<TH + 2 * NB, R0W, CIN, NT, E> pr_x;
  if constexpr (PRE) {
    patch_issue<PH, PW, (CPRE ? CPRE : E::KBLK), NT, E>(pr_pre, p.src, p.src_ld, p.preH, p.preW, 2 * (ty0 - NB) - 1, 2 * (tx0 - NB) - 1, wid, lane);
    if constexpr (CREST > 0)
      patch_issue<TH + 2 * NB, R0W, (CREST ? CREST : E::KBLK), NT, E>(pr_rest, p.src2, p.src2_ld, p.H, p.W, ty0 - NB, tx0 - NB, wid, lane);
  } else {
    patch_issue<TH + 2 * NB, R0W, CIN, NT, E>(pr_x, p.src, p.src_ld, p.H, p.W, ty0 - NB, tx0 - NB, wid, lane);
  }
  consts_issue<NT>(cregs, p.bias, p.n_bias);
  static_for<0, D>([&](auto gc) { wq_fetch<ST, D, decltype(gc)::value>(q, wbase, wid); });
  if constexpr (PRE) {
    patch_commit<PH, PW, (CPRE ? CPRE : E::KBLK), NT, E>(pr_pre, smem + p.off_p, wid, lane);
    if constexpr (CREST > 0) patch_commit<TH + 2 * NB, R0W, (CREST ? CREST : E::KBLK), NT, E>(pr_rest, smem + p.off_xr, wid, lane);
  } else {
    patch_commit<TH + 2 * NB, R0W, CIN, NT, E>(pr_x, smem + p.off_x, wid, lane);
  }
  consts_commit<NT>(cregs, bias_lds, p.n_bias);
  lds_barrier();
#else
  if constexpr (PRE) {
    load_patch<PH, PW, (CPRE ? CPRE : E::KBLK), NT, E>(smem + p.off_p, p.src, p.src_ld, p.preH, p.preW, 2 * (ty0 - NB) - 1,
                                                     2 * (tx0 - NB) - 1, p.zeros, wid, lane, p.src_lo, lds_lo);
    if constexpr (CREST > 0)
      load_patch<TH + 2 * NB, R0W, (CREST ? CREST : E::KBLK), NT, E>(smem + p.off_xr, p.src2, p.src2_ld, p.H, p.W, ty0 - NB, tx0 - NB,
                                                                   p.zeros, wid, lane, p.src2_lo, lds_lo);
  } else {
    load_patch<TH + 2 * NB, R0W, CIN, NT, E>(smem + p.off_x, p.src, p.src_ld, p.H, p.W, ty0 - NB, tx0 - NB, p.zeros, wid, lane, p.src_lo, lds_lo);
  }
  static_for<0, D>([&](auto gc) { wq_fetch<ST, D, decltype(gc)::value>(q, wbase, wid); });
  consts_issue<NT>(cregs, p.bias, p.n_bias);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the patch has landed (LDS-DMA is not tracked by the compiler)
  consts_commit<NT>(cregs, bias_lds, p.n_bias);
  lds_barrier();

#endif

  stamp(1);
  const Img Xi = Img{p.off_x, X.nch, X.sh, X.mask};
  const Img XRi = Img{p.off_xr, XR.nch, XR.sh, XR.mask};
  const Img Y = make_img(p.off_y, 2 * H_ / E::CH);   // a | b on R0
  const Img T = make_img(p.off_t, H_ / E::CH);       // t of the current bottleneck (R0, then R1)
  const Img U1 = make_img(p.off_u1, H_ / E::CH);     // NB == 2: first bottleneck's output on R1
  const Img U2 = make_img(p.off_u2, H_ / E::CH);     // last bottleneck's output on the tile
  auto in_image = [&](int iy, int ix) { return (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W; };
  // per-lane "inside the image" bits of the two halo'd regions, one bit per 16-pixel subtile (bit i: pixel 16 i + l15), computed
  // ONCE: the epilogues of the 1x1 steps in front of a 3x3 zero their output outside the image (the 3x3's zero padding) and
  // used to redo the row / column division and the range test per accumulator, behind an exec-masked branch
  unsigned inm0 = 0u, inm1 = 0u;
#pragma unroll
  for (int i = 0; i < (P0 + 15) / 16; ++i) {
    const int r = i * 16 + l15, ry = r / R0W, rx = r - ry * R0W;
    inm0 |= (r < P0 && in_image(ty0 - NB + ry, tx0 - NB + rx)) ? (1u << i) : 0u;
  }
  if constexpr (NB == 2) {
#pragma unroll
    for (int i = 0; i < (P1 + 15) / 16; ++i) {
      const int r = i * 16 + l15, ry = r / R1W, rx = r - ry * R1W;
      inm1 |= (r < P1 && in_image(ty0 - 1 + ry, tx0 - 1 + rx)) ? (1u << i) : 0u;
    }
  }
  auto masked = [](floatx4 v, unsigned bits, int sub) {   // v, or +0 where the lane's bit `sub` is clear: branch-free, exact
    const int m = -(int)((bits >> sub) & 1u);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __int_as_float(__float_as_int(v[r]) & m);
    return v;
  };

  auto run_step = [&](auto sc, auto pc, auto nc, const float* cst, auto baddr, auto epi) {
    dev::run_step<ST, D, decltype(sc)::value, decltype(pc)::value, E, decltype(nc)::value>(q, wbase, smem, wid, lane, baddr, epi, lds_lo, cst);
  };
  // (step, pixels, channels of the step's constant arrays, the arrays)
#define STEP(S, P, NCH) std::integral_constant<int, (S) + PRE>{}, std::integral_constant<int, (P)>{}, std::integral_constant<int, (NCH)>{}, CST(S)
  typedef typename E::acc_t acc_t;
  typedef StepConsts<E> KC;

  if constexpr (PRE) {
    // ---- pre-step: x = ReLU(3x3/s2 conv of the patch + b) on R0 -> the block's input image ----
    constexpr Img PI0 = make_img(0, (CPRE ? CPRE : E::KBLK) / E::CH);
    const Img PI = Img{p.off_p, PI0.nch, PI0.sh, PI0.mask};
    constexpr int CBP = (CPRE ? CPRE : E::KBLK) / E::KBLK;
    const float* c0 = bias_lds;
    dev::run_step<ST, D, 0, P0, E, CX>(q, wbase, smem, wid, lane,
        [&](int sub, auto kc) {
          constexpr int kb = decltype(kc)::value, tap = kb / CBP, cb = kb - tap * CBP, th3 = tap / 3;
          int r = sub * 16 + l15;
          r = r < P0 ? r : P0 - 1;
          const int ry = r / R0W, rx = r - ry * R0W;
          return PI.addr((2 * ry + th3) * PW + 2 * rx + (tap - th3 * 3), cb * 4 + lq);
        },
        [&](int sub, int n, const acc_t& acc, const KC& k) {
          const int r = sub * 16 + l15;
          if (r < P0) store4<E>(smem + img_at<E>(Xi, r, n), act_relu<E>(acc, k), k, lds_lo);
        }, lds_lo, c0);
    stamp(2);
  }

  // ---- step 0: a | b = ReLU(W12 x + b12) on R0 --------------------------------------------------------------------
  run_step(STEP(0, P0, 2 * H_),
      [&](int sub, auto kc) {
        constexpr int kb = decltype(kc)::value;
        const int r = sub * 16 + l15;
        if constexpr (kb < CX / E::KBLK) return Xi.addr(r < P0 ? r : P0 - 1, kb * 4 + lq);
        else return XRi.addr(r < P0 ? r : P0 - 1, (kb - CX / E::KBLK) * 4 + lq);
      },
      [&](int sub, int n, const acc_t& acc, const KC& k) {
        const int r = sub * 16 + l15;
        if (r < P0) store4<E>(smem + img_at<E>(Y, r, n), act_relu<E>(acc, k), k, lds_lo);
      });

  stamp(3);
  // ---- bottleneck 0 -----------------------------------------------------------------------------------------------
  // t = ReLU(Wb1 a + b) on R0, forced to 0 outside the image (zero padding of the 3x3 that follows)
  run_step(STEP(1, P0, H_),
      [&](int sub, auto kc) {
        const int r = sub * 16 + l15;
        return Y.addr(r < P0 ? r : P0 - 1, decltype(kc)::value * 4 + lq);
      },
      [&](int sub, int n, const acc_t& acc, const KC& k) {
        if constexpr (STAMPS) { if (sub == 0) stamp(12); }      // (debug twin: this step's K loop is done, its epilogue starts)
        const int r = sub * 16 + l15;
        const floatx4 v = masked(act_relu<E>(acc, k), inm0, sub);
        if (r < P0) store4<E>(smem + img_at<E>(T, r, n), v, k, lds_lo);
        if constexpr (STAMPS) { if (sub == (P0 + 15) / 16 - 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(13); } }   // (... and its LDS stores have landed)
      });
  stamp(4);
  if constexpr (NB == 1) {
    // u = ReLU(3x3(t) + b) + a on the tile
    run_step(STEP(2, PT, H_),
        [&](int sub, auto kc) {
          constexpr int kb = decltype(kc)::value, tap = kb / HB, cb = kb - tap * HB, th3 = tap / 3;
          int pp = sub * 16 + l15;
          pp = pp < PT ? pp : PT - 1;
          const int py = pp / TW, px = pp - py * TW;
          return T.addr((py + th3) * R0W + px + (tap - th3 * 3), cb * 4 + lq);
        },
        [&](int sub, int n, const acc_t& acc, const KC& k) {
          const int pp = sub * 16 + l15;
          if (pp >= PT) return;
          const int py = pp / TW, px = pp - py * TW;
          const floatx4 v = add_res<E>(act_relu<E>(acc, k), smem, Y, (py + 1) * R0W + px + 1, n, p.res_scale[0], lds_lo);
          store4<E>(smem + img_at<E>(U2, pp, n), v, k, lds_lo);
        });
  } else {
    // u1 = ReLU(3x3(t1) + b) + a on R1
    run_step(STEP(2, P1, H_),
        [&](int sub, auto kc) {
          constexpr int kb = decltype(kc)::value, tap = kb / HB, cb = kb - tap * HB, th3 = tap / 3;
          int pp = sub * 16 + l15;
          pp = pp < P1 ? pp : P1 - 1;
          const int py = pp / R1W, px = pp - py * R1W;
          return T.addr((py + th3) * R0W + px + (tap - th3 * 3), cb * 4 + lq);
        },
        [&](int sub, int n, const acc_t& acc, const KC& k) {
          const int pp = sub * 16 + l15;
          if (pp >= P1) return;
          const int py = pp / R1W, px = pp - py * R1W;
          const floatx4 v = add_res<E>(act_relu<E>(acc, k), smem, Y, (py + 1) * R0W + px + 1, n, p.res_scale[0], lds_lo);
          store4<E>(smem + img_at<E>(U1, pp, n), v, k, lds_lo);
        });
    stamp(5);
    // ---- bottleneck 1 ---------------------------------------------------------------------------------------------
    // t2 = ReLU(Wb1' u1 + b) on R1, 0 outside the image
    run_step(STEP(3, P1, H_),
        [&](int sub, auto kc) {
          const int r = sub * 16 + l15;
          return U1.addr(r < P1 ? r : P1 - 1, decltype(kc)::value * 4 + lq);
        },
        [&](int sub, int n, const acc_t& acc, const KC& k) {
          const int r = sub * 16 + l15;
          const floatx4 v = masked(act_relu<E>(acc, k), inm1, sub);
          if (r < P1) store4<E>(smem + img_at<E>(T, r, n), v, k, lds_lo);
        });
    stamp(6);
    // u2 = ReLU(3x3(t2) + b) + u1 on the tile
    run_step(STEP(4, PT, H_),
        [&](int sub, auto kc) {
          constexpr int kb = decltype(kc)::value, tap = kb / HB, cb = kb - tap * HB, th3 = tap / 3;
          int pp = sub * 16 + l15;
          pp = pp < PT ? pp : PT - 1;
          const int py = pp / TW, px = pp - py * TW;
          return T.addr((py + th3) * R1W + px + (tap - th3 * 3), cb * 4 + lq);
        },
        [&](int sub, int n, const acc_t& acc, const KC& k) {
          const int pp = sub * 16 + l15;
          if (pp >= PT) return;
          const int py = pp / TW, px = pp - py * TW;
          const floatx4 v = add_res<E>(act_relu<E>(acc, k), smem, U1, (py + 1) * R1W + px + 1, n, p.res_scale[1], lds_lo);
          store4<E>(smem + img_at<E>(U2, pp, n), v, k, lds_lo);
        });
  }

  stamp(7);
  // ---- last step: y = ReLU(W3 [u | b] + b3) on the tile -> staging image (linear rows) -> HBM ----------------------
  constexpr int ROWB = 2 * H_ * ESZ + 16;  // staged output row: 2h elements + 16 bytes of padding (bank spread)
  unsigned char* stage = smem + p.off_stage;
  constexpr int S3 = 1 + 2 * NB;
  run_step(STEP(S3, PT, 2 * H_),
      [&](int sub, auto kc) {
        constexpr int kb = decltype(kc)::value;
        int pp = sub * 16 + l15;
        pp = pp < PT ? pp : PT - 1;
        if constexpr (kb < HB) {                                  // k-blocks of channels [0, h): u
          return U2.addr(pp, kb * 4 + lq);
        } else {                                                  // k-blocks of channels [h, 2h): b = second half of Y
          const int py = pp / TW, px = pp - py * TW;
          return Y.addr((py + NB) * R0W + px + NB, H_ / E::CH + (kb - HB) * 4 + lq);
        }
      },
      [&](int sub, int n, const acc_t& acc, const KC& k) {
        const int pp = sub * 16 + l15;
        if (pp >= PT) return;
        store4<E>(stage + pp * ROWB + n * ESZ, act_relu<E>(acc, k), k, lds_lo);
      });
  stamp(8);
  typedef float vec16 __attribute__((ext_vector_type(4)));  // 16 opaque bytes
  constexpr int CPR = 2 * H_ * ESZ / 16;                    // 16-byte chunks per output pixel
  unsigned char* dst = static_cast<unsigned char*>(p.dst);
  for (int c = threadIdx.x; c < NPL * PT * CPR; c += NT) {
    const int pl = NPL == 1 ? 0 : c / (PT * CPR), cc = c - pl * (PT * CPR);      // (split: the hi tile, then the lo tile)
    const int pp = cc / CPR, ch = cc - pp * CPR;
    const int oy = ty0 + pp / TW, ox = tx0 + pp % TW;
    if (oy < p.H && ox < p.W) {
      const vec16 v = *reinterpret_cast<const vec16*>(stage + pl * lds_lo + pp * ROWB + ch * 16);
      *reinterpret_cast<vec16*>(dst + (NPL == 1 ? 0 : pl * p.dst_lo) + ((size_t)(oy * p.W + ox) * p.dst_ld) * ESZ + ch * 16) = v;
      if constexpr (!E::I8 && !E::SPLIT) {
        if (p.dst_q) {   // int8 twin of the output for the block's int8 consumers: stem_pool.hip quant_f16_i8_kernel's arithmetic
          const half8 hv = *reinterpret_cast<const half8*>(&v);
          unsigned int q[2] = {0u, 0u};
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            float t = __builtin_rintf((float)hv[r] * p.q_inv);
            t = t > 127.f ? 127.f : (t < -127.f ? -127.f : t);
            q[r >> 2] |= ((unsigned int)(int)t & 0xFFu) << (8 * (r & 3));
          }
          *reinterpret_cast<uint2*>(p.dst_q + (size_t)(oy * p.W + ox) * p.dst_q_ld + ch * 8) = make_uint2(q[0], q[1]);
        }
      }
    }
  }

  stamp(9);
  if constexpr (TAIL) {
    // ---- tail: lateral 1x1 (model.py:256,259: ConvBlock 2h -> h) on the block's output, still in the staging image,
    //      then nearest x2 upsample (model.py:145-147) in the store: each pixel's h channels go to its 2x2 block ----
    const Img YS = Img{p.off_stage, ROWB / 16, 0, 0};        // the staging image is linear: pitch ROWB, no swizzle
    typedef typename std::conditional<TAIL == 3, EltH, E>::type TE;   // element type of the tail's output
    constexpr int TSZ = TE::ESZ;
    constexpr int ROWT = H_ * TSZ + 16;
    constexpr int S4 = 2 + 2 * NB;
    unsigned char* tout = smem + p.off_tail;
    run_step(STEP(S4, PT, H_),
        [&](int sub, auto kc) {
          int pp = sub * 16 + l15;
          pp = pp < PT ? pp : PT - 1;
          return YS.addr(pp, decltype(kc)::value * 4 + lq);
        },
        [&](int sub, int n, const acc_t& acc, const KC& k) {
          const int pp = sub * 16 + l15;
          if (pp >= PT) return;
          store4<TE>(tout + pp * ROWT + n * TSZ, act_relu<E>(acc, k), k, lds_lo);
        });
    stamp(10);
    constexpr int CPT = H_ * TSZ / 16;
    unsigned char* dst2 = static_cast<unsigned char*>(p.dst2);
    if constexpr (TAIL == 1 || TAIL == 3) {
      const size_t px = (size_t)p.dst2_ld * TSZ, row = (size_t)(2 * p.W) * px;
      for (int c = threadIdx.x; c < NPL * PT * CPT; c += NT) {
        const int pl = NPL == 1 ? 0 : c / (PT * CPT), cc = c - pl * (PT * CPT);
        const int pp = cc / CPT, ch = cc - pp * CPT;
        const int oy = ty0 + pp / TW, ox = tx0 + pp % TW;
        if (oy < p.H && ox < p.W) {
          const vec16 v = *reinterpret_cast<const vec16*>(tout + pl * lds_lo + pp * ROWT + ch * 16);
          unsigned char* d = dst2 + (NPL == 1 ? 0 : pl * p.dst2_lo) + ((size_t)(2 * oy) * (2 * p.W) + 2 * ox) * px + ch * 16;
          *reinterpret_cast<vec16*>(d) = v;
          *reinterpret_cast<vec16*>(d + px) = v;
          *reinterpret_cast<vec16*>(d + row) = v;
          *reinterpret_cast<vec16*>(d + row + px) = v;
        }
      }
    } else {
      for (int c = threadIdx.x; c < NPL * PT * CPT; c += NT) {
        const int pl = NPL == 1 ? 0 : c / (PT * CPT), cc = c - pl * (PT * CPT);
        const int pp = cc / CPT, ch = cc - pp * CPT;
        const int oy = ty0 + pp / TW, ox = tx0 + pp % TW;
        if (oy < p.H && ox < p.W)
          *reinterpret_cast<vec16*>(dst2 + (NPL == 1 ? 0 : pl * p.dst2_lo) + ((size_t)(oy * p.W + ox) * p.dst2_ld) * TSZ + ch * 16) =
              *reinterpret_cast<const vec16*>(tout + pl * lds_lo + pp * ROWT + ch * 16);
      }
    }
  }
  if constexpr (STAMPS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(11);
    if (p.stamps && bid == ((p.tiles_x * p.tiles_y) >> 1) && threadIdx.x == 0) p.stamps[14] = wall_clock64();
  }
#undef STEP
#undef CST
}

template <int C>
struct HeadPlan {
  static constexpr int N = 3;
  static constexpr int kb(int s) { return s == 2 ? C / 32 : 9 * C / 32; }
  static constexpr int ns(int s) { return s == 2 ? 2 : 2 * C / 16; }
  static constexpr int wnt(int s) { return s == 2 ? 1 : 2; }   // wave tile 2 channel subtiles x up to 6 pixel subtiles
};


template <int C, int TH, int TW, int NW, int D>
__device__ __forceinline__ void head_fused_body(const HeadParams& p, int bid, unsigned char* smem) {
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
  const int tyi = fast_div(bid, p.tiles_x_magic), txi = bid - tyi * p.tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TW;

  const unsigned char* wbase = p.wstream + lane * 16;
  half8 q[D];

  float* bias_lds = reinterpret_cast<float*>(smem + p.off_bias);
  floatx4 cregs[kConstVecs];
  constexpr Img X = make_img(0, C / 8);
#if UNINA_BLOCK_PATCH_REGS
  PatchRegs<R0H, R0W, C, NT> pr_x;
  patch_issue<R0H, R0W, C, NT>(pr_x, p.src, p.src_ld, p.H, p.W, ty0 - 2, tx0 - 2, wid, lane);
  consts_issue<NT>(cregs, p.bias, p.n_bias);
  static_for<0, D>([&](auto gc) { wq_fetch<ST, D, decltype(gc)::value>(q, wbase, wid); });
  patch_commit<R0H, R0W, C, NT>(pr_x, smem + p.off_x, wid, lane);
  consts_commit<NT>(cregs, bias_lds, p.n_bias);
  lds_barrier();
#else
  load_patch<R0H, R0W, C, NT>(smem + p.off_x, p.src, p.src_ld, p.H, p.W, ty0 - 2, tx0 - 2, p.zeros, wid, lane);
  static_for<0, D>([&](auto gc) { wq_fetch<ST, D, decltype(gc)::value>(q, wbase, wid); });
  consts_issue<NT>(cregs, p.bias, p.n_bias);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the patch has landed
  consts_commit<NT>(cregs, bias_lds, p.n_bias);
  lds_barrier();

#endif

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


// ------------------------------------------------------------------------------------------------------------------
// head_ws_body: the same DetectionHead as head_fused_body, ROW-STREAMING with the weights stationary in registers
// (the idea of conv_igemm.hip's conv3x3_ws, chained over the head's two 3x3 layers). A workgroup owns a strip of TH x 14
// output pixels. Wave w keeps ALL weight blocks of channel subtile w of layer 0 (cls.0 | reg.0: 2C channels, K = 9C) and of
// layer 1 (cls.1 | reg.1: the same subtile index, its branch = w / (C/16)) in registers -- 2 x 9C/32 blocks = 144 VGPRs at
// C = 64 -- and walks down the input patch rows:
//   layer 0: the fragment of patch row rho (kx, cb) feeds h0 rows rho, rho-1, rho-2 (ky = 0, 1, 2); a finished h0 row (16
//            pixels wide = one MFMA subtile: hence 14 output columns) gets bias + ReLU, is zeroed outside the image (layer 1's
//            zero padding) and goes to a two-row LDS ring as fp16 -- every wave writes its 16 channels;
//   layer 1: one iteration later (a barrier in between) the h0 row feeds output rows i, i-1, i-2 the same way; a finished h1
//            row goes to the LDS h1 tile;
//   layer 2: after the last row, the 1x1 output convs read h1 from LDS and store the fp32 planes.
// LDS is read once per three MFMAs, with immediate offsets (padded pixel pitch, no xor swizzle: conv3x3_wsc_body explains
// why), every weight block is fetched once per workgroup, and each output receives its products in the order (ky, kx, cb)
// from a zero accumulator with the same fp16 rounding points as head_fused_body: bit-identical results.
template <int C, int TH, int NW>
__device__ __forceinline__ void head_ws_body(const HeadParams& p, int bid, unsigned char* smem) {
  static_assert(NW * 16 == 2 * C, "one wave per 16-channel subtile of the 2C-channel layers");
  typedef half8 frag;
  typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
  constexpr int TWO = 14, CB = C / 32, KBL = 9 * CB, NS = 2 * C / 16, SPR = 3 * CB;
  constexpr int XR = TH + 4, XW = 18, HR = TH + 2, NT = NW * 64;
  constexpr int PX = C * 2 + 32, PH = 2 * C * 2 + 32;            // padded pixel pitches of the x image and of the h0 / h1 images
                                                                 // (+32: conflict-free fragment reads, tools/probes/ldsbank_probe.hip)
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int tyi = fast_div(bid, p.tiles_x_magic), txi = bid - tyi * p.tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TWO;
  const int br = wid / (C / 16);                                  // layer-1 branch of this wave's subtile: 0 = cls, 1 = reg

  const unsigned char* wbase = p.wstream + lane * 16;
  frag w0[KBL], w1[KBL], w2[CB];
  auto ld0 = [&](auto kc) { w0[decltype(kc)::value] = *reinterpret_cast<const frag*>(wbase + (size_t)(decltype(kc)::value * NS + wid) * 1024); __builtin_amdgcn_sched_barrier(0); };
  auto ld1 = [&](auto kc) { w1[decltype(kc)::value] = *reinterpret_cast<const frag*>(wbase + (size_t)(KBL * NS + decltype(kc)::value * NS + wid) * 1024); __builtin_amdgcn_sched_barrier(0); };
  // request order = order of first use: layer 0's ky = 0 blocks, the patch, the rest of layer 0, layer 1, layer 2
  static_for<0, 3 * CB>(ld0);
  constexpr int nchx = C / 8, nslots = XR * XW * nchx, PITER = (nslots + NT - 1) / NT;
  uintx4 pv[PITER];
  {
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.src), 0, p.H * p.W * p.src_ld * 2, 0x00020000);
#pragma unroll
    for (int it = 0; it < PITER; ++it) {
      const int sl = it * NT + (int)threadIdx.x;
      const int r = sl / nchx, cs = sl - r * nchx;
      const int ry = r / XW, rx = r - ry * XW;
      const int iy = ty0 - 2 + ry, ix = tx0 - 2 + rx;
      const bool in = sl < nslots && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const unsigned off = (unsigned)((iy * p.W + ix) * p.src_ld * 2 + (cs << 4));
      pv[it] = __builtin_amdgcn_raw_buffer_load_b128(srs, in ? off : 0x40000000u, 0, 0);   // out of range: zeros (the conv's padding)
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.n_bias * 4, 0x00020000);
  const uintx4 cvec = __builtin_amdgcn_raw_buffer_load_b128(brs, threadIdx.x * 16, 0, 0);   // (past the array: zeros, never committed)
  __builtin_amdgcn_sched_barrier(0);
  static_for<3 * CB, KBL>(ld0);
  static_for<0, KBL>(ld1);
  static_for<0, CB>([&](auto kc) {   // layer 2: branch w & 1 (see below)
    w2[decltype(kc)::value] = *reinterpret_cast<const frag*>(wbase + (size_t)(2 * KBL * NS + decltype(kc)::value * 2 + (wid & 1)) * 1024);
    __builtin_amdgcn_sched_barrier(0);
  });
  float* bias_lds = reinterpret_cast<float*>(smem + p.off_bias);
#pragma unroll
  for (int it = 0; it < PITER; ++it) {
    const int sl = it * NT + (int)threadIdx.x;
    const int r = sl / nchx, cs = sl - r * nchx;
    if (sl < nslots) *reinterpret_cast<uintx4*>(smem + p.off_x + r * PX + cs * 16) = pv[it];
  }
  if ((int)threadIdx.x * 4 < p.n_bias) reinterpret_cast<uintx4*>(bias_lds)[threadIdx.x] = cvec;
  lds_barrier();

  const unsigned xb = (unsigned)(p.off_x + l15 * PX + lq * 16);
  const unsigned hb = (unsigned)(p.off_h0 + l15 * PH + (br * (C / 8) + lq) * 16);
  const unsigned hw = (unsigned)(p.off_h0 + l15 * PH + (wid * 16 + lq * 4) * 2);
  const unsigned h1w = (unsigned)(p.off_h1 + l15 * PH + (wid * 16 + lq * 4) * 2);
  const int nch = wid * 16 + lq * 4;                               // this lane's 4 channels of the 2C-channel layers
  const bool col_in = (unsigned)(tx0 - 1 + l15) < (unsigned)p.W;   // h0 column l15 is image column tx0 - 1 + l15
  const floatx4 bias0 = *reinterpret_cast<const floatx4*>(bias_lds + nch);            // (in registers: an LDS read in every row's
  const floatx4 bias1 = *reinterpret_cast<const floatx4*>(bias_lds + 2 * C + nch);    //  epilogue would wait for the whole LDS queue)
  floatx4 a0[4], a1[4];
  frag bx[SPR], bh[SPR];
  auto xfrag = [&](auto rc, auto sc) {
    constexpr int rho = decltype(rc)::value, st = decltype(sc)::value, kx = st / CB, cb = st % CB;
    return *reinterpret_cast<const frag*>(smem + xb + ((rho * XW + kx) * PX + cb * 64));
  };
  auto hfrag = [&](auto ic, auto sc) {
    constexpr int i = decltype(ic)::value, st = decltype(sc)::value, kx = st / CB, cb = st % CB;
    return *reinterpret_cast<const frag*>(smem + hb + (((i & 1) * XW + kx) * PH + cb * 64));
  };
  static_for<0, SPR>([&](auto sc) { bx[decltype(sc)::value] = xfrag(std::integral_constant<int, 0>{}, sc); });
  __builtin_amdgcn_sched_barrier(0);

  static_for<0, XR + 1>([&](auto rc) {
    constexpr int rho = decltype(rc)::value, i1 = rho - 3;          // layer 1 works on h0 row rho - 3 (published by the last barrier)
    if constexpr (i1 >= 0 && i1 < HR) {
      static_for<0, SPR>([&](auto sc) { bh[decltype(sc)::value] = hfrag(std::integral_constant<int, i1>{}, sc); });
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (rho < XR) {                                        // ---- layer 0 on patch row rho
      if constexpr (rho < HR) a0[rho & 3] = floatx4{0.f, 0.f, 0.f, 0.f};
      static_for<0, SPR>([&](auto sc) {
        constexpr int st = decltype(sc)::value, kx = st / CB, cb = st % CB;
        static_for<0, 3>([&](auto kyc) {
          constexpr int ky = decltype(kyc)::value, i = rho - ky;
          if constexpr (i >= 0 && i < HR) a0[i & 3] = EltH::mma(w0[(ky * 3 + kx) * CB + cb], bx[st], a0[i & 3]);
        });
      });
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (rho >= 2) {                                      // h0 row rho - 2 is complete
        constexpr int i = rho - 2;
        const floatx4 v = a0[i & 3] + bias0;
        const bool in = col_in && (unsigned)(ty0 - 1 + i) < (unsigned)p.H;
        half4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = (_Float16)(in && v[e] > 0.f ? v[e] : 0.f);
        *reinterpret_cast<half4*>(smem + hw + (i & 1) * XW * PH) = hv;
      }
      if constexpr (rho + 1 < XR) {
        static_for<0, SPR>([&](auto sc) { bx[decltype(sc)::value] = xfrag(std::integral_constant<int, rho + 1>{}, sc); });
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (i1 >= 0 && i1 < HR) {                              // ---- layer 1 on h0 row i1
      if constexpr (i1 < TH) a1[i1 & 3] = floatx4{0.f, 0.f, 0.f, 0.f};
      static_for<0, SPR>([&](auto sc) {
        constexpr int st = decltype(sc)::value, kx = st / CB, cb = st % CB;
        static_for<0, 3>([&](auto kyc) {
          constexpr int ky = decltype(kyc)::value, r = i1 - ky;
          if constexpr (r >= 0 && r < TH) a1[r & 3] = EltH::mma(w1[(ky * 3 + kx) * CB + cb], bh[st], a1[r & 3]);
        });
      });
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (i1 >= 2) {                                       // h1 row i1 - 2 is complete
        constexpr int r = i1 - 2;
        const floatx4 v = a1[r & 3] + bias1;
        half4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = (_Float16)(v[e] > 0.f ? v[e] : 0.f);
        *reinterpret_cast<half4*>(smem + h1w + r * 16 * PH) = hv;
      }
    }
    lds_barrier();
  });

  // ---- layer 2: raw outputs = W2 h1[branch] + b2, planar fp32. Wave w: branch w & 1, rows (w >> 1) + 4t.
  const int br2 = wid & 1;
  const float* bias_2 = bias_lds + 4 * C;
  const int M = p.H * p.W;
  const unsigned h1r = (unsigned)(p.off_h1 + l15 * PH + (br2 * (C / 8) + lq) * 16);
  const floatx4 b2 = *reinterpret_cast<const floatx4*>(bias_2 + br2 * 16 + lq * 4);
  float* dst0 = br2 ? p.out_reg : p.out_cls;
  const int cnt = br2 ? p.n_reg : p.n_cls;
#pragma unroll
  for (int t = 0; t < (TH + 3) / 4; ++t) {
    const int r = (wid >> 1) + 4 * t;
    if (r >= TH) break;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < CB; ++kb) acc = EltH::mma(w2[kb], *reinterpret_cast<const frag*>(smem + h1r + r * 16 * PH + kb * 64), acc);
    const int oy = ty0 + r, ox = tx0 + l15;
    if (l15 >= TWO || oy >= p.H || ox >= p.W) continue;
    const floatx4 v = acc + b2;
    float* dst = dst0 + (size_t)oy * p.W + ox;
    const int c = lq * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < cnt) dst[(size_t)(c + e) * M] = v[e];
  }
}

}  // namespace dev
}  // namespace unina
