// conv_pair.hip -- two consecutive 1x1 ConvBlocks as ONE launch (block-pipeline form, block_pipeline.h).
//
// The SPPF's output conv and the FPN lateral that follows it (model.py:130-132 SPPF_DLA.cv2, 512 -> 256, then
// model.py:256 Neck.lateral_p3, 256 -> 128, whose store does the nearest x2 upsample of model.py:145-147) are two
// launches of a few microseconds of work each on a 40 x 40 map: both sit on the launch + latency floor. A 1x1 conv
// needs no halo, so a workgroup that owns TH x TW pixels can run both: the input pixels are DMA'd into an LDS image,
// step 0 writes the first conv's output rows to a staging image (and to HBM: other ops may read it), step 1 reads
// them back as its B operand; weights stream L2 -> registers through the per-wave prefetch queue that keeps running
// across the step boundary. Same MFMA, K order, epilogue and rounding points as the per-op kernels -> bit-identical
// (tests/test_gpu_parity.py block-fusion tests). fp16 and int8 (EltI8) forms.
#include "block_kernels.h"

#include <cstdlib>
#include <cstring>

namespace unina {

using namespace dev;

extern __shared__ __align__(16) unsigned char pair_smem[];

template <int C0, int C1, int C2, int NW, int KBLK>
struct PairPlan {
  static constexpr int N = 2;
  static constexpr int kb(int s) { return (s == 0 ? C0 : C1) / KBLK; }
  static constexpr int ns(int s) { return (s == 0 ? C1 : C2) / 16; }
  static constexpr int wnt(int s) { return ns(s) <= NW ? 1 : ns(s) / NW; }
  static constexpr int nch(int s) { return 16 * ns(s); }
  static constexpr int cfirst(int s) { return s == 0 ? 0 : nch(0); }
};

// element-wise maximum of two 16-byte chunks (8 fp16 / 16 int8 values; SPPF inputs are post-ReLU, i.e. >= 0)
template <typename E>
__device__ __forceinline__ typename E::frag chunk_max(const typename E::frag& a, const typename E::frag& b) {
  if constexpr (E::I8) {
    typedef signed char c16 __attribute__((ext_vector_type(16)));
    const c16 x = __builtin_bit_cast(c16, a), y = __builtin_bit_cast(c16, b);
    return __builtin_bit_cast(typename E::frag, __builtin_elementwise_max(x, y));
  } else {
    return __builtin_elementwise_max(a, b);
  }
}

// POOL = 1: the three chained MaxPool2d(5, 1, 2) of SPPF_DLA (model.py:125, 129-131) run inside the launch too: the
// kernel reads only x = sppf.cv1's output (C0 / 4 channels) on the tile + 6-pixel halo, forms the 5x5 / 9x9 / 13x13
// clipped-window maxima (= the chained pools; zero fill outside the image is neutral for post-ReLU values) separably in
// LDS, and feeds [x | p1 | p2 | p3] to step 0 -- the 3 pooled maps never exist in HBM.
template <int C0, int C1, int C2, int TH, int TW, int NW, int D, int UP2, typename E, int POOL = 0>
__global__ __launch_bounds__(NW * 64) void conv_pair_kernel(const PairParams p) {
  typedef PairPlan<C0, C1, C2, NW, E::KBLK> PL;
  typedef StepTable<PL, NW> ST;
  static_assert(ST::valid(), "wave roles");
  static_assert(C0 % E::KBLK == 0 && C1 % E::KBLK == 0, "K blocks");
  static_assert(!E::SPLIT || (!POOL && !UNINA_BLOCK_PATCH_REGS), "split fp16: the pools run as their own launch (the halo'd x image and its vertical maxima do not fit LDS twice)");
  constexpr int PT = TH * TW, NT = NW * 64, ESZ = E::ESZ;
  constexpr int NPL = E::SPLIT ? 2 : 1;
  unsigned char* smem = pair_smem;
  const int lds_lo = E::SPLIT ? p.lds_lo : 0;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int bid = (int)blockIdx.x;
  const int tyi = fast_div(bid, p.tiles_x_magic), txi = bid - tyi * p.tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TW;

  const unsigned char* wbase = p.wstream + lane * 16;
  typename E::frag q[D];
  float* cst = reinterpret_cast<float*>(smem + p.off_bias);
  floatx4 cregs[kConstVecs];
  constexpr Img X = make_img(0, C0 / E::CH);
  constexpr int CXP = C0 / 4, NCX = CXP / E::CH;            // POOL: channels / chunks of x
  constexpr int RH = TH + 12, RW = TW + 12;                 // POOL: tile + 6-pixel halo
#if UNINA_BLOCK_PATCH_REGS
  PatchRegs<RH, RW, CXP, NT, E> pr_r;
  PatchRegs<TH, TW, C0, NT, E> pr_x;
  if constexpr (POOL) patch_issue<RH, RW, CXP, NT, E>(pr_r, p.src, p.src_ld, p.H, p.W, ty0 - 6, tx0 - 6, wid, lane);
  else patch_issue<TH, TW, C0, NT, E>(pr_x, p.src, p.src_ld, p.H, p.W, ty0, tx0, wid, lane);
  consts_issue<NT>(cregs, p.bias, p.n_bias);
  static_for<0, D>([&](auto gc) { wq_fetch<ST, D, decltype(gc)::value>(q, wbase, wid); });
  if constexpr (POOL) patch_commit<RH, RW, CXP, NT, E>(pr_r, smem + p.off_r, wid, lane);
  else patch_commit<TH, TW, C0, NT, E>(pr_x, smem + p.off_x, wid, lane);
  consts_commit<NT>(cregs, cst, p.n_bias);
  lds_barrier();
#else
  if constexpr (POOL)
    load_patch<RH, RW, CXP, NT, E>(smem + p.off_r, p.src, p.src_ld, p.H, p.W, ty0 - 6, tx0 - 6, p.zeros, wid, lane);
  else
    load_patch<TH, TW, C0, NT, E>(smem + p.off_x, p.src, p.src_ld, p.H, p.W, ty0, tx0, p.zeros, wid, lane, p.src_lo, lds_lo);
  static_for<0, D>([&](auto gc) { wq_fetch<ST, D, decltype(gc)::value>(q, wbase, wid); });
  consts_issue<NT>(cregs, p.bias, p.n_bias);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the patch has landed
  consts_commit<NT>(cregs, cst, p.n_bias);
  lds_barrier();

#endif

  const Img Xi = Img{p.off_x, X.nch, X.sh, X.mask};
  if constexpr (POOL && !E::SPLIT) {
    typedef typename E::frag frag;
    constexpr Img R0 = make_img(0, NCX);
    const Img R = Img{p.off_r, R0.nch, R0.sh, R0.mask};
    frag* V = reinterpret_cast<frag*>(smem + p.off_v);       // [3][TH][RW][NCX] vertical maxima (5 / 9 / 13 rows)
    constexpr int VN = TH * RW * NCX;
    for (int t = threadIdx.x; t < VN; t += NT) {
      const int c = t % NCX, rx = (t / NCX) % RW, ty = t / (NCX * RW);
      frag v[13];
#pragma unroll
      for (int d = 0; d < 13; ++d) v[d] = *reinterpret_cast<const frag*>(smem + R.addr((ty + d) * RW + rx, c));
      frag m = v[6];
#pragma unroll
      for (int d = 1; d <= 2; ++d) m = chunk_max<E>(m, chunk_max<E>(v[6 - d], v[6 + d]));
      V[t] = m;
#pragma unroll
      for (int d = 3; d <= 4; ++d) m = chunk_max<E>(m, chunk_max<E>(v[6 - d], v[6 + d]));
      V[VN + t] = m;
#pragma unroll
      for (int d = 5; d <= 6; ++d) m = chunk_max<E>(m, chunk_max<E>(v[6 - d], v[6 + d]));
      V[2 * VN + t] = m;
    }
    lds_barrier();
    for (int t = threadIdx.x; t < PT * NCX; t += NT) {
      const int c = t % NCX, pp = t / NCX, ty = pp / TW, tx = pp - ty * TW;
      const frag* row = V + (ty * RW + tx + 6) * NCX + c;     // centre column of this pixel in the v5 plane
      frag o5 = row[0], o9 = row[VN], o13 = row[2 * VN];
#pragma unroll
      for (int d = 1; d <= 6; ++d) {
        o13 = chunk_max<E>(o13, chunk_max<E>(row[2 * VN - d * NCX], row[2 * VN + d * NCX]));
        if (d <= 4) o9 = chunk_max<E>(o9, chunk_max<E>(row[VN - d * NCX], row[VN + d * NCX]));
        if (d <= 2) o5 = chunk_max<E>(o5, chunk_max<E>(row[-d * NCX], row[d * NCX]));
      }
      *reinterpret_cast<frag*>(smem + Xi.addr(pp, c)) = *reinterpret_cast<const frag*>(smem + R.addr((ty + 6) * RW + tx + 6, c));
      *reinterpret_cast<frag*>(smem + Xi.addr(pp, NCX + c)) = o5;
      *reinterpret_cast<frag*>(smem + Xi.addr(pp, 2 * NCX + c)) = o9;
      *reinterpret_cast<frag*>(smem + Xi.addr(pp, 3 * NCX + c)) = o13;
    }
    lds_barrier();
  }
  typedef typename E::acc_t acc_t;
  typedef float vec16 __attribute__((ext_vector_type(4)));  // 16 opaque bytes
  auto pixel = [&](int sub) {
    const int pp = sub * 16 + l15;
    return pp < PT ? pp : PT - 1;
  };

  // ---- step 0: y = ReLU(W0 x + b0) -> staging image (linear rows) -> HBM ----
  constexpr int ROWB = C1 * ESZ + 16;
  unsigned char* stage = smem + p.off_stage;
  const float* c0 = cst;
  run_step<ST, D, 0, PT, E, C1>(q, wbase, smem, wid, lane,
      [&](int sub, auto kc) { return Xi.addr(pixel(sub), decltype(kc)::value * 4 + lq); },
      [&](int sub, int n, const acc_t& acc, const StepConsts<E>& k) {
        const int pp = sub * 16 + l15;
        if (pp < PT) store4<E>(stage + pp * ROWB + n * ESZ, act_relu<E>(acc, k), k, lds_lo);
      }, lds_lo, c0);
  constexpr int CPR = C1 * ESZ / 16;
  unsigned char* dst = static_cast<unsigned char*>(p.dst);
  for (int c = threadIdx.x; c < NPL * PT * CPR; c += NT) {
    const int pl = NPL == 1 ? 0 : c / (PT * CPR), cc = c - pl * (PT * CPR);   // (split: the hi tile, then the lo tile)
    const int pp = cc / CPR, ch = cc - pp * CPR;
    const int oy = ty0 + pp / TW, ox = tx0 + pp % TW;
    if (oy < p.H && ox < p.W)
      *reinterpret_cast<vec16*>(dst + (NPL == 1 ? 0 : pl * p.dst_lo) + ((size_t)(oy * p.W + ox) * p.dst_ld) * ESZ + ch * 16) =
          *reinterpret_cast<const vec16*>(stage + pl * lds_lo + pp * ROWB + ch * 16);
  }

  // ---- step 1: z = ReLU(W1 y + b1) -> output tile -> HBM (x2 nearest upsample in the store when UP2) ----
  const Img YS = Img{p.off_stage, ROWB / 16, 0, 0};
  constexpr int ROWT = C2 * ESZ + 16;
  unsigned char* tout = smem + p.off_out;
  const float* c1 = cst + E::CM * PL::cfirst(1);
  run_step<ST, D, 1, PT, E, C2>(q, wbase, smem, wid, lane,
      [&](int sub, auto kc) { return YS.addr(pixel(sub), decltype(kc)::value * 4 + lq); },
      [&](int sub, int n, const acc_t& acc, const StepConsts<E>& k) {
        const int pp = sub * 16 + l15;
        if (pp < PT) store4<E>(tout + pp * ROWT + n * ESZ, act_relu<E>(acc, k), k, lds_lo);
      }, lds_lo, c1);
  constexpr int CPT = C2 * ESZ / 16;
  unsigned char* dst2b = static_cast<unsigned char*>(p.dst2);
  const size_t px = (size_t)p.dst2_ld * ESZ, row = (size_t)(2 * p.W) * px;
  for (int c = threadIdx.x; c < NPL * PT * CPT; c += NT) {
    const int pl = NPL == 1 ? 0 : c / (PT * CPT), cc = c - pl * (PT * CPT);
    const int pp = cc / CPT, ch = cc - pp * CPT;
    const int oy = ty0 + pp / TW, ox = tx0 + pp % TW;
    if (oy >= p.H || ox >= p.W) continue;
    const vec16 v = *reinterpret_cast<const vec16*>(tout + pl * lds_lo + pp * ROWT + ch * 16);
    unsigned char* dst2 = dst2b + (NPL == 1 ? 0 : pl * p.dst2_lo);
    if constexpr (UP2) {
      unsigned char* d = dst2 + ((size_t)(2 * oy) * (2 * p.W) + 2 * ox) * px + ch * 16;
      *reinterpret_cast<vec16*>(d) = v;
      *reinterpret_cast<vec16*>(d + px) = v;
      *reinterpret_cast<vec16*>(d + row) = v;
      *reinterpret_cast<vec16*>(d + row + px) = v;
    } else {
      *reinterpret_cast<vec16*>(dst2 + (size_t)(oy * p.W + ox) * px + ch * 16) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------- host side
namespace {

struct PClass {
  int dtype, c0, c1, c2, up2, th, tw, nw;
  const char* name;
  void (*fn)(const PairParams);
  int pool;
};
const PClass kPairClasses[] = {
    // backbone.sppf.cv2 -> neck.lateral_p3 (+ x2 upsample): 100 workgroups at 40^2
    {kF16, 512, 256, 128, 1, 4, 4, 8, "conv_pair<512,256,128,4x4,8w,up2>", conv_pair_kernel<512, 256, 128, 4, 4, 8, 16, 1, EltH>, 0},
    {kF16, 512, 256, 128, 1, 4, 4, 8, "conv_pair<pool5x3,512,256,128,4x4,8w,up2>", conv_pair_kernel<512, 256, 128, 4, 4, 8, 16, 1, EltH, 1>, 1},
    {kI8, 512, 256, 128, 1, 4, 4, 8, "conv_pair<i8,512,256,128,4x4,8w,up2>", conv_pair_kernel<512, 256, 128, 4, 4, 8, 8, 1, EltI8>, 0},
    {kI8, 512, 256, 128, 1, 4, 4, 8, "conv_pair<i8,pool5x3,512,256,128,4x4,8w,up2>", conv_pair_kernel<512, 256, 128, 4, 4, 8, 8, 1, EltI8, 1>, 1},
    // STRICT engines (split fp16): the pair only; the SPPF pools stay their own launch (sppf_pool_split_kernel)
    {kS16, 512, 256, 128, 1, 4, 4, 8, "conv_pair<s16,512,256,128,4x4,8w,up2>", conv_pair_kernel<512, 256, 128, 4, 4, 8, 8, 1, EltS>, 0},
};
const PClass* find_pclass(int dtype, int c0, int c1, int c2, int up2, int pool = 0) {
  for (const PClass& c : kPairClasses)
    if (c.dtype == dtype && c.c0 == c0 && c.c1 == c1 && c.c2 == c2 && c.up2 == up2 && c.pool == pool) return &c;
  return nullptr;
}
int align_up(int v, int a) { return (v + a - 1) / a * a; }

}  // namespace

hipError_t pair_init() {
  for (const PClass& c : kPairClasses) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

bool pair_supported(int dtype, int c0, int c1, int c2, int up2, int pool) { return find_pclass(dtype, c0, c1, c2, up2, pool) != nullptr; }

bool pair_layout(PairParams* p) {
  const PClass* c = find_pclass(p->dtype, p->c0, p->c1, p->c2, p->up2, p->pool);
  if (!c) return false;
  const int esz = p->dtype == kI8 ? 1 : 2, cm = p->dtype == kI8 ? 3 : 1, pt = c->th * c->tw;
  p->tiles_x = (p->W + c->tw - 1) / c->tw;
  p->tiles_y = (p->H + c->th - 1) / c->th;
  p->tiles_x_magic = div_magic((unsigned)p->tiles_x);
  p->n_bias = cm * (p->c1 + p->c2);
  int off = 0;
  p->off_bias = off; off += align_up(p->n_bias * 4, 1024);
  p->off_x = off;
  p->off_out = off;                                           // the output tile replaces the input patch (dead after step 0)
  const int x_bytes = align_up(pt * p->c0 * esz, 1024) + 1024, out_bytes = pt * (p->c2 * esz + 16);
  off += align_up(x_bytes > out_bytes ? x_bytes : out_bytes, 1024);
  p->off_stage = off; off += align_up(pt * (p->c1 * esz + 16), 1024);
  p->off_r = p->off_v = 0;
  if (p->pool) {   // x on the tile + 6-pixel halo, and the three planes of vertical maxima
    const int cx = p->c0 / 4, rh = c->th + 12, rw = c->tw + 12;
    p->off_r = off; off += align_up(rh * rw * cx * esz, 1024) + 1024;
    p->off_v = off; off += align_up(3 * c->th * rw * cx * esz, 1024);
  }
  p->lds_lo = 0;
  if (p->dtype == kS16) {   // every image has a lo twin: the whole image area once more behind itself
    p->lds_lo = off - p->off_x;
    off += p->lds_lo;
  }
  p->smem_bytes = off;
  return off <= 160 * 1024;
}

hipError_t pair_launch(const PairParams& p, hipStream_t stream) {
  const PClass* c = find_pclass(p.dtype, p.c0, p.c1, p.c2, p.up2, p.pool);
  if (!c) return hipErrorInvalidValue;
  hipLaunchKernelGGL(c->fn, dim3(p.tiles_x * p.tiles_y, 1, 1), dim3(c->nw * 64, 1, 1), p.smem_bytes, stream, p);
  return hipGetLastError();
}

const char* pair_kernel_name(const PairParams& p) {
  const PClass* c = find_pclass(p.dtype, p.c0, p.c1, p.c2, p.up2, p.pool);
  return c ? c->name : "conv_pair<?>";
}
int pair_block_threads(const PairParams& p) {
  const PClass* c = find_pclass(p.dtype, p.c0, p.c1, p.c2, p.up2, p.pool);
  return c ? c->nw * 64 : 0;
}

}  // namespace unina
