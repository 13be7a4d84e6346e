// block_pipeline.h -- shared device machinery of the block-fused kernels (c3k2_fused.hip, head_fused.hip). gfx950 only.
//
// A block kernel runs a short compile-time list of GEMM "steps" per workgroup. Activations (the MFMA B operand, pixels
// as columns) live in LDS images; weights (the A operand) stream from L2 straight into registers: a wave owns wnt
// 16-channel subtiles of a step's output, so its weight blocks form ONE flat sequence over all steps (packed by the
// host in consumption order) which is prefetched D blocks ahead through a circular register queue that keeps running
// across step boundaries. Every loop is unrolled at compile time and there is no wave-uniform branch around a global
// load, so queue slots are plain registers and the compiler's own counted s_waitcnt vmcnt tracks the prefetches.
#pragma once
#include <type_traits>

#include "mfma_common.h"

namespace unina {
namespace dev {

// Element types of a block kernel. A 1-KiB weight fragment block is 16 rows x 4 chunks of 16 bytes for both:
//   EltH : fp16, chunk = 8 channels,  block = 32 k, v_mfma_f32_16x16x32_f16, fp32 accumulators
//   EltI8: int8, chunk = 16 channels, block = 64 k, v_mfma_i32_16x16x64_i8, exact int32 accumulators (INT8 engines)
//   EltS : split fp16 (STRICT precision mode): every tensor is TWO fp16 images / planes of EltH's layout, hi and lo, the lo
//          one a fixed distance behind the hi one (LDS: `lds_lo` bytes, HBM: the buffer's lo-plane distance); a weight block
//          is the 2-KiB pair [hi | lo]; three v_mfma_f32_16x16x32_f16 per k block (hi*hi + hi*lo + lo*hi), fp32 accumulators
typedef int intx4 __attribute__((ext_vector_type(4)));
struct EltH {
  static constexpr bool I8 = false, SPLIT = false;
  static constexpr int CH = 8, KBLK = 32, ESZ = 2, CM = 1, WBLK = 1024;   // CM: fp32 constants per output channel (bias); WBLK: bytes per weight block
  typedef half8 frag;
  typedef floatx4 acc_t;
  static __device__ __forceinline__ acc_t mma(const frag& a, const frag& b, const acc_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ frag ld(const unsigned char* at, long long) { return *reinterpret_cast<const frag*>(at); }   // fragment at `at`
  static __device__ __forceinline__ frag ldw(const unsigned char* at) { return *reinterpret_cast<const frag*>(at); }             // this lane's 16 bytes of a weight block
};
struct EltI8 {
  static constexpr bool I8 = true, SPLIT = false;
  static constexpr int CH = 16, KBLK = 64, ESZ = 1, CM = 3, WBLK = 1024;  // bias | multiplier | 1 / s_out
  typedef intx4 frag;
  typedef intx4 acc_t;
  static __device__ __forceinline__ acc_t mma(const frag& a, const frag& b, const acc_t& c) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ frag ld(const unsigned char* at, long long) { return *reinterpret_cast<const frag*>(at); }
  static __device__ __forceinline__ frag ldw(const unsigned char* at) { return *reinterpret_cast<const frag*>(at); }
};
struct EltS {
  static constexpr bool I8 = false, SPLIT = true;
  static constexpr int CH = 8, KBLK = 32, ESZ = 2, CM = 1, WBLK = 2048;   // (CH / ESZ per plane)
  typedef half8x2 frag;
  typedef floatx4 acc_t;
  static __device__ __forceinline__ acc_t mma(const frag& a, const frag& b, const acc_t& c) { return mfma_split(a, b, c); }
  static __device__ __forceinline__ frag ld(const unsigned char* at, long long lo) {
    return frag{*reinterpret_cast<const half8*>(at), *reinterpret_cast<const half8*>(at + lo)};
  }
  static __device__ __forceinline__ frag ldw(const unsigned char* at) {
    return frag{*reinterpret_cast<const half8*>(at), *reinterpret_cast<const half8*>(at + 1024)};
  }
};

// LDS image: pixel row r owns nch 16-byte chunks (8 fp16 / 16 int8 channels each); chunk c lives at slot c ^ ((r >> sh) & mask)
// of its row, (sh, mask) chosen from the row pitch so that the 16 pixels of a fragment read hit 16 different bank slots.
struct Img {
  int base, nch, sh, mask;
  __device__ __forceinline__ int key(int row) const { return (row >> sh) & mask; }
  __device__ __forceinline__ int addr(int row, int chunk) const { return base + ((row * nch + (chunk ^ key(row))) << 4); }
};
__host__ __device__ constexpr Img make_img(int base, int nch) {
  // pitch = nch 16-byte slots; a ds_read_b128 group covers 16 slots' worth of banks
  return (nch % 16 == 0) ? Img{base, nch, 0, 15} : ((nch % 8 == 0) ? Img{base, nch, 1, 7} : Img{base, nch, 2, 3});
}

// Step table. PLAN provides: N (steps), kb(s) = K/32 (int8: K/64) weight blocks per channel subtile, ns(s) = output channels / 16,
// wnt(s) = subtiles per wave. Wave roles: waves_n = ns / wnt waves split the channels, waves_m = NW / waves_n split
// the pixels (those load the same weight blocks: L1 serves the repeats). Every wave works in every step.
template <typename PLAN, int NW>
struct StepTable {
  static constexpr int N = PLAN::N;
  static constexpr int kb(int s) { return PLAN::kb(s); }
  static constexpr int ns(int s) { return PLAN::ns(s); }
  static constexpr int wnt(int s) { return PLAN::wnt(s); }
  static constexpr int waves_n(int s) { return ns(s) / wnt(s); }
  static constexpr int waves_m(int s) { return NW / waves_n(s); }
  static constexpr bool valid() {
    for (int s = 0; s < N; ++s)
      if (ns(s) % wnt(s) || waves_n(s) > NW || NW % waves_n(s)) return false;
    return true;
  }
  static constexpr int first(int s) {  // index of step s' first block in a wave's flat weight sequence
    int t = 0;
    for (int i = 0; i < s; ++i) t += kb(i) * wnt(i);
    return t;
  }
  static constexpr int total() { return first(N); }
  static constexpr int blk(int s) {    // offset of step s in the packed stream, in 1-KiB blocks
    int t = 0;
    for (int i = 0; i < s; ++i) t += kb(i) * ns(i);
    return t;
  }
  static constexpr int step_of(int g) {
    int s = 0;
    while (s < N - 1 && g >= first(s + 1)) ++s;
    return s;
  }
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Element G of this wave's flat weight sequence -> queue slot G % D. wbase = stream + this lane's 16 bytes of a block.
template <typename ST, int D, int G, typename FRAG>
__device__ __forceinline__ void wq_fetch(FRAG (&q)[D], const unsigned char* wbase, int wid) {
  if constexpr (G < ST::total()) {
    constexpr int s = ST::step_of(G), e = G - ST::first(s), wnt = ST::wnt(s), kb = e / wnt, j = e - kb * wnt, ns = ST::ns(s);
    const int nsub = (wid % ST::waves_n(s)) * wnt + j;
    constexpr int WBLK = (int)sizeof(FRAG) * 64;   // 1 KiB, or the 2-KiB (hi | lo) pair of the split type
    if constexpr (sizeof(FRAG) == 32) q[G % D] = EltS::ldw(wbase + (size_t)(ST::blk(s) + kb * ns + nsub) * WBLK);
    else q[G % D] = *reinterpret_cast<const FRAG*>(wbase + (size_t)(ST::blk(s) + kb * ns + nsub) * WBLK);
  }
}

__device__ __forceinline__ void lds_barrier() {  // publishes this wave's LDS writes; does NOT drain the weight prefetches
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// A step's per-channel constants for ONE channel quad of a wave (channels n..n+3), in registers: bias (| multiplier | 1 / s_out
// for int8). run_step reads them from LDS BEFORE its K loop, so the epilogue starts from registers: read inside the epilogue
// (one LDS round trip per accumulator, each waited for) they were half of a short step -- the 1x1 steps of the stage-3 block:
// K loop 880 cycles, epilogue 1 110 (profiles/r03/block_phases_epilogue.txt).
template <typename E>
struct StepConsts {
  floatx4 bias, mult, inv;
};
template <typename E, int NCH>
__device__ __forceinline__ StepConsts<E> load_consts(const float* c, int n) {
  StepConsts<E> k;
  k.bias = *reinterpret_cast<const floatx4*>(c + n);
  if constexpr (E::I8) {
    k.mult = *reinterpret_cast<const floatx4*>(c + NCH + n);
    k.inv = *reinterpret_cast<const floatx4*>(c + 2 * NCH + n);
  }
  return k;
}

// One GEMM step S over P pixels.
//   baddr(sub, kc) : LDS byte address of this lane's 16-byte B fragment of pixel subtile `sub`, k-block kc (an
//                    std::integral_constant) -- the same for all of the wave's channel subtiles
//   epi(sub, n, acc [, k]): consumes channels n..n+3 of pixel sub*16 + (lane & 15); k = the step's constants for those channels
//                    (StepConsts, preloaded from `cst` = the step's [bias | ...] arrays of NCH channels in LDS) when NCH != 0
// Ends with the barrier that publishes the epilogue's LDS writes.
// lds_lo (EltS): byte distance from an LDS image to its lo twin.
template <typename ST, int D, int S, int P, typename E = EltH, int NCH = 0, typename BAddr, typename Epi>
__device__ __forceinline__ void run_step(typename E::frag (&q)[D], const unsigned char* wbase, const unsigned char* smem,
                                         int wid, int lane, BAddr baddr, Epi epi, int lds_lo = 0, const float* cst = nullptr) {
  typedef typename E::frag frag;
  typedef typename E::acc_t acc_t;
  constexpr int KB = ST::kb(S), G0 = ST::first(S), WN_T = ST::wnt(S), WVN = ST::waves_n(S), WVM = ST::waves_m(S);
  constexpr int MS = (P + 15) / 16, WM_T = (MS + WVM - 1) / WVM;
  constexpr bool DB = WM_T <= 6;   // B fragments double-buffered in registers when they fit comfortably
  const int wm = wid / WVN, wn = wid % WVN, lq = lane >> 4;
  acc_t acc[WN_T][WM_T];
  frag b[2][WM_T];
  auto& baddr_t = baddr;
  StepConsts<E> kc[WN_T];
  if constexpr (NCH != 0) {
#pragma unroll
    for (int j = 0; j < WN_T; ++j) kc[j] = load_consts<E, NCH>(cst, (wn * WN_T + j) * 16 + lq * 4);
  }
#pragma unroll
  for (int i = 0; i < WM_T; ++i) {
#pragma unroll
    for (int j = 0; j < WN_T; ++j) acc[j][i] = acc_t{0, 0, 0, 0};
    if constexpr (DB) b[0][i] = E::ld(smem + baddr_t(wm * WM_T + i, std::integral_constant<int, 0>{}), lds_lo);
  }
  static_for<0, KB>([&](auto kc) {
    constexpr int kb = decltype(kc)::value;
    frag a[WN_T];
#pragma unroll
    for (int j = 0; j < WN_T; ++j) a[j] = q[(G0 + kb * WN_T + j) % D];
    static_for<0, WN_T>([&](auto jc) { wq_fetch<ST, D, G0 + kb * WN_T + decltype(jc)::value + D>(q, wbase, wid); });
    if constexpr (DB) {
      if constexpr (kb + 1 < KB) {
#pragma unroll
        for (int i = 0; i < WM_T; ++i)
          b[(kb + 1) & 1][i] = E::ld(smem + baddr_t(wm * WM_T + i, std::integral_constant<int, kb + 1>{}), lds_lo);
      }
    } else {
#pragma unroll
      for (int i = 0; i < WM_T; ++i) b[kb & 1][i] = E::ld(smem + baddr_t(wm * WM_T + i, kc), lds_lo);
    }
#pragma unroll
    for (int j = 0; j < WN_T; ++j)
#pragma unroll
      for (int i = 0; i < WM_T; ++i) acc[j][i] = E::mma(a[j], b[kb & 1][i], acc[j][i]);
    // pin the iteration: left free, the scheduler sinks a step's weight requests behind its last MFMAs (cv1|cv2 of the stage-3
    // block: all 16 in one burst at the end, each costing the wave ~70 cycles of issue, and the next step then waits for data
    // that was only just requested -- the prefetch distance D was lost exactly where the steps are shortest)
    __builtin_amdgcn_sched_barrier(0);
  });
#pragma unroll
  for (int j = 0; j < WN_T; ++j)
#pragma unroll
    for (int i = 0; i < WM_T; ++i)
      if (wm * WM_T + i < MS) {
        if constexpr (NCH != 0) epi(wm * WM_T + i, (wn * WN_T + j) * 16 + lq * 4, acc[j][i], kc[j]);
        else epi(wm * WM_T + i, (wn * WN_T + j) * 16 + lq * 4, acc[j][i]);
      }
  lds_barrier();
}

__device__ __forceinline__ floatx4 bias_relu(const floatx4& acc, const float* bias_lds, int n) {
  floatx4 v = acc + *reinterpret_cast<const floatx4*>(bias_lds + n);
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
  return v;
}
__device__ __forceinline__ void store_h4(unsigned char* smem, const Img& im, int row, int n, const floatx4& v) {
  half4 hv;
#pragma unroll
  for (int r = 0; r < 4; ++r) hv[r] = (_Float16)v[r];
  *reinterpret_cast<half4*>(smem + im.addr(row, n >> 3) + (n & 4) * 2) = hv;
}
__device__ __forceinline__ floatx4 load_h4(const unsigned char* smem, const Img& im, int row, int n) {
  const half4 hv = *reinterpret_cast<const half4*>(smem + im.addr(row, n >> 3) + (n & 4) * 2);
  return floatx4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
}

// ---- element-generic epilogue pieces (the int8 forms repeat conv_igemm.hip's conv_epilogue operation for operation, so
// a block kernel and the per-op table give the same codes) ----
// y = ReLU(acc + bias)  |  ReLU(fma(acc, mult, bias)); `c` = the step's constants in LDS ([bias] | [bias | mult | inv]
// of NCH channels each)
template <typename E, int NCH>
__device__ __forceinline__ floatx4 act_relu(const typename E::acc_t& acc, const float* c, int n) {
  const floatx4 bias = *reinterpret_cast<const floatx4*>(c + n);
  floatx4 v;
  if constexpr (E::I8) {
    const floatx4 mult = *reinterpret_cast<const floatx4*>(c + NCH + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf((float)acc[r], mult[r], bias[r]);
  } else {
    v = acc + bias;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
  return v;
}
template <typename E>
__device__ __forceinline__ floatx4 act_relu(const typename E::acc_t& acc, const StepConsts<E>& k) {   // (constants in registers)
  floatx4 v;
  if constexpr (E::I8) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf((float)acc[r], k.mult[r], k.bias[r]);
  } else {
    v = acc + k.bias;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
  return v;
}
// v + residual (channels n..n+3 of image row `row`); int8: fma(code, s_res, v)
template <typename E>
__device__ __forceinline__ floatx4 add_res(floatx4 v, const unsigned char* smem, const Img& im, int row, int n, float res_scale, int lds_lo = 0) {
  if constexpr (E::SPLIT) {
    return v + load_h4(smem, im, row, n) + load_h4(smem + lds_lo, im, row, n);
  } else if constexpr (E::I8) {
    const int rv = *reinterpret_cast<const int*>(smem + im.addr(row, n >> 4) + (n & 15));
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf((float)(signed char)(rv >> (8 * r)), res_scale, v[r]);
    return v;
  } else {
    return v + load_h4(smem, im, row, n);
  }
}
// 4 consecutive channels -> memory at `at` (fp16: 8 bytes; int8: q = clamp(rne(v * inv), -127, 127), 4 bytes)
// (split fp16: 8 bytes of the hi value at `at`, 8 of the lo value `lo` bytes behind it)
template <typename E, int NCH>
__device__ __forceinline__ void store4(unsigned char* at, const floatx4& v, const float* c, int n, long long lo = 0) {
  if constexpr (E::SPLIT) {
    half4 hv, lv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      hv[r] = (_Float16)v[r];
      lv[r] = (_Float16)(v[r] - (float)hv[r]);
    }
    *reinterpret_cast<half4*>(at) = hv;
    *reinterpret_cast<half4*>(at + lo) = lv;
  } else if constexpr (E::I8) {
    const floatx4 inv = *reinterpret_cast<const floatx4*>(c + 2 * NCH + n);
    unsigned int q = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = __builtin_rintf(v[r] * inv[r]);  // round half to even
      t = t > 127.f ? 127.f : (t < -127.f ? -127.f : t);
      q |= ((unsigned int)(int)t & 0xFFu) << (8 * r);
    }
    *reinterpret_cast<unsigned int*>(at) = q;
  } else {
    half4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) hv[r] = (_Float16)v[r];
    *reinterpret_cast<half4*>(at) = hv;
  }
}
template <typename E, typename KE>
__device__ __forceinline__ void store4(unsigned char* at, const floatx4& v, const StepConsts<KE>& k, long long lo = 0) {   // (constants in registers)
  if constexpr (E::SPLIT) {
    half4 hv, lv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      hv[r] = (_Float16)v[r];
      lv[r] = (_Float16)(v[r] - (float)hv[r]);
    }
    *reinterpret_cast<half4*>(at) = hv;
    *reinterpret_cast<half4*>(at + lo) = lv;
  } else if constexpr (E::I8) {
    unsigned int q = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = __builtin_rintf(v[r] * k.inv[r]);  // round half to even
      t = t > 127.f ? 127.f : (t < -127.f ? -127.f : t);
      q |= ((unsigned int)(int)t & 0xFFu) << (8 * r);
    }
    *reinterpret_cast<unsigned int*>(at) = q;
  } else {
    half4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) hv[r] = (_Float16)v[r];
    *reinterpret_cast<half4*>(at) = hv;
  }
}
template <typename E>
__device__ __forceinline__ int img_at(const Img& im, int row, int n) {   // byte address of channel n (multiple of 4) of a row
  return im.addr(row, n / E::CH) + (n % E::CH) * E::ESZ;
}

// Per-channel constants (biases ...) global -> LDS in two halves, so that their loads are IN FLIGHT together with the
// patch DMA and the first weight blocks instead of in front of them: consts_issue() right after those are issued
// (branch-free: clamped index, every lane loads), consts_commit() after the s_waitcnt vmcnt(0) that the patch needs anyway.
// n is a multiple of 4, at most 4 * kConstVecs * NT floats; the array is 16-byte aligned.
constexpr int kConstVecs = 4;
template <int NT>
__device__ __forceinline__ void consts_issue(floatx4 (&v)[kConstVecs], const float* src, int n) {
  const int nv = n >> 2;
#pragma unroll
  for (int b = 0; b < kConstVecs; ++b) {
    int i = (int)threadIdx.x + b * NT;
    i = i < nv ? i : nv - 1;
    v[b] = reinterpret_cast<const floatx4*>(src)[i];
  }
}
template <int NT>
__device__ __forceinline__ void consts_commit(const floatx4 (&v)[kConstVecs], float* lds, int n) {
  const int nv = n >> 2;
#pragma unroll
  for (int b = 0; b < kConstVecs; ++b) {
    const int i = (int)threadIdx.x + b * NT;
    if (i < nv) reinterpret_cast<floatx4*>(lds)[i] = v[b];
  }
}

// Input patch -> LDS image by LDS-DMA: 16-byte slot s = (region pixel r of an RH x RW region whose origin is image
// pixel (y0, x0), chunk cs); out-of-image pixels read the zero page. NT threads; every wave must afterwards wait
// vmcnt(0) (its own DMAs) and pass a barrier before anyone reads the image.
// EltS: the lo plane (src_lo bytes behind the hi plane in HBM) lands lds_lo bytes behind the hi image.
template <int RH, int RW, int CIN, int NT, typename E = EltH>
__device__ __forceinline__ void load_patch(unsigned char* lds_img, const void* src_, int src_ld, int H, int W, int y0,
                                           int x0, const void* zeros, int wid, int lane, long long src_lo = 0, int lds_lo = 0) {
  constexpr Img X = make_img(0, CIN / E::CH);
  constexpr int nchx = CIN / E::CH, nslots = RH * RW * nchx;
  const unsigned char* src = static_cast<const unsigned char*>(src_);
  for (int s0 = wid * 64; s0 < nslots; s0 += NT) {
    const int s = s0 + lane;
    const unsigned char* g = static_cast<const unsigned char*>(zeros);
    if (s < nslots) {
      const int r = s / nchx, cs = s - r * nchx;
      const int ry = r / RW, rx = r - ry * RW;
      const int iy = y0 + ry, ix = x0 + rx;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        g = src + ((size_t)(iy * W + ix) * src_ld) * E::ESZ + ((cs ^ X.key(r)) << 4);
    }
    glds16(g, lds_img + s0 * 16);
    if constexpr (E::SPLIT) glds16(g == static_cast<const unsigned char*>(zeros) ? g : g + src_lo, lds_img + lds_lo + s0 * 16);
  }
}

#ifndef UNINA_BLOCK_PATCH_REGS
#define UNINA_BLOCK_PATCH_REGS 0   // block kernels / conv_pair: input patch through registers (1) or LDS-DMA (0, the default: same-box
                                   // A/B at 2 frames in flight: 8 270-8 470 frames/s with LDS-DMA against 8 130-8 200 through registers;
                                   // the register-queue 3x3 kernels (conv_igemm.hip) go the other way: 8 240-8 360 against 8 130-8 170)
#endif
// The same patch through REGISTERS: plain 16-byte global loads (all requested back to back: ~16 cycles of issue each, against
// the 60-185 cycles an LDS-DMA piece costs the issuing wave), then ds_write_b128 once they have landed -- the compiler's
// counted s_waitcnt vmcnt leaves whatever was requested after them (the weight queue) in flight. Same LDS image.
template <int RH, int RW, int CIN, int NT, typename E = EltH>
struct PatchRegs {
  static constexpr int nchx = CIN / E::CH, nslots = RH * RW * nchx, ITER = (nslots + NT - 1) / NT;
  floatx4 v[ITER];
};
// The form in use: a branch around each load (out-of-image pixels are not fetched). Same-box A/B (round 2, regq head
// pairs): LDS-DMA 19.8 us, this 18.9 us, the branch-free clamped form above 20.5 us per pair.
template <int RH, int RW, int CIN, int NT, typename E = EltH>
__device__ __forceinline__ void patch_issue(PatchRegs<RH, RW, CIN, NT, E>& pr, const void* src_, int src_ld, int H, int W, int y0,
                                              int x0, int wid, int lane) {
  typedef PatchRegs<RH, RW, CIN, NT, E> PR;
  constexpr Img X = make_img(0, CIN / E::CH);
  const unsigned char* src = static_cast<const unsigned char*>(src_);
#pragma unroll
  for (int it = 0; it < PR::ITER; ++it) {
    const int s = it * NT + wid * 64 + lane;
    pr.v[it] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (s < PR::nslots) {
      const int r = s / PR::nchx, cs = s - r * PR::nchx;
      const int ry = r / RW, rx = r - ry * RW;
      const int iy = y0 + ry, ix = x0 + rx;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        pr.v[it] = *reinterpret_cast<const floatx4*>(src + ((size_t)(iy * W + ix) * src_ld) * E::ESZ + ((cs ^ X.key(r)) << 4));
    }
  }
}
template <int RH, int RW, int CIN, int NT, typename E = EltH>
__device__ __forceinline__ void patch_commit(const PatchRegs<RH, RW, CIN, NT, E>& pr, unsigned char* lds_img, int wid, int lane) {
  typedef PatchRegs<RH, RW, CIN, NT, E> PR;
#pragma unroll
  for (int it = 0; it < PR::ITER; ++it) {
    const int s = it * NT + wid * 64 + lane;
    if (s < PR::nslots) *reinterpret_cast<floatx4*>(lds_img + s * 16) = pr.v[it];
  }
}

}  // namespace dev
}  // namespace unina
