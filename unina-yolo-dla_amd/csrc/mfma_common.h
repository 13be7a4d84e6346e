// mfma_common.h -- device-side helpers shared by the MFMA kernels (conv_igemm.hip, c3k2_fused.hip). gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace unina {
namespace dev {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// split fp16 (kS16, the STRICT precision mode): a fragment is the pair (hi, lo) of 16-byte fp16 fragments
struct half8x2 {
  half8 h, l;
};
// three MFMAs per k block into one fp32 accumulator, small terms first (the lo*lo term is below fp32 resolution)
__device__ __forceinline__ floatx4 mfma_split(const half8x2& a, const half8x2& b, floatx4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l, b.h, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.l, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.h, c, 0, 0, 0);
}

// In-block swizzle of a 1-KiB fragment block: the 16-byte slot of (row r, k-chunk c) is 4*r + (c ^ G[r>>2]), G = {0,2,3,1}
__device__ __forceinline__ int swz_g(int r16) { return (0x78 >> (2 * (r16 >> 2))) & 3; }

// LDS-DMA: every lane fetches 16 bytes from its own global address; the wave's 1 KiB lands at lds_wave_base + 16*lane.
__device__ __forceinline__ void glds16(const void* gptr, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// "At most n stages of LPT DMA instructions each may still be in flight" for a run-time (wave-uniform) n: the
// s_waitcnt immediate must be a constant, so the pipeline head / tail dispatch over the few possible values.
template <int LPT>
__device__ __forceinline__ void wait_stages(int n) {
  constexpr int kMax = 63;  // vmcnt is a 6-bit field
  switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<(LPT < kMax ? LPT : kMax)>(); break;
    case 2: wait_vmcnt<(2 * LPT < kMax ? 2 * LPT : kMax)>(); break;
    case 3: wait_vmcnt<(3 * LPT < kMax ? 3 * LPT : kMax)>(); break;
    case 4: wait_vmcnt<(4 * LPT < kMax ? 4 * LPT : kMax)>(); break;
    case 5: wait_vmcnt<(5 * LPT < kMax ? 5 * LPT : kMax)>(); break;
    default: wait_vmcnt<(6 * LPT < kMax ? 6 * LPT : kMax)>(); break;  // n >= 6: rings are at most 8 deep
  }
}

// x / d for 0 <= x, x * d < 2^32, with magic = ceil(2^32 / d) (0 encodes d == 1): one v_mul_hi_u32 instead of the
// ~40-instruction software integer division.
__device__ __forceinline__ int fast_div(int x, unsigned magic) {
  return magic ? (int)__umulhi((unsigned)x, magic) : x;
}

}  // namespace dev

inline unsigned div_magic(unsigned d) { return d > 1 ? (unsigned)(((1ull << 32) + d - 1) / d) : 0u; }

}  // namespace unina
