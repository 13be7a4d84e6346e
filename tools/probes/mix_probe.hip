// Instruction-mix probe for the weights-stationary 3x3 K loop: per step one ds_read_b128 (requested PF steps ahead), AV address
// VALU ops and three MFMAs on rolling accumulators; one wave per SIMD (256 threads), 192 steps, s_memtime per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <type_traits>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
extern __shared__ __align__(16) unsigned char smem[];
// LDS: 1 = read per step; VALU: address ops per step; NACC: rolling accumulators (MFMAs per step); NT threads
template <int LDS, int VALU, int NACC, int NT>
__global__ __launch_bounds__(NT) void k(const half8* in, floatx4* out, long long* t) {
  constexpr int STEPS = 192, PF = 3, NW = 36;
  half8 w[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) w[i] = in[threadIdx.x + NT * i];
  for (int i = threadIdx.x; i < 4096; i += NT) reinterpret_cast<floatx4*>(smem)[i] = floatx4{1.f, 2.f, 3.f, 4.f};
  floatx4 acc[4] = {};
  const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
  half8 b[PF + 1];
  auto addr = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    int row = (s / 4) * 18 + l15, ch = (s % 4) * 4 + lq;
    if constexpr (VALU == 0) return (s * 1024 + lane * 16) & 65535;
    else return ((row * 16 + (ch ^ (row & 15))) << 4) & 65535;
  };
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  static_for<0, PF>([&](auto sc) { b[decltype(sc)::value] = *reinterpret_cast<const half8*>(smem + addr(sc)); });
  if constexpr (!LDS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  const long long t0 = __builtin_amdgcn_s_memtime();
  static_for<0, STEPS>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if constexpr (LDS && s + PF < STEPS) b[(s + PF) % (PF + 1)] = *reinterpret_cast<const half8*>(smem + addr(std::integral_constant<int, s + PF>{}));
    static_for<0, NACC>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[(s * 3 + j) % NW], b[LDS ? s % (PF + 1) : j % PF], acc[j], 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
  });
  asm volatile("s_nop 0" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * NT + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if ((threadIdx.x & 63) == 0) t[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int LDS, int VALU, int NACC, int NT>
void run(const char* name, half8* in, floatx4* out, long long* t) {
  const int G = 256, NWV = NT / 64;
  std::vector<long long> h(G * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<LDS, VALU, NACC, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 1024);
  printf("%-44s", name);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(t, 0, G * 8 * 8);
    hipLaunchKernelGGL((k<LDS, VALU, NACC, NT>), dim3(G), dim3(NT), 65536 + 1024, 0, in, out, t);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), t, G * 8 * 8, hipMemcpyDeviceToHost);
    std::vector<long long> v;
    for (int g = 0; g < G; ++g) for (int wv = 0; wv < NWV; ++wv) v.push_back(h[g * 8 + wv]);
    std::sort(v.begin(), v.end());
    printf("  run %d: median %5.1f max %5.1f cyc/MFMA", rep, (double)v[v.size() / 2] / (192.0 * NACC), (double)v.back() / (192.0 * NACC));
  }
  printf("\n");
}
int main() {
  half8* in; floatx4* out; long long* t;
  hipMalloc(&in, 512 * 40 * 16); hipMemset(in, 0, 512 * 40 * 16);
  hipMalloc(&out, 256 * 512 * 16); hipMalloc(&t, 256 * 8 * 8);
  run<1, 1, 3, 256>("lds + addr valu + 3 mfma, 4 waves", in, out, t);
  run<1, 0, 3, 256>("lds (const addr) + 3 mfma, 4 waves", in, out, t);
  run<0, 0, 3, 256>("3 mfma only, 4 waves", in, out, t);
  run<0, 0, 4, 256>("4 mfma only (4 accumulators), 4 waves", in, out, t);
  run<1, 1, 4, 256>("lds + addr valu + 4 mfma, 4 waves", in, out, t);
  run<1, 1, 3, 512>("lds + addr valu + 3 mfma, 8 waves", in, out, t);
  run<0, 0, 3, 512>("3 mfma only, 8 waves", in, out, t);
  run<0, 0, 2, 256>("2 mfma only, 4 waves", in, out, t);
  run<0, 0, 1, 256>("1 mfma only, 4 waves", in, out, t);
  return 0;
}
