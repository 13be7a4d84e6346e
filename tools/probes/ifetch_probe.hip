// Instruction-fetch probe: the same MFMA stream as straight-line code and as a loop, one wave per SIMD, timed with
// s_memtime per wave. hipcc --offload-arch=gfx950 -O3 -o ifetch_probe ifetch_probe.hip && ./ifetch_probe
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
// N MFMAs total; BODY MFMAs per loop iteration (BODY == N: straight line); FILL extra v_add per MFMA
template <int N, int BODY, int FILL>
__global__ __launch_bounds__(256) void k(const half8* in, floatx4* out, long long* t) {
  half8 a[8], b = in[threadIdx.x + 2048];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = in[threadIdx.x + 256 * i];
  floatx4 acc[4] = {};
  float f = (float)threadIdx.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const long long t0 = __builtin_amdgcn_s_memtime();
#define M1(i) acc[(i) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i) & 7], b, acc[(i) & 3], 0, 0, 0); \
  if constexpr (FILL >= 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(1.0f));                       \
  if constexpr (FILL >= 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(1.0f));
#define U8 M1(0) M1(1) M1(2) M1(3) M1(4) M1(5) M1(6) M1(7)
#define U24 U8 U8 U8
#define U72 U24 U24 U24
#define U288 U72 U72 U72 U72
#define U1152 U288 U288 U288 U288
  if constexpr (BODY == 24) { for (int it = 0; it < N / 24; ++it) { U24 } }
  else if constexpr (BODY == 72) { for (int it = 0; it < N / 72; ++it) { U72 } }
  else if constexpr (BODY == 1152) { U1152 }
  else { U1152 U1152 }
  asm volatile("s_nop 0" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + f;
  if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int N, int BODY, int FILL>
void run(const char* name, half8* in, floatx4* out, long long* t) {
  const int G = 256;
  std::vector<long long> h(G * 4);
  printf("%-34s", name);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<N, BODY, FILL>), dim3(G), dim3(256), 0, 0, in, out, t);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), t, G * 4 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("  run %d: median %6.1f max %6.1f cyc/MFMA", rep, (double)h[G * 2] / N, (double)h[G * 4 - 1] / N);
  }
  printf("\n");
}
int main() {
  half8* in; floatx4* out; long long* t;
  hipMalloc(&in, 4096 * 16); hipMemset(in, 0, 4096 * 16);
  hipMalloc(&out, 256 * 256 * 16); hipMalloc(&t, 256 * 4 * 8);
  run<1152, 1152, 0>("straight 1152 mfma", in, out, t);
  run<1152, 24, 0>("loop 48 x 24 mfma", in, out, t);
  run<1152, 1152, 1>("straight 1152 mfma + 1 valu", in, out, t);
  run<1152, 24, 1>("loop 48 x 24 mfma + 1 valu", in, out, t);
  run<1152, 1152, 2>("straight 1152 mfma + 2 valu", in, out, t);
  run<1152, 24, 2>("loop 48 x 24 mfma + 2 valu", in, out, t);
  run<2304, 2304, 1>("straight 2304 mfma + 1 valu", in, out, t);
  run<2304, 72, 1>("loop 32 x 72 mfma + 1 valu", in, out, t);
  return 0;
}
