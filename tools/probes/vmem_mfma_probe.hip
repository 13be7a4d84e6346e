// Does a wave's vector-memory request overlap with its own MFMAs? One wave per SIMD (256 threads per workgroup, one workgroup per CU):
// every iteration issues ONE 1-KiB global load (L2-resident stream, results consumed at the very end) followed by N independent-
// accumulator MFMAs (v_mfma_f32_16x16x32_f16); cycles per iteration against N. If the request only costs an issue slot, the
// time is max(request rate, 16 N); if the wave sits in the request for its ~66 cycles, it is ~66 + 16 N.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int N, int LOADS_PER_IT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const floatx4* src, floatx4* out, long long* t, int iters) {
  const floatx4* p = src + threadIdx.x;
  constexpr int NACC = N > 16 ? 16 : (N < 1 ? 1 : N);   // independent accumulators = dependency distance in MFMAs
  floatx4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0, 0, 0, 0};
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x & 3); b[i] = (_Float16)1; }
  floatx4 v[8], sum = {0, 0, 0, 0};
  for (int i = 0; i < 8; ++i) v[i] = floatx4{0, 0, 0, 0};
  __builtin_amdgcn_s_barrier();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int l = 0; l < LOADS_PER_IT; ++l) {
        sum += v[(u * LOADS_PER_IT + l) & 7];                                        // consumes the load issued 8 requests ago
        v[(u * LOADS_PER_IT + l) & 7] = p[(size_t)((it + u) * LOADS_PER_IT + l) * (WAVES * 64)];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int m = 0; m < N; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  floatx4 s = sum;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int N, int L, int WAVES>
void run(const floatx4* src, floatx4* out, long long* t) {
  const int grid = 200, iters = 256;
  std::vector<long long> h(grid);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<N, L, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, src, out, t, iters);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), t, grid * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("  %d waves/CU, %d load(s) + %2d MFMAs per iteration: %6.1f cycles per iteration  (MFMAs alone would be %4d, loads alone ~%4d)\n", WAVES, L, N,
         (double)h[grid / 2] / iters, 16 * N * (WAVES / 4), 66 * L);
}
template <int N, int WAVES>
void tflops(const floatx4* src, floatx4* out, long long* t) {
  const int grid = 256, iters = 8192;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<N, 0, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, src, out, t, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<N, 0, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, src, out, t, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long ticks; hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost);
  const double fl = (double)grid * WAVES * iters * N * 16384.0;
  printf("  MFMA only, %2d waves/CU x %d MFMAs x %d iterations on %d CUs: %.3f ms wall = %.0f TFLOP/s; workgroup 0: %lld ticks = %.2f ticks per MFMA and SIMD, %.2f GHz\n", WAVES, N, iters, grid, ms,
         fl / ms * 1e-9, ticks, (double)ticks / ((double)iters * N * (WAVES / 4)), ticks / (ms * 1e6));
}
int main() {
  floatx4 *src, *out; long long* t;
  hipMalloc(&src, (size_t)64 << 20); hipMemset(src, 0, (size_t)64 << 20);
  hipMalloc(&out, 1024 * 1024 * 16); hipMalloc(&t, 4096 * 8);
  run<0, 1, 4>(src, out, t); run<1, 1, 4>(src, out, t); run<2, 1, 4>(src, out, t); run<3, 1, 4>(src, out, t); run<4, 1, 4>(src, out, t);
  run<6, 1, 4>(src, out, t); run<8, 1, 4>(src, out, t); run<12, 1, 4>(src, out, t);
  run<6, 0, 4>(src, out, t); run<6, 2, 4>(src, out, t);
  run<0, 1, 8>(src, out, t); run<3, 1, 8>(src, out, t); run<6, 1, 8>(src, out, t); run<6, 0, 8>(src, out, t); run<12, 1, 8>(src, out, t);
  tflops<1, 4>(src, out, t); tflops<2, 4>(src, out, t); tflops<3, 4>(src, out, t); tflops<4, 4>(src, out, t); tflops<6, 4>(src, out, t); tflops<8, 4>(src, out, t); tflops<12, 4>(src, out, t); tflops<16, 4>(src, out, t);
  tflops<1, 8>(src, out, t); tflops<3, 8>(src, out, t); tflops<6, 8>(src, out, t); tflops<12, 8>(src, out, t); tflops<6, 16>(src, out, t);
  return 0;
}
