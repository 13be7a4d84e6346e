// Per-CU ingest probe: every wave of a workgroup requests LOADS x 1 KiB (global_load_dwordx4, all back to back) and waits;
// s_memtime from the first request to vmcnt(0). Varies the waves per workgroup, the number of workgroups (CUs busy) and whether
// the workgroups read the SAME bytes (L2 hits after the first) or DISTINCT ones (L2 misses: Infinity Cache / HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float floatx4 __attribute__((ext_vector_type(4)));
// PERM: the lane -> 16-byte slot map the block / weights-stationary kernels used up to round 3 (lane (r = lane & 15, c = lane >> 4)
// reads slot 4r + (c ^ G[r >> 2]) of its wave's 1-KiB block: the LDS-image order of the packed weights) instead of slot = lane
template <int LOADS, int NT, bool PERM = false>
__global__ __launch_bounds__(NT) void k(const floatx4* src, size_t wg_stride_vec, floatx4* out, long long* t) {
  const int lane = threadIdx.x & 63, r = lane & 15, c = lane >> 4;
  const int g = (0x1320 >> (4 * (r >> 2))) & 3;   // G = (0, 2, 3, 1)
  const int slot = PERM ? 4 * r + (c ^ g) : lane;
  const floatx4* p = src + (size_t)blockIdx.x * wg_stride_vec + (threadIdx.x & ~63) + slot;
  floatx4 v[LOADS];
  __builtin_amdgcn_s_barrier();
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < LOADS; ++i) v[i] = p[(size_t)i * NT];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  floatx4 s = v[0];
#pragma unroll
  for (int i = 1; i < LOADS; ++i) s += v[i];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int LOADS, int NT>
__global__ __launch_bounds__(NT) void kbuf(const floatx4* src, size_t wg_stride_vec, floatx4* out, long long* t) {
  typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<floatx4*>(src + (size_t)blockIdx.x * wg_stride_vec), 0, 0x7fffffff, 0x00020000);
  const unsigned voff = threadIdx.x * 16;
  uintx4 v[LOADS];
  __builtin_amdgcn_s_barrier();
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < LOADS; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, i * NT * 16, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  uintx4 s = v[0];
#pragma unroll
  for (int i = 1; i < LOADS; ++i) s += v[i];
  out[blockIdx.x * NT + threadIdx.x] = __builtin_bit_cast(floatx4, s);
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int LOADS, int NT>
void runbuf(const floatx4* src, floatx4* out, long long* t, int grid) {
  std::vector<long long> h(grid);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((kbuf<LOADS, NT>), dim3(grid), dim3(NT), 0, 0, src, (size_t)0, out, t);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), t, grid * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("  BUFFER loads %2d waves x %2d KiB, %3d WGs, same bytes: median %6.0f cyc -> %5.1f B/clk/WG\n", NT / 64, LOADS, grid, (double)h[grid / 2], (double)LOADS * NT * 16 / h[grid / 2]);
}
template <int LOADS, int NT, bool PERM = false>
void run(const floatx4* src, size_t total_vec, floatx4* out, long long* t, int grid, bool distinct, floatx4* trash, size_t trash_vec) {
  const size_t per_wg = (size_t)LOADS * NT;
  const size_t stride = distinct ? per_wg : 0;
  std::vector<long long> h(grid);
  double best = 0;
  for (int rep = 0; rep < 3; ++rep) {
    if (distinct) hipMemset(trash, rep, trash_vec * 16);      // evict L2 (and part of the Infinity Cache)
    hipLaunchKernelGGL((k<LOADS, NT, PERM>), dim3(grid), dim3(NT), 0, 0, src, stride, out, t);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), t, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    best = (double)per_wg * 16 / h[grid / 2];
  }
  printf("  %s %2d waves x %2d KiB = %4zu KiB/WG, %3d WGs, %s: median %6.0f cyc -> %5.1f B/clk/WG (max %6lld cyc)\n", PERM ? "slot = 4r+(c^g)" : "slot = lane     ", NT / 64, LOADS,
         per_wg * 16 / 1024, grid, distinct ? "distinct bytes" : "same bytes    ", (double)h[grid / 2], best, h[grid - 1]);
}
int main() {
  const size_t total = (size_t)512 << 20;   // 512 MiB
  floatx4 *src, *out, *trash; long long* t;
  hipMalloc(&src, total); hipMemset(src, 1, total);
  hipMalloc(&trash, total); 
  hipMalloc(&out, 1024 * 1024 * 16); hipMalloc(&t, 4096 * 8);
  for (int grid : {32, 256}) { runbuf<32, 256>(src, out, t, grid); runbuf<64, 256>(src, out, t, grid); runbuf<32, 512>(src, out, t, grid); }
  for (int grid : {8, 100, 220}) {
    run<32, 512, false>(src, total / 16, out, t, grid, false, trash, total / 16);
    run<32, 512, true>(src, total / 16, out, t, grid, false, trash, total / 16);
    run<64, 256, false>(src, total / 16, out, t, grid, false, trash, total / 16);
    run<64, 256, true>(src, total / 16, out, t, grid, false, trash, total / 16);
  }
  if (getenv("PROBE_ALL"))
  for (int distinct = 0; distinct < 1; ++distinct)
    for (int grid : {8, 32, 128, 256, 512}) {
      run<32, 256>(src, total / 16, out, t, grid, distinct, trash, total / 16);
      run<64, 256>(src, total / 16, out, t, grid, distinct, trash, total / 16);
      run<32, 512>(src, total / 16, out, t, grid, distinct, trash, total / 16);
      run<16, 1024>(src, total / 16, out, t, grid, distinct, trash, total / 16);
    }
  return 0;
}
