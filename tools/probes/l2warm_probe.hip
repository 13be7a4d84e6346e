// Does an L2 warm-up by the PREVIOUS kernel survive the kernel boundary? Consumer: 100 workgroups x 8 waves stream the same
// W bytes through a 16-deep register queue per wave (the block kernels' weight stream); timed per workgroup with s_memtime.
//   cold      : after a 512 MiB memset (L2 and most of the Infinity Cache evicted)
//   mall      : after a 64 MiB memset (L2 evicted, W most likely still in the Infinity Cache)
//   prefetched: cold, then a prefetch kernel (every XCD reads all of W once), then the consumer as the NEXT kernel in the stream
//   warm      : the consumer again right after itself
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int D>
__global__ __launch_bounds__(512) void consumer(const floatx4* w, int nblk, floatx4* out, long long* t) {
  const floatx4* p = w + threadIdx.x;        // every wave reads its own 1-KiB blocks: block b of wave v = (b * 8 + v)
  floatx4 q[D], s = {0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < D; ++i) q[i] = p[(size_t)i * 512];
  for (int b = 0; b < nblk; b += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      s += q[i];
      const int nb = b + i + D;
      q[i] = p[(size_t)(nb < nblk ? nb : nblk - 1) * 512];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = s + q[0];
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void prefetch(const floatx4* w, size_t nvec, floatx4* sink) {
  // workgroup i runs on XCD i % 8: the workgroups of one XCD split W among themselves
  const int xw = blockIdx.x >> 3, nxw = gridDim.x >> 3;
  floatx4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)xw * 256 + threadIdx.x; i < nvec; i += (size_t)nxw * 256) s += w[i];
  if (s.x == 12345.f) sink[threadIdx.x] = s;
}
int main() {
  const size_t wbytes = 917 * 1024, big = (size_t)512 << 20;
  floatx4 *w, *out, *trash, *sink; long long* t;
  hipMalloc(&w, wbytes + 8192); hipMemset(w, 0, wbytes + 8192);
  hipMalloc(&trash, big); hipMalloc(&out, 100 * 512 * 16); hipMalloc(&sink, 4096); hipMalloc(&t, 100 * 8);
  const int nblk = (int)(wbytes / (512 * 16));
  std::vector<long long> h(100);
  auto report = [&](const char* name) {
    hipDeviceSynchronize();
    hipMemcpy(h.data(), t, 100 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-12s median %7lld cyc (%5.1f B/clk/WG), max %7lld\n", name, h[50], (double)wbytes / h[50], h[99]);
  };
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(trash, rep, big); hipDeviceSynchronize();
    hipLaunchKernelGGL(consumer<16>, dim3(100), dim3(512), 0, 0, w, nblk, out, t); report("cold");
    hipLaunchKernelGGL(consumer<16>, dim3(100), dim3(512), 0, 0, w, nblk, out, t); report("warm");
    hipMemset(trash, rep, (size_t)64 << 20); hipDeviceSynchronize();
    hipLaunchKernelGGL(consumer<16>, dim3(100), dim3(512), 0, 0, w, nblk, out, t); report("mall");
    hipMemset(trash, rep, big); hipDeviceSynchronize();
    hipLaunchKernelGGL(prefetch, dim3(256), dim3(256), 0, 0, w, wbytes / 16, sink);
    hipLaunchKernelGGL(consumer<16>, dim3(100), dim3(512), 0, 0, w, nblk, out, t); report("prefetched");
    hipMemset(trash, rep, (size_t)64 << 20); hipDeviceSynchronize();
    hipLaunchKernelGGL(prefetch, dim3(256), dim3(256), 0, 0, w, wbytes / 16, sink);
    hipLaunchKernelGGL(consumer<16>, dim3(100), dim3(512), 0, 0, w, nblk, out, t); report("mall+pref");
    hipMemset(trash, rep, (size_t)64 << 20); hipDeviceSynchronize();
    hipLaunchKernelGGL(consumer<32>, dim3(100), dim3(512), 0, 0, w, nblk, out, t); report("mall, D=32");
  }
  return 0;
}
