// LDS bank-conflict probe for ds_read_b128 fragment reads: lane (l15 = lane & 15, lq = lane >> 4) reads 16 bytes at
// l15 * P + f(lq, l15); 4 waves (one per SIMD) issue 512 independent reads each; cycles per read per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float floatx4 __attribute__((ext_vector_type(4)));
extern __shared__ __align__(16) unsigned char smem[];
__global__ __launch_bounds__(256) void k(int mode, int pitch, floatx4* out, long long* t) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<floatx4*>(smem)[i] = floatx4{1.f, 2.f, 3.f, 4.f};
  unsigned base;
  if (mode == 0) base = l15 * pitch + lq * 16;                       // padded pitch, chunks of a pixel contiguous
  else if (mode == 1) base = l15 * 256 + ((lq ^ l15) & 15) * 16;     // xor swizzle, 256-byte pixels (16 chunks)
  else if (mode == 2) base = l15 * pitch + lq * 64;                  // padded pitch, the 4 k-chunks 64 bytes apart
  else if (mode == 3) base = l15 * 512 + (((lq) ^ l15) & 15) * 16;   // xor swizzle, 512-byte pixels (32 chunks)
  else base = lane * 16;                                              // linear (conflict-free reference)
  __syncthreads();
  floatx4 acc = {0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 32; ++it) {
    floatx4 v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:4352\n ds_read_b128 %2, %8 offset:8704\n ds_read_b128 %3, %8 offset:13056\n"
                 "ds_read_b128 %4, %8 offset:17408\n ds_read_b128 %5, %8 offset:21760\n ds_read_b128 %6, %8 offset:26112\n ds_read_b128 %7, %8 offset:30464\n"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(base) : "memory");
    acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (lane == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
  floatx4* out; long long* t;
  hipMalloc(&out, 256 * 256 * 16); hipMalloc(&t, 256 * 4 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  struct { int mode, pitch; const char* name; } cases[] = {
      {4, 0, "linear lane*16 (reference)"}, {1, 0, "xor swizzle, 256-B pixels"}, {3, 0, "xor swizzle, 512-B pixels"},
      {0, 272, "pitch 272, lq*16"}, {0, 528, "pitch 528, lq*16"}, {0, 288, "pitch 288, lq*16"}, {0, 320, "pitch 320, lq*16"},
      {0, 264, "pitch 264 (8-B pad), lq*16"}, {2, 272, "pitch 272, lq*64"}, {2, 528, "pitch 528, lq*64"}, {0, 256, "pitch 256 (no pad), lq*16"},
      {0, 144, "pitch 144, lq*16 (C=64 image)"}, {0, 160, "pitch 160, lq*16"}};
  std::vector<long long> h(256 * 4);
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 65536, 0, c.mode, c.pitch, out, t);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), t, 256 * 4 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-34s %6.2f cycles per ds_read_b128 per wave (4 waves per CU)\n", c.name, (double)h[512] / 256.0);
  }
  return 0;
}
