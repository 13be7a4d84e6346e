// Host <-> GPU round trip floor: launch -> a one-wave kernel writes a word in pinned host memory -> the host sees it.
// (a) hipLaunchKernel on a stream, (b) hipGraphLaunch of a 1-kernel graph, (c) of a 15-kernel chain (first kernel writes
// flag A, last writes flag B: time to first kernel, and the chain's span).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void mark(volatile unsigned* flag, unsigned v) { if (threadIdx.x == 0) { *flag = v; __threadfence_system(); } }
__global__ void spin_work(int n, float* sink) { float a = threadIdx.x; for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f; if (a == 12345.f) *sink = a; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  volatile unsigned* flags; hipHostMalloc((void**)&flags, 64, hipHostMallocMapped);
  float* sink; hipMalloc(&sink, 4);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  auto stat = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  unsigned seq = 0;
  {  // (a)
    std::vector<double> t;
    for (int i = 0; i < 300; ++i) {
      ++seq; double a = now();
      hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, s, flags, seq);
      while (flags[0] != seq) {}
      t.push_back(now() - a);
    }
    printf("hipLaunchKernel -> host sees the word: p50 %.1f us\n", stat(t));
  }
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, s, flags, 7u);
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  {  // (b)
    std::vector<double> t;
    for (int i = 0; i < 300; ++i) {
      flags[0] = 0; double a = now();
      hipGraphLaunch(ge, s);
      while (flags[0] != 7u) {}
      t.push_back(now() - a);
    }
    printf("hipGraphLaunch (1 kernel) -> host sees the word: p50 %.1f us\n", stat(t));
  }
  hipGraph_t g2; hipGraphExec_t ge2;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, s, flags, 1u);
  for (int k = 0; k < 13; ++k) hipLaunchKernelGGL(spin_work, dim3(256), dim3(256), 0, s, 2000, sink);
  hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, s, flags + 16, 2u);
  hipStreamEndCapture(s, &g2); hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
  {  // (c)
    std::vector<double> t1, t2;
    for (int i = 0; i < 300; ++i) {
      flags[0] = 0; flags[16] = 0; double a = now();
      hipGraphLaunch(ge2, s);
      while (flags[0] != 1u) {}
      double b = now();
      while (flags[16] != 2u) {}
      t1.push_back(b - a); t2.push_back(now() - a);
    }
    printf("hipGraphLaunch (15-kernel chain): first kernel's word p50 %.1f us, last kernel's word p50 %.1f us\n", stat(t1), stat(t2));
  }
  // (d) the per-boundary floor of a dependent chain inside one graph: N launches of a kernel that does (almost) nothing, grids of
  //     1 / 220 / 800 workgroups x 256 threads; span from the first kernel's word to the last kernel's word, divided by N
  for (int wgs : {1, 220, 800}) {
    for (int work : {0, 2000}) {
      const int N = 32;
      hipGraph_t g3; hipGraphExec_t ge3;
      hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
      hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, s, flags, 1u);
      for (int k = 0; k < N; ++k) hipLaunchKernelGGL(spin_work, dim3(wgs), dim3(256), 0, s, work, sink);
      hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, s, flags + 16, 2u);
      hipStreamEndCapture(s, &g3); hipGraphInstantiate(&ge3, g3, nullptr, nullptr, 0);
      std::vector<double> t;
      for (int i = 0; i < 200; ++i) {
        flags[0] = 0; flags[16] = 0;
        hipGraphLaunch(ge3, s);
        while (flags[0] != 1u) {}
        double b = now();
        while (flags[16] != 2u) {}
        t.push_back((now() - b) / (N + 1));
      }
      printf("chain of %d launches, %3d workgroups x 256 threads, %4d dependent FMAs per thread: %.2f us per launch\n", N, wgs, work, stat(t));
      hipGraphExecDestroy(ge3); hipGraphDestroy(g3);
    }
  }
  return 0;
}
