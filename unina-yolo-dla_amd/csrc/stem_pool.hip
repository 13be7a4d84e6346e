// stem_pool.hip -- the non-GEMM ops of the forward graph (gfx950).
//   stem_conv_kernel   : backbone.stem  (model.py:175: ConvBlock(3, c1, k=3, s=2)) reading the fp32 NCHW image
//   sppf_pool_kernel   : SPPF_DLA's three chained MaxPool2d(5,1,2)  (model.py:125,129-131)
//   upsample2x_kernel  : Upsample(scale_factor=2, nearest)  (model.py:145-147), standalone form
#include "kernels.h"

namespace unina {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------- stem
// Cin = 3 makes K = 27: far too thin for a matrix-core tile, and the op is HBM-bound anyway
// (4.9 MB fp32 in, 6.6 MB fp16 out at 640^2 against 0.18 GFLOP). Plain fp32 FMAs, one thread per output pixel
// and ALL output channels: a wave's 27 input loads each cover 64 neighbouring pixels (stride-2 fp32: every fetched
// line is used), the 27x32 folded weights are LDS broadcast reads (ds_read_b128, same address in every lane), and
// each lane stores its pixel's 64 contiguous NHWC bytes.
template <typename T, int CO>
__global__ __launch_bounds__(256) void stem_conv_kernel(const StemParams p) {
  __shared__ __align__(16) float sw[27 * CO];
  __shared__ __align__(16) float sb[CO];
  for (int i = threadIdx.x; i < CO * 27; i += blockDim.x) sw[(i % 27) * CO + (i / 27)] = p.w[i];  // -> [k][co]
  for (int i = threadIdx.x; i < CO; i += blockDim.x) sb[i] = p.bias[i];
  __syncthreads();
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= p.Ho * p.Wo) return;
  const int oy = m / p.Wo, ox = m - oy * p.Wo;
  float x[27];
  const size_t plane = (size_t)p.H * p.W;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = oy * 2 + kh - 1;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = ox * 2 + kw - 1;
        const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        x[(c * 3 + kh) * 3 + kw] = ok ? p.src[c * plane + (size_t)iy * p.W + ix] : 0.f;
      }
    }
  float acc[CO];
#pragma unroll
  for (int r = 0; r < CO; ++r) acc[r] = sb[r];
#pragma unroll
  for (int k = 0; k < 27; ++k) {
#pragma unroll
    for (int r = 0; r < CO; r += 4) {
      const float4 w4 = *reinterpret_cast<const float4*>(&sw[k * CO + r]);
      acc[r + 0] = __builtin_fmaf(x[k], w4.x, acc[r + 0]);
      acc[r + 1] = __builtin_fmaf(x[k], w4.y, acc[r + 1]);
      acc[r + 2] = __builtin_fmaf(x[k], w4.z, acc[r + 2]);
      acc[r + 3] = __builtin_fmaf(x[k], w4.w, acc[r + 3]);
    }
  }
  T* d = static_cast<T*>(p.dst) + (size_t)m * p.dst_ld;
  constexpr int V = 16 / sizeof(T);  // elements per 16-byte store
  typedef T vec_t __attribute__((ext_vector_type(V)));
#pragma unroll
  for (int r = 0; r < CO; r += V) {
    vec_t hv;
#pragma unroll
    for (int q = 0; q < V; ++q) hv[q] = (T)(acc[r + q] > 0.f ? acc[r + q] : 0.f);
    *reinterpret_cast<vec_t*>(d + r) = hv;
  }
}

hipError_t stem_desc(const StemParams& p, LaunchDesc* out) {
  out->grid = dim3((p.Ho * p.Wo + 255) / 256);
  out->block = dim3(256);
  out->shmem = 0;
  if (p.Co == 32 && p.dtype == kF16) out->func = reinterpret_cast<const void*>(&stem_conv_kernel<half_t, 32>);
  else if (p.Co == 64 && p.dtype == kF16) out->func = reinterpret_cast<const void*>(&stem_conv_kernel<half_t, 64>);
  else if (p.Co == 32 && p.dtype == kF32) out->func = reinterpret_cast<const void*>(&stem_conv_kernel<float, 32>);
  else if (p.Co == 64 && p.dtype == kF32) out->func = reinterpret_cast<const void*>(&stem_conv_kernel<float, 64>);
  else return hipErrorInvalidValue;
  return hipSuccess;
}

hipError_t stem_launch(const StemParams& p, hipStream_t stream, dim3* grid_out, dim3* block_out) {
  LaunchDesc d;
  hipError_t e = stem_desc(p, &d);
  if (e != hipSuccess) return e;
  if (grid_out) *grid_out = d.grid;
  if (block_out) *block_out = d.block;
  StemParams copy = p;
  void* args[] = {&copy};
  return hipLaunchKernel(d.func, d.grid, d.block, args, d.shmem, stream);
}

// ---------------------------------------------------------------------------------------------- SPPF pool
// Three chained 5x5/s1/p2 max-pools of x equal the 5x5, 9x9 and 13x13 clipped-window maxima of x
// (max is associative and the implicit -inf padding never wins). The kernel is separable and LDS-staged:
// a block owns one output row y and a 64-channel slab:
//   phase 1: for each of the (up to) 13 input rows y-6..y+6 compute, per column, nothing yet -- rows are
//            just staged: LDS tile [13][W][64ch] would be 13*40*128 B = 66 KB at W=40, so instead the
//            VERTICAL maxima are formed first while streaming rows from global memory:
//              v5[x] = max rows y-2..y+2,  v9[x] = max rows y-4..y+4,  v13[x] = max rows y-6..y+6
//            (each thread owns (x, 8 channels): 13 coalesced 16-byte loads), written to LDS;
//   phase 2: horizontal maxima over v5/v9/v13 within +-2/4/6 columns read from LDS -> y1,y2,y3.
// fp16 max is exact, so this is bit-identical to chaining the pools in any precision.
template <typename T> struct PoolVec;
template <> struct PoolVec<half_t> {
  static constexpr int V = 8;
  typedef _Float16 vec __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ half_t lowest() { return (half_t)(-65504.0f); }
};
template <> struct PoolVec<float> {
  static constexpr int V = 4;
  typedef float vec __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ float lowest() { return -3.402823466e38f; }
};

template <> struct PoolVec<signed char> {
  static constexpr int V = 16;
  typedef signed char vec __attribute__((ext_vector_type(16)));
  static __device__ __forceinline__ signed char lowest() { return -128; }
};

template <typename T, int CH>  // CH channels per block (multiple of the 16-byte vector width)
__global__ __launch_bounds__(256) void sppf_pool_kernel(const PoolParams p) {
  typedef PoolVec<T> PV;
  typedef typename PV::vec vec_t;
  constexpr int V = PV::V, NV = CH / V;
  extern __shared__ __align__(16) unsigned char smem[];
  vec_t* v5 = reinterpret_cast<vec_t*>(smem);  // [W][NV]
  vec_t* v9 = v5 + p.W * NV;
  vec_t* v13 = v9 + p.W * NV;
  const int y = blockIdx.x;
  const int c0 = blockIdx.y * CH;
  const int nvec = p.W * NV;
  T* base = static_cast<T*>(p.buf);
  const T* x = base + p.coff + c0;
  vec_t lo;
#pragma unroll
  for (int i = 0; i < V; ++i) lo[i] = PV::lowest();  // below every finite value (post-ReLU inputs are >= 0)
  auto vmax = [](vec_t a, vec_t b) {
    vec_t r;
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = a[i] > b[i] ? a[i] : b[i];
    return r;
  };
  for (int t = threadIdx.x; t < nvec; t += blockDim.x) {
    const int xx = t / NV, cv = t % NV;
    vec_t m5 = lo, m9 = lo, m13 = lo;
#pragma unroll
    for (int dy = -6; dy <= 6; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= p.H) continue;
      const vec_t v = *reinterpret_cast<const vec_t*>(x + ((size_t)yy * p.W + xx) * p.ld + cv * V);
      m13 = vmax(m13, v);
      if (dy >= -4 && dy <= 4) m9 = vmax(m9, v);
      if (dy >= -2 && dy <= 2) m5 = vmax(m5, v);
    }
    v5[t] = m5;
    v9[t] = m9;
    v13[t] = m13;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nvec; t += blockDim.x) {
    const int xx = t / NV, cv = t % NV;
    vec_t o5 = lo, o9 = lo, o13 = lo;
#pragma unroll
    for (int dx = -6; dx <= 6; ++dx) {
      const int x2 = xx + dx;
      if (x2 < 0 || x2 >= p.W) continue;
      const int idx = x2 * NV + cv;
      o13 = vmax(o13, v13[idx]);
      if (dx >= -4 && dx <= 4) o9 = vmax(o9, v9[idx]);
      if (dx >= -2 && dx <= 2) o5 = vmax(o5, v5[idx]);
    }
    T* o = base + ((size_t)y * p.W + xx) * p.ld + p.coff + c0 + cv * V;
    *reinterpret_cast<vec_t*>(o + p.C) = o5;
    *reinterpret_cast<vec_t*>(o + 2 * p.C) = o9;
    *reinterpret_cast<vec_t*>(o + 3 * p.C) = o13;
  }
}

hipError_t sppf_pool_launch(const PoolParams& p, hipStream_t stream, dim3* grid_out, dim3* block_out) {
  constexpr int CH = 32;
  if (p.C % CH) return hipErrorInvalidValue;
  dim3 grid(p.H, p.C / CH), block(256);
  const size_t esz = p.dtype == kF32 ? 4 : (p.dtype == kI8 ? 1 : 2);
  const size_t smem = (size_t)3 * p.W * CH * esz;
  if (smem > 64 * 1024) return hipErrorInvalidValue;
  if (grid_out) *grid_out = grid;
  if (block_out) *block_out = block;
  if (p.dtype == kF32) sppf_pool_kernel<float, CH><<<grid, block, smem, stream>>>(p);
  else if (p.dtype == kI8) sppf_pool_kernel<signed char, CH><<<grid, block, smem, stream>>>(p);  // one scale per buffer: max commutes with it
  else sppf_pool_kernel<half_t, CH><<<grid, block, smem, stream>>>(p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- quantise
__global__ __launch_bounds__(256) void quant_f16_i8_kernel(const QuantParams p) {
  typedef signed char c16 __attribute__((ext_vector_type(16)));
  const size_t nvec = p.n / 16;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nvec; t += (size_t)gridDim.x * blockDim.x) {
    const half8 a = *reinterpret_cast<const half8*>(p.src + t * 16);
    const half8 b = *reinterpret_cast<const half8*>(p.src + t * 16 + 8);
    c16 q;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float x = __builtin_rintf((float)a[i] * p.inv_scale), y = __builtin_rintf((float)b[i] * p.inv_scale);
      x = x > 127.f ? 127.f : (x < -127.f ? -127.f : x);
      y = y > 127.f ? 127.f : (y < -127.f ? -127.f : y);
      q[i] = (signed char)(int)x;
      q[i + 8] = (signed char)(int)y;
    }
    *reinterpret_cast<c16*>(p.dst + t * 16) = q;
  }
}

hipError_t quant_launch(const QuantParams& p, hipStream_t stream) {
  if (p.n % 16) return hipErrorInvalidValue;
  size_t blocks = (p.n / 16 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  quant_f16_i8_kernel<<<(int)blocks, 256, 0, stream>>>(p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- upsample
__global__ __launch_bounds__(256) void upsample2x_kernel(const UpsampleParams p) {
  const int vec = p.C / 8;
  const size_t total = (size_t)4 * p.H * p.W * vec;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int cv = t % vec;
    const size_t pix = t / vec;
    const int ox = pix % (2 * p.W), oy = pix / (2 * p.W);
    const half8 v = *reinterpret_cast<const half8*>(p.src + ((size_t)(oy >> 1) * p.W + (ox >> 1)) * p.src_ld + cv * 8);
    *reinterpret_cast<half8*>(p.dst + pix * p.dst_ld + cv * 8) = v;
  }
}

hipError_t upsample2x_launch(const UpsampleParams& p, hipStream_t stream) {
  if (p.C % 8) return hipErrorInvalidValue;
  const size_t total = (size_t)4 * p.H * p.W * (p.C / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  upsample2x_kernel<<<blocks, 256, 0, stream>>>(p);
  return hipGetLastError();
}

}  // namespace unina
