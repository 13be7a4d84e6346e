// stem_pool.hip -- the non-GEMM ops of the forward graph (gfx950).
//   stem_conv_kernel   : backbone.stem  (model.py:175: ConvBlock(3, c1, k=3, s=2)) reading the fp32 NCHW image
//   sppf_pool_kernel   : SPPF_DLA's three chained MaxPool2d(5,1,2)  (model.py:125,129-131)
//   upsample2x_kernel  : Upsample(scale_factor=2, nearest)  (model.py:145-147), standalone form
#include "kernels.h"

#include <cstdint>
#include <cstdlib>
#include <type_traits>

namespace unina {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// Storage of one output element: fp16 / fp32, or (kS16, the split-fp16 precision mode) an fp16 pair hi + lo in two planes.
template <typename T> struct StemOut {
  static constexpr bool kSplit = false;
  static constexpr int kBytes = (int)sizeof(T);    // staged bytes per channel
  typedef T plane_t;
};
template <> struct StemOut<s16_t> {
  static constexpr bool kSplit = true;
  static constexpr int kBytes = 4;
  typedef half_t plane_t;
};

// ---------------------------------------------------------------------------------------------- stem
// Cin = 3 makes K = 27: far too thin for a matrix-core tile, and the op is HBM-bound anyway
// (4.9 MB fp32 in, 6.6 MB fp16 out at 640^2 against 0.18 GFLOP). Plain fp32 FMAs. TWO threads per output pixel,
// each with half of the output channels: 3 200 waves at 640^2 spread evenly over the 1 024 SIMDs (one thread per
// pixel gave 1.56 waves per SIMD, i.e. two uneven rounds), and the accumulators are float2 pairs so every FMA
// instruction is a v_pk_fma_f32 (two channels per issue slot; element-wise, so each channel still sees exactly the
// sequential fma chain k = 0..26 of the scalar form). A wave's 27 input loads each cover 64 neighbouring pixels
// (stride-2 fp32), the 27 x CO/2 folded weights of its channel half are scalar loads, and each thread stores its
// pixel's CO/2 contiguous NHWC channels.
typedef float floatx2 __attribute__((ext_vector_type(2)));

// Network-input pixel (y, x) of a camera frame: the arithmetic of preprocess.hip (cuda_preprocess.cu:99-128 plain BGRA,
// :144-204 half-pixel-centre bilinear resize), expression trees rounded exactly as written there.
__device__ __forceinline__ void camera_pixel(const StemParams& p, int y, int x, float (&rgb)[3]) {
#pragma clang fp contract(off)
  float r, g, b;
  if (p.src_kind == 1) {
    const uchar4 px = *reinterpret_cast<const uchar4*>(p.cam + (size_t)y * p.cam_pitch + (size_t)x * 4);  // B,G,R,A
    r = (float)px.z; g = (float)px.y; b = (float)px.x;
  } else {
    const int sw = p.cam_w, sh = p.cam_h;
    const float scale_x = (float)sw / p.W, scale_y = (float)sh / p.H;
    float sx = (x + 0.5f) * scale_x - 0.5f, sy = (y + 0.5f) * scale_y - 0.5f;
    sx = fmaxf(0.0f, fminf(sx, sw - 1.0f));
    sy = fmaxf(0.0f, fminf(sy, sh - 1.0f));
    const int x0 = (int)sx, y0 = (int)sy;
    const int x1 = min(x0 + 1, sw - 1), y1 = min(y0 + 1, sh - 1);
    const float fx = sx - x0, fy = sy - y0;
    const float w00 = (1.0f - fx) * (1.0f - fy), w01 = fx * (1.0f - fy), w10 = (1.0f - fx) * fy, w11 = fx * fy;
    const uchar4 p00 = *reinterpret_cast<const uchar4*>(p.cam + (size_t)y0 * p.cam_pitch + (size_t)x0 * 4);
    const uchar4 p01 = *reinterpret_cast<const uchar4*>(p.cam + (size_t)y0 * p.cam_pitch + (size_t)x1 * 4);
    const uchar4 p10 = *reinterpret_cast<const uchar4*>(p.cam + (size_t)y1 * p.cam_pitch + (size_t)x0 * 4);
    const uchar4 p11 = *reinterpret_cast<const uchar4*>(p.cam + (size_t)y1 * p.cam_pitch + (size_t)x1 * 4);
    r = w00 * p00.z + w01 * p01.z + w10 * p10.z + w11 * p11.z;
    g = w00 * p00.y + w01 * p01.y + w10 * p10.y + w11 * p11.y;
    b = w00 * p00.x + w01 * p01.x + w10 * p10.x + w11 * p11.x;
  }
  rgb[0] = ((r / 255.0f) - p.norm.mean_r) / p.norm.std_r;
  rgb[1] = ((g / 255.0f) - p.norm.mean_g) / p.norm.std_g;
  rgb[2] = ((b / 255.0f) - p.norm.mean_b) / p.norm.std_b;
}

template <typename T, int CO>
__global__ __launch_bounds__(256) void stem_conv_kernel(const StemParams p) {
  constexpr int CH = CO / 2;  // channels per thread
  // A WAVE owns 64 consecutive output pixels and ONE half of the output channels (even waves the first, odd waves the
  // second), so the 27 x CH folded weights it needs are wave-uniform: scalar loads (s_load into SGPRs, read by the FMAs as
  // scalar operands), no LDS at all. (The previous form kept both halves in a wave and read the weights from an LDS copy,
  // 108 ds_read_b128 per thread; measured the same 12.5 us at 640^2 -- the kernel is not LDS-bound -- but this one needs
  // no LDS, no barrier and 70 instead of 84 VGPRs.)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c0 = (wave & 1) * CH;
  const int m = blockIdx.x * 128 + (wave >> 1) * 64 + (int)(threadIdx.x & 63);
  if (m >= p.Ho * p.Wo) return;
  const float* __restrict__ wt = p.wt + c0;
  const float* __restrict__ bs = p.bias + c0;
  const int oy = m / p.Wo, ox = m - oy * p.Wo;
  float x[27];
  const size_t plane = (size_t)p.H * p.W;
  if (p.src_kind == 0) {
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
  } else {
    // camera frame: the network-input pixel (iy, ix) is computed as preprocess.hip would have written it
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = oy * 2 + kh - 1;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = ox * 2 + kw - 1;
        float rgb[3] = {0.f, 0.f, 0.f};
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) camera_pixel(p, iy, ix, rgb);
#pragma unroll
        for (int c = 0; c < 3; ++c) x[(c * 3 + kh) * 3 + kw] = rgb[c];
      }
    }
  }
  floatx2 acc[CH / 2];
#pragma unroll
  for (int r = 0; r < CH / 2; ++r) acc[r] = floatx2{bs[2 * r], bs[2 * r + 1]};
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const floatx2 xk = {x[k], x[k]};
#pragma unroll
    for (int r = 0; r < CH / 2; ++r) acc[r] = __builtin_elementwise_fma(xk, floatx2{wt[k * CO + 2 * r], wt[k * CO + 2 * r + 1]}, acc[r]);
  }
  typedef typename StemOut<T>::plane_t PT;
  PT* d = static_cast<PT*>(p.dst) + (size_t)m * p.dst_ld + c0;
  constexpr int V = 16 / sizeof(PT);  // elements per 16-byte store
  typedef PT vec_t __attribute__((ext_vector_type(V)));
#pragma unroll
  for (int r = 0; r < CH; r += V) {
    vec_t hv, lv;
#pragma unroll
    for (int q = 0; q < V; ++q) {
      float a = acc[(r + q) >> 1][(r + q) & 1];
      a = a > 0.f ? a : 0.f;
      hv[q] = (PT)a;
      lv[q] = (PT)(a - (float)hv[q]);
    }
    *reinterpret_cast<vec_t*>(d + r) = hv;
    if constexpr (StemOut<T>::kSplit) *reinterpret_cast<vec_t*>(reinterpret_cast<unsigned char*>(d + r) + p.dst_lo) = lv;
  }
}

// ---- tiled form (the default) ----------------------------------------------------------------------------------
// The one-thread-per-pixel form above issues 27 stride-2 dword loads per thread, twice (both channel halves): 173 000
// wave-level load instructions per 640^2 frame, each a 512-byte span of which half is used, and fetches its 27 x CO/2
// weights with scalar loads that the compiler cannot keep in flight (54 s_load_dwordx16, each behind an
// s_waitcnt lgkmcnt(0)): 13.4 us at 640^2 for 11.5 MB (0.86 TB/s). Here a 128-thread workgroup owns 2 x 64 output pixels (800 workgroups at 640^2: three per CU, so that one's loads overlap another's arithmetic):
//   1. its input footprint (3 planes x 5 rows x 132 columns fp32, 8 KB) comes in as aligned 16-byte loads along W --
//      ~5 000 wave-level loads per frame, every byte used -- into LDS, next to the weights ([27][CO] + bias, 3.5 KB); a
//      camera frame is pre-processed once per footprint pixel instead of once per tap that touches it;
//   2. thread = TWO output pixels (the tile's two rows, one column) x one HALF of the channels: the same sequential fma chain
//      per channel (k = (c, kh, kw) = 0..26 from the bias, v_pk_fma_f32 on channel pairs), inputs from LDS, weights as LDS
//      broadcast reads shared by the two pixels and read one tap ahead;
//   3. the tile's NHWC rows are staged in LDS and leave as contiguous 16-byte-per-lane stores (full lines).
constexpr int kStemNT = 128;   // threads per workgroup
constexpr int kStemTH = 2, kStemTW = 64, kStemPR = 2 * kStemTH + 1, kStemPW = 2 * kStemTW + 4;   // patch rows / columns
template <typename T, int CO>
struct StemTile {
  static constexpr int NPATCH = 3 * kStemPR * kStemPW;                // floats
  static constexpr int NW = 27 * CO + CO;                             // weights [27][CO] then bias [CO]
  static constexpr int ROWB = CO * StemOut<T>::kBytes + 16;           // padded staging row (bytes) of one output pixel (split: [hi | lo])
  static constexpr int STAGE = kStemTH * kStemTW * ROWB;              // the staging tile reuses the patch + weight area
  static constexpr int IN_BYTES = (NPATCH + NW) * 4;
  static constexpr unsigned SMEM = (unsigned)(STAGE > IN_BYTES ? STAGE : IN_BYTES);
};
template <typename T, int CO>
__global__ __launch_bounds__(kStemNT) void stem_tile_kernel(const StemParams p) {
  typedef StemTile<T, CO> ST;
  extern __shared__ __align__(16) unsigned char stem_smem[];
  float* patch = reinterpret_cast<float*>(stem_smem);          // [3][kStemPR][kStemPW]
  float* wl = patch + ST::NPATCH;                               // [27][CO], then bias [CO]
  const int tid = threadIdx.x;
  const int tiles_x = (p.Wo + kStemTW - 1) / kStemTW;
  const int ty0 = ((int)blockIdx.x / tiles_x) * kStemTH, tx0 = ((int)blockIdx.x % tiles_x) * kStemTW;
  const int iy0 = 2 * ty0 - 1, ix0 = 2 * tx0 - 4;               // image coordinates of patch (row 0, column 0)
  for (int i = tid; i < 27 * CO / 4; i += kStemNT) reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(p.wt)[i];
  if (tid < CO) wl[27 * CO + tid] = p.bias[tid];
  if (p.src_kind == 0) {
    const size_t plane = (size_t)p.H * p.W;
    constexpr int NV = 3 * kStemPR * (kStemPW / 4);
    for (int v = tid; v < NV; v += kStemNT) {
      const int c = v / (kStemPR * (kStemPW / 4)), rem = v - c * (kStemPR * (kStemPW / 4));
      const int r = rem / (kStemPW / 4), q = rem - r * (kStemPW / 4);
      const int iy = iy0 + r, ix = ix0 + 4 * q;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (iy >= 0 && iy < p.H) {
        const float* row = p.src + c * plane + (size_t)iy * p.W;
        if (ix >= 0 && ix + 3 < p.W) {
          val = *reinterpret_cast<const float4*>(row + ix);    // W % 4 == 0 and a 16-byte aligned tensor: aligned
        } else {
          if (ix >= 0 && ix < p.W) val.x = row[ix];
          if (ix + 1 >= 0 && ix + 1 < p.W) val.y = row[ix + 1];
          if (ix + 2 >= 0 && ix + 2 < p.W) val.z = row[ix + 2];
          if (ix + 3 >= 0 && ix + 3 < p.W) val.w = row[ix + 3];
        }
      }
      *reinterpret_cast<float4*>(patch + (c * kStemPR + r) * kStemPW + 4 * q) = val;
    }
  } else {
    // camera frame: every footprint pixel is pre-processed once, as preprocess.hip would have written it
    for (int e = tid; e < kStemPR * kStemPW; e += kStemNT) {
      const int r = e / kStemPW, j = e - r * kStemPW;
      const int iy = iy0 + r, ix = ix0 + j;
      float rgb[3] = {0.f, 0.f, 0.f};
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) camera_pixel(p, iy, ix, rgb);
#pragma unroll
      for (int c = 0; c < 3; ++c) patch[(c * kStemPR + r) * kStemPW + j] = rgb[c];
    }
  }
  __syncthreads();
  // a WAVE = the tile's two rows x 64 columns x ONE half of the output channels: 1 600 waves at 640^2, so that
  // most SIMDs interleave two (a single wave issues a VALU instruction every 4 cycles, two waves every 2)
  constexpr int CH = CO / 2;
  const int tx = tid & 63, c0 = (tid >> 6) * CH;   // (tile rows 0 and 1)
  float x[2][27];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const float* row = patch + (c * kStemPR + 2 * h + kh) * kStemPW + 2 * tx + 3;
        x[h][(c * 3 + kh) * 3 + 0] = row[0];
        const floatx2 v12 = *reinterpret_cast<const floatx2*>(row + 1);   // (8-byte aligned: 2 * tx + 4)
        x[h][(c * 3 + kh) * 3 + 1] = v12[0];
        x[h][(c * 3 + kh) * 3 + 2] = v12[1];
      }
  // (v_pk_fma_f32 on channel pairs; a scalar v_fma_f32 build of this loop measured 10.3 us against 9.5 us)
  floatx2 acc[2][CH / 2];
#pragma unroll
  for (int r = 0; r < CH / 2; ++r) {
    acc[0][r] = *reinterpret_cast<const floatx2*>(wl + 27 * CO + c0 + 2 * r);
    acc[1][r] = acc[0][r];
  }
  // the weights of tap k + 1 are read (LDS broadcast: the same address in every lane) while tap k is computed: left to
  // itself the compiler issues two reads, waits for them, uses them -- one exposed LDS latency per 8 FMAs, and with one
  // or two waves per SIMD nothing else covers it (measured: 13.5 us, slower than the per-pixel form)
  float4 wbuf[2][CH / 4];
#pragma unroll
  for (int j = 0; j < CH / 4; ++j) wbuf[0][j] = *reinterpret_cast<const float4*>(wl + c0 + 4 * j);
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    if (k + 1 < 27) {
#pragma unroll
      for (int j = 0; j < CH / 4; ++j) wbuf[(k + 1) & 1][j] = *reinterpret_cast<const float4*>(wl + (k + 1) * CO + c0 + 4 * j);
    }
    __builtin_amdgcn_sched_barrier(0);
    const floatx2 x0 = {x[0][k], x[0][k]}, x1 = {x[1][k], x[1][k]};
#pragma unroll
    for (int j = 0; j < CH / 4; ++j) {
      const float4 w4 = wbuf[k & 1][j];
      acc[0][2 * j] = __builtin_elementwise_fma(x0, floatx2{w4.x, w4.y}, acc[0][2 * j]);
      acc[0][2 * j + 1] = __builtin_elementwise_fma(x0, floatx2{w4.z, w4.w}, acc[0][2 * j + 1]);
      acc[1][2 * j] = __builtin_elementwise_fma(x1, floatx2{w4.x, w4.y}, acc[1][2 * j]);
      acc[1][2 * j + 1] = __builtin_elementwise_fma(x1, floatx2{w4.z, w4.w}, acc[1][2 * j + 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();                                              // every thread has read its inputs: the input area is free
  typedef typename StemOut<T>::plane_t PT;
  constexpr bool SPLIT = StemOut<T>::kSplit;
  constexpr int V = 16 / (int)sizeof(PT);                       // elements per 16-byte chunk
  typedef PT vec_t __attribute__((ext_vector_type(V)));
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < CH; r += V) {
      vec_t hv, lv;
#pragma unroll
      for (int q = 0; q < V; ++q) {
        float a = acc[h][(r + q) >> 1][(r + q) & 1];
        a = a > 0.f ? a : 0.f;
        hv[q] = (PT)a;
        lv[q] = (PT)(a - (float)hv[q]);
      }
      *reinterpret_cast<vec_t*>(stem_smem + (h * kStemTW + tx) * ST::ROWB + (c0 + r) * (int)sizeof(PT)) = hv;
      if constexpr (SPLIT) *reinterpret_cast<vec_t*>(stem_smem + (h * kStemTW + tx) * ST::ROWB + CO * 2 + (c0 + r) * 2) = lv;
    }
  __syncthreads();
  constexpr int CPL = CO * (int)sizeof(PT) / 16;                // 16-byte chunks per pixel and plane
  constexpr int CPP = SPLIT ? 2 * CPL : CPL;                    // ... per pixel
  typedef float vec16 __attribute__((ext_vector_type(4)));
  unsigned char* dst = static_cast<unsigned char*>(p.dst);
#pragma unroll
  for (int i = 0; i < CPP; ++i) {
    const int chunk = i * kStemNT + tid;
    const int pl = chunk / CPP, part = chunk - pl * CPP;        // tile pixel (row-major), chunk of its channels
    const int oy = ty0 + (pl >> 6), ox = tx0 + (pl & 63);
    const int plane = part / CPL, pc = part - plane * CPL;      // (split: the staged row is [hi chunks | lo chunks])
    if (oy < p.Ho && ox < p.Wo)
      *reinterpret_cast<vec16*>(dst + plane * p.dst_lo + ((size_t)(oy * p.Wo + ox) * p.dst_ld) * sizeof(PT) + pc * 16) =
          *reinterpret_cast<const vec16*>(stem_smem + pl * ST::ROWB + part * 16);
  }
}

template <typename T, int CO>
constexpr unsigned stem_tile_smem() { return StemTile<T, CO>::SMEM; }

// per device: the fp32 / wide instantiations stage more than the default 64 KB of dynamic LDS
hipError_t stem_init() {
  struct { const void* fn; unsigned smem; } ks[] = {
      {reinterpret_cast<const void*>(&stem_tile_kernel<half_t, 32>), stem_tile_smem<half_t, 32>()},
      {reinterpret_cast<const void*>(&stem_tile_kernel<half_t, 64>), stem_tile_smem<half_t, 64>()},
      {reinterpret_cast<const void*>(&stem_tile_kernel<float, 32>), stem_tile_smem<float, 32>()},
      {reinterpret_cast<const void*>(&stem_tile_kernel<float, 64>), stem_tile_smem<float, 64>()},
      {reinterpret_cast<const void*>(&stem_tile_kernel<s16_t, 32>), stem_tile_smem<s16_t, 32>()},
      {reinterpret_cast<const void*>(&stem_tile_kernel<s16_t, 64>), stem_tile_smem<s16_t, 64>()}};
  for (const auto& k : ks) {
    hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.smem);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t stem_desc(const StemParams& p, LaunchDesc* out) {
  if (!p.wt) return hipErrorInvalidValue;
  out->block = dim3(256);
  static const bool legacy = getenv("UNINA_STEM_V1") && getenv("UNINA_STEM_V1")[0] == '1';   // the one-thread-per-pixel form
  // The form depends on the SHAPE only, never on the pointer of the frame at hand: the captured frame graph's stem node is
  // re-pointed per frame (hipGraphExecKernelNodeSetParams) and must keep its function, grid and block. The tiled kernel's aligned
  // 16-byte row loads need W % 4 == 0 and a 16-byte aligned tensor -- which unina_set_tensor_address / unina_infer guarantee
  // (they refuse any other address with UNINA_ERR_ARG).
  if (!legacy && (p.W & 3) == 0 && p.dst_ld == p.Co) {
    out->block = dim3(kStemNT);
    out->grid = dim3(((p.Ho + kStemTH - 1) / kStemTH) * ((p.Wo + kStemTW - 1) / kStemTW));
    if (p.Co == 32 && p.dtype == kF16) { out->func = reinterpret_cast<const void*>(&stem_tile_kernel<half_t, 32>); out->shmem = stem_tile_smem<half_t, 32>(); }
    else if (p.Co == 64 && p.dtype == kF16) { out->func = reinterpret_cast<const void*>(&stem_tile_kernel<half_t, 64>); out->shmem = stem_tile_smem<half_t, 64>(); }
    else if (p.Co == 32 && p.dtype == kF32) { out->func = reinterpret_cast<const void*>(&stem_tile_kernel<float, 32>); out->shmem = stem_tile_smem<float, 32>(); }
    else if (p.Co == 64 && p.dtype == kF32) { out->func = reinterpret_cast<const void*>(&stem_tile_kernel<float, 64>); out->shmem = stem_tile_smem<float, 64>(); }
    else if (p.Co == 32 && p.dtype == kS16) { out->func = reinterpret_cast<const void*>(&stem_tile_kernel<s16_t, 32>); out->shmem = stem_tile_smem<s16_t, 32>(); }
    else if (p.Co == 64 && p.dtype == kS16) { out->func = reinterpret_cast<const void*>(&stem_tile_kernel<s16_t, 64>); out->shmem = stem_tile_smem<s16_t, 64>(); }
    else return hipErrorInvalidValue;
    return hipSuccess;
  }
  out->grid = dim3((p.Ho * p.Wo + 127) / 128);        // 128 pixels x two channel halves per 256-thread workgroup
  out->shmem = 0;
  if (p.Co == 32 && p.dtype == kF16) out->func = reinterpret_cast<const void*>(&stem_conv_kernel<half_t, 32>);
  else if (p.Co == 64 && p.dtype == kF16) out->func = reinterpret_cast<const void*>(&stem_conv_kernel<half_t, 64>);
  else if (p.Co == 32 && p.dtype == kF32) out->func = reinterpret_cast<const void*>(&stem_conv_kernel<float, 32>);
  else if (p.Co == 64 && p.dtype == kF32) out->func = reinterpret_cast<const void*>(&stem_conv_kernel<float, 64>);
  else if (p.Co == 32 && p.dtype == kS16) out->func = reinterpret_cast<const void*>(&stem_conv_kernel<s16_t, 32>);
  else if (p.Co == 64 && p.dtype == kS16) out->func = reinterpret_cast<const void*>(&stem_conv_kernel<s16_t, 64>);
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
  // Rows / columns outside the image are CLAMPED to the border instead of skipped: the border row (column) always
  // lies inside the clipped window, so the duplicate never changes a maximum -- and without the per-row branches the
  // 13 loads are independent and issued back to back (with them they serialised: 13 x L2 latency per thread).
  (void)lo;
  for (int t = threadIdx.x; t < nvec; t += blockDim.x) {
    const int xx = t / NV, cv = t % NV;
    vec_t v[13];
#pragma unroll
    for (int dy = -6; dy <= 6; ++dy) {
      int yy = y + dy;
      yy = yy < 0 ? 0 : (yy >= p.H ? p.H - 1 : yy);
      v[dy + 6] = *reinterpret_cast<const vec_t*>(x + ((size_t)yy * p.W + xx) * p.ld + cv * V);
    }
    vec_t m5 = v[6];
#pragma unroll
    for (int d = 1; d <= 2; ++d) m5 = vmax(m5, vmax(v[6 - d], v[6 + d]));
    vec_t m9 = m5;
#pragma unroll
    for (int d = 3; d <= 4; ++d) m9 = vmax(m9, vmax(v[6 - d], v[6 + d]));
    vec_t m13 = m9;
#pragma unroll
    for (int d = 5; d <= 6; ++d) m13 = vmax(m13, vmax(v[6 - d], v[6 + d]));
    v5[t] = m5;
    v9[t] = m9;
    v13[t] = m13;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nvec; t += blockDim.x) {
    const int xx = t / NV, cv = t % NV;
    auto col = [&](int dx) {
      int x2 = xx + dx;
      x2 = x2 < 0 ? 0 : (x2 >= p.W ? p.W - 1 : x2);
      return x2 * NV + cv;
    };
    vec_t o5 = v5[col(0)], o9 = v9[col(0)], o13 = v13[col(0)];
#pragma unroll
    for (int d = 1; d <= 6; ++d) {
      o13 = vmax(o13, vmax(v13[col(-d)], v13[col(d)]));
      if (d <= 4) o9 = vmax(o9, vmax(v9[col(-d)], v9[col(d)]));
      if (d <= 2) o5 = vmax(o5, vmax(v5[col(-d)], v5[col(d)]));
    }
    T* o = base + ((size_t)y * p.W + xx) * p.ld + p.coff + c0 + cv * V;
    *reinterpret_cast<vec_t*>(o + p.C) = o5;
    *reinterpret_cast<vec_t*>(o + 2 * p.C) = o9;
    *reinterpret_cast<vec_t*>(o + 3 * p.C) = o13;
  }
}

// split fp16 (kS16): the same separable scheme on (hi, lo) pairs. A pair's value hi + lo fits an fp32 (11 + 11 mantissa bits, lo
// below half an ulp of hi), so the maxima are taken on that fp32 value (exactly) and the winner is split again on the way out
// (fp16(v) and fp16(v - fp16(v)): the pair it came from, up to the choice at a rounding tie -- the same VALUE either way).
template <int CH>
__global__ __launch_bounds__(256) void sppf_pool_split_kernel(const PoolParams p) {
  typedef float vec_t __attribute__((ext_vector_type(8)));
  constexpr int V = 8, NV = CH / V;
  extern __shared__ __align__(16) unsigned char smem[];
  vec_t* v5 = reinterpret_cast<vec_t*>(smem);  // [W][NV]
  vec_t* v9 = v5 + p.W * NV;
  vec_t* v13 = v9 + p.W * NV;
  const int y = blockIdx.x;
  const int c0 = blockIdx.y * CH;
  const int nvec = p.W * NV;
  half_t* base = static_cast<half_t*>(p.buf);
  const half_t* x = base + p.coff + c0;
  auto vmax = [](vec_t a, vec_t b) {
    vec_t r;
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = a[i] > b[i] ? a[i] : b[i];
    return r;
  };
  auto load = [&](const half_t* at) {
    const half8 h = *reinterpret_cast<const half8*>(at);
    const half8 l = *reinterpret_cast<const half8*>(reinterpret_cast<const unsigned char*>(at) + p.lo);
    vec_t r;
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = (float)h[i] + (float)l[i];
    return r;
  };
  auto store = [&](half_t* at, const vec_t& v) {
    half8 h, l;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      h[i] = (half_t)v[i];
      l[i] = (half_t)(v[i] - (float)h[i]);
    }
    *reinterpret_cast<half8*>(at) = h;
    *reinterpret_cast<half8*>(reinterpret_cast<unsigned char*>(at) + p.lo) = l;
  };
  for (int t = threadIdx.x; t < nvec; t += blockDim.x) {
    const int xx = t / NV, cv = t % NV;
    vec_t v[13];
#pragma unroll
    for (int dy = -6; dy <= 6; ++dy) {
      int yy = y + dy;
      yy = yy < 0 ? 0 : (yy >= p.H ? p.H - 1 : yy);
      v[dy + 6] = load(x + ((size_t)yy * p.W + xx) * p.ld + cv * V);
    }
    vec_t m5 = v[6];
#pragma unroll
    for (int d = 1; d <= 2; ++d) m5 = vmax(m5, vmax(v[6 - d], v[6 + d]));
    vec_t m9 = m5;
#pragma unroll
    for (int d = 3; d <= 4; ++d) m9 = vmax(m9, vmax(v[6 - d], v[6 + d]));
    vec_t m13 = m9;
#pragma unroll
    for (int d = 5; d <= 6; ++d) m13 = vmax(m13, vmax(v[6 - d], v[6 + d]));
    v5[t] = m5;
    v9[t] = m9;
    v13[t] = m13;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nvec; t += blockDim.x) {
    const int xx = t / NV, cv = t % NV;
    auto col = [&](int dx) {
      int x2 = xx + dx;
      x2 = x2 < 0 ? 0 : (x2 >= p.W ? p.W - 1 : x2);
      return x2 * NV + cv;
    };
    vec_t o5 = v5[col(0)], o9 = v9[col(0)], o13 = v13[col(0)];
#pragma unroll
    for (int d = 1; d <= 6; ++d) {
      o13 = vmax(o13, vmax(v13[col(-d)], v13[col(d)]));
      if (d <= 4) o9 = vmax(o9, vmax(v9[col(-d)], v9[col(d)]));
      if (d <= 2) o5 = vmax(o5, vmax(v5[col(-d)], v5[col(d)]));
    }
    half_t* o = base + ((size_t)y * p.W + xx) * p.ld + p.coff + c0 + cv * V;
    store(o + p.C, o5);
    store(o + 2 * p.C, o9);
    store(o + 3 * p.C, o13);
  }
}

hipError_t sppf_pool_launch(const PoolParams& p, hipStream_t stream, dim3* grid_out, dim3* block_out) {
  constexpr int CH = 32;
  if (p.C % CH) return hipErrorInvalidValue;
  dim3 grid(p.H, p.C / CH), block(256);
  const size_t esz = (p.dtype == kF32 || p.dtype == kS16) ? 4 : (p.dtype == kI8 ? 1 : 2);
  const size_t smem = (size_t)3 * p.W * CH * esz;
  if (smem > 64 * 1024) return hipErrorInvalidValue;
  if (grid_out) *grid_out = grid;
  if (block_out) *block_out = block;
  if (p.dtype == kS16) sppf_pool_split_kernel<CH><<<grid, block, smem, stream>>>(p);
  else if (p.dtype == kF32) sppf_pool_kernel<float, CH><<<grid, block, smem, stream>>>(p);
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
