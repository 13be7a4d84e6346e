// preprocess.hip -- camera-buffer pre-processing on the GPU (gfx950): the step right before the engine in the
// reference's processGpuBuffer (perception_node.cpp:601-604), behind the reference's C API names. The ARITHMETIC per
// pixel is that of ros2_ws/src/perception/src/cuda_preprocess.cu (it has to be: the results are compared bit for bit
// with oracle/preprocess_oracle.c, and unina_infer_bgra's in-stem form must equal the two-step form):
//   plain BGRA   :99-128   u8 BGRA (pitched) -> fp32 RGB planar, ((v/255) - mean)/std
//   BGRA resize  :144-204  half-pixel-centre bilinear, clamp to [0, src-1], same normalise
//   NV12         :212-253  BT.601 (1.402 / 0.344136 / 0.714136 / 1.772), clamp, normalise
// The DATA MOVEMENT is not the reference's one-thread-per-pixel form: all three are HBM-bound byte movers (4 B/px in,
// 12 B/px out), so ONE kernel template, thread = FOUR consecutive output pixels of a row: the no-resize paths read the
// quad with one 16-byte load (BGRA) or one dword of luma + one dword of chroma (NV12: 2 chroma pairs for 4 pixels), every
// path writes one 16-byte store per output plane (a wave: 1 KiB contiguous per plane per instruction instead of 256 B),
// and a flat grid-stride loop over the quads replaces the 2-D grid. Row tails (width % 4) and unaligned tensors fall back to
// scalar accesses inside the same kernel. Built with -ffp-contract=off so the expression trees round exactly as written.
#include <cstdio>
#include <hip/hip_runtime.h>

#include "../../include/unina_mi355.h"

#pragma clang fp contract(off)

namespace {

enum Mode : int { kPlain = 0, kResize = 1, kNv12 = 2 };

struct PreParams {
  const uint8_t* in;      // BGRA (plain / resize) or the Y plane (NV12)
  const uint8_t* uv;      // NV12: interleaved chroma plane
  float* out;             // [3][dh][dw]
  int sw, sh, pitch;      // source size / pitch (NV12: luma pitch)
  int uv_pitch;
  int dw, dh;             // output size (== source size except for kResize)
  NormParams norm;
};

__device__ __forceinline__ void normalise(float r, float g, float b, const NormParams& p, float (&o)[3]) {
  o[0] = ((r / 255.0f) - p.mean_r) / p.std_r;
  o[1] = ((g / 255.0f) - p.mean_g) / p.std_g;
  o[2] = ((b / 255.0f) - p.mean_b) / p.std_b;
}

// one output pixel of the bilinear resize (cuda_preprocess.cu:155-198)
__device__ __forceinline__ void resize_pixel(const PreParams& q, int dx, int dy, float (&o)[3]) {
  const float scale_x = (float)q.sw / q.dw, scale_y = (float)q.sh / q.dh;
  float sx = (dx + 0.5f) * scale_x - 0.5f, sy = (dy + 0.5f) * scale_y - 0.5f;
  sx = fmaxf(0.0f, fminf(sx, q.sw - 1.0f));
  sy = fmaxf(0.0f, fminf(sy, q.sh - 1.0f));
  const int x0 = (int)sx, y0 = (int)sy;
  const int x1 = min(x0 + 1, q.sw - 1), y1 = min(y0 + 1, q.sh - 1);
  const float fx = sx - x0, fy = sy - y0;
  const float w00 = (1.0f - fx) * (1.0f - fy), w01 = fx * (1.0f - fy), w10 = (1.0f - fx) * fy, w11 = fx * fy;
  const uchar4 p00 = *reinterpret_cast<const uchar4*>(q.in + (size_t)y0 * q.pitch + (size_t)x0 * 4);
  const uchar4 p01 = *reinterpret_cast<const uchar4*>(q.in + (size_t)y0 * q.pitch + (size_t)x1 * 4);
  const uchar4 p10 = *reinterpret_cast<const uchar4*>(q.in + (size_t)y1 * q.pitch + (size_t)x0 * 4);
  const uchar4 p11 = *reinterpret_cast<const uchar4*>(q.in + (size_t)y1 * q.pitch + (size_t)x1 * 4);
  const float r = w00 * p00.z + w01 * p01.z + w10 * p10.z + w11 * p11.z;
  const float g = w00 * p00.y + w01 * p01.y + w10 * p10.y + w11 * p11.y;
  const float b = w00 * p00.x + w01 * p01.x + w10 * p10.x + w11 * p11.x;
  normalise(r, g, b, q.norm, o);
}

// BT.601 (cuda_preprocess.cu:229-247)
__device__ __forceinline__ void nv12_pixel(float Y, float U, float V, const NormParams& n, float (&o)[3]) {
  float r = Y + 1.402f * V;
  float g = Y - 0.344136f * U - 0.714136f * V;
  float b = Y + 1.772f * U;
  r = fmaxf(0.0f, fminf(255.0f, r));
  g = fmaxf(0.0f, fminf(255.0f, g));
  b = fmaxf(0.0f, fminf(255.0f, b));
  normalise(r, g, b, n, o);
}

template <int MODE>
__global__ __launch_bounds__(256) void preprocess_quads_kernel(const PreParams q) {
  const int qpr = (q.dw + 3) >> 2;                         // quads per output row
  const long long nquads = (long long)qpr * q.dh;
  const size_t plane = (size_t)q.dw * q.dh;
  // wide accesses only where they are aligned: every row of the output starts 16-byte aligned iff dw % 4 == 0 and the
  // tensor does; every row of the source iff the pitch and the base allow it
  const bool wide_out = (q.dw & 3) == 0 && ((uintptr_t)q.out & 15) == 0;
  const bool wide_in = MODE == kPlain ? ((q.pitch & 15) == 0 && ((uintptr_t)q.in & 15) == 0)
                                      : ((q.pitch & 3) == 0 && (q.uv_pitch & 3) == 0 && ((uintptr_t)q.in & 3) == 0 && ((uintptr_t)q.uv & 3) == 0);
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nquads; t += (long long)gridDim.x * blockDim.x) {
    const int y = (int)(t / qpr), x = (int)(t - (long long)y * qpr) * 4;
    const int n = q.dw - x < 4 ? q.dw - x : 4;             // pixels of this quad inside the row
    float o[4][3];
    if constexpr (MODE == kPlain) {
      uchar4 px[4];
      if (wide_in && n == 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(q.in + (size_t)y * q.pitch + (size_t)x * 4);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) px[i] = make_uchar4(w[i] & 255u, (w[i] >> 8) & 255u, (w[i] >> 16) & 255u, w[i] >> 24);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          px[i] = i < n ? *reinterpret_cast<const uchar4*>(q.in + (size_t)y * q.pitch + (size_t)(x + i) * 4) : make_uchar4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) normalise((float)px[i].z, (float)px[i].y, (float)px[i].x, q.norm, o[i]);   // B,G,R,A in memory
    } else if constexpr (MODE == kResize) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < n) resize_pixel(q, x + i, y, o[i]);
    } else {
      unsigned char yy[4], uu[4];
      const uint8_t* yrow = q.in + (size_t)y * q.pitch + x;
      const uint8_t* crow = q.uv + (size_t)(y / 2) * q.uv_pitch + x;     // x is even: pairs (U,V) of pixels x, x+1 | x+2, x+3
      if (wide_in && n == 4) {
        const unsigned yw = *reinterpret_cast<const unsigned*>(yrow), cw = *reinterpret_cast<const unsigned*>(crow);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          yy[i] = (unsigned char)(yw >> (8 * i));
          uu[i] = (unsigned char)(cw >> (8 * i));
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          yy[i] = i < n ? yrow[i] : 0;
          uu[i] = (i & ~1) < n ? crow[i] : 0;               // (a pair is read whenever its first pixel is inside the row)
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) nv12_pixel((float)yy[i], uu[(i & ~1)] - 128.0f, uu[(i & ~1) + 1] - 128.0f, q.norm, o[i]);
    }
    const size_t idx = (size_t)y * q.dw + x;
    if (wide_out) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        *reinterpret_cast<float4*>(q.out + c * plane + idx) = make_float4(o[0][c], o[1][c], o[2][c], o[3][c]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < n) {
#pragma unroll
          for (int c = 0; c < 3; ++c) q.out[c * plane + idx + i] = o[i][c];
        }
    }
  }
}

template <int MODE>
hipError_t launch_quads(const PreParams& q, hipStream_t stream) {
  const long long nquads = (long long)((q.dw + 3) / 4) * q.dh;
  long long blocks = (nquads + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;                 // grid-stride beyond 16 workgroups per CU
  preprocess_quads_kernel<MODE><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(q);
  return hipGetLastError();
}

}  // namespace

extern "C" {

NormParams create_norm_params_imagenet(void) {  // cuda_preprocess.cu:262 (defaults :73-75)
  NormParams p = {0.485f, 0.456f, 0.406f, 0.229f, 0.224f, 0.225f};
  return p;
}

NormParams create_norm_params(float mean_r, float mean_g, float mean_b, float std_r, float std_g, float std_b) {
  NormParams p = {mean_r, mean_g, mean_b, std_r, std_g, std_b};
  return p;
}

hipError_t preprocess_bgra_resize(const uint8_t* d_input, float* d_output, int src_width, int src_height, int src_pitch,
                                  int dst_width, int dst_height, NormParams params, hipStream_t stream) {
  if (!d_input || !d_output || src_width <= 0 || src_height <= 0 || dst_width <= 0 || dst_height <= 0 || src_pitch < 4 * src_width)
    return hipErrorInvalidValue;
  PreParams q = {d_input, nullptr, d_output, src_width, src_height, src_pitch, 0, dst_width, dst_height, params};
  return launch_quads<kResize>(q, stream);
}

hipError_t preprocess_bgra(const uint8_t* d_input, float* d_output, int width, int height, int pitch, NormParams params,
                           hipStream_t stream) {
  if (!d_input || !d_output || width <= 0 || height <= 0 || pitch < 4 * width || (pitch & 3)) return hipErrorInvalidValue;
  PreParams q = {d_input, nullptr, d_output, width, height, pitch, 0, width, height, params};
  return launch_quads<kPlain>(q, stream);
}

hipError_t preprocess_nv12(const uint8_t* d_y_plane, const uint8_t* d_uv_plane, float* d_output, int width, int height,
                           int y_pitch, int uv_pitch, NormParams params, hipStream_t stream) {
  if (!d_y_plane || !d_uv_plane || !d_output || width <= 0 || height <= 0 || y_pitch < width || uv_pitch < width)
    return hipErrorInvalidValue;
  PreParams q = {d_y_plane, d_uv_plane, d_output, width, height, y_pitch, uv_pitch, width, height, params};
  return launch_quads<kNv12>(q, stream);
}

float* allocate_preprocess_buffer(int width, int height) {  // nullptr on failure (cuda_preprocess.cu:395-405)
  float* d = nullptr;
  if (width <= 0 || height <= 0) return nullptr;
  hipError_t err = hipMalloc(&d, (size_t)3 * width * height * sizeof(float));
  if (err != hipSuccess) {
    fprintf(stderr, "Failed to allocate preprocess buffer: %s\n", hipGetErrorString(err));
    return nullptr;
  }
  return d;
}

void free_preprocess_buffer(float* d_buffer) {
  if (d_buffer) (void)hipFree(d_buffer);
}

hipStream_t create_preprocess_stream(void) {  // nullptr on failure (cuda_preprocess.cu:419-428)
  hipStream_t s = nullptr;
  hipError_t err = hipStreamCreate(&s);
  if (err != hipSuccess) {
    fprintf(stderr, "Failed to create HIP stream: %s\n", hipGetErrorString(err));
    return nullptr;
  }
  return s;
}

void destroy_preprocess_stream(hipStream_t stream) {
  if (stream) (void)hipStreamDestroy(stream);
}

}  // extern "C"
