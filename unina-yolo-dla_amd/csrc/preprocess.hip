// preprocess.hip -- camera-buffer pre-processing on the GPU (gfx950): the step right before the engine in the
// reference's processGpuBuffer (perception_node.cpp:601-604). Same C API and the same arithmetic as
// ros2_ws/src/perception/src/cuda_preprocess.cu:
//   bgra_to_rgb_normalize_kernel         :99-128   u8 BGRA (pitched) -> fp32 RGB planar, ((v/255) - mean)/std
//   resize_bgra_to_rgb_normalize_kernel  :144-204  half-pixel-centre bilinear, clamp to [0, src-1], same normalise
//   nv12_to_rgb_normalize_kernel         :212-253  BT.601 (1.402 / 0.344136 / 0.714136 / 1.772), clamp, normalise
// All three are HBM-bound byte movers (4 B/px in, 12 B/px out): one thread per output pixel, x fastest, so every
// wave reads 256 contiguous input bytes (no-resize paths) and writes 256 contiguous bytes to each output plane.
// Built with -ffp-contract=off so the expression trees round exactly as written (oracle/preprocess_oracle.c).
#include <cstdio>
#include <hip/hip_runtime.h>

#include "../../include/unina_mi355.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ void write_norm(float* out, size_t plane, size_t idx, float r, float g, float b,
                                           const NormParams& p) {
  out[idx] = ((r / 255.0f) - p.mean_r) / p.std_r;
  out[plane + idx] = ((g / 255.0f) - p.mean_g) / p.std_g;
  out[2 * plane + idx] = ((b / 255.0f) - p.mean_b) / p.std_b;
}

__global__ __launch_bounds__(256) void bgra_to_rgb_normalize_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                                    int width, int height, int pitch, NormParams p) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= width || y >= height) return;
  const uchar4 px = *reinterpret_cast<const uchar4*>(in + (size_t)y * pitch + (size_t)x * 4);  // B,G,R,A
  write_norm(out, (size_t)width * height, (size_t)y * width + x, (float)px.z, (float)px.y, (float)px.x, p);
}

__global__ __launch_bounds__(256) void resize_bgra_to_rgb_normalize_kernel(const uint8_t* __restrict__ in,
                                                                           float* __restrict__ out, int sw, int sh,
                                                                           int pitch, int dw, int dh, NormParams p) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y * blockDim.y + threadIdx.y;
  if (dx >= dw || dy >= dh) return;
  const float scale_x = (float)sw / dw, scale_y = (float)sh / dh;
  float sx = (dx + 0.5f) * scale_x - 0.5f, sy = (dy + 0.5f) * scale_y - 0.5f;
  sx = fmaxf(0.0f, fminf(sx, sw - 1.0f));
  sy = fmaxf(0.0f, fminf(sy, sh - 1.0f));
  const int x0 = (int)sx, y0 = (int)sy;
  const int x1 = min(x0 + 1, sw - 1), y1 = min(y0 + 1, sh - 1);
  const float fx = sx - x0, fy = sy - y0;
  const float w00 = (1.0f - fx) * (1.0f - fy), w01 = fx * (1.0f - fy), w10 = (1.0f - fx) * fy, w11 = fx * fy;
  const uchar4 p00 = *reinterpret_cast<const uchar4*>(in + (size_t)y0 * pitch + (size_t)x0 * 4);
  const uchar4 p01 = *reinterpret_cast<const uchar4*>(in + (size_t)y0 * pitch + (size_t)x1 * 4);
  const uchar4 p10 = *reinterpret_cast<const uchar4*>(in + (size_t)y1 * pitch + (size_t)x0 * 4);
  const uchar4 p11 = *reinterpret_cast<const uchar4*>(in + (size_t)y1 * pitch + (size_t)x1 * 4);
  const float r = w00 * p00.z + w01 * p01.z + w10 * p10.z + w11 * p11.z;
  const float g = w00 * p00.y + w01 * p01.y + w10 * p10.y + w11 * p11.y;
  const float b = w00 * p00.x + w01 * p01.x + w10 * p10.x + w11 * p11.x;
  write_norm(out, (size_t)dw * dh, (size_t)dy * dw + dx, r, g, b, p);
}

__global__ __launch_bounds__(256) void nv12_to_rgb_normalize_kernel(const uint8_t* __restrict__ yp,
                                                                    const uint8_t* __restrict__ uvp,
                                                                    float* __restrict__ out, int width, int height,
                                                                    int y_pitch, int uv_pitch, NormParams p) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= width || y >= height) return;
  const float Y = yp[(size_t)y * y_pitch + x];
  const size_t uv = (size_t)(y / 2) * uv_pitch + (size_t)(x / 2) * 2;
  const float U = uvp[uv] - 128.0f, V = uvp[uv + 1] - 128.0f;
  float r = Y + 1.402f * V;
  float g = Y - 0.344136f * U - 0.714136f * V;
  float b = Y + 1.772f * U;
  r = fmaxf(0.0f, fminf(255.0f, r));
  g = fmaxf(0.0f, fminf(255.0f, g));
  b = fmaxf(0.0f, fminf(255.0f, b));
  write_norm(out, (size_t)width * height, (size_t)y * width + x, r, g, b, p);
}

inline dim3 grid_for(int w, int h, dim3 b) { return dim3((w + b.x - 1) / b.x, (h + b.y - 1) / b.y); }

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
  const dim3 block(64, 4);
  resize_bgra_to_rgb_normalize_kernel<<<grid_for(dst_width, dst_height, block), block, 0, stream>>>(
      d_input, d_output, src_width, src_height, src_pitch, dst_width, dst_height, params);
  return hipGetLastError();
}

hipError_t preprocess_bgra(const uint8_t* d_input, float* d_output, int width, int height, int pitch, NormParams params,
                           hipStream_t stream) {
  if (!d_input || !d_output || width <= 0 || height <= 0 || pitch < 4 * width || (pitch & 3)) return hipErrorInvalidValue;
  const dim3 block(64, 4);
  bgra_to_rgb_normalize_kernel<<<grid_for(width, height, block), block, 0, stream>>>(d_input, d_output, width, height, pitch, params);
  return hipGetLastError();
}

hipError_t preprocess_nv12(const uint8_t* d_y_plane, const uint8_t* d_uv_plane, float* d_output, int width, int height,
                           int y_pitch, int uv_pitch, NormParams params, hipStream_t stream) {
  if (!d_y_plane || !d_uv_plane || !d_output || width <= 0 || height <= 0 || y_pitch < width || uv_pitch < width)
    return hipErrorInvalidValue;
  const dim3 block(64, 4);
  nv12_to_rgb_normalize_kernel<<<grid_for(width, height, block), block, 0, stream>>>(d_y_plane, d_uv_plane, d_output, width,
                                                                                   height, y_pitch, uv_pitch, params);
  return hipGetLastError();
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
