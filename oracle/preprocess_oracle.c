/*
 * preprocess_oracle.c -- CPU ORACLE for the camera pre-process (test infrastructure, NOT product code).
 * Scalar restatement of ros2_ws/src/perception/src/cuda_preprocess.cu:
 *   bgra_to_rgb_normalize_kernel :99-128, resize_bgra_to_rgb_normalize_kernel :144-204,
 *   nv12_to_rgb_normalize_kernel :212-253.
 * The reference kernels need nvcc (absent) and the reference holds no vector for them: parity unpinned beyond
 * this line-by-line restatement. Built with -ffp-contract=off (each operation rounds once, as written).
 */
#include <math.h>
#include <stdint.h>

typedef struct { float mean_r, mean_g, mean_b, std_r, std_g, std_b; } uo_norm;

static void write_norm(float *out, long plane, long idx, float r, float g, float b, const uo_norm *p) {
  out[idx] = ((r / 255.0f) - p->mean_r) / p->std_r;
  out[plane + idx] = ((g / 255.0f) - p->mean_g) / p->std_g;
  out[2 * plane + idx] = ((b / 255.0f) - p->mean_b) / p->std_b;
}

void uo_preprocess_bgra(const uint8_t *in, float *out, int width, int height, int pitch, const uo_norm *p) {
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const uint8_t *px = in + (long)y * pitch + (long)x * 4;
      write_norm(out, (long)width * height, (long)y * width + x, px[2], px[1], px[0], p);
    }
}

void uo_preprocess_bgra_resize(const uint8_t *in, float *out, int sw, int sh, int pitch, int dw, int dh, const uo_norm *p) {
  for (int dy = 0; dy < dh; ++dy)
    for (int dx = 0; dx < dw; ++dx) {
      float scale_x = (float)sw / dw, scale_y = (float)sh / dh;
      float sx = (dx + 0.5f) * scale_x - 0.5f, sy = (dy + 0.5f) * scale_y - 0.5f;
      sx = fmaxf(0.0f, fminf(sx, sw - 1.0f));
      sy = fmaxf(0.0f, fminf(sy, sh - 1.0f));
      int x0 = (int)sx, y0 = (int)sy;
      int x1 = x0 + 1 < sw - 1 ? x0 + 1 : sw - 1, y1 = y0 + 1 < sh - 1 ? y0 + 1 : sh - 1;
      float fx = sx - x0, fy = sy - y0;
      float w00 = (1.0f - fx) * (1.0f - fy), w01 = fx * (1.0f - fy), w10 = (1.0f - fx) * fy, w11 = fx * fy;
      const uint8_t *p00 = in + (long)y0 * pitch + (long)x0 * 4, *p01 = in + (long)y0 * pitch + (long)x1 * 4;
      const uint8_t *p10 = in + (long)y1 * pitch + (long)x0 * 4, *p11 = in + (long)y1 * pitch + (long)x1 * 4;
      float r = w00 * p00[2] + w01 * p01[2] + w10 * p10[2] + w11 * p11[2];
      float g = w00 * p00[1] + w01 * p01[1] + w10 * p10[1] + w11 * p11[1];
      float b = w00 * p00[0] + w01 * p01[0] + w10 * p10[0] + w11 * p11[0];
      write_norm(out, (long)dw * dh, (long)dy * dw + dx, r, g, b, p);
    }
}

void uo_preprocess_nv12(const uint8_t *yp, const uint8_t *uvp, float *out, int width, int height, int y_pitch,
                        int uv_pitch, const uo_norm *p) {
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      float Y = yp[(long)y * y_pitch + x];
      long uv = (long)(y / 2) * uv_pitch + (long)(x / 2) * 2;
      float U = uvp[uv] - 128.0f, V = uvp[uv + 1] - 128.0f;
      float r = Y + 1.402f * V;
      float g = Y - 0.344136f * U - 0.714136f * V;
      float b = Y + 1.772f * U;
      r = fmaxf(0.0f, fminf(255.0f, r));
      g = fmaxf(0.0f, fminf(255.0f, g));
      b = fmaxf(0.0f, fminf(255.0f, b));
      write_norm(out, (long)width * height, (long)y * width + x, r, g, b, p);
    }
}
