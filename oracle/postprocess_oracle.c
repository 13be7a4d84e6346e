/*
 * postprocess_oracle.c -- CPU ORACLE, decode + NMS (test infrastructure, NOT product code).
 *
 * Restates the reference post-process with its two variants made explicit:
 *   CPU twin   ros2_ws/src/perception/include/postprocess.hpp
 *                compute_iou :28-39, nms :44-67, apply_conformal_prediction :77-85, decode_head :94-145
 *   GPU file   ros2_ws/src/perception/src/gpu_postprocess.cu
 *                sigmoid :62-64, compute_iou_gpu :69-83, decode_yolo_head_kernel :102-199,
 *                nms_kernel :207-231, ConfidenceComparator :236-242, MAX_DETECTIONS :25
 *
 * The engine implements SURVEY.md App. D: GPU-file thresholds (>=, +1e-6f, strict-confidence
 * suppression, cap 1024) with the CPU header's deterministic order (row-major enumeration
 * P2->P3->P4, stable sort, sequential greedy NMS). uo_semantics_cpu_header() reproduces
 * postprocess.hpp exactly and is pinned against the compiled header in oracle/_ref.
 */
#include "unina_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

uo_pp_semantics uo_semantics_engine(void) {
  uo_pp_semantics s = {1, 1e-6f, 1, 1024};
  return s;
}
uo_pp_semantics uo_semantics_cpu_header(void) {
  uo_pp_semantics s = {0, 0.0f, 0, 0};
  return s;
}

/* gpu_postprocess.cu:62-64 / postprocess.hpp:108-112 */
float uo_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

/* gpu_postprocess.cu:69-83 (eps=1e-6f) / postprocess.hpp:28-39 (eps=0) */
float uo_iou(const uo_det *a, const uo_det *b, float eps) {
  float ix1 = fmaxf(a->x1, b->x1), iy1 = fmaxf(a->y1, b->y1);
  float ix2 = fminf(a->x2, b->x2), iy2 = fminf(a->y2, b->y2);
  if (ix1 >= ix2 || iy1 >= iy2) return 0.0f;
  float inter = (ix2 - ix1) * (iy2 - iy1);
  float area_a = (a->x2 - a->x1) * (a->y2 - a->y1);
  float area_b = (b->x2 - b->x1) * (b->y2 - b->y1);
  if (eps != 0.0f) return inter / (area_a + area_b - inter + eps);
  return inter / (area_a + area_b - inter);
}

/* decode_head (postprocess.hpp:94-145) / decode_yolo_head_kernel (gpu_postprocess.cu:102-199) */
void uo_decode_head(const float *cls, const float *reg, int w, int h, int stride, int num_classes, float conf_thr,
                    float q, const uo_pp_semantics *sem, uo_det *out, int *n) {
  const int hw = w * h;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const int idx = y * w + x;
      int best = -1;
      float max_conf = 0.0f; /* first max wins ties; starts from (0.0,-1) */
      for (int c = 0; c < num_classes; ++c) {
        float conf = uo_sigmoid(cls[c * hw + idx]);
        if (conf > max_conf) {
          max_conf = conf;
          best = c;
        }
      }
      int keep = sem->ge_threshold ? (max_conf >= conf_thr) : (max_conf > conf_thr);
      if (!keep) continue;
      const float fs = (float)stride;
      float xc = ((float)x + 0.5f) * fs, yc = ((float)y + 0.5f) * fs;
      float l = reg[0 * hw + idx] * fs, t = reg[1 * hw + idx] * fs;
      float r = reg[2 * hw + idx] * fs, b = reg[3 * hw + idx] * fs;
      uo_det d;
      d.x1 = xc - l;
      d.y1 = yc - t;
      d.x2 = xc + r;
      d.y2 = yc + b;
      if (q > 0.0f) { /* dilation uses the pre-dilation w,h for all four edges */
        float bw = d.x2 - d.x1, bh = d.y2 - d.y1;
        float dw = bw * q, dh = bh * q;
        d.x1 -= dw;
        d.y1 -= dh;
        d.x2 += dw;
        d.y2 += dh;
      }
      d.confidence = max_conf;
      d.class_id = best;
      d.valid = 1;
      d._pad = 0;
      out[(*n)++] = d;
    }
}

/* stable merge sort by confidence descending (ties keep enumeration order) */
static void merge_sort(uo_det *a, uo_det *tmp, int n) {
  if (n < 2) return;
  int m = n / 2;
  merge_sort(a, tmp, m);
  merge_sort(a + m, tmp, n - m);
  int i = 0, j = m, k = 0;
  while (i < m && j < n) tmp[k++] = (a[j].confidence > a[i].confidence) ? a[j++] : a[i++];
  while (i < m) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, sizeof(uo_det) * (size_t)n);
}

int uo_sort_nms(uo_det *dets, int n, float iou_thr, const uo_pp_semantics *sem, uo_det *out) {
  if (n <= 0) return 0;
  uo_det *tmp = malloc(sizeof(uo_det) * (size_t)n);
  merge_sort(dets, tmp, n);
  free(tmp);
  if (sem->max_det > 0 && n > sem->max_det) n = sem->max_det; /* keep the max_det best (ties: enumeration order) */
  unsigned char *sup = calloc((size_t)n, 1);
  int kept = 0;
  /* sequential greedy, class-aware, strict IoU > thr (postprocess.hpp:55-65) */
  for (int i = 0; i < n; ++i) {
    if (sup[i]) continue;
    out[kept] = dets[i];
    out[kept].valid = 1;
    out[kept]._pad = 0;
    ++kept;
    for (int j = i + 1; j < n; ++j) {
      if (sup[j] || dets[i].class_id != dets[j].class_id) continue;
      if (sem->strict_conf && !(dets[i].confidence > dets[j].confidence)) continue; /* gpu_postprocess.cu:224 */
      if (uo_iou(&dets[i], &dets[j], sem->iou_eps) > iou_thr) sup[j] = 1;
    }
  }
  free(sup);
  return kept;
}

int uo_postprocess(const float *const heads[6], const int grid_w[3], const int grid_h[3], const int strides[3],
                   int num_classes, float conf_thr, float iou_thr, float q, const uo_pp_semantics *sem, uo_det *out,
                   int *n_candidates) {
  int total = 0;
  for (int i = 0; i < 3; ++i) total += grid_w[i] * grid_h[i];
  uo_det *cand = malloc(sizeof(uo_det) * (size_t)(total > 0 ? total : 1));
  int n = 0;
  for (int i = 0; i < 3; ++i) /* node call order P2,P3,P4: perception_node.cpp:630-640 */
    uo_decode_head(heads[2 * i], heads[2 * i + 1], grid_w[i], grid_h[i], strides[i], num_classes, conf_thr, q, sem,
                   cand, &n);
  if (n_candidates) *n_candidates = n;
  uo_det *kept = malloc(sizeof(uo_det) * (size_t)(n > 0 ? n : 1));
  int k = uo_sort_nms(cand, n, iou_thr, sem, kept);
  memcpy(out, kept, sizeof(uo_det) * (size_t)k);
  free(kept);
  free(cand);
  return k;
}
