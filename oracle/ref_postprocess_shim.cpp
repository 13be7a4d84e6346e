// ref_postprocess_shim.cpp -- builds oracle/_ref/libref_postprocess.so from the REFERENCE header
// where it lies (/root/reference/.../include/postprocess.hpp, included via -I, never copied).
// Only a C-ABI doorway is added here so tests can drive the reference's own decode_head()/nms().
// Built only in the dev container (the reference tree does not exist on the GPU box).
#include "postprocess.hpp"

#include <cstring>

extern "C" {
struct ref_det { float x1, y1, x2, y2, confidence; int class_id; };

// decode the three heads in node order (P2,P3,P4) then nms(); returns count, fills out[cap].
int ref_postprocess(const float* const heads[6], const int grid_w[3], const int grid_h[3], const int strides[3],
                    int num_classes, float conf_thr, float iou_thr, float q, ref_det* out, int cap,
                    int* n_candidates) {
  std::vector<Detection> dets;
  for (int i = 0; i < 3; ++i)
    decode_head(heads[2 * i], heads[2 * i + 1], grid_w[i], grid_h[i], strides[i], num_classes, conf_thr, q, dets);
  if (n_candidates) *n_candidates = static_cast<int>(dets.size());
  std::vector<Detection> kept = nms(dets, iou_thr);
  int n = static_cast<int>(kept.size()) < cap ? static_cast<int>(kept.size()) : cap;
  for (int i = 0; i < n; ++i) {
    out[i].x1 = kept[i].x1; out[i].y1 = kept[i].y1; out[i].x2 = kept[i].x2; out[i].y2 = kept[i].y2;
    out[i].confidence = kept[i].confidence; out[i].class_id = kept[i].class_id;
  }
  return static_cast<int>(kept.size());
}

float ref_iou(const ref_det* a, const ref_det* b) {
  Detection da{a->x1, a->y1, a->x2, a->y2, a->confidence, a->class_id};
  Detection db{b->x1, b->y1, b->x2, b->y2, b->confidence, b->class_id};
  return compute_iou(da, db);
}
}
