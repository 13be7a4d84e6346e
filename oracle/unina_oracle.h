/*
 * unina_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C fp32 restatement of the reference's algorithm for the hot path:
 *   - forward graph:  /root/reference/unina_yolo_dla/model.py:23-365  (graph (A));
 *                     /root/reference/unina_yolo_dla/qat.py:225-491     (graph (B), the QAT model, in float)
 *   - decode + NMS:   ros2_ws/src/perception/include/postprocess.hpp:28-145 (CPU twin)
 *                     ros2_ws/src/perception/src/gpu_postprocess.cu:62-83,102-251 (GPU deltas)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library. The shipped engine (libunina_mi355.so) never links or calls it.
 *
 * Parity pin: tests/test_oracle_golden.py checks this code against tensors produced by
 * importing the reference model.py (tests/golden/make_golden.py) and against the
 * reference postprocess.hpp compiled into oracle/_ref (oracle/Makefile).
 */
#ifndef UNINA_ORACLE_H
#define UNINA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- state dict (UNSD container, see unina-yolo-dla_amd/statedict.py) ---- */
typedef struct uo_statedict uo_statedict;
uo_statedict *uo_sd_load(const char *path);
void uo_sd_free(uo_statedict *sd);
int uo_sd_count(const uo_statedict *sd);
/* returns NULL if absent; dims[] gets up to 4 entries */
const float *uo_sd_get(const uo_statedict *sd, const char *name, int *ndim, int dims[4]);

/* ---- forward (model.py:347-365) ---- */
typedef struct uo_run uo_run;
/* x: [3,H,W] fp32 (batch 1). keep_all!=0 keeps every module output by name
 * (e.g. "backbone.stem", "neck.up1", "backbone.sppf.pool2"); otherwise only the
 * six heads "p2_cls".."p4_reg" are kept. nthreads<=0: OpenMP default. */
uo_run *uo_forward(const uo_statedict *sd, const float *x, int H, int W, int num_classes,
                   int base_channels, int lite_p2, int keep_all, int nthreads);
/* Graph (B): UNINA_YOLO_DLA_QAT.forward (qat.py:443-491) in float (the reference's own behaviour without
 * pytorch-quantization); module names as in qat.py ("stem", "stage1_c3k2.cv1", "head_p2_cls.0", ...), synthetic
 * names "up_p5/up_p4/up_p3", "cat_fpn1..3", "cat_pan1..2", "<sppf>.pool1..3". H, W multiples of 32. */
uo_run *uo_forward_qat(const uo_statedict *sd, const float *x, int H, int W, int num_classes,
                       int base_channels, int keep_all, int nthreads);
const float *uo_run_get(const uo_run *r, const char *name, int *c, int *h, int *w);
int uo_run_count(const uo_run *r);
const char *uo_run_name(const uo_run *r, int i);
const char *uo_last_error(void);
void uo_run_free(uo_run *r);

/* ---- post-process ---- */
/* Same 32-byte record as the reference's GpuDetection (gpu_postprocess.h:27-33). */
typedef struct {
  float x1, y1, x2, y2;
  float confidence;
  int class_id;
  int valid;
  int _pad;
} uo_det;

/* The reference's CPU header and GPU file disagree on three details;
 * the oracle exposes them so BOTH can be reproduced:
 *   ge_threshold : keep cell if conf >= thr (gpu_postprocess.cu:132) else conf > thr (postprocess.hpp:116)
 *   iou_eps      : added to the IoU denominator (1e-6f in gpu_postprocess.cu:82, 0 in postprocess.hpp:38)
 *   strict_conf  : box i may only suppress j if conf_i > conf_j (gpu_postprocess.cu:224)
 *   max_det      : cap (MAX_DETECTIONS = 1024, gpu_postprocess.cu:25); <=0 = no cap */
typedef struct {
  int ge_threshold;
  float iou_eps;
  int strict_conf;
  int max_det;
} uo_pp_semantics;

/* The semantics the engine implements: GPU-file thresholds, CPU-header greedy order. */
uo_pp_semantics uo_semantics_engine(void);
/* The semantics of postprocess.hpp exactly. */
uo_pp_semantics uo_semantics_cpu_header(void);

float uo_sigmoid(float x);
float uo_iou(const uo_det *a, const uo_det *b, float eps);

/* Row-major decode of one head, appending to out[*n...] (no cap; caller sizes out for w*h more). */
void uo_decode_head(const float *cls, const float *reg, int w, int h, int stride, int num_classes,
                    float conf_thr, float conformal_q, const uo_pp_semantics *sem, uo_det *out, int *n);

/* Stable sort by confidence desc (ties keep enumeration order), cap, greedy class-aware NMS,
 * compaction. dets[0..n) is reordered in place; kept records are written to out (valid=1,_pad=0).
 * Returns the number kept. */
int uo_sort_nms(uo_det *dets, int n, float iou_thr, const uo_pp_semantics *sem, uo_det *out);

/* Whole post-process: heads in order P2,P3,P4 (perception_node.cpp:630-640). heads[2*i]=cls, heads[2*i+1]=reg. */
int uo_postprocess(const float *const heads[6], const int grid_w[3], const int grid_h[3],
                   const int strides[3], int num_classes, float conf_thr, float iou_thr,
                   float conformal_q, const uo_pp_semantics *sem, uo_det *out, int *n_candidates);

#ifdef __cplusplus
}
#endif
#endif
