/*
 * unina_mi355.h -- C ABI of libunina_mi355.so, the MI355X (gfx950) engine for the
 * UNINA-YOLO-DLA detector hot path: forward graph + decode/NMS.
 *
 * This is the drop-in boundary. Every entry point names the reference interface it
 * replaces (paths relative to /root/reference/unina_yolo_dla/):
 *
 *   engine object ......... class TensorRTEngine          ros2_ws/src/perception/src/perception_node.cpp:223-351
 *   post-process C API .... extern "C" block              ros2_ws/src/perception/include/gpu_postprocess.h:36-84
 *   detection record ...... struct GpuDetection           ros2_ws/src/perception/include/gpu_postprocess.h:27-33
 *   tensor names .......... "images", "p2_cls".."p4_reg"  perception_node.cpp:612-618, model.py:382-383
 *
 * Conventions: plain C, no exceptions cross the boundary, 0 == success everywhere.
 * Device pointers are ordinary HIP device pointers; streams are hipStream_t.
 * A handle may be used by one thread at a time; distinct handles (e.g. one per GPU,
 * or two on one GPU to keep two frames in flight) are fully independent.
 */
#ifndef UNINA_MI355_H
#define UNINA_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef UNINA_NO_HIP_HEADERS /* for pure-C consumers / ctypes documentation builds */
typedef struct ihipStream_t *hipStream_t;
typedef int hipError_t;
#else
#include <hip/hip_runtime_api.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define MAX_DETECTIONS 1024 /* gpu_postprocess.h:24 */

/* 32-byte record, identical layout to the reference (gpu_postprocess.h:27-33).
 * Coordinates are xyxy in input-tensor pixels. */
#if defined(__cplusplus)
struct alignas(32) GpuDetection {
#else
struct GpuDetection {
#endif
  float x1, y1, x2, y2;
  float confidence;
  int class_id;
  int valid; /* 1 = kept */
  int _pad;
};
typedef struct GpuDetection GpuDetection;

/* ------------------------------------------------------------------ error codes */
enum {
  UNINA_OK = 0,
  UNINA_ERR_IO = 1,          /* engine file unreadable                    */
  UNINA_ERR_FORMAT = 2,      /* bad magic / version / table               */
  UNINA_ERR_HIP = 3,         /* a HIP call failed (see unina_last_error)  */
  UNINA_ERR_ARG = 4,         /* null / misaligned / unknown tensor name   */
  UNINA_ERR_STATE = 5,       /* e.g. "images" not bound before enqueue    */
  UNINA_ERR_UNSUPPORTED = 6  /* layer shape the kernels do not cover      */
};

typedef struct unina_engine unina_engine_t;

/* ------------------------------------------------------------------ engine (TensorRTEngine role) */

/* Replaces TensorRTEngine::load(engine_path, logger, dla_core) (perception_node.cpp:228-259).
 * `path` is an engine file written by unina-yolo-dla_amd/export.py (folded weights + op table);
 * `device_id` plays the role of dla_core: which accelerator the handle lives on. */
int unina_load_engine(const char *path, int device_id, unina_engine_t **out);

/* Replaces TensorRTEngine::unload() / the destructor (perception_node.cpp:226,261-266). NULL is a no-op. */
void unina_unload_engine(unina_engine_t *e);

/* Replaces TensorRTEngine::getInputDimensions(w,h) (perception_node.cpp:297-325); also reports num_classes,
 * which the node hard-codes to 4 (perception_node.cpp:630-639). Any out pointer may be NULL. */
int unina_engine_input_dims(const unina_engine_t *e, int *width, int *height, int *num_classes);

/* Replaces setInputTensorAddress / setOutputTensorAddress (perception_node.cpp:268-276).
 * Names: "images" fp32 [1,3,H,W]; "p2_cls","p2_reg","p3_cls","p3_reg","p4_cls","p4_reg" fp32 planar
 * [1,C,H/s,W/s], s = 4/8/16. Pointers must be 16-byte aligned device pointers owned by the caller.
 * Outputs left unbound are written to engine-owned buffers (see unina_tensor_address). */
int unina_set_tensor_address(unina_engine_t *e, const char *name, void *device_ptr);

/* Current device address + element count of a named tensor (engine-owned default or the bound one). */
int unina_tensor_address(const unina_engine_t *e, const char *name, void **device_ptr, size_t *num_floats);

/* Replaces TensorRTEngine::enqueueV3(stream) (perception_node.cpp:278-282, call site :621):
 * runs the forward graph asynchronously on `stream`, producing the six raw head tensors. */
int unina_enqueue(unina_engine_t *e, hipStream_t stream);

/* The fused path the north star names: forward + sigmoid/argmax/threshold + TLBR decode + conformal dilation
 * + sort + class-aware NMS + compaction, all on the GPU, no host round-trip in between. It replaces
 * perception_node.cpp:612-656 (bind, enqueueV3, reset_detection_counter, 3 x decode_yolo_head,
 * get_detection_count, run_gpu_nms, copy_valid_detections_to_host).
 *
 *   d_images_nchw : device, fp32 [1,3,H,W] (same as binding "images"); NULL = keep the current binding
 *   out           : HOST buffer for up to MAX_DETECTIONS records (sorted by confidence, valid=1)
 *   out_count     : HOST int
 * Synchronous: returns after the records are in `out`. The post-process writes the count and the kept records straight
 * into a pinned host block and then a completion word (system-scope release); the call spins on that word, so the
 * latency path has neither a D2H copy command nor a stream synchronisation. `stream` may still hold the tail of the
 * frame's launch for a few microseconds after the call returns (work submitted to it later is ordered as usual).
 * UNINA_HOST_RESULT=0 / UNINA_HOST_POLL=0 restore the device buffer + copy / hipStreamSynchronize.
 *
 * NOTE on the six head tensors ("p2_cls" ... "p4_reg"): unina_infer / unina_infer_async / unina_infer_bgra compute the heads'
 * output convs INSIDE the decode launch and do NOT write the fp32 planes of those heads (P3 / P4 always; P2 too unless its
 * head runs as one fused kernel). After these calls unina_tensor_address / unina_debug_read_buffer of such a plane return
 * whatever an earlier unina_enqueue left there (or zeros). A caller that wants the raw heads (the TensorRT-shaped path of
 * perception_node.cpp:612-640) calls unina_enqueue, which always writes all six; UNINA_POST_FOLD=0 makes the frame path
 * write them too. */
int unina_infer(unina_engine_t *e, const float *d_images_nchw, float conf_threshold, float iou_threshold,
                float conformal_q, GpuDetection *out, int *out_count, hipStream_t stream);

/* Normalisation constants of the pre-process (pre-process C API below; also taken by unina_infer_bgra). */
typedef struct {
  float mean_r, mean_g, mean_b;
  float std_r, std_g, std_b;
} NormParams; /* cuda_preprocess.h:38-45 */

/* `n_calls` serial unina_infer calls over a ring of `n_ring` device frames, each timed on the host's steady clock from entry to
 * return (records copied out): the latency a C / C++ caller sees, free of a binding layer's per-call cost. lat_us[n_calls]. */
int unina_serial_latency(unina_engine_t *e, const float *const *d_frames, int n_ring, int n_calls, float conf_threshold,
                         float iou_threshold, float conformal_q, double *lat_us, hipStream_t stream);

/* Camera frame -> detections in one call: unina_infer with the pre-process of cuda_preprocess.h:50-112
 * (preprocess_bgra when the frame has the network's size, preprocess_bgra_resize otherwise) computed inside the
 * stem kernel, i.e. perception_node.cpp:601-656 (preprocess_bgra_resize ... copy_valid_detections_to_host) as ONE
 * graph launch without the fp32 tensor in between. Identical results to the two-step form.
 *   d_bgra : device, pitched BGRA8 (src_pitch bytes per row, multiple of 4, >= 4 * src_width) */
int unina_infer_bgra(unina_engine_t *e, const uint8_t *d_bgra, int src_width, int src_height, int src_pitch,
                     const NormParams *norm, float conf_threshold, float iou_threshold, float conformal_q,
                     GpuDetection *out, int *out_count, hipStream_t stream);

/* Asynchronous variant: results stay on the device (d_out: MAX_DETECTIONS records, d_out_count: one int);
 * nothing is synchronised. Used to pipeline frames and to feed the RCCL gather without touching the host. */
int unina_infer_async(unina_engine_t *e, const float *d_images_nchw, float conf_threshold, float iou_threshold,
                      float conformal_q, GpuDetection *d_out, int *d_out_count, hipStream_t stream);

/* Decode + NMS only, on head tensors already on the device (the six bound/owned outputs). */
int unina_postprocess_async(unina_engine_t *e, float conf_threshold, float iou_threshold, float conformal_q,
                            GpuDetection *d_out, int *d_out_count, hipStream_t stream);

/* Error text of the last failing call on this handle (never NULL). With e == NULL: last load failure. */
const char *unina_last_error(const unina_engine_t *e);

/* ------------------------------------------------------------------ introspection / measurement */

typedef struct unina_op_info {
  char name[96];      /* reference module path(s), e.g. "head_p3.cls_branch.0+head_p3.reg_branch.0" */
  char kernel[64];    /* device kernel (template instantiation) that executes it */
  int kind;           /* 1 conv, 2 stem, 3 sppf pool, 4 upsample */
  int m, n, k;        /* implicit-GEMM shape (pixels, out channels, taps*in channels); 0 for non-conv */
  double flops;       /* 2*m*n*k */
  double bytes;       /* algorithmic bytes: inputs once + weights once + outputs once */
  int grid, block;    /* launch geometry */
} unina_op_info;

int unina_op_count(const unina_engine_t *e);
int unina_get_op_info(const unina_engine_t *e, int index, unina_op_info *info);

/* Times every op of the forward with HIP events on `stream`, INSIDE the frame sequence: each of the `iters`
 * repetitions replays ops 0..i-1 and then times op i alone, so the op meets the cache state of a real frame (weights
 * not L2-resident). Event-to-event time of a single launch: like rocprofv3's kernel duration it includes the
 * dispatch cost (the two agree within a few percent). ms_per_op[i] = mean milliseconds of op i (0 for ops that run
 * inside a fused block's launch). Used by bench.py's roofline leg. */
int unina_profile_ops(unina_engine_t *e, int iters, float *ms_per_op, hipStream_t stream);

/* The same for the post-process launches of a frame (each repetition replays the whole forward first): ms2[0] = decode
 * launch (including the head output convs folded into it; ops the frame leaves out because of that report 0 in
 * unina_profile_ops), ms2[1] = pair tiles + greedy scan + output launch (0 if the post-process is one launch). */
int unina_profile_post(unina_engine_t *e, int iters, float conf_threshold, float iou_threshold, float conformal_q, float *ms2,
                       hipStream_t stream);

/* Tile-configuration control of the implicit-GEMM conv kernel (autotuning, tests). cfg = -1 restores the heuristic.
 * Returns UNINA_ERR_UNSUPPORTED if the configuration does not fit the op's shape. */
int unina_conv_config_count(void);
const char *unina_conv_config_name(int cfg);
int unina_set_op_config(unina_engine_t *e, int op_index, int cfg);
/* Times every fitting configuration of every conv op (`iters` launches each, in the frame sequence like
 * unina_profile_ops, HIP events on `stream`) and keeps the
 * fastest -- the engine-build "tactic selection" the reference leaves to TensorRT (export_trt.py:459-468).
 * Needs "images" bound. Results are bit-identical under every configuration. */
int unina_autotune(unina_engine_t *e, int iters, hipStream_t stream);

/* Block fusion (fp16 engines). At load the engine recognises every C3k2 block of the op table (the reference's
 * model.py:76-110: cv1|cv2 -> n x Bottleneck -> cv3) and, while fusion is on (the default; UNINA_FUSE=0 in the
 * environment starts with it off), runs each as ONE launch that keeps the block's intermediates in LDS
 * (csrc/c3k2_fused.hip) -- the role of TensorRT's layer fusion in the reference's engine build. Results are
 * bit-identical either way; with fusion on the blocks' internal buffers are not written (unina_debug_read_buffer of
 * e.g. "neck.pan_c3k2_1.cat" then returns stale data), so per-layer checks and calibration switch it off.
 * unina_fusion_groups: number of blocks currently running fused (0 when off / none recognised). */
int unina_set_fusion(unina_engine_t *e, int enable);
int unina_fusion_groups(const unina_engine_t *e);
/* Load-time analysis alone, no device needed: number of C3k2 blocks of the engine file that would run fused
 * (negative = -error code). */
int unina_debug_fusable_groups(const char *path);

/* Copies an internal activation buffer to the host as fp32 NCHW ([C,H,W]) -- parity tests only.
 * `name` is a buffer name from the engine file (e.g. "p3_fused"); returns UNINA_ERR_ARG if unknown. */
int unina_debug_read_buffer(unina_engine_t *e, const char *name, float *host_out, size_t capacity_floats,
                            int *c, int *h, int *w);

/* Debug: wall_clock64 stamps (100 MHz) of the phases of the last unina_infer's post-process launches; only written
 * when the environment has UNINA_POST_STAMPS=1. out16 (16 values): [7] launch 1 starts (workgroup 0), [0] its decode is
 * done, [1] last arriver found, [2] candidates gathered (launch 1 ends), [3] launch 2 starts, [4] its last arriver
 * found (all pair tiles done), [8] ranks / masks loaded, [9] per-class row lists built, [5] greedy scan done, [6] output
 * written. (The older launch forms use [0..6] as decode, hand-off, gather, sort, masks, scan, output.) */
int unina_debug_post_stamps(unina_engine_t *e, long long *out16);

/* Debug: one launch of conv op `op_index` with in-kernel s_memtime stamps of a mid-grid workgroup:
 * out5 = start, prologue issued, first operands usable, K loop done, stores drained (shader-clock ticks). */
int unina_debug_conv_stamps(unina_engine_t *e, int op_index, long long *out5, hipStream_t stream);

/* Debug: the same for the DUAL conv launch led by op `op_index` (two independent convs as one grid, e.g. the P3 | P4 head
 * layers): out16[0..7] = stamps of conv A's mid workgroup, out16[8..15] = conv B's (each: start, loads issued, patch
 * landed, K loop done, stores issued, 2 x 100 MHz reference). UNINA_ERR_ARG if the op does not lead such a launch. */
int unina_debug_dual_stamps(unina_engine_t *e, int op_index, long long *out16, hipStream_t stream);
/* Debug: the same launch with every workgroup's start / end on the 100 MHz wall clock (out[2*i], out[2*i+1]); returns the
   grid size or a negative error code. */
int unina_debug_dual_timeline(unina_engine_t* e, int op_index, long long* out, int cap, hipStream_t stream);
/* Debug: the fused C3k2 block led by op `op_index` run once as a stamped twin (after the ops in front of it): out16[k] =
 * shader-clock stamp of its mid workgroup after step k (0 entry, 1 patch landed, 2 pre-conv, 3 cv1|cv2, 4 b0.cv1, 5 b0.cv2,
 * 6 b1.cv1, 7 last bottleneck, 8 cv3, 9 output stored, 10 tail conv, 11 drained), [14] / [15] = 100 MHz clock at end / entry.
 * Only the 40x40-level blocks have twins (UNINA_ERR_HIP otherwise). */
int unina_debug_block_stamps(unina_engine_t *e, int op_index, long long *out16, hipStream_t stream);

/* Library/build identification: "unina_mi355 <version> gfx950". */
const char *unina_version(void);

/* ------------------------------------------------------------------ multi-GPU: the gather of detection slots, over RCCL
 * SURVEY.md section 8e. The path shards by frame: one process per GPU, the engine file loaded by each, NO data-path collective;
 * the only exchange is an all-gather of fixed-size detection records (e.g. k slots of 8 + 8 * MAX_DETECTIONS int32 words, as
 * bench.py / gather.py use) so that one rank -- the node that publishes -- sees every GPU's detections. The reference has no
 * counterpart (one GPU, one stream: perception_node.cpp:472,802); these entry points are what its node would call with
 * several MI355X (INTEGRATION.md "Several GPUs"). RCCL is loaded on first use: single-GPU consumers never touch it, and
 * unina_comm_* return UNINA_ERR_UNSUPPORTED (message in unina_comm_last_error) where librccl is absent.
 * Issue the gather on a stream of its own behind an event of the frames it carries, as gather.SlotRing does: the call only
 * enqueues (RCCL semantics), inference streams need not wait for it. */
#define UNINA_COMM_ID_BYTES 128
typedef struct unina_comm unina_comm;
/* rank 0: 128 opaque bytes to hand to every rank out of band (file, socket, MPI, ROS parameter ...) */
int unina_comm_unique_id(void *id128);
/* every rank, collectively: joins the communicator `id128` as `rank` of `world` on HIP device `device_id` */
int unina_comm_init(unina_comm **out, const void *id128, int rank, int world, int device_id);
/* d_send: bytes_per_rank bytes on this rank's device; d_recv: world * bytes_per_rank bytes, rank-major. Enqueued on `stream`. */
int unina_comm_all_gather(unina_comm *c, const void *d_send, void *d_recv, size_t bytes_per_rank, hipStream_t stream);
int unina_comm_rank(const unina_comm *c);
int unina_comm_world(const unina_comm *c);
void unina_comm_destroy(unina_comm *c);
const char *unina_comm_last_error(void);

/* ------------------------------------------------------------------ post-process C API (gpu_postprocess.h:42-80)
 * Same seven symbols, same argument meaning, hipError_t/hipStream_t in place of cudaError_t/cudaStream_t
 * (ABI-identical: int + pointer), so perception_node.cpp:627-656 recompiles unchanged under HIP.
 * Semantics are the deterministic ones of SURVEY.md App. D (the reference kernel's atomic append order and
 * racy NMS are not reproducible by construction). One process-global workspace, like the reference
 * (gpu_postprocess.cu:56-57); the engine handles above each own a private one instead. */
hipError_t init_postprocess_resources(void);
hipError_t cleanup_postprocess_resources(void);
hipError_t reset_detection_counter(hipStream_t stream);
hipError_t get_detection_count(int *count, hipStream_t stream);
hipError_t decode_yolo_head(const float *d_cls, const float *d_reg, GpuDetection *d_detections, int grid_w,
                            int grid_h, int stride, int num_classes, float conf_threshold, float conformal_q,
                            hipStream_t stream);
hipError_t run_gpu_nms(GpuDetection *d_detections, int num_detections, float iou_threshold, hipStream_t stream);
hipError_t copy_valid_detections_to_host(const GpuDetection *d_detections, GpuDetection *h_detections,
                                         int num_detections, int *out_valid_count, hipStream_t stream);

/* ------------------------------------------------------------------ pre-process C API (cuda_preprocess.h:50-112)
 * The step right before the engine in processGpuBuffer (perception_node.cpp:601-604): camera buffer -> fp32 RGB
 * planar "images" tensor. Same names, argument order and error behaviour as the reference (allocators and the
 * stream factory return NULL on failure, cuda_preprocess.cu:395-428); arithmetic per cuda_preprocess.cu:99-253. */

NormParams create_norm_params_imagenet(void);
NormParams create_norm_params(float mean_r, float mean_g, float mean_b, float std_r, float std_g, float std_b);
hipError_t preprocess_bgra_resize(const uint8_t *d_input, float *d_output, int src_width, int src_height,
                                  int src_pitch, int dst_width, int dst_height, NormParams params,
                                  hipStream_t stream);
hipError_t preprocess_bgra(const uint8_t *d_input, float *d_output, int width, int height, int pitch,
                           NormParams params, hipStream_t stream);
hipError_t preprocess_nv12(const uint8_t *d_y_plane, const uint8_t *d_uv_plane, float *d_output, int width,
                           int height, int y_pitch, int uv_pitch, NormParams params, hipStream_t stream);
float *allocate_preprocess_buffer(int width, int height);
void free_preprocess_buffer(float *d_buffer);
hipStream_t create_preprocess_stream(void);
void destroy_preprocess_stream(hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UNINA_MI355_H */
