// node_harness -- a compiled C++ consumer of include/unina_mi355.h: the per-frame body of the reference's ROS 2 node
// (PerceptionNodeLifecycle::processGpuBuffer, ros2_ws/src/perception/src/perception_node.cpp:581-689, and its engine
// wrapper class TensorRTEngine, :223-351) with ROS, the ZED SDK and TensorRT taken away and libunina_mi355.so put in.
// It includes ONLY the public header (plus the HIP runtime API for the buffers the node owns, :473-483, 696-707), so
// that the Mi355Engine shim and the three call sequences of INTEGRATION.md are compiled and run, not just documented.
//
//   node_harness <engine.une> <frame.bgra> <src_w> <src_h> <pitch> <mode A|B|C> <out.bin> [conf iou q]
//
//   mode A: preprocess_bgra_resize -> bind seven tensors -> enqueueV3 -> reset_detection_counter, 3 x decode_yolo_head,
//           get_detection_count, run_gpu_nms, copy_valid_detections_to_host   (the node's own sequence, :601-656)
//   mode B: preprocess_bgra_resize -> unina_infer                              (fused path)
//   mode C: unina_infer_bgra                                                   (camera frame in, detections out)
//
// <out.bin>: int32 count, then count 32-byte GpuDetection records -- what the node would publish (:659-678).
// Built by __graft_entry__.build() with hipcc, linked with -lunina_mi355; tests/test_gpu_node_harness.py runs it as a
// child process and compares the bytes with the ctypes path.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "unina_mi355.h"

#define HIP_CHECK(call)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      std::fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      std::exit(3);                                                                           \
    }                                                                                         \
  } while (0)

// replaces TensorRTEngine (perception_node.cpp:223-351); same method names, the C ABI underneath
class Mi355Engine {
 public:
  ~Mi355Engine() { unload(); }
  bool load(const std::string &engine_path, int device /* was dla_core */) {                       // :228-259
    return unina_load_engine(engine_path.c_str(), device, &e_) == UNINA_OK;
  }
  void unload() {                                                                                   // :261-266
    unina_unload_engine(e_);
    e_ = nullptr;
  }
  void setInputTensorAddress(const char *name, void *p) { unina_set_tensor_address(e_, name, p); }  // :268
  void setOutputTensorAddress(const char *name, void *p) { unina_set_tensor_address(e_, name, p); } // :273
  bool enqueueV3(hipStream_t s) { return unina_enqueue(e_, s) == UNINA_OK; }                        // :278-282
  bool isLoaded() const { return e_ != nullptr; }                                                   // :284
  bool getInputDimensions(int &w, int &h) const {                                                   // :297-325
    return unina_engine_input_dims(e_, &w, &h, nullptr) == UNINA_OK;
  }
  int numClasses() const {
    int nc = 0;
    unina_engine_input_dims(e_, nullptr, nullptr, &nc);
    return nc;
  }
  unina_engine_t *handle() { return e_; }

 private:
  unina_engine_t *e_ = nullptr;
};

struct GpuBufferHandle {  // perception_node.cpp:357-368 (the camera buffer the SDK hands over)
  void *device_ptr;
  int width, height, pitch;
};

class Node {
 public:
  bool configure(const char *engine_path, float conf, float iou, float q) {                         // on_configure, :410-500
    confidence_threshold_ = conf;
    iou_threshold_ = iou;
    conformal_q_ = q;
    if (!engine_.load(engine_path, 0)) {
      std::fprintf(stderr, "load failed: %s\n", unina_last_error(nullptr));
      return false;
    }
    if (!engine_.getInputDimensions(input_width_, input_height_)) return false;
    num_classes_ = engine_.numClasses();
    stream_ = create_preprocess_stream();                                                            // :472
    if (!stream_) return false;
    d_preprocess_output_ = allocate_preprocess_buffer(input_width_, input_height_);                  // :473-474
    if (!d_preprocess_output_) return false;
    norm_params_ = create_norm_params_imagenet();                                                    // :392-395
    p2_w_ = input_width_ / 4, p2_h_ = input_height_ / 4;                                             // :460-465
    p3_w_ = input_width_ / 8, p3_h_ = input_height_ / 8;
    p4_w_ = input_width_ / 16, p4_h_ = input_height_ / 16;
    auto alloc = [](float **p, size_t n) { HIP_CHECK(hipMalloc(reinterpret_cast<void **>(p), n * sizeof(float))); };
    alloc(&d_p2_cls_, size_t(num_classes_) * p2_w_ * p2_h_), alloc(&d_p2_reg_, size_t(4) * p2_w_ * p2_h_);   // :696-707
    alloc(&d_p3_cls_, size_t(num_classes_) * p3_w_ * p3_h_), alloc(&d_p3_reg_, size_t(4) * p3_w_ * p3_h_);
    alloc(&d_p4_cls_, size_t(num_classes_) * p4_w_ * p4_h_), alloc(&d_p4_reg_, size_t(4) * p4_w_ * p4_h_);
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d_detections_), MAX_DETECTIONS * sizeof(GpuDetection)));  // :481-482
    h_detections_.resize(MAX_DETECTIONS);                                                            // :483
    if (init_postprocess_resources() != hipSuccess) return false;                                    // :478
    return true;
  }

  // the node's own sequence, perception_node.cpp:601-656; returns the number of published detections or -1
  int processGpuBuffer_A(const GpuBufferHandle &buffer) {
    if (buffer.pitch % 4 != 0) return -1;
    hipError_t err = preprocess_bgra_resize(static_cast<const uint8_t *>(buffer.device_ptr), d_preprocess_output_,
                                            buffer.width, buffer.height, buffer.pitch, input_width_, input_height_,
                                            norm_params_, stream_);
    if (err != hipSuccess) return -1;
    engine_.setInputTensorAddress("images", d_preprocess_output_);
    engine_.setOutputTensorAddress("p2_cls", d_p2_cls_);
    engine_.setOutputTensorAddress("p2_reg", d_p2_reg_);
    engine_.setOutputTensorAddress("p3_cls", d_p3_cls_);
    engine_.setOutputTensorAddress("p3_reg", d_p3_reg_);
    engine_.setOutputTensorAddress("p4_cls", d_p4_cls_);
    engine_.setOutputTensorAddress("p4_reg", d_p4_reg_);
    if (!engine_.enqueueV3(stream_)) return -1;
    reset_detection_counter(stream_);
    decode_yolo_head(d_p2_cls_, d_p2_reg_, d_detections_, p2_w_, p2_h_, 4, num_classes_, confidence_threshold_,
                     conformal_q_, stream_);
    decode_yolo_head(d_p3_cls_, d_p3_reg_, d_detections_, p3_w_, p3_h_, 8, num_classes_, confidence_threshold_,
                     conformal_q_, stream_);
    decode_yolo_head(d_p4_cls_, d_p4_reg_, d_detections_, p4_w_, p4_h_, 16, num_classes_, confidence_threshold_,
                     conformal_q_, stream_);
    int num_detections = 0;
    get_detection_count(&num_detections, stream_);
    HIP_CHECK(hipStreamSynchronize(stream_));
    int valid_count = 0;
    if (num_detections > 0) {
      num_detections = std::min(num_detections, int(MAX_DETECTIONS));
      run_gpu_nms(d_detections_, num_detections, iou_threshold_, stream_);
      copy_valid_detections_to_host(d_detections_, h_detections_.data(), num_detections, &valid_count, stream_);
    }
    return valid_count;
  }

  // INTEGRATION.md Option B: pre-process, then ONE call for perception_node.cpp:612-656
  int processGpuBuffer_B(const GpuBufferHandle &buffer) {
    hipError_t err = preprocess_bgra_resize(static_cast<const uint8_t *>(buffer.device_ptr), d_preprocess_output_,
                                            buffer.width, buffer.height, buffer.pitch, input_width_, input_height_,
                                            norm_params_, stream_);
    if (err != hipSuccess) return -1;
    int n = 0;
    int rc = unina_infer(engine_.handle(), d_preprocess_output_, confidence_threshold_, iou_threshold_, conformal_q_,
                         h_detections_.data(), &n, stream_);
    if (rc != UNINA_OK) {
      std::fprintf(stderr, "unina_infer: %s\n", unina_last_error(engine_.handle()));
      return -1;
    }
    return n;
  }

  // INTEGRATION.md Option C: camera frame in, detections out (perception_node.cpp:601-656 as one call)
  int processGpuBuffer_C(const GpuBufferHandle &buffer) {
    int n = 0;
    int rc = unina_infer_bgra(engine_.handle(), static_cast<const uint8_t *>(buffer.device_ptr), buffer.width,
                              buffer.height, buffer.pitch, &norm_params_, confidence_threshold_, iou_threshold_,
                              conformal_q_, h_detections_.data(), &n, stream_);
    if (rc != UNINA_OK) {
      std::fprintf(stderr, "unina_infer_bgra: %s\n", unina_last_error(engine_.handle()));
      return -1;
    }
    return n;
  }

  void cleanup() {                                                                                   // :709-751
    if (stream_) hipStreamSynchronize(stream_);
    cleanup_postprocess_resources();
    for (float *p : {d_p2_cls_, d_p2_reg_, d_p3_cls_, d_p3_reg_, d_p4_cls_, d_p4_reg_})
      if (p) hipFree(p);
    if (d_detections_) hipFree(d_detections_);
    free_preprocess_buffer(d_preprocess_output_);
    engine_.unload();
    if (stream_) destroy_preprocess_stream(stream_);
  }

  const std::vector<GpuDetection> &detections() const { return h_detections_; }
  int width() const { return input_width_; }
  int height() const { return input_height_; }

 private:
  Mi355Engine engine_;
  hipStream_t stream_ = nullptr;
  NormParams norm_params_{};
  int input_width_ = 0, input_height_ = 0, num_classes_ = 0;
  int p2_w_ = 0, p2_h_ = 0, p3_w_ = 0, p3_h_ = 0, p4_w_ = 0, p4_h_ = 0;
  float confidence_threshold_ = 0.5f, iou_threshold_ = 0.45f, conformal_q_ = 0.1f;   // :387-389
  float *d_preprocess_output_ = nullptr;
  float *d_p2_cls_ = nullptr, *d_p2_reg_ = nullptr, *d_p3_cls_ = nullptr, *d_p3_reg_ = nullptr, *d_p4_cls_ = nullptr,
        *d_p4_reg_ = nullptr;
  GpuDetection *d_detections_ = nullptr;
  std::vector<GpuDetection> h_detections_;
};

int main(int argc, char **argv) {
  if (argc < 8) {
    std::fprintf(stderr, "usage: %s engine.une frame.bgra src_w src_h pitch A|B|C out.bin [conf iou q]\n", argv[0]);
    return 2;
  }
  const char *engine_path = argv[1], *frame_path = argv[2], *out_path = argv[7];
  const int src_w = std::atoi(argv[3]), src_h = std::atoi(argv[4]), pitch = std::atoi(argv[5]);
  const char mode = argv[6][0];
  const float conf = argc > 8 ? std::atof(argv[8]) : 0.5f, iou = argc > 9 ? std::atof(argv[9]) : 0.45f,
              q = argc > 10 ? std::atof(argv[10]) : 0.1f;
  if (src_w <= 0 || src_h <= 0 || pitch < 4 * src_w) {
    std::fprintf(stderr, "bad frame geometry\n");
    return 2;
  }
  std::vector<uint8_t> frame(size_t(pitch) * src_h);
  FILE *f = std::fopen(frame_path, "rb");
  if (!f || std::fread(frame.data(), 1, frame.size(), f) != frame.size()) {
    std::fprintf(stderr, "cannot read %zu bytes from %s\n", frame.size(), frame_path);
    return 2;
  }
  std::fclose(f);

  Node node;
  if (!node.configure(engine_path, conf, iou, q)) return 1;
  GpuBufferHandle buf{nullptr, src_w, src_h, pitch};
  HIP_CHECK(hipMalloc(&buf.device_ptr, frame.size()));
  HIP_CHECK(hipMemcpy(buf.device_ptr, frame.data(), frame.size(), hipMemcpyHostToDevice));

  int n = -1;
  for (int rep = 0; rep < 2; ++rep) {  // twice: the second frame runs on warm state, like every frame after the first
    n = mode == 'A' ? node.processGpuBuffer_A(buf) : mode == 'B' ? node.processGpuBuffer_B(buf) : node.processGpuBuffer_C(buf);
    if (n < 0) {
      std::fprintf(stderr, "frame dropped (mode %c)\n", mode);
      return 1;
    }
  }
  FILE *o = std::fopen(out_path, "wb");
  if (!o) return 2;
  int32_t count = n;
  std::fwrite(&count, sizeof(count), 1, o);
  std::fwrite(node.detections().data(), sizeof(GpuDetection), size_t(n), o);
  std::fclose(o);
  std::printf("mode %c: %dx%d -> %dx%d, %d detections\n", mode, src_w, src_h, node.width(), node.height(), n);
  HIP_CHECK(hipFree(buf.device_ptr));
  node.cleanup();
  return 0;
}
