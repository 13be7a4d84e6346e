// engine.hip -- engine handle behind the C ABI of include/unina_mi355.h.
//
// Role in the reference: class TensorRTEngine (ros2_ws/src/perception/src/perception_node.cpp:223-351) plus
// the per-frame body of processGpuBuffer (perception_node.cpp:612-656). Here the "plan" is an explicit op
// table (engine_format.h) executed as hand-written HIP kernels; the whole forward is captured once into a
// hipGraph and replayed per frame (52 launches -> one graph launch), the post-process is one more launch.
//
// Memory: one HBM blob for all folded weights (10 MB fp16), one arena for the NHWC fp16 activation buffers
// (86 MB at 640^2 -- every tensor keeps its own slot; with 288 GB there is nothing to gain from aliasing and
// distinct slots keep two handles = two frames in flight trivially independent).
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "engine_format.h"
#include "kernels.h"

using namespace unina;

namespace {

thread_local std::string g_load_error = "";

struct Buffer {
  BufferDesc d;
  void* ptr = nullptr;    // current device address (engine-owned or caller-bound)
  void* owned = nullptr;  // engine-owned allocation (inside the arena), if any
  size_t bytes = 0;
};

struct PlannedOp {
  OpDesc d;
  ConvParams cp;
  ConvLaunch cl;
  StemParams sp;
  PoolParams pp;
  QuantParams qp;
  unina_op_info info;
  // fused C3k2 block (c3k2_fused.hip): role 1 = first op of a fusable group (launches the whole block when fusion is
  // on), 2 = absorbed by the group's first op (no launch of its own when fusion is on), 0 = ordinary op
  int fuse_role = 0;
  int fuse_kind = 0;        // role 1: 1 = C3k2 block (c3k2_fused.hip), 2 = DetectionHead (head_fused.hip), 4 = conv pair (conv_pair.hip)
  HeadParams hp;
  PairParams pr;            // fuse_kind 4: two consecutive 1x1 convs (conv_pair.hip)
  int group_last = -1;      // role 1: index of the group's last op (cv3 / the head's output convs)
  int dual_with = -1;       // >= 0: this conv and conv `dual_with` (independent, same kernel family) run as ONE grid
  int dual_kind = -1;
  bool dual_absorbed = false;   // runs inside an earlier op's dual launch
  int tail_op = -1;         // role 1, C3k2: index of the 1x1 conv that runs as the block kernel's last step
  int tail_kind = 0;        // 1: lateral 1x1 + x2 upsample store; 2: plain 1x1 ConvBlock (same resolution); 3: 1 with int8 in, fp16 out
  uint64_t wt_off = 0;      // stem: blob offset of the transposed weights [27][Co] (appended at load)
  uint64_t w_lane_off[2] = {0, 0};   // conv: blob offset of each slice's weights in LANE order (appended at load; 0 = none)
  int fuse_pre = 0;         // role 1, C3k2: 1 = this op is the 3x3/s2 conv in front of the block (the block's cv1|cv2 is the NEXT op)
  int quant_op = -1;        // role 1, fp16 C3k2 in an INT8 engine: the QUANT op of the block's output that the kernel's store absorbs
  int hid = 0, nb = 0;      // role 1: hidden width, bottleneck count
  uint64_t stream_off = 0, fbias_off = 0;   // role 1: blob offsets of the packed stage stream / concatenated biases
  C3k2Params fp;
};

struct DeviceResult {  // what the fused post-process writes; one D2H brings count + records
  int count;
  int candidates;
  unsigned int seq;    // host block only: the sequence number the post-process stores last (unina_infer's completion word)
  int pad[5];
  GpuDetection det[MAX_DETECTIONS];
  long long pad_stamps[16]; // debug phase stamps of the post-process kernel (UNINA_POST_STAMPS=1) / of a conv or a dual conv launch
};

}  // namespace

struct unina_engine {
  int device = 0;
  FileHeader h;
  std::vector<Buffer> bufs;
  std::vector<PlannedOp> ops;
  void* d_blob = nullptr;
  void* d_arena = nullptr;
  void* d_zeros = nullptr;          // zero page read by out-of-image taps
  std::vector<int> force_cfg;       // per-op tile configuration override (-1 = heuristic)
  bool fuse = false;                // run fusable C3k2 groups as one launch each (fp16 engines; UNINA_FUSE=0 / unina_set_fusion)
  int n_groups = 0;                 // fusable groups found at load
  int images_buf = -1;
  int out_buf[6] = {-1, -1, -1, -1, -1, -1};
  // post-process workspace
  GpuDetection* d_cand = nullptr;
  int* d_block_count = nullptr;
  unsigned int* d_ticket = nullptr;
  void* d_post_ws = nullptr;         // two-launch post-process workspace (sorted candidates, mask tiles, second ticket)
  DeviceResult* d_result = nullptr;
  DeviceResult* h_result = nullptr;  // pinned
  DeviceResult* h_result_dev = nullptr;  // the same block as the device addresses it (hipHostGetDevicePointer)
  bool camera_active = false;            // inside unina_infer_bgra: the stem reads a camera frame, "images" need not be bound
  unsigned int result_seq = 0;           // unina_infer calls so far
  unsigned int* done_flag = nullptr;     // set around unina_infer's launch: the post-process signals completion there
  unsigned int done_value = 0;
  int post_blocks = 0;
  // graph
  bool use_graph = true;
  bool plan_dirty = true;
  hipStream_t capture_stream = nullptr;
  double t_submit_us = 0, t_wait_us = 0, t_copy_us = 0;   // UNINA_TIMING=1: host-side split of unina_infer (printed at unload)
  long t_calls = 0;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  // full-frame graph (unina_infer / unina_infer_async): stem + forward + post-process as ONE hipGraphLaunch. The two
  // nodes whose parameters change from frame to frame (input pointer; thresholds / output buffers) are re-pointed with
  // hipGraphExecKernelNodeSetParams instead of being launched separately (which cost a ~9 us bubble in front of the
  // post-process and a separate submission for the stem).
  bool full_graph = true;
  hipGraph_t fgraph = nullptr;
  hipGraphExec_t fexec = nullptr;
  hipGraphNode_t stem_node = nullptr, post_node = nullptr, post_node2 = nullptr;
  bool post_split = true;            // post-process as two launches (UNINA_POST_SPLIT=0: everything in one workgroup)
  int post_mode = 2;                 // 2: the two-launch form (compact candidate list, sort-free NMS); 0 (UNINA_POST_SPLIT=0): everything in one launch
  bool fold_heads = true;            // full-frame graph: the heads' output convs run inside the decode launch (UNINA_POST_FOLD=0: off)
  int fold_op[3] = {-1, -1, -1};     // per head: the output-conv op the decode launch absorbs (-1: the head is read from its planes)
  int stem_op = -1;
  StemParams f_stem;
  PostParams f_post;
  std::string err;
};

namespace {

int fail(unina_engine* e, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf; else g_load_error = buf;
  return code;
}

#define HIPCHK(e, call)                                                                          \
  do {                                                                                           \
    hipError_t _err = (call);                                                                    \
    if (_err != hipSuccess) return fail((e), UNINA_ERR_HIP, "%s: %s", #call, hipGetErrorString(_err)); \
  } while (0)

size_t buffer_bytes(const BufferDesc& d) {
  const size_t n = (size_t)d.h * d.w * d.c;
  return d.dtype == kBufF16Nhwc ? n * 2 : (d.dtype == kBufI8Nhwc ? n : n * 4);   // (kBufS16Nhwc: two fp16 planes, hi then lo)
}
// split fp16 buffers: byte distance from the hi plane to the lo plane
long long lo_plane(const BufferDesc& d) { return d.dtype == kBufS16Nhwc ? (long long)d.h * d.w * d.c * 2 : 0; }

// element type of an NHWC activation buffer (-1: not an activation buffer)
int act_dtype_of(uint32_t buf_dtype) {
  return buf_dtype == kBufF16Nhwc ? kF16 : (buf_dtype == kBufF32Nhwc ? kF32 : (buf_dtype == kBufI8Nhwc ? kI8 : (buf_dtype == kBufS16Nhwc ? kS16 : -1)));
}
size_t dtype_size(int dt) { return dt == kF32 ? 4 : (dt == kI8 ? 1 : 2); }        // bytes per element of one plane (addressing)
double dtype_bytes(int dt) { return dt == kS16 ? 4.0 : (double)dtype_size(dt); }   // bytes per value (traffic accounting)

int engine_dtype(const unina_engine* e);

void drop_graph(unina_engine* e) {
  if (e->exec) (void)hipGraphExecDestroy(e->exec);
  if (e->graph) (void)hipGraphDestroy(e->graph);
  if (e->fexec) (void)hipGraphExecDestroy(e->fexec);
  if (e->fgraph) (void)hipGraphDestroy(e->fgraph);
  e->exec = nullptr;
  e->graph = nullptr;
  e->fexec = nullptr;
  e->fgraph = nullptr;
  e->stem_node = e->post_node = e->post_node2 = nullptr;
}

int engine_dtype(const unina_engine* e) { return e->h.precision == kFp32 ? kF32 : (e->h.precision == kSplit16 ? kS16 : kF16); }

// ---- dependency analysis over the op table (buffer id + channel range granularity) ----
struct Region {
  int buf, c0, c1;
};
bool overlaps(const Region& a, const Region& b) { return a.buf == b.buf && a.c0 < b.c1 && b.c0 < a.c1; }

void op_regions(const unina_engine* e, size_t i, std::vector<Region>* reads, std::vector<Region>* writes) {
  const OpDesc& d = e->ops[i].d;
  reads->clear();
  writes->clear();
  if (e->fuse && e->ops[i].fuse_role == 2) return;
  if (e->ops[i].dual_absorbed) return;
  if (e->ops[i].dual_with >= 0 && !e->ops[i].fuse_role) {   // the pair's reads / writes, at the leader's position
    const int pair[2] = {(int)i, e->ops[i].dual_with};
    for (int k : pair) {
      const OpDesc& dk = e->ops[k].d;
      for (uint32_t s = 0; s < dk.nseg; ++s) {
        reads->push_back({(int)dk.src_buf, (int)dk.seg[s].src_coff, (int)(dk.seg[s].src_coff + dk.cin)});
        writes->push_back({(int)dk.seg[s].dst_buf, (int)dk.seg[s].dst_coff, (int)(dk.seg[s].dst_coff + dk.seg[s].n_count)});
      }
    }
    return;
  }
  if (e->fuse && e->ops[i].fuse_role == 1) {
    const OpDesc& last = e->ops[e->ops[i].group_last].d;
    reads->push_back({(int)d.src_buf, (int)d.seg[0].src_coff, (int)(d.seg[0].src_coff + d.cin)});
    if (e->ops[i].fuse_kind == 1 && e->ops[i].fuse_pre) {   // pre-conv: the block also reads the part of its input the pre-conv does not produce
      const OpDesc& blk = e->ops[i + 1].d;
      if (d.seg[0].n_count < blk.cin)
        reads->push_back({(int)blk.src_buf, (int)(blk.seg[0].src_coff + d.seg[0].n_count), (int)(blk.seg[0].src_coff + blk.cin)});
    }
    if (e->ops[i].fuse_kind == 4) {   // the first conv's output goes to HBM too
      const OpDesc& c1 = e->ops[i + e->ops[i].fuse_pre].d;
      writes->push_back({(int)c1.seg[0].dst_buf, (int)c1.seg[0].dst_coff, (int)(c1.seg[0].dst_coff + c1.seg[0].n_count)});
    }
    for (uint32_t s = 0; s < last.nseg; ++s)
      writes->push_back({(int)last.seg[s].dst_buf, (int)last.seg[s].dst_coff, (int)(last.seg[s].dst_coff + last.seg[s].n_count)});
    if (e->ops[i].tail_op >= 0) {
      const SegDesc& ts = e->ops[e->ops[i].tail_op].d.seg[0];
      writes->push_back({(int)ts.dst_buf, (int)ts.dst_coff, (int)(ts.dst_coff + ts.n_count)});
    }
    if (e->ops[i].quant_op >= 0) {
      const SegDesc& qs = e->ops[e->ops[i].quant_op].d.seg[0];
      writes->push_back({(int)qs.dst_buf, (int)qs.dst_coff, (int)(qs.dst_coff + qs.n_count)});
    }
    if (e->ops[i].dual_with >= 0) {   // block dual: the partner group's reads / writes happen here too
      const PlannedOp& hb = e->ops[e->ops[i].dual_with];
      const OpDesc& hl = e->ops[hb.group_last].d;
      reads->push_back({(int)hb.d.src_buf, (int)hb.d.seg[0].src_coff, (int)(hb.d.seg[0].src_coff + hb.d.cin)});
      for (uint32_t s = 0; s < hl.nseg; ++s)
        writes->push_back({(int)hl.seg[s].dst_buf, (int)hl.seg[s].dst_coff, (int)(hl.seg[s].dst_coff + hl.seg[s].n_count)});
    }
    return;
  }
  if (d.kind == kOpSppfPool) {
    const int c0 = (int)d.seg[0].src_coff, C = (int)d.cin;
    reads->push_back({(int)d.src_buf, c0, c0 + C});
    writes->push_back({(int)d.src_buf, c0 + C, c0 + 4 * C});
    return;
  }
  for (uint32_t s = 0; s < d.nseg; ++s) {
    const SegDesc& sd = d.seg[s];
    const int cin = d.kind == kOpStem ? 3 : (int)d.cin;
    reads->push_back({(int)d.src_buf, (int)sd.src_coff, (int)sd.src_coff + cin});
    writes->push_back({(int)sd.dst_buf, (int)sd.dst_coff, (int)(sd.dst_coff + sd.n_count)});
    if (d.res_buf >= 0) reads->push_back({d.res_buf, d.res_coff, d.res_coff + (int)sd.n_count});
  }
}

// (Re)computes kernel parameters from the current buffer addresses. Element types are properties of the BUFFERS
// (fp16 / fp32 / int8 NHWC): a conv runs in the type of its source buffer and converts to the type of each
// destination buffer in its epilogue, so fp16, fp32 and mixed int8/fp16 engines share one planner.
// Which heads' output convs (model.py:292,299: Conv2d(C, nc | 4, 1) with bias, the `.2` layers) can run inside the decode
// launch of the post-process (postprocess.hip, fold): a plain fp16 1x1 conv op of two slices that write the head's cls
// and reg planes, at most 16 output channels each. A head whose output conv already lives in a fused launch (the P2
// head kernel) is read from its planes instead.
void find_fold_ops(unina_engine* e) {
  for (int h = 0; h < 3; ++h) e->fold_op[h] = -1;
  if (!e->fold_heads || e->post_mode != 2) return;
  for (size_t i = 0; i < e->ops.size(); ++i) {
    const PlannedOp& op = e->ops[i];
    if (op.d.kind != kOpConv || (e->fuse && op.fuse_role)) continue;
    const ConvParams& c = op.cp;
    if ((c.dtype != kF16 && c.dtype != kS16) || c.ksize != 1 || c.stride != 1 || c.relu || c.res || c.nseg != 2 || (c.Cin & 31)) continue;
    for (int h = 0; h < 3; ++h) {
      const void* pc = e->bufs[e->out_buf[2 * h]].ptr;
      const void* pr = e->bufs[e->out_buf[2 * h + 1]].ptr;
      if (!pc || !pr || c.seg[0].dst_planar != pc || c.seg[1].dst_planar != pr) continue;
      if (c.seg[0].n_count > 16 || c.seg[1].n_count > 16 || c.seg[0].mult || c.seg[1].mult) continue;
      if (c.M != (int)(e->bufs[e->out_buf[2 * h]].d.w * e->bufs[e->out_buf[2 * h]].d.h)) continue;
      e->fold_op[h] = (int)i;
    }
  }
}

int plan(unina_engine* e) {
  const char* blob = static_cast<const char*>(e->d_blob);
  for (size_t i = 0; i < e->ops.size(); ++i) {
    PlannedOp& op = e->ops[i];
    const OpDesc& d = op.d;
    const Buffer& src = e->bufs[d.src_buf];
    unina_op_info& info = op.info;
    memset(&info, 0, sizeof info);
    snprintf(info.name, sizeof info.name, "%s", d.name);
    info.kind = (int)d.kind;
    if (d.kind == kOpConv) {
      ConvParams& p = op.cp;
      memset(&p, 0, sizeof p);
      const int dt = act_dtype_of(src.d.dtype);
      if (dt < 0) return fail(e, UNINA_ERR_FORMAT, "op %zu: source is not an activation buffer", i);
      const size_t esz = dtype_size(dt);
      p.dtype = dt;
      p.src = src.ptr;
      p.src_lo = lo_plane(src.d);
      p.src_ld = (int)src.d.c;
      p.H = (int)d.in_h; p.W = (int)d.in_w; p.Cin = (int)d.cin;
      p.Ho = (int)d.out_h; p.Wo = (int)d.out_w; p.M = p.Ho * p.Wo;
      p.ksize = (int)d.ksize; p.stride = (int)d.stride; p.pad = (int)d.ksize / 2;
      p.relu = (int)d.relu;
      if (d.res_buf >= 0) {
        const Buffer& rb = e->bufs[d.res_buf];
        const int rdt = act_dtype_of(rb.d.dtype);
        if (rdt < 0) return fail(e, UNINA_ERR_FORMAT, "op %zu: residual is not an activation buffer", i);
        p.res = static_cast<const char*>(rb.ptr) + (size_t)d.res_coff * dtype_size(rdt);
        p.res_ld = (int)rb.d.c;
        p.res_dtype = rdt;
        p.res_lo = lo_plane(rb.d);
        p.res_scale = rb.d.scale;
        if ((rdt == kS16) != (dt == kS16)) return fail(e, UNINA_ERR_FORMAT, "op %zu: split-fp16 conv with a residual of another type", i);
      }
      p.nseg = (int)d.nseg;
      p.zeros = e->d_zeros;
      p.force_cfg = i < e->force_cfg.size() ? e->force_cfg[i] : -1;
      int ntot = 0;
      double out_bytes = 0;
      for (int s = 0; s < p.nseg; ++s) {
        const SegDesc& sd = d.seg[s];
        const Buffer& db = e->bufs[sd.dst_buf];
        ConvSeg& cs = p.seg[s];
        cs.w = blob + sd.w_off;
        cs.w_lane = op.w_lane_off[s] ? blob + op.w_lane_off[s] : nullptr;
        cs.bias = reinterpret_cast<const float*>(blob + sd.b_off);
        cs.mult = sd.m_off ? reinterpret_cast<const float*>(blob + sd.m_off) : nullptr;
        if (dt == kI8 && !cs.mult) return fail(e, UNINA_ERR_FORMAT, "op %zu: int8 conv without multipliers", i);
        cs.src_coff = (int)sd.src_coff;
        cs.n_count = (int)sd.n_count;
        cs.up2 = (sd.flags & kSegUp2) ? 1 : 0;
        if (sd.flags & kSegPlanarF32) {
          if (db.d.dtype != kBufF32Planar) return fail(e, UNINA_ERR_FORMAT, "op %zu: planar slice into non-planar buffer", i);
          cs.dst_planar = static_cast<float*>(db.ptr);
          cs.dst = nullptr;
          cs.dst_ld = 0;
          cs.out_dtype = kF32;
          out_bytes += 4.0 * sd.n_count * p.M;
        } else {
          const int odt = act_dtype_of(db.d.dtype);
          const uint32_t al = odt == kI8 ? 16 : 8;   // a 16-byte store chunk must not straddle the slice
          if (odt < 0 || sd.n_count % al || sd.dst_coff % al || db.d.c % al)
            return fail(e, UNINA_ERR_UNSUPPORTED, "op %zu: NHWC slice needs channel counts/offsets that are multiples of %u", i, al);
          cs.dst = static_cast<char*>(db.ptr) + (size_t)sd.dst_coff * dtype_size(odt);
          cs.dst_planar = nullptr;
          cs.dst_ld = (int)db.d.c;
          cs.out_dtype = odt;
          cs.dst_lo = lo_plane(db.d);
          if ((odt == kS16) != (dt == kS16)) return fail(e, UNINA_ERR_FORMAT, "op %zu: split-fp16 conv into a buffer of another type", i);
          cs.out_inv_scale = odt == kI8 ? 1.0f / db.d.scale : 1.0f;
          out_bytes += dtype_bytes(odt) * sd.n_count * p.M * (cs.up2 ? 4 : 1);
        }
        ntot += (int)sd.n_count;
      }
      const int kb = dt == kF32 ? 16 : (dt == kI8 ? 64 : 32);
      if (p.Cin % kb || (p.src_ld * esz) % 16) return fail(e, UNINA_ERR_UNSUPPORTED, "op %zu (%s): Cin %% %d != 0", i, d.name, kb);
      if (p.force_cfg >= 0 && !conv_config_valid(p, p.force_cfg)) p.force_cfg = -1;
      op.cl = conv_plan(p);
      const int K = p.ksize * p.ksize * p.Cin;
      info.m = p.M; info.n = ntot; info.k = K;
      info.flops = 2.0 * p.M * (double)ntot * K;
      // algorithmic bytes: each distinct input element once, weights once, outputs once, residual once
      const bool shared_src = p.nseg == 1 || d.seg[0].src_coff == d.seg[1].src_coff;
      info.bytes = dtype_bytes(dt) * p.H * p.W * p.Cin * (shared_src ? 1 : p.nseg) + dtype_bytes(dt) * ntot * K + 4.0 * ntot + out_bytes +
                   (p.res ? dtype_bytes(p.res_dtype) * p.M * ntot : 0.0);
      snprintf(info.kernel, sizeof info.kernel, "%s", op.cl.kernel_name);
      info.grid = (int)(op.cl.grid.x * op.cl.grid.y);
      info.block = (int)op.cl.block.x;
    } else if (d.kind == kOpStem) {
      const SegDesc& sd = d.seg[0];
      const Buffer& db = e->bufs[sd.dst_buf];
      StemParams& p = op.sp;
      const int odt = act_dtype_of(db.d.dtype);
      if (odt != kF16 && odt != kF32 && odt != kS16) return fail(e, UNINA_ERR_UNSUPPORTED, "stem output must be fp16, fp32 or split fp16");
      p.dtype = odt;
      p.dst_lo = lo_plane(db.d);
      p.src = static_cast<const float*>(src.ptr);
      p.w = reinterpret_cast<const float*>(blob + sd.w_off);
      p.wt = reinterpret_cast<const float*>(blob + op.wt_off);
      p.bias = reinterpret_cast<const float*>(blob + sd.b_off);
      p.dst = static_cast<char*>(db.ptr) + (size_t)sd.dst_coff * dtype_size(odt);
      p.H = (int)d.in_h; p.W = (int)d.in_w; p.Ho = (int)d.out_h; p.Wo = (int)d.out_w;
      p.Co = (int)sd.n_count; p.dst_ld = (int)db.d.c;
      info.m = p.Ho * p.Wo; info.n = p.Co; info.k = 27;
      info.flops = 2.0 * info.m * info.n * 27;
      info.bytes = 4.0 * 3 * p.H * p.W + dtype_bytes(odt) * info.m * p.Co;
      {
        LaunchDesc sl;
        const bool tiled = stem_desc(p, &sl) == hipSuccess && sl.block.x == 128;
        snprintf(info.kernel, sizeof info.kernel, "%s<%s,%d>", tiled ? "stem_tile_kernel" : "stem_conv_kernel", odt == kF32 ? "f32" : (odt == kS16 ? "s16" : "f16"), p.Co);
        info.grid = tiled ? (int)sl.grid.x : (2 * info.m + 255) / 256;
        info.block = tiled ? 128 : 256;
      }
    } else if (d.kind == kOpSppfPool) {
      PoolParams& p = op.pp;
      const int dt = act_dtype_of(src.d.dtype);
      if (dt < 0) return fail(e, UNINA_ERR_FORMAT, "op %zu: pool on a non-activation buffer", i);
      p.dtype = dt;
      p.buf = src.ptr;
      p.lo = lo_plane(src.d);
      p.H = (int)d.in_h; p.W = (int)d.in_w; p.C = (int)d.cin; p.ld = (int)src.d.c; p.coff = (int)d.seg[0].src_coff;
      info.bytes = dtype_bytes(dt) * p.H * p.W * p.C * 4;
      snprintf(info.kernel, sizeof info.kernel, dt == kS16 ? "sppf_pool_split_kernel<%s32>" : "sppf_pool_kernel<%s,32>", dt == kF32 ? "f32" : (dt == kI8 ? "i8" : (dt == kS16 ? "" : "f16")));
      info.grid = p.H * (p.C / 32);
      info.block = 256;
    } else if (d.kind == kOpQuant) {
      const Buffer& db = e->bufs[d.seg[0].dst_buf];
      if (src.d.dtype != kBufF16Nhwc || db.d.dtype != kBufI8Nhwc || src.d.c != db.d.c || src.d.h != db.d.h || src.d.w != db.d.w)
        return fail(e, UNINA_ERR_FORMAT, "op %zu: QUANT needs an fp16 source and an int8 twin of the same shape", i);
      QuantParams& p = op.qp;
      p.src = static_cast<const half_t*>(src.ptr);
      p.dst = static_cast<signed char*>(db.ptr);
      p.n = (size_t)src.d.h * src.d.w * src.d.c;
      p.inv_scale = 1.0f / db.d.scale;
      info.bytes = 3.0 * p.n;
      snprintf(info.kernel, sizeof info.kernel, "quant_f16_i8_kernel");
      info.grid = (int)((p.n / 16 + 255) / 256);
      info.block = 256;
    } else {
      return fail(e, UNINA_ERR_UNSUPPORTED, "op %zu: kind %u not executable", i, d.kind);
    }
  }
  // fused groups: parameters of the one-launch form; op infos describe what actually runs
  for (size_t i = 0; i < e->ops.size(); ++i) {
    PlannedOp& op = e->ops[i];
    if (op.fuse_role != 1) continue;
    if (op.fuse_kind == 4) {
      const OpDesc& a = e->ops[i + op.fuse_pre].d;
      PlannedOp& zb = e->ops[op.group_last];
      const OpDesc& z = zb.d;
      const Buffer& src = e->bufs[a.src_buf];
      const Buffer& mid = e->bufs[a.seg[0].dst_buf];
      const Buffer& out = e->bufs[z.seg[0].dst_buf];
      PairParams& f = op.pr;
      memset(&f, 0, sizeof f);
      f.dtype = act_dtype_of(src.d.dtype);
      const size_t esz = dtype_size(f.dtype);
      f.src = static_cast<const char*>(src.ptr) + a.seg[0].src_coff * esz;
      f.src_ld = (int)src.d.c;
      f.c0 = (int)a.cin; f.c1 = (int)a.seg[0].n_count; f.c2 = (int)z.seg[0].n_count;
      f.H = (int)a.in_h; f.W = (int)a.in_w;
      f.dst = static_cast<char*>(mid.ptr) + a.seg[0].dst_coff * esz;
      f.dst_ld = (int)mid.d.c;
      f.dst2 = static_cast<char*>(out.ptr) + z.seg[0].dst_coff * esz;
      f.dst2_ld = (int)out.d.c;
      f.up2 = (z.seg[0].flags & kSegUp2) ? 1 : 0;
      f.src_lo = lo_plane(src.d); f.dst_lo = lo_plane(mid.d); f.dst2_lo = lo_plane(out.d);
      f.pool = op.fuse_pre;
      f.wstream = reinterpret_cast<const unsigned char*>(blob + op.stream_off);
      f.bias = reinterpret_cast<const float*>(blob + op.fbias_off);
      f.zeros = e->d_zeros;
      if (!pair_layout(&f)) return fail(e, UNINA_ERR_UNSUPPORTED, "op %zu: fused conv pair does not fit", i);
      if (!e->fuse) continue;
      if (op.fuse_pre) {   // the pool op is the group's head: the infos of cv2 move onto it
        unina_op_info& ai = e->ops[i + 1].info;
        op.info.flops = ai.flops;
        op.info.bytes = ai.bytes - dtype_bytes(f.dtype) * f.H * f.W * 3 * (f.c0 / 4);   // the pooled maps are formed in LDS
        op.info.m = ai.m; op.info.n = ai.n;
        ai.flops = 0; ai.bytes = 0; ai.grid = 0;
        snprintf(ai.kernel, sizeof ai.kernel, "(fused into op %zu)", i);
      }
      op.info.flops += zb.info.flops;
      op.info.bytes += zb.info.bytes - dtype_bytes(f.dtype) * f.H * f.W * f.c1;     // the second conv reads the first's output from LDS
      op.info.grid = f.tiles_x * f.tiles_y;
      op.info.block = pair_block_threads(f);
      op.info.k = 0;
      snprintf(op.info.kernel, sizeof op.info.kernel, "%s", pair_kernel_name(f));
      snprintf(op.info.name, sizeof op.info.name, "%.40s+%.40s", a.name, z.name);
      zb.info.flops = 0;
      zb.info.bytes = 0;
      zb.info.grid = 0;
      snprintf(zb.info.kernel, sizeof zb.info.kernel, "(fused into op %zu)", i);
      continue;
    }
    if (op.fuse_kind == 2) {
      const OpDesc& a = op.d;
      const OpDesc& z = e->ops[op.group_last].d;
      const Buffer& src = e->bufs[a.src_buf];
      HeadParams& f = op.hp;
      memset(&f, 0, sizeof f);
      f.src = static_cast<const half_t*>(src.ptr) + a.seg[0].src_coff;
      f.src_ld = (int)src.d.c;
      f.C = (int)a.cin;
      f.H = (int)a.in_h;
      f.W = (int)a.in_w;
      f.out_cls = static_cast<float*>(e->bufs[z.seg[0].dst_buf].ptr);
      f.out_reg = static_cast<float*>(e->bufs[z.seg[1].dst_buf].ptr);
      f.n_cls = (int)z.seg[0].n_count;
      f.n_reg = (int)z.seg[1].n_count;
      f.wstream = reinterpret_cast<const unsigned char*>(blob + op.stream_off);
      f.bias = reinterpret_cast<const float*>(blob + op.fbias_off);
      f.zeros = e->d_zeros;
      if (!head_layout(&f)) return fail(e, UNINA_ERR_UNSUPPORTED, "op %zu: fused head does not fit", i);
      if (!e->fuse) continue;
      unina_op_info& info = op.info;
      double flops = 0, wbytes = 0;
      for (int k = (int)i; k <= op.group_last; ++k) {
        flops += e->ops[k].info.flops;
        wbytes += 2.0 * e->ops[k].info.n * e->ops[k].info.k + 4.0 * e->ops[k].info.n;
        if (k > (int)i) {
          unina_op_info& ai = e->ops[k].info;
          ai.flops = 0;
          ai.bytes = 0;
          ai.grid = 0;
          snprintf(ai.kernel, sizeof ai.kernel, "(fused into op %zu)", i);
        }
      }
      info.flops = flops;
      info.bytes = 2.0 * f.H * f.W * f.C + wbytes + 4.0 * f.H * f.W * (f.n_cls + f.n_reg);
      info.n = f.n_cls + f.n_reg;
      info.k = 0;
      info.grid = f.tiles_x * f.tiles_y;
      info.block = head_block_threads(f.C);
      snprintf(info.kernel, sizeof info.kernel, "%s", head_kernel_name(f.C));
      snprintf(info.name, sizeof info.name, "%.*s[head]", (int)(strchr(a.name, '.') ? strchr(a.name, '.') - a.name : 60), a.name);
      continue;
    }
    const OpDesc& a = e->ops[i + op.fuse_pre].d;   // the block's cv1|cv2
    const OpDesc& z = e->ops[op.group_last].d;
    const OpDesc& in = op.d;                       // the op whose input the kernel reads (the pre-conv, or cv1|cv2 itself)
    const Buffer& src = e->bufs[in.src_buf];
    const Buffer& dst = e->bufs[z.seg[0].dst_buf];
    C3k2Params& f = op.fp;
    memset(&f, 0, sizeof f);
    f.dtype = act_dtype_of(src.d.dtype);
    const size_t fesz = dtype_size(f.dtype);
    const double fb = dtype_bytes(f.dtype);
    f.src = static_cast<const char*>(src.ptr) + in.seg[0].src_coff * fesz;
    f.src_ld = (int)src.d.c;
    if (op.fuse_pre) {
      f.cpre = (int)in.cin;
      f.preH = (int)in.in_h;
      f.preW = (int)in.in_w;
      f.cx = (int)in.seg[0].n_count;
      const Buffer& xb = e->bufs[a.src_buf];   // the rest of the block's input (a concat's other part) still comes from HBM
      f.src2 = static_cast<const char*>(xb.ptr) + (a.seg[0].src_coff + f.cx) * fesz;
      f.src2_ld = (int)xb.d.c;
      f.src2_lo = lo_plane(xb.d);
    }
    f.src_lo = lo_plane(src.d);
    f.dst_lo = lo_plane(dst.d);
    f.Cin = (int)a.cin;
    f.H = (int)a.in_h;
    f.W = (int)a.in_w;
    f.dst = static_cast<char*>(dst.ptr) + z.seg[0].dst_coff * fesz;
    f.dst_ld = (int)dst.d.c;
    for (int b = 0; b < op.nb; ++b) {   // int8: scale of each bottleneck's shortcut tensor (the 3x3's residual buffer)
      const OpDesc& c2 = e->ops[i + op.fuse_pre + 2 + 2 * b].d;
      f.res_scale[b] = c2.res_buf >= 0 ? e->bufs[c2.res_buf].d.scale : 1.0f;
    }
    f.wstream = reinterpret_cast<const unsigned char*>(blob + op.stream_off);
    f.bias = reinterpret_cast<const float*>(blob + op.fbias_off);
    f.zeros = e->d_zeros;
    f.hid = op.hid;
    f.nb = op.nb;
    if (op.tail_op >= 0) {
      const SegDesc& ts = e->ops[op.tail_op].d.seg[0];
      const Buffer& tb = e->bufs[ts.dst_buf];
      f.tail = op.tail_kind;
      f.dst2 = static_cast<char*>(tb.ptr) + ts.dst_coff * (op.tail_kind == 3 ? 2 : fesz);
      f.dst2_ld = (int)tb.d.c;
      f.dst2_lo = lo_plane(tb.d);
    }
    if (op.quant_op >= 0) {
      const Buffer& qb = e->bufs[e->ops[op.quant_op].d.seg[0].dst_buf];
      f.dst_q = static_cast<signed char*>(qb.ptr);
      f.dst_q_ld = (int)qb.d.c;
      f.q_inv = 1.0f / qb.d.scale;   // as the QUANT op's own parameter
    }
    if (!c3k2_layout(&f)) return fail(e, UNINA_ERR_UNSUPPORTED, "op %zu: fused C3k2 block does not fit", i);
    if (!e->fuse) continue;
    if (op.quant_op >= 0) {
      unina_op_info& qi = e->ops[op.quant_op].info;
      qi.flops = 0;
      qi.bytes = 0;
      qi.grid = 0;
      snprintf(qi.kernel, sizeof qi.kernel, "(fused into op %zu)", i);
    }
    unina_op_info& info = op.info;
    double flops = 0, wbytes = 0;
    const int last_op = op.tail_op >= 0 ? op.tail_op : op.group_last;
    for (int k = (int)i; k <= last_op; ++k) {
      flops += e->ops[k].info.flops;
      wbytes += fb * e->ops[k].info.n * e->ops[k].info.k + 4.0 * e->ops[k].info.n;
      if (k > (int)i) {
        unina_op_info& ai = e->ops[k].info;
        ai.flops = 0;
        ai.bytes = 0;
        ai.grid = 0;
        snprintf(ai.kernel, sizeof ai.kernel, "(fused into op %zu)", i);
      }
    }
    info.flops = flops;
    info.bytes = (f.cpre ? fb * (f.preH * f.preW * f.cpre + f.H * f.W * (f.Cin - f.cx)) : fb * f.H * f.W * f.Cin) + wbytes + fb * f.H * f.W * 2 * f.hid +   // input once, weights once, output once
                 (f.tail ? fb * (f.tail == 1 ? 4 : 1) * f.H * f.W * f.hid : 0.0);            // (+ the tail conv's output)
    info.n = 2 * f.hid;
    info.k = 0;
    info.grid = f.tiles_x * f.tiles_y;
    info.block = c3k2_block_threads(f.hid, f.nb, f.Cin, f.tail, f.dtype, f.cpre, f.cx);
    snprintf(info.kernel, sizeof info.kernel, "%s", c3k2_kernel_name(f.hid, f.nb, f.Cin, f.tail, f.dtype, f.cpre, f.cx));
    snprintf(info.name, sizeof info.name, "%.*s[c3k2 x%d]", (int)(strchr(a.name, '+') ? strchr(a.name, '+') - a.name - 4 : 60), a.name, f.nb);
  }
  // dual launches: pair independent convs of one kernel family (the P3 / P4 head layers) into one grid each. The later
  // op moves to the earlier one's position, so nothing between them may feed it or touch what it writes.
  for (auto& op : e->ops) {
    op.dual_with = op.dual_kind = -1;
    op.dual_absorbed = false;
  }
  if (e->fuse && !getenv("UNINA_NO_DUAL")) {
    const size_t n = e->ops.size();
    std::vector<Region> ri, wi, rj, wj;
    // a fused C3k2 block and the fused head that does not depend on it (pan_c3k2_2 and head_p2): block_dual.hip
    for (size_t i = 0; i < n; ++i) {
      PlannedOp& a = e->ops[i];
      if (a.fuse_role != 1 || a.fuse_kind != 1 || a.dual_with >= 0) continue;
      for (size_t j = i + 1; j < n; ++j) {
        PlannedOp& b = e->ops[j];
        if (b.fuse_role != 1 || b.fuse_kind != 2 || b.dual_absorbed || !block_dual_match(a.fp, b.hp)) continue;
        op_regions(e, j, &rj, &wj);
        bool legal = true;
        for (size_t k = i; k < j && legal; ++k) {
          op_regions(e, k, &ri, &wi);
          for (const Region& w : wi) {
            for (const Region& r : rj) legal = legal && !overlaps(w, r);
            for (const Region& w2 : wj) legal = legal && !overlaps(w, w2);
          }
          for (const Region& r : ri)
            for (const Region& w2 : wj) legal = legal && !overlaps(r, w2);
        }
        if (!legal) continue;
        a.dual_with = (int)j;
        b.dual_absorbed = true;
        a.info.flops += b.info.flops;
        a.info.bytes += b.info.bytes;
        a.info.grid += b.info.grid;
        snprintf(a.info.kernel, sizeof a.info.kernel, "%s", block_dual_name(a.fp.dtype, a.fp.cpre));
        b.info.flops = 0;
        b.info.bytes = 0;
        b.info.grid = 0;
        snprintf(b.info.kernel, sizeof b.info.kernel, "(dual launch with op %zu)", i);
        break;
      }
    }
    for (size_t i = 0; i < n; ++i) {
      PlannedOp& a = e->ops[i];
      if (a.d.kind != kOpConv || a.fuse_role || a.dual_absorbed || a.dual_with >= 0 || a.cp.force_cfg >= 0) continue;
      for (size_t j = i + 1; j < n; ++j) {
        PlannedOp& b = e->ops[j];
        if (b.d.kind != kOpConv || b.fuse_role || b.dual_absorbed || b.dual_with >= 0 || b.cp.force_cfg >= 0) continue;
        const int kind = conv_dual_match(a.cp, b.cp);
        if (kind < 0) continue;
        op_regions(e, j, &rj, &wj);
        bool legal = true;
        for (size_t k = i; k < j && legal; ++k) {
          op_regions(e, k, &ri, &wi);
          for (const Region& w : wi) {
            for (const Region& r : rj) legal = legal && !overlaps(w, r);
            for (const Region& w2 : wj) legal = legal && !overlaps(w, w2);
          }
          for (const Region& r : ri)
            for (const Region& w2 : wj) legal = legal && !overlaps(r, w2);
        }
        if (!legal) {
          // the other direction: the EARLIER op sinks to the later one's position (an INT8 engine's P3 output conv
          // waits for the P4 head's): nothing in between may read what it writes or write what it reads / writes
          const int kind2 = conv_dual_match(b.cp, a.cp);
          if (kind2 < 0) continue;
          std::vector<Region> ra, wa;
          op_regions(e, i, &ra, &wa);
          bool sink = true;
          for (size_t k = i + 1; k <= j && sink; ++k) {
            op_regions(e, k, &ri, &wi);
            for (const Region& w : wi) {
              for (const Region& r : ra) sink = sink && !overlaps(w, r);
              for (const Region& w2 : wa) sink = sink && !overlaps(w, w2);
            }
            for (const Region& r : ri)
              for (const Region& w2 : wa) sink = sink && !overlaps(r, w2);
          }
          if (!sink) continue;
          b.dual_with = (int)i;
          b.dual_kind = kind2;
          a.dual_absorbed = true;
          b.info.flops += a.info.flops;
          b.info.bytes += a.info.bytes;
          b.info.grid += a.info.grid;
          snprintf(b.info.kernel, sizeof b.info.kernel, "%s", conv_dual_name(kind2));
          a.info.flops = 0;
          a.info.bytes = 0;
          a.info.grid = 0;
          snprintf(a.info.kernel, sizeof a.info.kernel, "(dual launch with op %zu)", j);
          break;
        }
        a.dual_with = (int)j;
        a.dual_kind = kind;
        b.dual_absorbed = true;
        int grid = 0;
        (void)grid;
        a.info.flops += b.info.flops;
        a.info.bytes += b.info.bytes;
        a.info.grid += b.info.grid;
        snprintf(a.info.kernel, sizeof a.info.kernel, "%s", conv_dual_name(kind));
        b.info.flops = 0;
        b.info.bytes = 0;
        b.info.grid = 0;
        snprintf(b.info.kernel, sizeof b.info.kernel, "(dual launch with op %zu)", i);
        break;
      }
    }
  }
  find_fold_ops(e);
  e->plan_dirty = false;
  drop_graph(e);
  return UNINA_OK;
}

hipError_t launch_op(unina_engine* e, size_t i, hipStream_t s) {
  PlannedOp& op = e->ops[i];
  if (op.dual_absorbed) return hipSuccess;
  if (e->fuse && op.fuse_role == 1 && op.dual_with >= 0) return block_dual_launch(op.fp, e->ops[op.dual_with].hp, s);
  if (e->fuse && op.fuse_role == 1 && op.fuse_kind == 4) return pair_launch(op.pr, s);
  if (e->fuse && op.fuse_role == 1) return op.fuse_kind == 2 ? head_launch(op.hp, s) : c3k2_launch(op.fp, s);
  if (op.dual_with >= 0) return conv_dual_launch(op.dual_kind, op.cp, e->ops[op.dual_with].cp, s);
  if (e->fuse && op.fuse_role == 2) return hipSuccess;   // runs inside its group's launch
  switch (op.d.kind) {
    case kOpConv: return conv_launch(op.cp, op.cl, s);
    case kOpStem: return stem_launch(op.sp, s);
    case kOpSppfPool: return sppf_pool_launch(op.pp, s);
    case kOpQuant: return quant_launch(op.qp, s);
    default: return hipErrorInvalidValue;
  }
}

// Ops that read the caller's "images" tensor (the stem) stay OUTSIDE the captured graph: a camera pipeline hands
// over a different input buffer every frame, and re-binding must not cost a re-capture.
bool is_eager(const unina_engine* e, size_t i) { return (int)e->ops[i].d.src_buf == e->images_buf; }

int launch_all(unina_engine* e, hipStream_t s, int which = 0 /*0 all, 1 eager only, 2 graph part only*/) {
  for (size_t i = 0; i < e->ops.size(); ++i) {
    if ((which == 1 && !is_eager(e, i)) || (which == 2 && is_eager(e, i))) continue;
    hipError_t err = launch_op(e, i, s);
    if (err != hipSuccess) return fail(e, UNINA_ERR_HIP, "op %zu (%s): %s", i, e->ops[i].d.name, hipGetErrorString(err));
  }
  return UNINA_OK;
}

// Captures the graph part of the forward as a plain chain on the capture stream. (Independent branches as parallel graph
// paths -- heads | PAN path on side streams with event edges -- were implemented and measured in round 1: 1 821 / 2 503
// frames/s with 2 / 3 paths against 5 837 as a chain; removed.)
int capture(unina_engine* e) {
  drop_graph(e);
  HIPCHK(e, hipStreamBeginCapture(e->capture_stream, hipStreamCaptureModeThreadLocal));
  int rc = UNINA_OK;
  for (size_t j = 0; j < e->ops.size() && rc == UNINA_OK; ++j) {
    if (is_eager(e, j)) continue;
    const hipError_t err = launch_op(e, j, e->capture_stream);
    if (err != hipSuccess) rc = fail(e, UNINA_ERR_HIP, "op %zu (%s): %s", j, e->ops[j].d.name, hipGetErrorString(err));
  }
  const hipError_t end = hipStreamEndCapture(e->capture_stream, &e->graph);
  if (rc != UNINA_OK) return rc;
  if (end != hipSuccess) return fail(e, UNINA_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(end));
  HIPCHK(e, hipGraphInstantiate(&e->exec, e->graph, nullptr, nullptr, 0));
  return UNINA_OK;
}

// ---- fusable C3k2 groups (model.py:76-110) -------------------------------------------------------------------------
// The exporter lowers a C3k2 block to  [cv1|cv2 (1x1, two slices)] -> n x [b.cv1 (1x1), b.cv2 (3x3, +residual)] ->
// [cv3 (1x1 over the concat buffer)]. This recognises that shape in the op table (structurally: buffers, slices and
// residual wiring must match exactly, and no op outside the group may read the group's intermediates), packs the
// group's weights into the stage stream of c3k2_fused.hip and appends it to the (host copy of the) blob.
bool is_plain_conv(const OpDesc& d, uint32_t k, uint32_t nseg, bool int8 = false) {
  if (d.kind != kOpConv || d.ksize != k || d.stride != 1 || !d.relu || d.nseg != nseg) return false;
  for (uint32_t s = 0; s < nseg; ++s)
    if (d.seg[s].flags || (d.seg[s].m_off != 0) != int8 || d.seg[s].n_pad != d.seg[s].n_count) return false;
  return true;
}

void find_c3k2_groups(unina_engine* e, std::vector<char>* blob) {
  const size_t n = e->ops.size();
  for (size_t i = 0; i + 3 < n; ++i) {
    const OpDesc& a = e->ops[i].d;
    // fp16 block, or (INT8 engines) a block whose input, output and intermediates are all int8 code tensors
    const uint32_t bdt = e->bufs[a.src_buf].d.dtype;
    if (bdt != kBufF16Nhwc && bdt != kBufI8Nhwc && bdt != kBufS16Nhwc) continue;
    const bool i8 = bdt == kBufI8Nhwc;
    const int dt = i8 ? kI8 : (bdt == kBufS16Nhwc ? kS16 : kF16);
    const uint32_t al = i8 ? 16 : 8;   // channels per 16-byte chunk
    if (!is_plain_conv(a, 1, 2, i8) || a.res_buf >= 0) continue;
    const uint32_t h = a.seg[0].n_count;
    if (a.seg[1].n_count != h || a.seg[0].src_coff != a.seg[1].src_coff) continue;
    if (a.seg[0].src_coff % al || e->bufs[a.src_buf].d.c % al) continue;
    // bottlenecks
    uint32_t cur_buf = a.seg[0].dst_buf, cur_coff = a.seg[0].dst_coff;
    size_t j = i + 1;
    int nb = 0;
    std::vector<uint32_t> inter = {a.seg[0].dst_buf, a.seg[1].dst_buf};
    while (j + 1 < n && nb < 2) {
      const OpDesc& c1 = e->ops[j].d;
      const OpDesc& c2 = e->ops[j + 1].d;
      if (!is_plain_conv(c1, 1, 1, i8) || !is_plain_conv(c2, 3, 1, i8)) break;
      if (c1.res_buf >= 0 || c1.cin != h || c1.seg[0].n_count != h || c1.src_buf != cur_buf || c1.seg[0].src_coff != cur_coff) break;
      if (c2.cin != h || c2.seg[0].n_count != h || c2.src_buf != c1.seg[0].dst_buf || c2.seg[0].src_coff != c1.seg[0].dst_coff) break;
      if (c2.res_buf != (int)cur_buf || c2.res_coff != (int)cur_coff) break;
      inter.push_back(c1.seg[0].dst_buf);
      inter.push_back(c2.seg[0].dst_buf);
      cur_buf = c2.seg[0].dst_buf;
      cur_coff = c2.seg[0].dst_coff;
      j += 2;
      ++nb;
    }
    if (nb < 1 || j >= n) continue;
    const OpDesc& z = e->ops[j].d;  // cv3 over [last bottleneck | cv2]
    if (!is_plain_conv(z, 1, 1, i8) || z.res_buf >= 0 || z.cin != 2 * h || z.seg[0].n_count != 2 * h) continue;
    if (z.src_buf != cur_buf || z.src_buf != a.seg[1].dst_buf || cur_coff != z.seg[0].src_coff || a.seg[1].dst_coff != cur_coff + h) continue;
    if (e->bufs[z.seg[0].dst_buf].d.dtype != bdt || z.seg[0].dst_coff % al || e->bufs[z.seg[0].dst_buf].d.c % al) continue;
    if (a.in_h != z.out_h || a.in_w != z.out_w) continue;
    if (!c3k2_supported((int)h, nb, (int)a.cin, 0, dt)) continue;
    // an FPN block is followed by its lateral conv (model.py:256,259: ConvBlock 1x1, 2h -> h) whose store does the
    // nearest x2 upsample: it becomes the block kernel's last step when that class exists
    size_t jt = j;   // last op of the group incl. the tail
    if (j + 1 < n && !i8) {
      const OpDesc& t = e->ops[j + 1].d;
      if (t.kind == kOpConv && t.ksize == 1 && t.stride == 1 && t.relu && t.nseg == 1 && t.res_buf < 0 && t.seg[0].flags == kSegUp2 &&
          !t.seg[0].m_off && t.seg[0].n_pad == t.seg[0].n_count && t.cin == 2 * h && t.seg[0].n_count == h &&
          t.src_buf == z.seg[0].dst_buf && t.seg[0].src_coff == z.seg[0].dst_coff &&
          e->bufs[t.seg[0].dst_buf].d.dtype == bdt && t.seg[0].dst_coff % 8 == 0 && e->bufs[t.seg[0].dst_buf].d.c % 8 == 0 &&
          t.seg[0].dst_buf != z.seg[0].dst_buf && c3k2_supported((int)h, nb, (int)a.cin, 1, dt))
        jt = j + 1;
    }
    // ... or by a plain 1x1 ConvBlock 2h -> h on its output (stage3_c3k2 -> sppf.cv1, model.py:215-216), fp16 or int8
    int tail_kind = jt > j ? 1 : 0;
    if (jt == j && j + 1 < n) {
      const OpDesc& t = e->ops[j + 1].d;
      if (is_plain_conv(t, 1, 1, i8) && t.res_buf < 0 && t.cin == 2 * h && t.seg[0].n_count == h && t.src_buf == z.seg[0].dst_buf &&
          t.seg[0].src_coff == z.seg[0].dst_coff && e->bufs[t.seg[0].dst_buf].d.dtype == bdt && t.seg[0].dst_coff % al == 0 &&
          e->bufs[t.seg[0].dst_buf].d.c % al == 0 && t.seg[0].dst_buf != z.seg[0].dst_buf && t.in_h == z.out_h && t.in_w == z.out_w &&
          c3k2_supported((int)h, nb, (int)a.cin, 2, dt)) {
        jt = j + 1;
        tail_kind = 2;
      }
    }
    // INT8 engines: an int8 FPN block whose lateral (int8 conv) writes the fp16 concat buffer of a narrow fp16 block
    if (jt == j && i8 && j + 1 < n) {
      const OpDesc& t = e->ops[j + 1].d;
      if (t.kind == kOpConv && t.ksize == 1 && t.stride == 1 && t.relu && t.nseg == 1 && t.res_buf < 0 && t.seg[0].flags == kSegUp2 &&
          t.seg[0].m_off && t.seg[0].n_pad == t.seg[0].n_count && t.cin == 2 * h && t.seg[0].n_count == h &&
          t.src_buf == z.seg[0].dst_buf && t.seg[0].src_coff == z.seg[0].dst_coff &&
          e->bufs[t.seg[0].dst_buf].d.dtype == kBufF16Nhwc && t.seg[0].dst_coff % 8 == 0 && e->bufs[t.seg[0].dst_buf].d.c % 8 == 0 &&
          c3k2_supported((int)h, nb, (int)a.cin, 3, dt)) {
        jt = j + 1;
        tail_kind = 3;
      }
    }
    // INT8 engines: an fp16 block whose output gets an int8 twin from the QUANT op that follows (mixed readers)
    int quant_op = -1;
    if (!i8 && dt == kF16 && jt + 1 < n) {
      const OpDesc& qd = e->ops[jt + 1].d;
      if (qd.kind == kOpQuant && qd.src_buf == z.seg[0].dst_buf && z.seg[0].dst_coff == 0 && e->bufs[z.seg[0].dst_buf].d.c == 2 * h &&
          qd.nseg == 1 && qd.seg[0].dst_coff == 0 && e->bufs[qd.seg[0].dst_buf].d.dtype == kBufI8Nhwc &&
          e->bufs[qd.seg[0].dst_buf].d.c == 2 * h && !e->ops[jt + 1].fuse_role)
        quant_op = (int)(jt + 1);
    }
    // the 3x3 / stride-2 ConvBlock that produces the block's input (stage1_conv -> stage1_block) as a first step, when the
    // block is its only reader (the input tensor then never exists in HBM) and a class exists
    size_t i0 = i;   // first op of the group
    if (i > 0 && !e->ops[i - 1].fuse_role) {
      const OpDesc& pz = e->ops[i - 1].d;
      const uint32_t cx = pz.seg[0].n_count;   // the pre-conv writes the FIRST cx channels of the block's input (all of it, or
                                               // the down-sampling half of a PAN concat)
      bool ok = pz.kind == kOpConv && pz.ksize == 3 && pz.stride == 2 && pz.relu && pz.nseg == 1 && pz.res_buf < 0 && !pz.seg[0].flags &&
                (pz.seg[0].m_off != 0) == i8 && pz.seg[0].n_pad == pz.seg[0].n_count && pz.seg[0].dst_buf == a.src_buf &&
                pz.seg[0].dst_coff == a.seg[0].src_coff && cx <= a.cin && cx % al == 0 &&
                e->bufs[pz.src_buf].d.dtype == bdt && pz.seg[0].src_coff % al == 0 && e->bufs[pz.src_buf].d.c % al == 0 &&
                pz.out_h == a.in_h && pz.out_w == a.in_w && pz.src_buf != a.src_buf &&
                c3k2_supported((int)h, nb, (int)a.cin, tail_kind, dt, (int)pz.cin, (int)cx);
      // its output must be private to the group: nothing else reads or writes those channels
      const Region out{(int)a.src_buf, (int)pz.seg[0].dst_coff, (int)(pz.seg[0].dst_coff + cx)};
      for (size_t k = 0; k < n && ok; ++k) {
        if (k >= i - 1 && k <= j) continue;
        const OpDesc& o = e->ops[k].d;
        if (o.kind == kOpConv || o.kind == kOpQuant || o.kind == kOpSppfPool || o.kind == kOpUpsample)
          for (uint32_t sgi = 0; sgi < o.nseg; ++sgi) {
            const int rc0 = (int)o.seg[sgi].src_coff, rc1 = rc0 + (int)(o.kind == kOpConv ? o.cin : o.seg[sgi].n_count);
            if (overlaps(out, Region{(int)o.src_buf, rc0, o.kind == kOpSppfPool ? (int)e->bufs[o.src_buf].d.c : rc1})) ok = false;
            if (overlaps(out, Region{(int)o.seg[sgi].dst_buf, (int)o.seg[sgi].dst_coff, (int)(o.seg[sgi].dst_coff + o.seg[sgi].n_count)})) ok = false;
          }
        if (o.res_buf == (int)a.src_buf) ok = false;
      }
      if (e->bufs[a.src_buf].d.flags & (kBufInput | kBufOutput)) ok = false;
      if (ok) i0 = i - 1;
    }
    const OpDesc& first = e->ops[i0].d;
    // the group's intermediates must be private to it, and must not be its own input or output
    bool priv = true;
    for (uint32_t b : inter) {
      if (e->bufs[b].d.dtype != bdt) priv = false;   // (an INT8 engine's fp16 conv may still write an int8 buffer)
      if (b == first.src_buf || b == z.seg[0].dst_buf || (e->bufs[b].d.flags & (kBufInput | kBufOutput))) priv = false;
      for (size_t k = 0; k < n && priv; ++k) {
        if (k >= i0 && k <= j) continue;
        const OpDesc& o = e->ops[k].d;
        if (o.src_buf == b || o.res_buf == (int)b) priv = false;
        for (uint32_t s = 0; s < o.nseg; ++s)
          if (o.seg[s].dst_buf == b) priv = false;
      }
    }
    if (!priv) continue;
    // pack
    std::vector<C3k2Conv> convs;
    for (size_t k = i0; k <= jt; ++k) {
      const OpDesc& o = e->ops[k].d;
      C3k2Conv cv;
      memset(&cv, 0, sizeof cv);
      for (uint32_t s = 0; s < o.nseg; ++s) {
        cv.w[s] = reinterpret_cast<const unsigned char*>(blob->data() + o.seg[s].w_off);
        cv.bias[s] = reinterpret_cast<const float*>(blob->data() + o.seg[s].b_off);
        cv.n[s] = (int)o.seg[s].n_count;
        if (i8) {
          cv.mult[s] = reinterpret_cast<const float*>(blob->data() + o.seg[s].m_off);
          cv.out_inv[s] = 1.0f / e->bufs[o.seg[s].dst_buf].d.scale;   // as plan() computes ConvSeg::out_inv_scale
        }
      }
      cv.K = (int)(o.ksize * o.ksize * o.cin);
      convs.push_back(cv);
    }
    std::vector<unsigned char> stream;
    std::vector<float> bias;
    if (!c3k2_pack((int)h, nb, (int)a.cin, tail_kind, convs.data(), &stream, &bias, dt, i0 < i ? (int)first.cin : 0,
                   i0 < i ? (int)first.seg[0].n_count : 0)) continue;
    blob->resize((blob->size() + 255) & ~(size_t)255);
    const uint64_t so = blob->size();
    blob->insert(blob->end(), stream.begin(), stream.end());
    blob->resize((blob->size() + 255) & ~(size_t)255);
    const uint64_t bo = blob->size();
    blob->insert(blob->end(), reinterpret_cast<const char*>(bias.data()), reinterpret_cast<const char*>(bias.data() + bias.size()));
    PlannedOp& head = e->ops[i0];
    head.fuse_role = 1;
    head.fuse_kind = 1;
    head.fuse_pre = i0 < i ? 1 : 0;
    head.group_last = (int)j;
    head.hid = (int)h;
    head.nb = nb;
    head.stream_off = so;
    head.fbias_off = bo;
    head.tail_op = jt > j ? (int)jt : -1;
    head.tail_kind = tail_kind;
    head.quant_op = quant_op;
    if (quant_op >= 0) e->ops[quant_op].fuse_role = 2;
    for (size_t k = i0 + 1; k <= jt; ++k) e->ops[k].fuse_role = 2;
    ++e->n_groups;
    i = jt;
  }
}

// Two consecutive 1x1 ConvBlocks outside any block group (sppf.cv2 -> lateral_p3, whose store does the x2 upsample):
// one launch when conv_pair.hip has a class for the channel counts. fp16 or all-int8.
void find_pair_groups(unina_engine* e, std::vector<char>* blob) {
  const size_t n = e->ops.size();
  for (size_t i = 0; i + 1 < n; ++i) {
    if (e->ops[i].fuse_role || e->ops[i + 1].fuse_role) continue;
    const OpDesc& a = e->ops[i].d;
    const OpDesc& z = e->ops[i + 1].d;
    if (a.kind != kOpConv || z.kind != kOpConv) continue;
    // SPPF: the pool op right in front (x -> [p1 | p2 | p3] next to x in the same buffer) whose 4-way concat `a` reads can
    // run inside the pair's launch when nothing else reads the pooled maps
    int pool = 0;
    if (i > 0 && !e->ops[i - 1].fuse_role && e->ops[i - 1].d.kind == kOpSppfPool) {
      const OpDesc& pl = e->ops[i - 1].d;
      bool ok = pl.src_buf == a.src_buf && pl.seg[0].src_coff == a.seg[0].src_coff && a.cin == 4 * pl.cin;
      const Region pooled{(int)pl.src_buf, (int)(pl.seg[0].src_coff + pl.cin), (int)(pl.seg[0].src_coff + 4 * pl.cin)};
      for (size_t k = 0; k < n && ok; ++k) {
        if (k == i - 1 || k == i) continue;
        const OpDesc& o = e->ops[k].d;
        if (o.kind == kOpStem) continue;
        for (uint32_t sgi = 0; sgi < o.nseg; ++sgi) {
          const int rc0 = (int)o.seg[sgi].src_coff, rc1 = rc0 + (int)(o.kind == kOpConv ? o.cin : o.seg[sgi].n_count);
          if (overlaps(pooled, Region{(int)o.src_buf, rc0, rc1})) ok = false;
          if (overlaps(pooled, Region{(int)o.seg[sgi].dst_buf, (int)o.seg[sgi].dst_coff, (int)(o.seg[sgi].dst_coff + o.seg[sgi].n_count)})) ok = false;
        }
        if (o.res_buf == (int)pl.src_buf) ok = false;
      }
      if (ok) pool = 1;
    }
    const uint32_t bdt = e->bufs[a.src_buf].d.dtype;
    if (bdt != kBufF16Nhwc && bdt != kBufI8Nhwc && bdt != kBufS16Nhwc) continue;
    const bool i8 = bdt == kBufI8Nhwc;
    const uint32_t al = i8 ? 16 : 8;
    if (!is_plain_conv(a, 1, 1, i8) || a.res_buf >= 0) continue;
    if (z.ksize != 1 || z.stride != 1 || !z.relu || z.nseg != 1 || z.res_buf >= 0 || (z.seg[0].flags & ~(uint32_t)kSegUp2) ||
        (z.seg[0].m_off != 0) != i8 || z.seg[0].n_pad != z.seg[0].n_count) continue;
    if (z.src_buf != a.seg[0].dst_buf || z.seg[0].src_coff != a.seg[0].dst_coff || z.cin != a.seg[0].n_count) continue;
    if (e->bufs[a.seg[0].dst_buf].d.dtype != bdt || e->bufs[z.seg[0].dst_buf].d.dtype != bdt) continue;
    if (a.seg[0].src_coff % al || e->bufs[a.src_buf].d.c % al || a.seg[0].dst_coff % al || e->bufs[a.seg[0].dst_buf].d.c % al ||
        z.seg[0].dst_coff % al || e->bufs[z.seg[0].dst_buf].d.c % al) continue;
    if (z.seg[0].dst_buf == a.seg[0].dst_buf || z.seg[0].dst_buf == a.src_buf || a.seg[0].dst_buf == a.src_buf) continue;
    const int dt = i8 ? kI8 : (bdt == kBufS16Nhwc ? kS16 : kF16), up2 = (z.seg[0].flags & kSegUp2) ? 1 : 0;
    if (pool && !pair_supported(dt, (int)a.cin, (int)a.seg[0].n_count, (int)z.seg[0].n_count, up2, 1)) pool = 0;
    if (!pair_supported(dt, (int)a.cin, (int)a.seg[0].n_count, (int)z.seg[0].n_count, up2, pool)) continue;
    C3k2Conv cv[2];
    memset(cv, 0, sizeof cv);
    const OpDesc* od[2] = {&a, &z};
    for (int k = 0; k < 2; ++k) {
      const SegDesc& sg = od[k]->seg[0];
      cv[k].w[0] = reinterpret_cast<const unsigned char*>(blob->data() + sg.w_off);
      cv[k].bias[0] = reinterpret_cast<const float*>(blob->data() + sg.b_off);
      cv[k].n[0] = (int)sg.n_count;
      cv[k].K = (int)od[k]->cin;
      if (i8) {
        cv[k].mult[0] = reinterpret_cast<const float*>(blob->data() + sg.m_off);
        cv[k].out_inv[0] = 1.0f / e->bufs[sg.dst_buf].d.scale;
      }
    }
    std::vector<unsigned char> stream;
    std::vector<float> bias;
    block_pack(cv, 2, &stream, &bias, dt);
    blob->resize((blob->size() + 255) & ~(size_t)255);
    const uint64_t so = blob->size();
    blob->insert(blob->end(), stream.begin(), stream.end());
    blob->resize((blob->size() + 255) & ~(size_t)255);
    const uint64_t bo = blob->size();
    blob->insert(blob->end(), reinterpret_cast<const char*>(bias.data()), reinterpret_cast<const char*>(bias.data() + bias.size()));
    PlannedOp& head = e->ops[i - pool];
    head.fuse_role = 1;
    head.fuse_kind = 4;
    head.fuse_pre = pool;            // 1: this op is the SPPF pool, the conv pair are the next two ops
    head.group_last = (int)(i + 1);
    head.stream_off = so;
    head.fbias_off = bo;
    if (pool) e->ops[i].fuse_role = 2;
    e->ops[i + 1].fuse_role = 2;
    ++e->n_groups;
    ++i;
  }
}

// A DetectionHead (model.py:274-303) in the exporter's table: [cls.0|reg.0 3x3, same input, two slices of h0] ->
// [cls.1|reg.1 3x3, slice i reads h0's slice i, writes h1's slice i] -> [cls.2|reg.2 1x1 without ReLU into the two
// fp32 planar outputs]. Fused when the channel width has a head_fused.hip class and h0 / h1 are private.
void find_head_groups(unina_engine* e, std::vector<char>* blob) {
  const size_t n = e->ops.size();
  for (size_t i = 0; i + 2 < n; ++i) {
    const OpDesc& a = e->ops[i].d;
    const OpDesc& b = e->ops[i + 1].d;
    const OpDesc& c = e->ops[i + 2].d;
    if (e->ops[i].fuse_role || e->ops[i + 1].fuse_role || e->ops[i + 2].fuse_role) continue;
    if (!is_plain_conv(a, 3, 2) || !is_plain_conv(b, 3, 2) || a.res_buf >= 0 || b.res_buf >= 0) continue;
    const uint32_t C = a.cin;
    if (a.seg[0].n_count != C || a.seg[1].n_count != C || a.seg[0].src_coff != a.seg[1].src_coff) continue;
    const uint32_t h0 = a.seg[0].dst_buf;
    if (a.seg[1].dst_buf != h0 || a.seg[0].dst_coff != 0 || a.seg[1].dst_coff != C || e->bufs[h0].d.c != 2 * C) continue;
    if (b.cin != C || b.src_buf != h0 || b.seg[0].src_coff != 0 || b.seg[1].src_coff != C || b.seg[0].n_count != C || b.seg[1].n_count != C) continue;
    const uint32_t h1 = b.seg[0].dst_buf;
    if (b.seg[1].dst_buf != h1 || b.seg[0].dst_coff != 0 || b.seg[1].dst_coff != C || e->bufs[h1].d.c != 2 * C || h1 == h0) continue;
    if (c.kind != kOpConv || c.ksize != 1 || c.stride != 1 || c.relu || c.nseg != 2 || c.res_buf >= 0 || c.cin != C || c.src_buf != h1) continue;
    if (c.seg[0].src_coff != 0 || c.seg[1].src_coff != C) continue;
    bool ok = true;
    for (int s = 0; s < 2; ++s)
      ok = ok && (c.seg[s].flags == kSegPlanarF32) && !c.seg[s].m_off && c.seg[s].n_pad == 16 && c.seg[s].n_count <= 16 &&
           c.seg[s].dst_coff == 0 && e->bufs[c.seg[s].dst_buf].d.dtype == kBufF32Planar;
    if (!ok || e->bufs[a.src_buf].d.dtype != kBufF16Nhwc || !head_supported((int)C)) continue;
    if (a.in_h != c.out_h || a.in_w != c.out_w) continue;
    for (uint32_t hb : {h0, h1}) {
      if (e->bufs[hb].d.dtype != kBufF16Nhwc) ok = false;
      if (hb == a.src_buf || (e->bufs[hb].d.flags & (kBufInput | kBufOutput))) ok = false;
      for (size_t k = 0; k < n && ok; ++k) {
        if (k >= i && k <= i + 2) continue;
        const OpDesc& o = e->ops[k].d;
        if (o.src_buf == hb || o.res_buf == (int)hb) ok = false;
        for (uint32_t s = 0; s < o.nseg; ++s)
          if (o.seg[s].dst_buf == hb) ok = false;
      }
    }
    if (!ok) continue;
    std::vector<C3k2Conv> convs;
    for (size_t k = i; k <= i + 2; ++k) {
      const OpDesc& o = e->ops[k].d;
      C3k2Conv cv;
      memset(&cv, 0, sizeof cv);
      for (uint32_t s = 0; s < 2; ++s) {
        cv.w[s] = reinterpret_cast<const unsigned char*>(blob->data() + o.seg[s].w_off);
        cv.bias[s] = reinterpret_cast<const float*>(blob->data() + o.seg[s].b_off);
        cv.n[s] = (int)o.seg[s].n_pad;
      }
      cv.K = (int)(o.ksize * o.ksize * o.cin);
      convs.push_back(cv);
    }
    std::vector<unsigned char> stream;
    std::vector<float> bias;
    block_pack(convs.data(), 3, &stream, &bias);
    blob->resize((blob->size() + 255) & ~(size_t)255);
    const uint64_t so = blob->size();
    blob->insert(blob->end(), stream.begin(), stream.end());
    blob->resize((blob->size() + 255) & ~(size_t)255);
    const uint64_t bo = blob->size();
    blob->insert(blob->end(), reinterpret_cast<const char*>(bias.data()), reinterpret_cast<const char*>(bias.data() + bias.size()));
    PlannedOp& head = e->ops[i];
    head.fuse_role = 1;
    head.fuse_kind = 2;
    head.group_last = (int)i + 2;
    head.hid = (int)C;
    head.stream_off = so;
    head.fbias_off = bo;
    e->ops[i + 1].fuse_role = e->ops[i + 2].fuse_role = 2;
    ++e->n_groups;
    i += 2;
  }
}

// ---- full-frame graph ------------------------------------------------------------------------------------------------
hipError_t last_captured_node(hipStream_t st, hipGraphNode_t* node) {
  hipStreamCaptureStatus status;
  unsigned long long id = 0;
  hipGraph_t g = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  hipError_t err = hipStreamGetCaptureInfo_v2(st, &status, &id, &g, &deps, &ndeps);
  if (err != hipSuccess) return err;
  if (status != hipStreamCaptureStatusActive || ndeps != 1) return hipErrorInvalidValue;
  *node = deps[0];
  return hipSuccess;
}

int capture_full(unina_engine* e, const PostParams& pp) {
  if (e->fexec) (void)hipGraphExecDestroy(e->fexec);
  if (e->fgraph) (void)hipGraphDestroy(e->fgraph);
  e->fexec = nullptr;
  e->fgraph = nullptr;
  e->stem_node = e->post_node = e->post_node2 = nullptr;
  e->stem_op = -1;
  hipStream_t st = e->capture_stream;
  HIPCHK(e, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  hipError_t err = hipSuccess;
  int rc = UNINA_OK;
  for (size_t j = 0; j < e->ops.size() && err == hipSuccess; ++j) {
    if (pp.mode == 2) {   // the heads' output convs that the decode launch computes itself are left out of the graph
      auto used = [&](int k) {
        for (int h = 0; h < 3; ++h)
          if (e->fold_op[h] == k && pp.h1[h] != nullptr) return true;
        return false;
      };
      const PlannedOp& oj = e->ops[j];
      const bool dual_leader = oj.dual_with >= 0 && !(e->fuse && oj.fuse_role);
      if (used((int)j) && (!dual_leader || used(oj.dual_with))) continue;
    }
    err = launch_op(e, j, st);
    if (err != hipSuccess) {
      rc = fail(e, UNINA_ERR_HIP, "op %zu (%s): %s", j, e->ops[j].d.name, hipGetErrorString(err));
      break;
    }
    if (is_eager(e, j) && e->ops[j].d.kind == kOpStem && e->stem_op < 0) {
      err = last_captured_node(st, &e->stem_node);
      e->stem_op = (int)j;
    }
  }
  LaunchDesc pd[2];
  const int npost = postprocess_desc(pp, pd);
  if (npost < 1) err = hipErrorInvalidValue;
  for (int k = 0; k < npost && err == hipSuccess; ++k) {
    PostParams copy = pp;
    void* args[] = {&copy};
    err = hipLaunchKernel(pd[k].func, pd[k].grid, pd[k].block, args, pd[k].shmem, st);
    if (err == hipSuccess) err = last_captured_node(st, k == 0 ? &e->post_node : &e->post_node2);
  }
  hipError_t end = hipStreamEndCapture(st, &e->fgraph);
  if (rc != UNINA_OK) return rc;
  if (err != hipSuccess) return fail(e, UNINA_ERR_HIP, "full-frame graph capture: %s", hipGetErrorString(err));
  if (end != hipSuccess) return fail(e, UNINA_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(end));
  if (e->stem_op < 0 || !e->stem_node || !e->post_node) return fail(e, UNINA_ERR_STATE, "full-frame graph: stem / post-process node not found");
  HIPCHK(e, hipGraphInstantiate(&e->fexec, e->fgraph, nullptr, nullptr, 0));
  e->f_stem = e->ops[e->stem_op].sp;
  e->f_post = pp;
  return UNINA_OK;
}

template <typename P>
hipError_t set_node(hipGraphExec_t exec, hipGraphNode_t node, const LaunchDesc& d, const P& params) {
  P copy = params;
  void* args[] = {&copy};
  hipKernelNodeParams np;
  memset(&np, 0, sizeof np);
  np.func = const_cast<void*>(d.func);
  np.gridDim = d.grid;
  np.blockDim = d.block;
  np.sharedMemBytes = d.shmem;
  np.kernelParams = args;
  return hipGraphExecKernelNodeSetParams(exec, node, &np);
}

// One graph launch for the whole frame; re-points the stem / post-process nodes when their parameters changed.
int launch_full(unina_engine* e, const PostParams& pp, hipStream_t stream) {
  if (!e->fexec) {
    int rc = capture_full(e, pp);
    if (rc != UNINA_OK) return rc;
  } else {
    const PlannedOp& so = e->ops[e->stem_op];
    const StemParams& sp = so.sp;
    if (memcmp(&sp, &e->f_stem, sizeof sp)) {
      LaunchDesc d;
      {
        HIPCHK(e, stem_desc(sp, &d));
        HIPCHK(e, set_node(e->fexec, e->stem_node, d, sp));
      }
      e->f_stem = sp;
    }
    if (memcmp(&pp, &e->f_post, sizeof pp)) {
      LaunchDesc d[2];
      const int npost = postprocess_desc(pp, d);
      if (npost < 1 || (npost == 2) != (e->post_node2 != nullptr)) return fail(e, UNINA_ERR_STATE, "post-process launch shape changed");
      HIPCHK(e, set_node(e->fexec, e->post_node, d[0], pp));
      if (npost == 2) HIPCHK(e, set_node(e->fexec, e->post_node2, d[1], pp));
      e->f_post = pp;
    }
  }
  HIPCHK(e, hipGraphLaunch(e->fexec, stream));
  return UNINA_OK;
}

// Mean duration of op `i` measured INSIDE the frame sequence: every repetition replays ops 0..i-1 first, so the op
// finds the caches as it does in a real frame (weights cold in L2, inputs just written). Timing back-to-back repeats
// of one launch instead keeps its weights L2-resident and ranks weight-heavy configurations wrongly (measured: a
// configuration that re-reads 2.4 MB of weights from 25 workgroup columns won the warm timing at 13.4 us and ran
// 20.8 us in the frame). `launch` issues op i (so that the autotuner can substitute a candidate configuration).
template <typename F>
int time_in_sequence(unina_engine* e, size_t i, int iters, hipStream_t stream, hipEvent_t a, hipEvent_t b, F launch, float* ms_out) {
  float total = 0.f;
  for (int it = 0; it < iters; ++it) {
    for (size_t k = 0; k < i; ++k) {
      hipError_t err = launch_op(e, k, stream);
      if (err != hipSuccess) return fail(e, UNINA_ERR_HIP, "op %zu (%s): %s", k, e->ops[k].d.name, hipGetErrorString(err));
    }
    HIPCHK(e, hipEventRecord(a, stream));
    HIPCHK(e, launch());
    HIPCHK(e, hipEventRecord(b, stream));
    HIPCHK(e, hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(e, hipEventElapsedTime(&ms, a, b));
    total += ms;
  }
  *ms_out = total / (float)iters;
  return UNINA_OK;
}

int find_buffer(const unina_engine* e, const char* name) {
  for (size_t i = 0; i < e->bufs.size(); ++i)
    if (!strncmp(e->bufs[i].d.name, name, sizeof e->bufs[i].d.name)) return (int)i;
  return -1;
}

float half_bits_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
  uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FF, bits;
  if (exp == 0) {
    if (man == 0) bits = sign;
    else {
      int sh = 0;
      while (!(man & 0x400)) { man <<= 1; ++sh; }
      man &= 0x3FF;
      bits = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (man << 13);
    }
  } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
  else bits = sign | ((exp + 112) << 23) | (man << 13);
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

int fill_post_params(unina_engine* e, PostParams* pp, float conf, float iou, float q, GpuDetection* d_out, int* d_count,
                     int* d_cand_count, bool fold = false) {
  memset(pp, 0, sizeof *pp);
  for (int i = 0; i < 3; ++i) {
    const Buffer& c = e->bufs[e->out_buf[2 * i]];
    const Buffer& r = e->bufs[e->out_buf[2 * i + 1]];
    pp->cls[i] = static_cast<const float*>(c.ptr);
    pp->reg[i] = static_cast<const float*>(r.ptr);
    pp->gw[i] = (int)c.d.w;
    pp->gh[i] = (int)c.d.h;
    pp->stride[i] = (int)e->h.strides[i];
  }
  pp->num_classes = (int)e->h.num_classes;
  pp->conf_thr = conf;
  pp->iou_thr = iou;
  pp->conformal_q = q;
  pp->cand = e->d_cand;
  pp->block_count = e->d_block_count;
  pp->ticket = e->d_ticket;
  pp->mode = e->post_split ? e->post_mode : 0;
  if (pp->mode) post_bind_workspace(pp, e->d_post_ws);
  if (pp->mode == 2) {
    for (int h = 0; h < 3 && fold; ++h) {
      if (e->fold_op[h] < 0) continue;
      const ConvParams& c = e->ops[e->fold_op[h]].cp;
      pp->h1[h] = c.src;
      pp->h1_lo[h] = c.dtype == kS16 ? c.src_lo : 0;
      pp->h1_ld[h] = c.src_ld;
      pp->h1_c[h] = c.Cin;
      for (int k = 0; k < 2; ++k) {
        pp->h1_coff[h][k] = c.seg[k].src_coff;
        pp->w2[h][k] = static_cast<const unsigned char*>(c.seg[k].w_lane);
        pp->b2[h][k] = c.seg[k].bias;
      }
    }
    if (!post_plan_blocks(pp)) {     // does not fit the block table: everything in one launch
      pp->mode = 0;
      pp->ws_box = nullptr;
      for (int h = 0; h < 3; ++h) pp->h1[h] = nullptr;
    }
  }
  pp->out = d_out;
  pp->out_count = d_count;
  pp->out_candidates = d_cand_count;
  pp->stamps = getenv("UNINA_POST_STAMPS") ? reinterpret_cast<long long*>(e->d_result->pad_stamps) : nullptr;
  pp->done_flag = e->done_flag;
  pp->done_value = e->done_value;
  return UNINA_OK;
}

}  // namespace

extern "C" {

#ifndef UNINA_SOURCE_HASH
#define UNINA_SOURCE_HASH "unhashed"
#endif
// "... src:<hash of the sources this binary was built from>" (build.py source_hash(); tests compare it with the tree's)
const char* unina_version(void) { return "unina_mi355 0.3.0 gfx950 src:" UNINA_SOURCE_HASH; }

const char* unina_last_error(const unina_engine_t* e) { return e ? e->err.c_str() : g_load_error.c_str(); }

int unina_load_engine(const char* path, int device_id, unina_engine_t** out) {
  if (!path || !out) return fail(nullptr, UNINA_ERR_ARG, "unina_load_engine: null argument");
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return fail(nullptr, UNINA_ERR_IO, "cannot open %s", path);
  unina_engine* e = new unina_engine();
  e->device = device_id;
  auto bail = [&](int code, const char* msg) {
    g_load_error = msg;
    if (f) fclose(f);
    unina_unload_engine(e);
    return code;
  };
  if (fread(&e->h, sizeof e->h, 1, f) != 1) return bail(UNINA_ERR_FORMAT, "truncated header");
  if (memcmp(e->h.magic, kMagic, 8)) return bail(UNINA_ERR_FORMAT, "bad magic (not a UNINAENG file)");
  if (e->h.version != kVersion) return bail(UNINA_ERR_FORMAT, "unsupported engine file version");
  if (e->h.precision != kFp16 && e->h.precision != kFp32 && e->h.precision != kInt8 && e->h.precision != kSplit16)
    return bail(UNINA_ERR_UNSUPPORTED, "unknown engine precision");
  if (e->h.n_heads != 3 || e->h.n_buffers == 0 || e->h.n_buffers > 4096 || e->h.n_ops == 0 || e->h.n_ops > 4096)
    return bail(UNINA_ERR_FORMAT, "implausible table sizes");
  e->bufs.resize(e->h.n_buffers);
  for (auto& b : e->bufs)
    if (fread(&b.d, sizeof b.d, 1, f) != 1) return bail(UNINA_ERR_FORMAT, "truncated buffer table");
  e->ops.resize(e->h.n_ops);
  for (auto& o : e->ops)
    if (fread(&o.d, sizeof o.d, 1, f) != 1) return bail(UNINA_ERR_FORMAT, "truncated op table");
  std::vector<char> blob(e->h.blob_bytes);
  if (e->h.blob_bytes && fread(blob.data(), 1, blob.size(), f) != blob.size()) return bail(UNINA_ERR_FORMAT, "truncated weight blob");
  fclose(f);
  f = nullptr;
  const size_t file_blob_bytes = blob.size();

  // table validation (indices, offsets) before anything touches the GPU
  for (auto& o : e->ops) {
    o.d.name[sizeof o.d.name - 1] = 0;
    if (o.d.src_buf >= e->h.n_buffers || o.d.nseg < 1 || o.d.nseg > 2 || (o.d.res_buf >= (int)e->h.n_buffers))
      return bail(UNINA_ERR_FORMAT, "op table: bad buffer index");
    for (uint32_t s = 0; s < o.d.nseg; ++s) {
      const SegDesc& sd = o.d.seg[s];
      if (sd.dst_buf >= e->h.n_buffers) return bail(UNINA_ERR_FORMAT, "op table: bad destination buffer");
      if (o.d.kind == kOpConv) {
        const uint32_t sdt = e->bufs[o.d.src_buf].d.dtype;
        const uint64_t wbytes = (uint64_t)sd.n_pad * o.d.ksize * o.d.ksize * o.d.cin * ((sdt == kBufF32Nhwc || sdt == kBufS16Nhwc) ? 4 : (sdt == kBufI8Nhwc ? 1 : 2));
        if (sd.m_off && (sd.m_off + (uint64_t)sd.n_pad * 4 > e->h.blob_bytes || sd.m_off % 16)) return bail(UNINA_ERR_FORMAT, "op table: multiplier offset outside blob");
        if (sd.w_off + wbytes > e->h.blob_bytes || sd.b_off + (uint64_t)sd.n_pad * 4 > e->h.blob_bytes || sd.w_off % 16 || sd.b_off % 16)
          return bail(UNINA_ERR_FORMAT, "op table: weight offset outside blob");
        const BufferDesc& sb = e->bufs[o.d.src_buf].d;
        const BufferDesc& db = e->bufs[sd.dst_buf].d;
        if (sd.src_coff + o.d.cin > sb.c || sb.h != o.d.in_h || sb.w != o.d.in_w) return bail(UNINA_ERR_FORMAT, "op table: source slice outside buffer");
        const uint32_t mul = (sd.flags & kSegUp2) ? 2 : 1;
        if (sd.dst_coff + sd.n_count > db.c || db.h != o.d.out_h * mul || db.w != o.d.out_w * mul) return bail(UNINA_ERR_FORMAT, "op table: destination slice outside buffer");
      }
    }
  }
  for (size_t i = 0; i < e->bufs.size(); ++i) {
    e->bufs[i].d.name[sizeof e->bufs[i].d.name - 1] = 0;
    e->bufs[i].bytes = buffer_bytes(e->bufs[i].d);
    if (e->bufs[i].d.flags & kBufInput) e->images_buf = (int)i;
  }
  static const char* kOut[6] = {"p2_cls", "p2_reg", "p3_cls", "p3_reg", "p4_cls", "p4_reg"};
  for (int i = 0; i < 6; ++i) {
    e->out_buf[i] = find_buffer(e, kOut[i]);
    if (e->out_buf[i] < 0) return bail(UNINA_ERR_FORMAT, "engine file lacks a head output buffer");
  }
  if (e->images_buf < 0) return bail(UNINA_ERR_FORMAT, "engine file lacks the images input buffer");

  // convs: a LANE-order twin of every slice's weight blocks (kernels.h weight_block_to_lane_order) for the kernels that take
  // weights straight into registers (conv3x3_regq / conv3x3_ws / the decode launch's output convs); the file's order stays
  // what the LDS-DMA kernels copy into LDS
  for (auto& op : e->ops) {
    if (op.d.kind != kOpConv) continue;
    const uint32_t sdt = e->bufs[op.d.src_buf].d.dtype;
    for (uint32_t s = 0; s < op.d.nseg; ++s) {
      const SegDesc& sd = op.d.seg[s];
      const uint64_t wbytes = (uint64_t)sd.n_pad * op.d.ksize * op.d.ksize * op.d.cin * ((sdt == kBufF32Nhwc || sdt == kBufS16Nhwc) ? 4 : (sdt == kBufI8Nhwc ? 1 : 2));
      if (wbytes == 0 || wbytes % 1024) continue;
      blob.resize((blob.size() + 255) & ~(size_t)255);
      op.w_lane_off[s] = blob.size();
      blob.resize(blob.size() + wbytes);
      for (uint64_t b = 0; b < wbytes; b += 1024)
        weight_block_to_lane_order(reinterpret_cast<const unsigned char*>(blob.data()) + sd.w_off + b,
                                   reinterpret_cast<unsigned char*>(blob.data()) + op.w_lane_off[s] + b);
    }
  }

  // stem: a [27][Co] transposed copy of its weights (wave-uniform scalar loads in stem_conv_kernel)
  for (auto& op : e->ops) {
    if (op.d.kind != kOpStem) continue;
    const SegDesc& sd = op.d.seg[0];
    const uint32_t co = sd.n_count;
    if (sd.w_off + (uint64_t)co * 27 * 4 > blob.size()) return bail(UNINA_ERR_FORMAT, "stem weights out of range");
    std::vector<float> wt((size_t)27 * co);
    const float* w = reinterpret_cast<const float*>(blob.data() + sd.w_off);
    for (uint32_t c = 0; c < co; ++c)
      for (int k = 0; k < 27; ++k) wt[(size_t)k * co + c] = w[(size_t)c * 27 + k];
    blob.resize((blob.size() + 255) & ~(size_t)255);
    op.wt_off = blob.size();
    blob.insert(blob.end(), reinterpret_cast<const char*>(wt.data()), reinterpret_cast<const char*>(wt.data() + wt.size()));
  }
  // fusable groups: their packed weight streams are appended to the blob before upload. The matchers accept all-fp16
  // groups (in an INT8 engine: the carved-out P2 head, train.py:779) and, for C3k2 blocks, all-int8 groups
  if (e->h.precision == kFp16 || e->h.precision == kInt8 || e->h.precision == kSplit16) {
    find_c3k2_groups(e, &blob);
    find_head_groups(e, &blob);
    find_pair_groups(e, &blob);
    const char* fz = getenv("UNINA_FUSE");
    e->fuse = e->n_groups > 0 && !(fz && fz[0] == '0');
  }
  (void)file_blob_bytes;

  // ---- device side ----
  hipError_t err;
#define LOADCHK(call)                                                            \
  if ((err = (call)) != hipSuccess) {                                            \
    std::string m = std::string(#call) + ": " + hipGetErrorString(err);          \
    return bail(UNINA_ERR_HIP, m.c_str());                                       \
  }
  LOADCHK(hipSetDevice(device_id));
  LOADCHK(hipMalloc(&e->d_blob, blob.size() ? blob.size() : 16));
  LOADCHK(hipMemcpy(e->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
  size_t arena = 0;
  for (auto& b : e->bufs)
    if (!(b.d.flags & kBufInput)) arena += (b.bytes + 255) & ~(size_t)255;
  LOADCHK(conv_init());
  LOADCHK(c3k2_init());
  LOADCHK(head_init());
  LOADCHK(pair_init());
  LOADCHK(block_dual_init());
  LOADCHK(post_init());
  LOADCHK(stem_init());
  LOADCHK(hipMalloc(&e->d_zeros, 256));
  LOADCHK(hipMemset(e->d_zeros, 0, 256));
  LOADCHK(hipMalloc(&e->d_arena, arena ? arena : 256));
  LOADCHK(hipMemset(e->d_arena, 0, arena ? arena : 256));
  size_t off = 0;
  for (auto& b : e->bufs) {
    if (b.d.flags & kBufInput) continue;
    b.owned = static_cast<char*>(e->d_arena) + off;
    b.ptr = b.owned;
    off += (b.bytes + 255) & ~(size_t)255;
  }
  int gw[3], gh[3];
  for (int i = 0; i < 3; ++i) {
    gw[i] = (int)e->bufs[e->out_buf[2 * i]].d.w;
    gh[i] = (int)e->bufs[e->out_buf[2 * i]].d.h;
  }
  e->post_blocks = post_num_blocks(gw, gh);
  if (e->post_blocks > kPostBlock) return bail(UNINA_ERR_UNSUPPORTED, "input too large for the post-process workspace");
  {  // candidate segments: post_blocks x 1024 records (modes 0 / 1) or up to 1024 workgroups x 256 (mode 2)
    const size_t recs = (size_t)e->post_blocks * kPostBlock > (size_t)1024 * kPost2Block ? (size_t)e->post_blocks * kPostBlock : (size_t)1024 * kPost2Block;
    LOADCHK(hipMalloc(&e->d_cand, sizeof(GpuDetection) * recs));
    LOADCHK(hipMalloc(&e->d_block_count, sizeof(int) * (size_t)(e->post_blocks > 1024 ? e->post_blocks : 1024)));
  }
  LOADCHK(hipMalloc(&e->d_ticket, sizeof(unsigned int)));
  LOADCHK(hipMemset(e->d_ticket, 0, sizeof(unsigned int)));
  LOADCHK(hipMalloc(&e->d_post_ws, post_workspace_bytes()));
  LOADCHK(hipMemset(e->d_post_ws, 0, post_workspace_bytes()));
  LOADCHK(hipMalloc(&e->d_result, sizeof(DeviceResult)));
  LOADCHK(hipMemset(e->d_result, 0, sizeof(DeviceResult)));
  LOADCHK(hipHostMalloc(&e->h_result, sizeof(DeviceResult), hipHostMallocMapped));
  memset(e->h_result, 0, sizeof(DeviceResult));
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_result_dev), e->h_result, 0) != hipSuccess) {
    e->h_result_dev = nullptr;
    (void)hipGetLastError();
  }
  LOADCHK(hipStreamCreateWithFlags(&e->capture_stream, hipStreamNonBlocking));
  LOADCHK(hipDeviceSynchronize());
#undef LOADCHK
  const char* ng = getenv("UNINA_NO_GRAPH");
  e->use_graph = !(ng && ng[0] == '1');
  if (const char* fg = getenv("UNINA_FULL_GRAPH")) e->full_graph = fg[0] != '0';
  if (const char* ps = getenv("UNINA_POST_SPLIT")) e->post_split = ps[0] != '0';
  if (const char* pf = getenv("UNINA_POST_FOLD")) e->fold_heads = pf[0] != '0';
  e->plan_dirty = true;
  *out = e;
  return UNINA_OK;
}

void unina_unload_engine(unina_engine_t* e) {
  if (!e) return;
  if (e->t_calls) fprintf(stderr, "unina_infer host split over %ld calls: submit (params + hipGraphLaunch) %.2f us, wait for the completion word %.2f us, record copy %.2f us\n", e->t_calls, e->t_submit_us / e->t_calls, e->t_wait_us / e->t_calls, e->t_copy_us / e->t_calls);
  (void)hipSetDevice(e->device);
  drop_graph(e);
  if (e->capture_stream) (void)hipStreamDestroy(e->capture_stream);
  void* dev[] = {e->d_blob, e->d_arena, e->d_zeros, e->d_cand, e->d_block_count, e->d_ticket, e->d_post_ws, e->d_result};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (e->h_result) (void)hipHostFree(e->h_result);
  delete e;
}

int unina_engine_input_dims(const unina_engine_t* e, int* width, int* height, int* num_classes) {
  if (!e) return UNINA_ERR_ARG;
  if (width) *width = (int)e->h.in_w;
  if (height) *height = (int)e->h.in_h;
  if (num_classes) *num_classes = (int)e->h.num_classes;
  return UNINA_OK;
}

int unina_set_tensor_address(unina_engine_t* e, const char* name, void* device_ptr) {
  if (!e || !name) return UNINA_ERR_ARG;
  const int i = find_buffer(e, name);
  if (i < 0 || !(e->bufs[i].d.flags & (kBufInput | kBufOutput))) return fail(e, UNINA_ERR_ARG, "unknown I/O tensor '%s'", name);
  if (!device_ptr && (e->bufs[i].d.flags & kBufOutput)) device_ptr = e->bufs[i].owned;  // NULL = back to the engine-owned buffer
  if (((uintptr_t)device_ptr) & 15) return fail(e, UNINA_ERR_ARG, "tensor '%s': address must be 16-byte aligned", name);
  if (e->bufs[i].ptr != device_ptr) {
    e->bufs[i].ptr = device_ptr;
    if (i == e->images_buf && !e->plan_dirty) {
      for (size_t k = 0; k < e->ops.size(); ++k)  // only the eager (stem) ops read it: patch them, keep the graph
        if (is_eager(e, k)) e->ops[k].sp.src = static_cast<const float*>(device_ptr);
    } else {
      e->plan_dirty = true;
    }
  }
  return UNINA_OK;
}

int unina_tensor_address(const unina_engine_t* e, const char* name, void** device_ptr, size_t* num_floats) {
  if (!e || !name) return UNINA_ERR_ARG;
  const int i = find_buffer(e, name);
  if (i < 0 || !(e->bufs[i].d.flags & (kBufInput | kBufOutput))) return UNINA_ERR_ARG;
  if (device_ptr) *device_ptr = e->bufs[i].ptr;
  if (num_floats) *num_floats = (size_t)e->bufs[i].d.h * e->bufs[i].d.w * e->bufs[i].d.c;
  return UNINA_OK;
}

int unina_enqueue(unina_engine_t* e, hipStream_t stream) {
  if (!e) return UNINA_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->bufs[e->images_buf].ptr && !e->camera_active) return fail(e, UNINA_ERR_STATE, "tensor 'images' is not bound");
  if (e->plan_dirty) {
    int rc = plan(e);
    if (rc != UNINA_OK) return rc;
  }
  if (!e->use_graph) return launch_all(e, stream);
  if (!e->exec) {
    int rc = capture(e);
    if (rc != UNINA_OK) return rc;
  }
  int rc = launch_all(e, stream, 1);
  if (rc != UNINA_OK) return rc;
  HIPCHK(e, hipGraphLaunch(e->exec, stream));
  return UNINA_OK;
}

int unina_postprocess_async(unina_engine_t* e, float conf, float iou, float q, GpuDetection* d_out, int* d_out_count,
                            hipStream_t stream) {
  if (!e || !d_out || !d_out_count) return UNINA_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  PostParams pp;
  fill_post_params(e, &pp, conf, iou, q, d_out, d_out_count, &e->d_result->candidates);
  HIPCHK(e, postprocess_launch(pp, stream));
  return UNINA_OK;
}

int unina_infer_async(unina_engine_t* e, const float* d_images, float conf, float iou, float q, GpuDetection* d_out,
                      int* d_out_count, hipStream_t stream) {
  if (!e) return UNINA_ERR_ARG;
  if (d_images) {
    int rc = unina_set_tensor_address(e, "images", const_cast<float*>(d_images));
    if (rc != UNINA_OK) return rc;
  }
  if (e->full_graph && e->use_graph) {
    if (!d_out || !d_out_count) return UNINA_ERR_ARG;
    HIPCHK(e, hipSetDevice(e->device));
    if (!e->bufs[e->images_buf].ptr && !e->camera_active) return fail(e, UNINA_ERR_STATE, "tensor 'images' is not bound");
    if (e->plan_dirty) {
      int rc = plan(e);
      if (rc != UNINA_OK) return rc;
    }
    PostParams pp;
    fill_post_params(e, &pp, conf, iou, q, d_out, d_out_count, &e->d_result->candidates, /*fold=*/true);
    return launch_full(e, pp, stream);
  }
  int rc = unina_enqueue(e, stream);
  if (rc != UNINA_OK) return rc;
  return unina_postprocess_async(e, conf, iou, q, d_out, d_out_count, stream);
}

int unina_infer(unina_engine_t* e, const float* d_images, float conf, float iou, float q, GpuDetection* out,
                int* out_count, hipStream_t stream) {
  if (!e || !out || !out_count) return UNINA_ERR_ARG;
  // The post-process writes its compacted output (write-only: count + n records) straight into the pinned, device-mapped
  // host block: no D2H copy command (a blit-kernel launch of its own, and all 1024 slots) on the latency path.
  // UNINA_HOST_RESULT=0 restores the device buffer + copy.
  static const bool host_result = !(getenv("UNINA_HOST_RESULT") && getenv("UNINA_HOST_RESULT")[0] == '0');
  // ... and the host does not wait for the stream either: the last block stores this call's sequence number behind the
  // records (system-scope release) and the host spins on that word -- the runtime's completion signal of the frame graph
  // arrives microseconds later. UNINA_HOST_POLL=0 waits with hipStreamSynchronize instead.
  static const bool host_poll = !(getenv("UNINA_HOST_POLL") && getenv("UNINA_HOST_POLL")[0] == '0');
  bool copied = false;
  if (host_result && e->h_result_dev) {
    unsigned int seq = 0;
    if (host_poll) {
      seq = ++e->result_seq ? e->result_seq : ++e->result_seq;   // never 0
      e->done_flag = &e->h_result_dev->seq;
      e->done_value = seq;
    }
    static const bool timing = getenv("UNINA_TIMING") != nullptr;
    const auto ta = std::chrono::steady_clock::now();
    int rc = unina_infer_async(e, d_images, conf, iou, q, e->h_result_dev->det, &e->h_result_dev->count, stream);
    e->done_flag = nullptr;
    if (rc != UNINA_OK) return rc;
    const auto tb = std::chrono::steady_clock::now();
    if (host_poll) {
      volatile unsigned int* flag = &e->h_result->seq;
      const auto t0 = std::chrono::steady_clock::now();
      bool seen = false;
      for (unsigned long it = 1;; ++it) {
        if (*flag == seq) { seen = true; break; }
        __builtin_ia32_pause();
        if ((it & 0xFFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      if (!seen) HIPCHK(e, hipStreamSynchronize(stream));   // (a faulted launch reports here)
    } else {
      HIPCHK(e, hipStreamSynchronize(stream));
    }
    if (timing) {
      const auto tc = std::chrono::steady_clock::now();
      const int n0 = e->h_result->count;
      if (n0 >= 0 && n0 <= MAX_DETECTIONS) {
        memcpy(out, e->h_result->det, sizeof(GpuDetection) * (size_t)n0);
        copied = true;
      }
      const auto td = std::chrono::steady_clock::now();
      e->t_submit_us += std::chrono::duration<double, std::micro>(tb - ta).count();
      e->t_wait_us += std::chrono::duration<double, std::micro>(tc - tb).count();
      e->t_copy_us += std::chrono::duration<double, std::micro>(td - tc).count();
      ++e->t_calls;
    }
  } else {
    int rc = unina_infer_async(e, d_images, conf, iou, q, e->d_result->det, &e->d_result->count, stream);
    if (rc != UNINA_OK) return rc;
    HIPCHK(e, hipMemcpyAsync(e->h_result, e->d_result, sizeof(DeviceResult), hipMemcpyDeviceToHost, stream));
    HIPCHK(e, hipStreamSynchronize(stream));
  }
  int n = e->h_result->count;
  if (n < 0 || n > MAX_DETECTIONS) return fail(e, UNINA_ERR_STATE, "post-process returned count %d", n);
  if (!copied) memcpy(out, e->h_result->det, sizeof(GpuDetection) * (size_t)n);   // (the UNINA_TIMING branch has copied them already)
  *out_count = n;
  return UNINA_OK;
}

// `n_calls` serial unina_infer calls over a ring of `n_ring` input frames, each timed on the host's steady clock from entry to
// return (detections copied out): the latency a C / C++ caller of this ABI sees, without a binding layer's per-call cost
// (the ctypes path adds ~12 us). lat_us[n_calls].
int unina_serial_latency(unina_engine_t* e, const float* const* d_frames, int n_ring, int n_calls, float conf, float iou, float q,
                         double* lat_us, hipStream_t stream) {
  if (!e || !d_frames || !lat_us || n_ring < 1 || n_calls < 1) return UNINA_ERR_ARG;
  std::vector<GpuDetection> out(MAX_DETECTIONS);
  int n = 0;
  for (int i = 0; i < n_calls; ++i) {
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = unina_infer(e, d_frames[i % n_ring], conf, iou, q, out.data(), &n, stream);
    lat_us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (rc != UNINA_OK) return rc;
  }
  return UNINA_OK;
}

// Camera frame -> detections: unina_infer with the pre-process (preprocess.hip: BGRA -> RGB, optional half-pixel-centre
// bilinear resize, normalise) computed inside the stem kernel instead of written to an fp32 tensor by one launch and
// read back by the next (4 B/px in instead of 12 B/px out + 12 B/px in, one launch less). Same arithmetic, so the
// detections are those of preprocess_bgra[_resize] + unina_infer bit for bit (tests/test_gpu_preprocess.py).
int unina_infer_bgra(unina_engine_t* e, const uint8_t* d_bgra, int src_width, int src_height, int src_pitch,
                     const NormParams* norm, float conf, float iou, float q, GpuDetection* out, int* out_count,
                     hipStream_t stream) {
  if (!e || !d_bgra || !norm || !out || !out_count) return UNINA_ERR_ARG;
  if (src_width <= 0 || src_height <= 0 || src_pitch < 4 * src_width || (src_pitch & 3) || ((uintptr_t)d_bgra & 3))
    return fail(e, UNINA_ERR_ARG, "unina_infer_bgra: bad frame geometry");
  HIPCHK(e, hipSetDevice(e->device));
  if (e->plan_dirty) {
    int rc = plan(e);
    if (rc != UNINA_OK) return rc;
  }
  int nstem = 0;
  for (size_t k = 0; k < e->ops.size(); ++k) {
    PlannedOp& op = e->ops[k];
    if (!is_eager(e, k) || op.d.kind != kOpStem) continue;
    op.sp.src_kind = (src_width == op.sp.W && src_height == op.sp.H) ? 1 : 2;
    op.sp.cam = d_bgra;
    op.sp.cam_w = src_width;
    op.sp.cam_h = src_height;
    op.sp.cam_pitch = src_pitch;
    op.sp.norm = *norm;
    ++nstem;
  }
  if (!nstem) return fail(e, UNINA_ERR_STATE, "no stem op reads the input tensor");
  e->camera_active = true;
  const int rc = unina_infer(e, nullptr, conf, iou, q, out, out_count, stream);
  e->camera_active = false;
  for (size_t k = 0; k < e->ops.size(); ++k) {
    PlannedOp& op = e->ops[k];
    if (!is_eager(e, k) || op.d.kind != kOpStem) continue;
    op.sp.src_kind = 0;
    op.sp.cam = nullptr;
    op.sp.cam_w = op.sp.cam_h = op.sp.cam_pitch = 0;
    memset(&op.sp.norm, 0, sizeof op.sp.norm);
  }
  return rc;
}

int unina_debug_post_stamps(unina_engine_t* e, long long* out8) {
  if (!e || !out8) return UNINA_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, hipDeviceSynchronize());
  HIPCHK(e, hipMemcpy(out8, e->d_result->pad_stamps, sizeof(long long) * 16, hipMemcpyDeviceToHost));
  return UNINA_OK;
}

// Debug: runs conv op `op_index` once with in-kernel s_memtime stamps on (one workgroup in the middle of the grid):
// out5 = shader-clock ticks at start / prologue issued / first data usable / K loop done / stores drained.
int unina_debug_conv_stamps(unina_engine_t* e, int op_index, long long* out5, hipStream_t stream) {
  if (!e || !out5 || op_index < 0 || op_index >= (int)e->ops.size()) return UNINA_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  if (e->plan_dirty) {
    int rc = plan(e);
    if (rc != UNINA_OK) return rc;
  }
  if (e->ops[op_index].d.kind != kOpConv) return fail(e, UNINA_ERR_ARG, "op %d is not a convolution", op_index);
  ConvParams p = e->ops[op_index].cp;
  p.stamps = reinterpret_cast<long long*>(e->d_result->pad_stamps);
  HIPCHK(e, hipMemsetAsync(p.stamps, 0, sizeof(long long) * 8, stream));
  HIPCHK(e, conv_launch(p, e->ops[op_index].cl, stream));
  HIPCHK(e, hipStreamSynchronize(stream));
  HIPCHK(e, hipMemcpy(out5, p.stamps, sizeof(long long) * 8, hipMemcpyDeviceToHost));  // 5 shader-clock stamps + 2 at 100 MHz + entry
  return UNINA_OK;
}

// One launch of the DUAL conv launch led by op `op_index` (fusion on: e.g. the P3 | P4 head layers) with in-kernel stamps of
// the mid workgroup of each of its two convs: out16[0..7] conv A, out16[8..15] conv B, each as unina_debug_conv_stamps.
int unina_debug_dual_stamps(unina_engine_t* e, int op_index, long long* out16, hipStream_t stream) {
  if (!e || !out16 || op_index < 0 || op_index >= (int)e->ops.size()) return UNINA_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  if (e->plan_dirty) {
    int rc = plan(e);
    if (rc != UNINA_OK) return rc;
  }
  const PlannedOp& op = e->ops[op_index];
  if (op.d.kind != kOpConv || op.fuse_role || op.dual_with < 0 || op.dual_absorbed)
    return fail(e, UNINA_ERR_ARG, "op %d does not lead a dual conv launch", op_index);
  ConvParams pa = op.cp, pb = e->ops[op.dual_with].cp;
  pa.stamps = reinterpret_cast<long long*>(e->d_result->pad_stamps);
  pb.stamps = pa.stamps + 8;
  HIPCHK(e, hipMemsetAsync(pa.stamps, 0, sizeof(long long) * 16, stream));
  HIPCHK(e, conv_dual_launch(op.dual_kind, pa, pb, stream));
  HIPCHK(e, hipStreamSynchronize(stream));
  HIPCHK(e, hipMemcpy(out16, pa.stamps, sizeof(long long) * 16, hipMemcpyDeviceToHost));
  return UNINA_OK;
}

// Debug: the fused C3k2 block led by op `op_index`, launched once as its stamped twin right after the ops in front of it (cold
// weights, as in a frame): out16[k] = shader-clock stamp of the mid workgroup after step k (block_kernels.h), [14] / [15] =
// the 100 MHz wall clock at its end / entry. Only the 40^2 blocks have twins.
int unina_debug_block_stamps(unina_engine_t* e, int op_index, long long* out16, hipStream_t stream) {
  if (!e || !out16 || op_index < 0 || op_index >= (int)e->ops.size()) return UNINA_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  if (e->plan_dirty) {
    int rc = plan(e);
    if (rc != UNINA_OK) return rc;
  }
  const PlannedOp& op = e->ops[op_index];
  if (!e->fuse || op.fuse_role != 1 || op.fuse_kind != 1) return fail(e, UNINA_ERR_ARG, "op %d does not lead a fused C3k2 block", op_index);
  for (int i = 0; i < op_index; ++i) HIPCHK(e, launch_op(e, (size_t)i, stream));
  C3k2Params p = op.fp;
  p.stamps = reinterpret_cast<long long*>(e->d_result->pad_stamps);
  HIPCHK(e, hipMemsetAsync(p.stamps, 0, sizeof(long long) * 16, stream));
  HIPCHK(e, c3k2_launch_stamped(p, stream));
  HIPCHK(e, hipStreamSynchronize(stream));
  HIPCHK(e, hipMemcpy(out16, p.stamps, sizeof(long long) * 16, hipMemcpyDeviceToHost));
  return UNINA_OK;
}

namespace unina {
__global__ void debug_stamp_kernel(long long* t) { if (threadIdx.x == 0) *t = wall_clock64(); }
}
// Same launch with EVERY workgroup's start / end on the 100 MHz wall clock: out[2*i], out[2*i+1] for workgroup i in BLOCK-ID
// order (the weights-stationary pairs put conv B's workgroups first, the register-queue pairs conv A's), then the clock of a
// one-wave marker kernel enqueued right before and of one right after the launch (dispatch gaps); cap >= 2 * grid + 2
// values. Returns the grid size (> 0) or a negative error code.
int unina_debug_dual_timeline(unina_engine_t* e, int op_index, long long* out, int cap, hipStream_t stream) {
  if (!e || !out || op_index < 0 || op_index >= (int)e->ops.size()) return -UNINA_ERR_ARG;
  if (hipSetDevice(e->device) != hipSuccess) return -UNINA_ERR_HIP;
  if (e->plan_dirty) {
    int rc = plan(e);
    if (rc != UNINA_OK) return -rc;
  }
  const PlannedOp& op = e->ops[op_index];
  if (op.d.kind != kOpConv || op.fuse_role || op.dual_with < 0 || op.dual_absorbed) return -UNINA_ERR_ARG;
  ConvParams pa = op.cp, pb = e->ops[op.dual_with].cp;
  // every workgroup writes wg_times[2 * blockIdx.x ..]: the buffer is sized from the grid, known BEFORE anything is launched
  const int grid = conv_dual_grid(op.dual_kind, pa, pb);
  if (grid < 1 || 2 * grid + 2 > cap) return -UNINA_ERR_ARG;
  long long* d = nullptr;
  const size_t words = 18 + 2 * (size_t)grid;
  if (hipMalloc(&d, sizeof(long long) * words) != hipSuccess) return -UNINA_ERR_HIP;
  long long* marks = d + 16 + 2 * grid;
  pa.stamps = d;
  pb.stamps = d + 8;
  pa.wg_times = pb.wg_times = d + 16;
  int launched = 0;
  hipError_t he = hipMemsetAsync(d, 0, sizeof(long long) * words, stream);
  if (he == hipSuccess) {
    hipLaunchKernelGGL(unina::debug_stamp_kernel, dim3(1), dim3(64), 0, stream, marks);
    he = hipGetLastError();
  }
  if (he == hipSuccess) he = conv_dual_launch(op.dual_kind, pa, pb, stream, &launched);
  if (he == hipSuccess) {
    hipLaunchKernelGGL(unina::debug_stamp_kernel, dim3(1), dim3(64), 0, stream, marks + 1);
    he = hipGetLastError();
  }
  if (he == hipSuccess) he = hipStreamSynchronize(stream);
  if (he == hipSuccess && launched != grid) he = hipErrorInvalidValue;     // (the planner's grid is the launch's)
  if (he == hipSuccess) he = hipMemcpy(out, d + 16, sizeof(long long) * 2 * grid, hipMemcpyDeviceToHost);
  if (he == hipSuccess) he = hipMemcpy(out + 2 * grid, marks, sizeof(long long) * 2, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  return he == hipSuccess ? grid : -UNINA_ERR_HIP;
}

int unina_op_count(const unina_engine_t* e) { return e ? (int)e->ops.size() : -1; }

int unina_get_op_info(const unina_engine_t* ce, int index, unina_op_info* info) {
  unina_engine* e = const_cast<unina_engine*>(ce);
  if (!e || !info || index < 0 || index >= (int)e->ops.size()) return UNINA_ERR_ARG;
  if (e->plan_dirty) {
    // planning needs addresses only for pointers; shapes/flops are valid regardless
    int rc = plan(e);
    if (rc != UNINA_OK) return rc;
  }
  *info = e->ops[index].info;
  return UNINA_OK;
}

// Load-time analysis only (no device is touched): how many C3k2 blocks of the engine file would run fused.
int unina_debug_fusable_groups(const char* path) {
  if (!path) return -UNINA_ERR_ARG;
  FILE* f = fopen(path, "rb");
  if (!f) return -UNINA_ERR_IO;
  unina_engine e;
  bool ok = fread(&e.h, sizeof e.h, 1, f) == 1 && !memcmp(e.h.magic, kMagic, 8) && e.h.version == kVersion &&
            e.h.n_buffers && e.h.n_buffers <= 4096 && e.h.n_ops && e.h.n_ops <= 4096;
  std::vector<char> blob;
  if (ok) {
    e.bufs.resize(e.h.n_buffers);
    e.ops.resize(e.h.n_ops);
    for (auto& b : e.bufs) ok = ok && fread(&b.d, sizeof b.d, 1, f) == 1;
    for (auto& o : e.ops) ok = ok && fread(&o.d, sizeof o.d, 1, f) == 1;
    blob.resize(e.h.blob_bytes);
    ok = ok && (!e.h.blob_bytes || fread(blob.data(), 1, blob.size(), f) == blob.size());
  }
  fclose(f);
  if (!ok) return -UNINA_ERR_FORMAT;
  for (auto& o : e.ops) {
    if (o.d.src_buf >= e.h.n_buffers || o.d.nseg < 1 || o.d.nseg > 2 || o.d.res_buf >= (int)e.h.n_buffers) return -UNINA_ERR_FORMAT;
    for (uint32_t s = 0; s < o.d.nseg; ++s)
      if (o.d.seg[s].dst_buf >= e.h.n_buffers || o.d.seg[s].w_off > blob.size() || o.d.seg[s].b_off > blob.size()) return -UNINA_ERR_FORMAT;
  }
  if (e.h.precision == kFp16 || e.h.precision == kInt8) {
    find_c3k2_groups(&e, &blob);
    find_head_groups(&e, &blob);
    find_pair_groups(&e, &blob);
  }
  return e.n_groups;
}

int unina_set_fusion(unina_engine_t* e, int enable) {
  if (!e) return UNINA_ERR_ARG;
  const bool on = enable && e->n_groups > 0;
  if (on != e->fuse) {
    e->fuse = on;
    e->plan_dirty = true;
  }
  return UNINA_OK;
}

int unina_fusion_groups(const unina_engine_t* e) { return e ? (e->fuse ? e->n_groups : 0) : -1; }

int unina_conv_config_count(void) { return (int)kCfgCount; }

const char* unina_conv_config_name(int cfg) { return conv_config_name(cfg, kF16); }

int unina_set_op_config(unina_engine_t* e, int op_index, int cfg) {
  if (!e || op_index < 0 || op_index >= (int)e->ops.size()) return UNINA_ERR_ARG;
  if (e->plan_dirty) {
    int rc = plan(e);
    if (rc != UNINA_OK) return rc;
  }
  if (e->ops[op_index].d.kind != kOpConv) return fail(e, UNINA_ERR_ARG, "op %d is not a convolution", op_index);
  if (cfg >= 0 && !conv_config_valid(e->ops[op_index].cp, cfg)) return fail(e, UNINA_ERR_UNSUPPORTED, "config %d does not fit op %d", cfg, op_index);
  if (e->force_cfg.size() < e->ops.size()) e->force_cfg.assign(e->ops.size(), -1);
  e->force_cfg[op_index] = cfg;
  e->plan_dirty = true;
  return UNINA_OK;
}

// Tactic selection by timing, like the reference's TensorRT build step (export_trt.py:459-468) does on the target:
// every conv op is timed with every tile configuration that fits it and keeps the fastest. Results do not change
// (each output element accumulates its K terms in the same order under every configuration).
int unina_autotune(unina_engine_t* e, int iters, hipStream_t stream) {
  if (!e || iters < 1) return UNINA_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->bufs[e->images_buf].ptr && !e->camera_active) return fail(e, UNINA_ERR_STATE, "tensor 'images' is not bound");
  if (e->plan_dirty) {
    int rc = plan(e);
    if (rc != UNINA_OK) return rc;
  }
  int rc = launch_all(e, stream);
  if (rc != UNINA_OK) return rc;
  if (e->force_cfg.size() < e->ops.size()) e->force_cfg.assign(e->ops.size(), -1);
  hipEvent_t a, b;
  HIPCHK(e, hipEventCreate(&a));
  HIPCHK(e, hipEventCreate(&b));
  for (size_t i = 0; i < e->ops.size(); ++i) {
    if (e->ops[i].d.kind != kOpConv || (e->fuse && e->ops[i].fuse_role) || e->ops[i].dual_with >= 0 || e->ops[i].dual_absorbed) continue;
    float best = 1e30f;
    int best_cfg = -1;
    for (int cfg = 0; cfg < (int)kCfgCount; ++cfg) {
      if (!conv_config_valid(e->ops[i].cp, cfg)) continue;
      const ConvLaunch l = conv_plan_with(e->ops[i].cp, cfg);
      float ms = 0.f;
      rc = time_in_sequence(e, i, iters, stream, a, b, [&]() { return conv_launch(e->ops[i].cp, l, stream); }, &ms);
      if (rc != UNINA_OK) return rc;
      if (ms < best) {
        best = ms;
        best_cfg = cfg;
      }
    }
    e->force_cfg[i] = best_cfg;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  e->plan_dirty = true;
  return UNINA_OK;
}

int unina_profile_ops(unina_engine_t* e, int iters, float* ms_per_op, hipStream_t stream) {
  if (!e || !ms_per_op || iters < 1) return UNINA_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->bufs[e->images_buf].ptr && !e->camera_active) return fail(e, UNINA_ERR_STATE, "tensor 'images' is not bound");
  if (e->plan_dirty) {
    int rc = plan(e);
    if (rc != UNINA_OK) return rc;
  }
  int rc = launch_all(e, stream);  // warm-up: every buffer holds real activations
  if (rc != UNINA_OK) return rc;
  hipEvent_t a, b;
  HIPCHK(e, hipEventCreate(&a));
  HIPCHK(e, hipEventCreate(&b));
  PostParams pp;   // which head output convs the frame's decode launch computes itself (they are not launched in a frame)
  fill_post_params(e, &pp, 0.5f, 0.45f, 0.1f, e->d_result->det, &e->d_result->count, &e->d_result->candidates, e->full_graph && e->use_graph);
  auto folded = [&](int k) {
    for (int h = 0; h < 3; ++h)
      if (pp.mode == 2 && e->fold_op[h] == k && pp.h1[h] != nullptr) return true;
    return false;
  };
  for (size_t i = 0; i < e->ops.size(); ++i) {
    const PlannedOp& op = e->ops[i];
    const bool dual_leader = op.dual_with >= 0 && !(e->fuse && op.fuse_role);
    if ((e->fuse && op.fuse_role == 2) || op.dual_absorbed || (folded((int)i) && (!dual_leader || folded(op.dual_with)))) {
      ms_per_op[i] = 0.f;
      continue;
    }
    rc = time_in_sequence(e, i, iters, stream, a, b, [&]() { return launch_op(e, i, stream); }, &ms_per_op[i]);
    if (rc != UNINA_OK) return rc;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return UNINA_OK;
}

// The post-process launches of a frame, timed like unina_profile_ops (HIP events on `stream`, each repetition replays the
// whole forward first): ms2[0] = decode launch (with the folded head output convs), ms2[1] = pair tiles + scan + output
// (0 when the post-process is one launch). Thresholds as given.
int unina_profile_post(unina_engine_t* e, int iters, float conf, float iou, float q, float* ms2, hipStream_t stream) {
  if (!e || !ms2 || iters < 1) return UNINA_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->bufs[e->images_buf].ptr && !e->camera_active) return fail(e, UNINA_ERR_STATE, "tensor 'images' is not bound");
  if (e->plan_dirty) {
    int rc = plan(e);
    if (rc != UNINA_OK) return rc;
  }
  PostParams pp;
  fill_post_params(e, &pp, conf, iou, q, e->d_result->det, &e->d_result->count, &e->d_result->candidates, true);
  pp.done_flag = nullptr;
  LaunchDesc d[2];
  const int npost = postprocess_desc(pp, d);
  if (npost < 1) return fail(e, UNINA_ERR_STATE, "post-process launch shape");
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  ms2[0] = ms2[1] = 0.f;
  // (the events are destroyed on every path out of here)
  auto run = [&]() -> int {
    for (auto& x : ev) HIPCHK(e, hipEventCreate(&x));
    for (int it = 0; it < iters; ++it) {
      int rc = launch_all(e, stream);
      if (rc != UNINA_OK) return rc;
      for (int k = 0; k < npost; ++k) {
        PostParams copy = pp;
        void* args[] = {&copy};
        HIPCHK(e, hipEventRecord(ev[k], stream));
        HIPCHK(e, hipLaunchKernel(d[k].func, d[k].grid, d[k].block, args, d[k].shmem, stream));
      }
      HIPCHK(e, hipEventRecord(ev[npost], stream));
      HIPCHK(e, hipEventSynchronize(ev[npost]));
      for (int k = 0; k < npost; ++k) {
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
        ms2[k] += ms / (float)iters;
      }
    }
    return UNINA_OK;
  };
  const int rc = run();
  for (auto& x : ev)
    if (x) (void)hipEventDestroy(x);
  return rc;
}

int unina_debug_read_buffer(unina_engine_t* e, const char* name, float* host_out, size_t capacity, int* c, int* h, int* w) {
  if (!e || !name) return UNINA_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  const int i = find_buffer(e, name);
  if (i < 0) return fail(e, UNINA_ERR_ARG, "unknown buffer '%s'", name);
  const Buffer& b = e->bufs[i];
  const size_t n = (size_t)b.d.h * b.d.w * b.d.c;
  if (c) *c = (int)b.d.c;
  if (h) *h = (int)b.d.h;
  if (w) *w = (int)b.d.w;
  if (capacity < n) return fail(e, UNINA_ERR_ARG, "buffer '%s' needs %zu floats", name, n);  // dims are still reported
  HIPCHK(e, hipDeviceSynchronize());
  if (b.d.dtype == kBufI8Nhwc) {  // returned as real values (code * scale)
    std::vector<signed char> tmp(n);
    HIPCHK(e, hipMemcpy(tmp.data(), b.ptr, n, hipMemcpyDeviceToHost));
    const size_t hw = (size_t)b.d.h * b.d.w;
    for (size_t p = 0; p < hw; ++p)
      for (size_t ch = 0; ch < b.d.c; ++ch) host_out[ch * hw + p] = (float)tmp[p * b.d.c + ch] * b.d.scale;
  } else if (b.d.dtype == kBufF32Nhwc) {
    std::vector<float> tmp(n);
    HIPCHK(e, hipMemcpy(tmp.data(), b.ptr, n * 4, hipMemcpyDeviceToHost));
    const size_t hw = (size_t)b.d.h * b.d.w;
    for (size_t p = 0; p < hw; ++p)
      for (size_t ch = 0; ch < b.d.c; ++ch) host_out[ch * hw + p] = tmp[p * b.d.c + ch];
  } else if (b.d.dtype == kBufF16Nhwc) {
    std::vector<uint16_t> tmp(n);
    HIPCHK(e, hipMemcpy(tmp.data(), b.ptr, n * 2, hipMemcpyDeviceToHost));
    const size_t hw = (size_t)b.d.h * b.d.w;
    for (size_t p = 0; p < hw; ++p)
      for (size_t ch = 0; ch < b.d.c; ++ch) host_out[ch * hw + p] = half_bits_to_float(tmp[p * b.d.c + ch]);
  } else if (b.d.dtype == kBufS16Nhwc) {   // value = hi + lo
    std::vector<uint16_t> tmp(2 * n);
    HIPCHK(e, hipMemcpy(tmp.data(), b.ptr, n * 4, hipMemcpyDeviceToHost));
    const size_t hw = (size_t)b.d.h * b.d.w;
    for (size_t p = 0; p < hw; ++p)
      for (size_t ch = 0; ch < b.d.c; ++ch)
        host_out[ch * hw + p] = half_bits_to_float(tmp[p * b.d.c + ch]) + half_bits_to_float(tmp[n + p * b.d.c + ch]);
  } else {
    HIPCHK(e, hipMemcpy(host_out, b.ptr, n * 4, hipMemcpyDeviceToHost));
  }
  return UNINA_OK;
}

}  // extern "C"
