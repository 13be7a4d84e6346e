// postprocess.hip -- the detection post-process on the GPU (gfx950).
//
// (1) postprocess_kernel: the WHOLE post-process as ONE launch. Replaces, per frame, the reference's
//     reset_detection_counter + 3 x decode_yolo_head_kernel + get_detection_count (host sync) + thrust::sort +
//     nms_kernel + cub::DeviceSelect::If (+2 host syncs)
//     (ros2_ws/src/perception/src/gpu_postprocess.cu:102-199, 207-251, 332-442; call sites
//     perception_node.cpp:627-656) with a single launch and no host round-trip.
// (2) the reference's seven-function C API (gpu_postprocess.h:42-80) on top of the same device code, so the
//     node's calling sequence links unchanged.
//
// Semantics (oracle: oracle/postprocess_oracle.c with uo_semantics_engine()):
//   decode   conf = 1/(1+expf(-logit)); first-max-wins argmax from (0.0,-1); keep conf >= thr
//            (gpu_postprocess.cu:118-132); box = ((x+.5)s - l*s, (y+.5)s - t*s, (x+.5)s + r*s, (y+.5)s + b*s)
//            (:141-152); conformal dilation with the pre-dilation w,h (:155-162)
//   order    candidates enumerated P2->P3->P4, row-major; if more than MAX_DETECTIONS pass, the fused kernel
//            keeps the MAX_DETECTIONS highest confidences (ties: enumeration order) -- a deterministic
//            refinement of the reference's atomic-order cap (:178-197)
//   sort     confidence descending, stable
//   NMS      sequential greedy, class-aware, IoU > thr with +1e-6f in the denominator (:69-83),
//            only a strictly higher confidence suppresses (:224)
//   output   compacted, sorted, valid=1, _pad=0
//
// Structure of (1): every block decodes 1024 cells and writes its survivors, in order, to its own segment of
// the workspace; the block that draws the last arrival ticket (agent-scope release/acquire hand-off) gathers
// the segments in enumeration order, sorts in LDS (bitonic, 64-bit keys), runs the NMS with 64x1024-bit
// suppression masks per 64-row chunk, and compacts with wave ballots.
// Built with -ffp-contract=off: the box arithmetic must round exactly like the reference's scalar code.
#include "kernels.h"

#pragma clang fp contract(off)

namespace unina {

namespace {

constexpr int kT = kPostBlock;  // threads per block == cells per block
constexpr int kWaves = kT / 64;
constexpr int kMaxDet = MAX_DETECTIONS;
constexpr int kWords = kMaxDet / 64;

struct Smem {
  int scan[kT + 1];                  // block-count prefix (exclusive)
  unsigned long long keys[kMaxDet];  // (conf bits << 32) | (0xFFFFFFFF - gathered position)
  float x1[kMaxDet], y1[kMaxDet], x2[kMaxDet], y2[kMaxDet], conf[kMaxDet];
  int cls[kMaxDet];
  unsigned long long mask[64][kWords];
  unsigned long long removed[kWords];
  unsigned long long rownz;
  int wave_cnt[kWaves];
  int hist[256];
  int is_last;
  int misc[4];
};

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float iou_eps(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2,
                                         float by2) {
  const float ix1 = fmaxf(ax1, bx1), iy1 = fmaxf(ay1, by1);
  const float ix2 = fminf(ax2, bx2), iy2 = fminf(ay2, by2);
  if (ix1 >= ix2 || iy1 >= iy2) return 0.0f;
  const float inter = (ix2 - ix1) * (iy2 - iy1);
  const float area_a = (ax2 - ax1) * (ay2 - ay1);
  const float area_b = (bx2 - bx1) * (by2 - by1);
  return inter / (area_a + area_b - inter + 1e-6f);
}

// exclusive prefix of a 0/1 flag over the block, in thread order; returns this thread's offset, total in *total
__device__ __forceinline__ int block_rank(bool flag, Smem& s, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned long long b = __ballot(flag);
  const int rank = __popcll(b & ((1ull << lane) - 1ull));
  __syncthreads();  // protect wave_cnt from the previous use
  if (lane == 0) s.wave_cnt[wid] = __popcll(b);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    const int c = s.wave_cnt[w];
    if (w < wid) off += c;
    tot += c;
  }
  *total = tot;
  return off + rank;
}

// one cell of one head -> (pass, record)
__device__ __forceinline__ bool decode_cell(const float* __restrict__ cls, const float* __restrict__ reg, int gw, int gh,
                                            int stride, int num_classes, float conf_thr, float q, int idx,
                                            GpuDetection* d) {
  const int hw = gw * gh;
  float max_conf = 0.0f;
  int best = -1;
  for (int c = 0; c < num_classes; ++c) {
    const float conf = sigmoidf(cls[(size_t)c * hw + idx]);
    if (conf > max_conf) {
      max_conf = conf;
      best = c;
    }
  }
  if (!(max_conf >= conf_thr)) return false;
  const int y = idx / gw, x = idx - y * gw;
  const float fs = (float)stride;
  const float xc = ((float)x + 0.5f) * fs, yc = ((float)y + 0.5f) * fs;
  const float l = reg[idx] * fs, t = reg[(size_t)hw + idx] * fs;
  const float r = reg[(size_t)2 * hw + idx] * fs, b = reg[(size_t)3 * hw + idx] * fs;
  d->x1 = xc - l;
  d->y1 = yc - t;
  d->x2 = xc + r;
  d->y2 = yc + b;
  if (q > 0.0f) {
    const float bw = d->x2 - d->x1, bh = d->y2 - d->y1;
    const float dw = bw * q, dh = bh * q;
    d->x1 -= dw;
    d->y1 -= dh;
    d->x2 += dw;
    d->y2 += dh;
  }
  d->confidence = max_conf;
  d->class_id = best;
  d->valid = 1;
  d->_pad = 0;
  return true;
}

// Publishes this block's global stores and draws an arrival ticket; returns true in exactly one block (the last
// to arrive), with every other block's stores visible to it (agent-scope release / acquire).
__device__ __forceinline__ bool arrive_and_check_last(unsigned int* ticket, Smem& s) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s.is_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s.is_last) return false;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  return true;
}

// exclusive prefix of block_count[0..nblocks) into s.scan[0..nblocks]; returns the total
__device__ __forceinline__ int scan_block_counts(const int* block_count, int nblocks, Smem& s) {
  const int tid = threadIdx.x;
  const int v = tid < nblocks ? __hip_atomic_load(block_count + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
  s.scan[tid + 1] = v;
  if (tid == 0) s.scan[0] = 0;
  __syncthreads();
  for (int off = 1; off < kT; off <<= 1) {
    const int add = (tid + 1 > off) ? s.scan[tid + 1 - off] : 0;
    __syncthreads();
    s.scan[tid + 1] += add;
    __syncthreads();
  }
  return s.scan[nblocks];
}

// enumeration position e -> candidate record in the per-block segments (s.scan = exclusive prefix of block counts)
__device__ __forceinline__ const GpuDetection* cand_at(const GpuDetection* cand, const Smem& s, int nblocks, int e) {
  int lo = 0, hi = nblocks - 1;  // largest b with scan[b] <= e
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (s.scan[mid] <= e) lo = mid; else hi = mid - 1;
  }
  return cand + (size_t)lo * kT + (e - s.scan[lo]);
}

__device__ __forceinline__ void put(Smem& s, int pos, const GpuDetection* c) {
  s.x1[pos] = c->x1; s.y1[pos] = c->y1; s.x2[pos] = c->x2; s.y2[pos] = c->y2;
  s.conf[pos] = c->confidence; s.cls[pos] = c->class_id;
}

__device__ __forceinline__ int gathered_pos(const Smem& s, int i) {
  return (int)(0xFFFFFFFFu - (unsigned int)(s.keys[i] & 0xFFFFFFFFull));
}

// Records 0..n) are in s.{x1..cls} in enumeration order. Stable sort by confidence (descending) and greedy NMS.
// On return (after a barrier) s.keys holds the sorted order and s.removed the suppression bits by SORTED index.
__device__ void sort_and_nms(Smem& s, int n, float iou_thr) {
  const int tid = threadIdx.x;
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  if (tid < np2)
    s.keys[tid] = tid < n ? (((unsigned long long)__float_as_uint(s.conf[tid]) << 32) | (0xFFFFFFFFu - (unsigned)tid)) : 0ull;
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int partner = tid ^ j;
      if (tid < np2 && partner > tid) {
        const unsigned long long a = s.keys[tid], b = s.keys[partner];
        const bool desc = (tid & k) == 0;
        if (desc ? (a < b) : (a > b)) {
          s.keys[tid] = b;
          s.keys[partner] = a;
        }
      }
      __syncthreads();
    }
  }
  const int nw = (n + 63) >> 6;
  unsigned long long removed_reg = 0ull;  // lane w (< kWords) of wave 0 owns word w
  for (int c = 0; c < nw; ++c) {
    if (tid == 0) s.rownz = 0ull;
    __syncthreads();
    {
      const int r = tid >> 4, w = tid & 15;  // 64 rows x 16 words of 64 columns
      const int i = c * 64 + r;
      unsigned long long bits = 0ull;
      if (i < n && w >= c && w < nw) {
        const int pi = gathered_pos(s, i);
        const float ax1 = s.x1[pi], ay1 = s.y1[pi], ax2 = s.x2[pi], ay2 = s.y2[pi], ac = s.conf[pi];
        const int acls = s.cls[pi];
        for (int b = 0; b < 64; ++b) {
          const int j = w * 64 + b;
          if (j <= i || j >= n) continue;
          const int pj = gathered_pos(s, j);
          if (s.cls[pj] != acls) continue;
          if (!(ac > s.conf[pj])) continue;  // only a strictly higher confidence suppresses
          if (iou_eps(ax1, ay1, ax2, ay2, s.x1[pj], s.y1[pj], s.x2[pj], s.y2[pj]) > iou_thr) bits |= 1ull << b;
        }
      }
      s.mask[r][w] = bits;
      if (bits) atomicOr(&s.rownz, 1ull << r);
    }
    __syncthreads();
    if (tid < 64) {  // wave 0 resolves this chunk in order, visiting only the rows that suppress something
      unsigned long long todo = s.rownz;
      unsigned long long cur = __shfl(removed_reg, c);
      while (todo) {
        const int r = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        if ((cur >> r) & 1ull) continue;  // row r was itself suppressed: it suppresses nothing
        if (tid < kWords) removed_reg |= s.mask[r][tid];
        cur = __shfl(removed_reg, c);
      }
    }
    __syncthreads();
  }
  if (tid < kWords) s.removed[tid] = removed_reg;
  __syncthreads();
}

}  // namespace

// ================================================================================================ fused kernel
__global__ __launch_bounds__(kPostBlock) void postprocess_kernel(const PostParams p) {
  __shared__ Smem s;
  const int tid = threadIdx.x;
  const int n0 = p.gw[0] * p.gh[0], n1 = p.gw[1] * p.gh[1], n2 = p.gw[2] * p.gh[2];
  const int ncells = n0 + n1 + n2;

  // ---- phase 1: decode (all blocks) ----
  {
    const int g = blockIdx.x * kT + tid;
    bool pass = false;
    GpuDetection d;
    if (g < ncells) {
      const int h = g < n0 ? 0 : (g < n0 + n1 ? 1 : 2);
      const int idx = g - (h == 0 ? 0 : (h == 1 ? n0 : n0 + n1));
      pass = decode_cell(p.cls[h], p.reg[h], p.gw[h], p.gh[h], p.stride[h], p.num_classes, p.conf_thr, p.conformal_q,
                         idx, &d);
    }
    int total;
    const int pos = block_rank(pass, s, &total);
    if (pass) p.cand[(size_t)blockIdx.x * kT + pos] = d;
    if (tid == 0) p.block_count[blockIdx.x] = total;
  }
  if (!arrive_and_check_last(p.ticket, s)) return;

  // ---- phase 2 (one block): gather in enumeration order ----
  const int nblocks = gridDim.x;  // <= kT (checked on the host)
  const int total = scan_block_counts(p.block_count, nblocks, s);
  int n;
  if (total <= kMaxDet) {
    n = total;
    if (tid < n) put(s, tid, cand_at(p.cand, s, nblocks, tid));
  } else {
    // overflow: keep the kMaxDet largest confidences, ties by enumeration order. Radix select on the bit
    // patterns (confidences are positive floats, so the patterns order like the values).
    n = kMaxDet;
    unsigned int prefix = 0, pmask = 0;
    int want = kMaxDet;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) s.hist[tid] = 0;
      __syncthreads();
      for (int e = tid; e < total; e += kT) {
        const unsigned int key = __float_as_uint(cand_at(p.cand, s, nblocks, e)->confidence);
        if ((key & pmask) == prefix) atomicAdd(&s.hist[(key >> shift) & 255u], 1);
      }
      __syncthreads();
      if (tid == 0) {
        int acc = 0, b = 255;
        for (; b > 0; --b) {
          if (acc + s.hist[b] >= want) break;
          acc += s.hist[b];
        }
        s.misc[0] = b;
        s.misc[1] = want - acc;
      }
      __syncthreads();
      prefix |= (unsigned int)s.misc[0] << shift;
      pmask |= 255u << shift;
      want = s.misc[1];
      __syncthreads();
    }
    const unsigned int T = prefix;  // bits of the kMaxDet-th largest confidence; take `want` of its ties
    int base_sel = 0, base_eq = 0;
    for (int e0 = 0; e0 < total; e0 += kT) {
      const int e = e0 + tid;
      const GpuDetection* c = e < total ? cand_at(p.cand, s, nblocks, e) : nullptr;
      const unsigned int key = c ? __float_as_uint(c->confidence) : 0u;
      const bool eq = c && key == T;
      int tot_eq, tot_sel;
      const int rank_eq = block_rank(eq, s, &tot_eq);
      const bool sel = c && (key > T || (eq && base_eq + rank_eq < want));
      const int pos = base_sel + block_rank(sel, s, &tot_sel);
      if (sel) put(s, pos, c);
      base_sel += tot_sel;
      base_eq += tot_eq;
    }
  }
  __syncthreads();

  sort_and_nms(s, n, p.iou_thr);

  // ---- compaction + output ----
  const bool kept = tid < n && !((s.removed[tid >> 6] >> (tid & 63)) & 1ull);
  int nkept;
  const int opos = block_rank(kept, s, &nkept);
  if (kept) {
    const int my = gathered_pos(s, tid);
    GpuDetection d;
    d.x1 = s.x1[my]; d.y1 = s.y1[my]; d.x2 = s.x2[my]; d.y2 = s.y2[my];
    d.confidence = s.conf[my];
    d.class_id = s.cls[my];
    d.valid = 1;
    d._pad = 0;
    p.out[opos] = d;
  }
  if (tid == 0) {
    *p.out_count = nkept;
    if (p.out_candidates) *p.out_candidates = total;
    __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  }
}

int post_num_blocks(const int gw[3], const int gh[3]) {
  const int cells = gw[0] * gh[0] + gw[1] * gh[1] + gw[2] * gh[2];
  return (cells + kPostBlock - 1) / kPostBlock;
}

hipError_t postprocess_launch(const PostParams& p, hipStream_t stream) {
  const int nb = post_num_blocks(p.gw, p.gh);
  if (nb < 1 || nb > kPostBlock) return hipErrorInvalidValue;
  postprocess_kernel<<<nb, kPostBlock, 0, stream>>>(p);
  return hipGetLastError();
}

// ================================================================================================ step-wise API
// The reference's seven entry points (gpu_postprocess.h:42-80 / gpu_postprocess.cu:267-442), deterministic:
//   decode_yolo_head appends the head's survivors in row-major order at the running counter (records past
//   MAX_DETECTIONS are dropped, the counter keeps counting -- like the reference's `if (det_idx < MAX_DETECTIONS)`);
//   run_gpu_nms sorts the first n records in place (stable, by confidence) and clears `valid` on suppressed ones;
//   copy_valid_detections_to_host compacts `valid != 0` and copies count + records (two syncs, as the reference).
namespace {

struct StepWorkspace {
  int* d_count = nullptr;            // running detection counter (gpu_postprocess.cu:48)
  GpuDetection* d_cand = nullptr;    // per-block segments for the ordered append
  int* d_block_count = nullptr;
  unsigned int* d_ticket = nullptr;
  GpuDetection* d_compact = nullptr; // compacted output (gpu_postprocess.cu:49)
  int* d_num_selected = nullptr;     // gpu_postprocess.cu:50
  int cand_blocks = 0;
};
StepWorkspace g_ws;  // one process-global workspace, like the reference (gpu_postprocess.cu:56-57)
constexpr int kStepMaxBlocks = 1024;

__global__ __launch_bounds__(kPostBlock) void decode_head_append_kernel(const float* cls, const float* reg,
                                                                        GpuDetection* dets, int* d_count,
                                                                        GpuDetection* cand, int* block_count,
                                                                        unsigned int* ticket, int gw, int gh, int stride,
                                                                        int num_classes, float conf_thr, float q) {
  __shared__ Smem s;
  const int tid = threadIdx.x;
  const int idx = blockIdx.x * kT + tid;
  bool pass = false;
  GpuDetection d;
  if (idx < gw * gh) pass = decode_cell(cls, reg, gw, gh, stride, num_classes, conf_thr, q, idx, &d);
  int total;
  const int pos = block_rank(pass, s, &total);
  if (pass) cand[(size_t)blockIdx.x * kT + pos] = d;
  if (tid == 0) block_count[blockIdx.x] = total;
  if (!arrive_and_check_last(ticket, s)) return;
  const int nblocks = gridDim.x;
  const int tot = scan_block_counts(block_count, nblocks, s);
  const int base = *d_count;
  for (int e = tid; e < tot; e += kT)
    if (base + e < kMaxDet) dets[base + e] = *cand_at(cand, s, nblocks, e);
  __syncthreads();
  if (tid == 0) {
    *d_count = base + tot;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(kPostBlock) void nms_inplace_kernel(GpuDetection* dets, int n, float iou_thr) {
  __shared__ Smem s;
  const int tid = threadIdx.x;
  if (tid < n) put(s, tid, dets + tid);
  __syncthreads();
  sort_and_nms(s, n, iou_thr);
  if (tid < n) {
    const int my = gathered_pos(s, tid);
    GpuDetection d;
    d.x1 = s.x1[my]; d.y1 = s.y1[my]; d.x2 = s.x2[my]; d.y2 = s.y2[my];
    d.confidence = s.conf[my];
    d.class_id = s.cls[my];
    d.valid = ((s.removed[tid >> 6] >> (tid & 63)) & 1ull) ? 0 : 1;
    d._pad = 0;
    dets[tid] = d;
  }
}

__global__ __launch_bounds__(kPostBlock) void compact_valid_kernel(const GpuDetection* dets, int n, GpuDetection* out,
                                                                   int* num_selected) {
  __shared__ Smem s;
  const int tid = threadIdx.x;
  const bool v = tid < n && dets[tid].valid != 0;  // IsValidDetection (gpu_postprocess.cu:247-251)
  int tot;
  const int pos = block_rank(v, s, &tot);
  if (v) out[pos] = dets[tid];
  if (tid == 0) *num_selected = tot;
}

}  // namespace
}  // namespace unina

using unina::g_ws;

extern "C" {

hipError_t init_postprocess_resources(void) {
  hipError_t err;
  if ((err = hipMalloc(&g_ws.d_count, sizeof(int))) != hipSuccess) return err;
  if ((err = hipMalloc(&g_ws.d_cand, sizeof(GpuDetection) * (size_t)unina::kStepMaxBlocks * unina::kPostBlock)) != hipSuccess) return err;
  if ((err = hipMalloc(&g_ws.d_block_count, sizeof(int) * unina::kStepMaxBlocks)) != hipSuccess) return err;
  if ((err = hipMalloc(&g_ws.d_ticket, sizeof(unsigned int))) != hipSuccess) return err;
  if ((err = hipMalloc(&g_ws.d_compact, sizeof(GpuDetection) * MAX_DETECTIONS)) != hipSuccess) return err;
  if ((err = hipMalloc(&g_ws.d_num_selected, sizeof(int))) != hipSuccess) return err;
  if ((err = hipMemset(g_ws.d_count, 0, sizeof(int))) != hipSuccess) return err;
  if ((err = hipMemset(g_ws.d_ticket, 0, sizeof(unsigned int))) != hipSuccess) return err;
  g_ws.cand_blocks = unina::kStepMaxBlocks;
  return hipSuccess;
}

hipError_t cleanup_postprocess_resources(void) {
  void* ptrs[] = {g_ws.d_count, g_ws.d_cand, g_ws.d_block_count, g_ws.d_ticket, g_ws.d_compact, g_ws.d_num_selected};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  g_ws = unina::StepWorkspace{};
  return hipSuccess;
}

hipError_t reset_detection_counter(hipStream_t stream) {
  if (!g_ws.d_count) return hipErrorNotInitialized;
  return hipMemsetAsync(g_ws.d_count, 0, sizeof(int), stream);  // (the reference copies a stack zero: App. C #6)
}

hipError_t get_detection_count(int* count, hipStream_t stream) {
  if (!g_ws.d_count) return hipErrorNotInitialized;
  return hipMemcpyAsync(count, g_ws.d_count, sizeof(int), hipMemcpyDeviceToHost, stream);
}

hipError_t decode_yolo_head(const float* d_cls, const float* d_reg, GpuDetection* d_detections, int grid_w, int grid_h,
                            int stride, int num_classes, float conf_threshold, float conformal_q, hipStream_t stream) {
  if (!g_ws.d_count) return hipErrorNotInitialized;
  const int cells = grid_w * grid_h;
  const int nb = (cells + unina::kPostBlock - 1) / unina::kPostBlock;
  if (cells <= 0 || nb > g_ws.cand_blocks) return hipErrorInvalidValue;
  unina::decode_head_append_kernel<<<nb, unina::kPostBlock, 0, stream>>>(
      d_cls, d_reg, d_detections, g_ws.d_count, g_ws.d_cand, g_ws.d_block_count, g_ws.d_ticket, grid_w, grid_h, stride,
      num_classes, conf_threshold, conformal_q);
  return hipGetLastError();
}

hipError_t run_gpu_nms(GpuDetection* d_detections, int num_detections, float iou_threshold, hipStream_t stream) {
  if (num_detections == 0) return hipSuccess;
  if (num_detections < 0 || num_detections > MAX_DETECTIONS) return hipErrorInvalidValue;
  unina::nms_inplace_kernel<<<1, unina::kPostBlock, 0, stream>>>(d_detections, num_detections, iou_threshold);
  return hipGetLastError();
}

hipError_t copy_valid_detections_to_host(const GpuDetection* d_detections, GpuDetection* h_detections,
                                         int num_detections, int* out_valid_count, hipStream_t stream) {
  if (num_detections == 0) {
    *out_valid_count = 0;
    return hipSuccess;
  }
  if (!g_ws.d_compact) return hipErrorNotInitialized;
  if (num_detections < 0 || num_detections > MAX_DETECTIONS) return hipErrorInvalidValue;
  unina::compact_valid_kernel<<<1, unina::kPostBlock, 0, stream>>>(d_detections, num_detections, g_ws.d_compact,
                                                                    g_ws.d_num_selected);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return err;
  int valid_count = 0;
  if ((err = hipMemcpyAsync(&valid_count, g_ws.d_num_selected, sizeof(int), hipMemcpyDeviceToHost, stream)) != hipSuccess) return err;
  if ((err = hipStreamSynchronize(stream)) != hipSuccess) return err;
  valid_count = valid_count > MAX_DETECTIONS ? MAX_DETECTIONS : valid_count;
  *out_valid_count = valid_count;
  if (valid_count > 0) {
    if ((err = hipMemcpyAsync(h_detections, g_ws.d_compact, sizeof(GpuDetection) * (size_t)valid_count, hipMemcpyDeviceToHost, stream)) != hipSuccess) return err;
    if ((err = hipStreamSynchronize(stream)) != hipSuccess) return err;
  }
  return hipSuccess;
}

}  // extern "C"
