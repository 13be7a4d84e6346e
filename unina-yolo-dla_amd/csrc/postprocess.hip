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
// the segments in enumeration order, rank-sorts them in LDS (64-bit keys), builds the upper-triangular
// suppression bitmap with broadcast LDS reads (16 waves x 64x64-bit tiles), scans it greedily, and compacts
// with wave ballots.
// Built with -ffp-contract=off: the box arithmetic must round exactly like the reference's scalar code.
#include "kernels.h"
#include "mfma_common.h"

#pragma clang fp contract(off)

namespace unina {

namespace {

constexpr int kT = kPostBlock;  // threads per block == cells per block
constexpr int kWaves = kT / 64;
constexpr int kMaxDet = MAX_DETECTIONS;
constexpr int kWords = kMaxDet / 64;
constexpr int kTriWords = 64 * (kWords * (kWords + 1) / 2);  // upper-triangular 64x64-bit tiles, worst case

// All LDS of the post-process kernels lives in ONE dynamic array carved into this struct (108 KiB).
struct Smem {
  int scan[kT + 8];                   // block-count prefix (exclusive)
  unsigned long long keys[kMaxDet];   // (conf bits << 32) | (0xFFFFFFFF - enumeration position)
  float4 box[kMaxDet];                // x1,y1,x2,y2   (enumeration order, then sorted order)
  float2 cc[kMaxDet];                 // confidence, class id (bit pattern)
  unsigned long long mask[kTriWords]; // suppression bits: tile (c,w>=c), row r  ->  mask[tri(c) + r*(nw-c) + (w-c)]
  unsigned long long rownz[kWords];   // per 64-row chunk: rows with a non-empty mask
  unsigned long long removed[kWords];
  int wave_cnt[kWaves];
  int hist[256];
  int is_last;
  int misc[4];
};
constexpr size_t kPostSmemBytes = sizeof(Smem);

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

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

// one cell of one head -> (pass, record). cls_at(c) / reg_at(k) deliver the cell's raw head values (from the fp32 planes,
// or from the registers / LDS of the launch that has just computed them): ONE arithmetic sequence for every form.
template <typename FC, typename FR>
__device__ __forceinline__ bool decode_core(FC cls_at, FR reg_at, int gw, int stride, int num_classes, float conf_thr, float q,
                                            int idx, GpuDetection* d) {
  float max_conf = 0.0f;
  int best = -1;
  for (int c = 0; c < num_classes; ++c) {
    const float conf = sigmoidf(cls_at(c));
    if (conf > max_conf) {
      max_conf = conf;
      best = c;
    }
  }
  if (!(max_conf >= conf_thr)) return false;
  const int y = idx / gw, x = idx - y * gw;
  const float fs = (float)stride;
  const float xc = ((float)x + 0.5f) * fs, yc = ((float)y + 0.5f) * fs;
  const float l = reg_at(0) * fs, t = reg_at(1) * fs;
  const float r = reg_at(2) * fs, b = reg_at(3) * fs;
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

__device__ __forceinline__ bool decode_cell(const float* __restrict__ cls, const float* __restrict__ reg, int gw, int gh,
                                            int stride, int num_classes, float conf_thr, float q, int idx,
                                            GpuDetection* d) {
  const int hw = gw * gh;
  return decode_core([&](int c) { return cls[(size_t)c * hw + idx]; }, [&](int k) { return reg[(size_t)k * hw + idx]; }, gw,
                     stride, num_classes, conf_thr, q, idx, d);
}

// Publishes this block's global stores and draws an arrival ticket; returns true in exactly one block (the last
// to arrive), with every other block's stores visible to it (agent-scope release / acquire).
__device__ __forceinline__ bool arrive_and_check_last(unsigned int* ticket, int* is_last /*LDS*/, int arrivals = 0 /*0: the whole grid*/) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *is_last = (t == (arrivals ? (unsigned)arrivals : gridDim.x) - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (!*is_last) return false;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  return true;
}

// The same hand-off WITHOUT the two fences (buffer_wbl2 / buffer_inv: ~1.7 us each on this part, MI355X_MICROARCH.md "Workgroup
// dispatch ... visibility"): valid when EVERY handed-off byte is stored with sc1 (st_sc1 below, or an agent-scope atomic) and
// EVERY load of them by the last arriver is an sc1 load (ld_sc1 / an agent-scope atomic load) -- the guide's measured form "one
// lane of each storing workgroup adds to ONE counter after every storing wave's vmcnt(0) wait and the workgroup barrier; the
// workgroup whose add came last loads after its add has returned, its other waves after a barrier". One workgroup per CU.
__device__ __forceinline__ bool arrive_and_check_last_sc1(unsigned int* ticket, int* is_last /*LDS*/, int arrivals) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's sc1 stores and atomics have left
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *is_last = (t == (unsigned)arrivals - 1u) ? 1 : 0;
  }
  __syncthreads();
  return *is_last != 0;
}
__device__ __forceinline__ void st_sc1(unsigned long long* at, unsigned long long v) {
  __hip_atomic_store(at, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long* at) {
  return __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(float2* at, float2 v) {
  st_sc1(reinterpret_cast<unsigned long long*>(at), ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x));
}
__device__ __forceinline__ float2 ld_sc1(const float2* at) {
  const unsigned long long u = ld_sc1(reinterpret_cast<const unsigned long long*>(at));
  return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
}
__device__ __forceinline__ void st_sc1(float4* at, float4 v) {
  st_sc1(reinterpret_cast<float2*>(at), make_float2(v.x, v.y));
  st_sc1(reinterpret_cast<float2*>(at) + 1, make_float2(v.z, v.w));
}
__device__ __forceinline__ float4 ld_sc1(const float4* at) {
  const float2 a = ld_sc1(reinterpret_cast<const float2*>(at)), b = ld_sc1(reinterpret_cast<const float2*>(at) + 1);
  return make_float4(a.x, a.y, b.x, b.y);
}

// exclusive prefix of block_count[0..nblocks) into s.scan[0..nblocks]; returns the total. nblocks <= kT.
__device__ __forceinline__ int scan_block_counts(const int* block_count, int nblocks, Smem& s) {
  const int tid = threadIdx.x;
  if (tid < 64) {  // one wave: serial over 64-entry chunks, shuffle scan inside a chunk
    int carry = 0;
    for (int base = 0; base < nblocks; base += 64) {
      const int i = base + tid;
      const int v = i < nblocks ? __hip_atomic_load(block_count + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (tid >= off) incl += t;
      }
      if (i < nblocks) s.scan[i] = carry + incl - v;
      carry += __shfl(incl, 63);
    }
    if (tid == 0) s.scan[nblocks] = carry;
  }
  __syncthreads();
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
  s.box[pos] = make_float4(c->x1, c->y1, c->x2, c->y2);
  s.cc[pos] = make_float2(c->confidence, __int_as_float(c->class_id));
}

__device__ __forceinline__ GpuDetection get(const Smem& s, int pos, int valid) {
  GpuDetection d;
  const float4 b = s.box[pos];
  const float2 c = s.cc[pos];
  d.x1 = b.x; d.y1 = b.y; d.x2 = b.z; d.y2 = b.w;
  d.confidence = c.x;
  d.class_id = __float_as_int(c.y);
  d.valid = valid;
  d._pad = 0;
  return d;
}

__device__ __forceinline__ int tri_off(int c, int nw) { return 64 * (c * nw - (c * (c - 1)) / 2); }

// Sequential greedy pass over the suppression bitmap by ONE wave: walks the 64-row chunks in order and visits only
// rows that suppress something; a row that was itself suppressed suppresses nothing. lane w (< kWords) owns word w.
// (A variant that walks a chunk's diagonal tile on the scalar unit with v_readlane and merges the off-diagonal words
// with LDS atomics measured slower: 9.0 vs 4.4 us at n = 489.)
__device__ __forceinline__ void greedy_scan(const unsigned long long* mask, const unsigned long long* rownz,
                                            unsigned long long* removed, int nw, int lane) {
  unsigned long long removed_reg = 0ull;
  // word c of the removed set lives in lane c: a wave-uniform v_readlane (c is a scalar loop counter), not a bpermute
  auto word_of = [&](int c) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)removed_reg, c);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(removed_reg >> 32), c);
    return ((unsigned long long)hi << 32) | lo;
  };
  for (int c = 0; c < nw; ++c) {
    unsigned long long todo = rownz[c];
    unsigned long long cur = word_of(c);
    const int base = tri_off(c, nw), stride = nw - c;
    const bool mine = lane >= c && lane < nw;
    const unsigned long long* row0 = mask + base + (mine ? lane - c : 0);
    // the next candidate row is known before the current one is decided: its words are fetched one row ahead
    int r = todo ? __ffsll((long long)todo) - 1 : 0;
    unsigned long long next_words = row0[r * stride];
    while (todo) {
      todo &= todo - 1ull;
      const unsigned long long words = next_words;
      const int rn = todo ? __ffsll((long long)todo) - 1 : r;
      next_words = row0[rn * stride];
      if (!((cur >> r) & 1ull)) {   // (a row that was itself suppressed suppresses nothing)
        if (mine) removed_reg |= words;
        cur = word_of(c);
      }
      r = rn;
    }
  }
  if (lane < kWords) removed[lane] = removed_reg;
}

// lane ^ J within a wave, J a compile-time power of two < 64
template <int J>
__device__ __forceinline__ unsigned xor_lane(unsigned v, int lane) {
  if constexpr (J == 1) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  } else if constexpr (J == 2) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  } else if constexpr (J == 8) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);   // row_ror:8 (rotation by half a row = xor 8)
  } else if constexpr (J == 4) {
    const unsigned a = (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x124, 0xF, 0xF, true);   // row_ror:4  : lane i <- lane (i - 4) mod 16
    const unsigned b = (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x12C, 0xF, 0xF, true);   // row_ror:12 : lane i <- lane (i + 4) mod 16
    return (lane & 4) ? a : b;
  } else if constexpr (J == 16) {
    // v_permlane16_swap (gfx950): swaps the odd rows of its first operand with the even rows of its second. With v in
    // both: [0] = rows (r0, r0, r2, r2), [1] = rows (r1, r1, r3, r3) -> an even row takes [1], an odd row [0].
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (lane & 16) ? r[0] : r[1];
  } else {
    // v_permlane32_swap: swaps the upper half of its first operand with the lower half of its second:
    // [0] = (lo, lo), [1] = (hi, hi) -> the lower half takes [1], the upper half [0].
    static_assert(J == 32, "intra-wave distance");
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (lane & 32) ? r[0] : r[1];
  }
}

template <int K, int J>
__device__ __forceinline__ void sort_step(unsigned long long& key, int tid, unsigned long long* (&buf)[2], int& pb) {
  unsigned long long other;
  if constexpr (J >= 64) {
    buf[pb][tid] = key;
    __syncthreads();
    other = buf[pb][tid ^ J];
    pb ^= 1;
  } else {
    const int lane = tid & 63;
    const unsigned lo = xor_lane<J>((unsigned)key, lane), hi = xor_lane<J>((unsigned)(key >> 32), lane);
    other = ((unsigned long long)hi << 32) | lo;
  }
  const bool take_max = ((tid & J) == 0) == ((tid & K) == 0);   // lower index of a descending pair keeps the larger key
  key = take_max ? (key > other ? key : other) : (key < other ? key : other);
  if constexpr (J > 1) sort_step<K, J / 2>(key, tid, buf, pb);
}

// stages K, 2K, ... up to N (N a power of two <= kT, block-uniform)
template <int K>
__device__ __forceinline__ void sort_stages(unsigned long long& key, int N, int tid, unsigned long long* (&buf)[2], int& pb) {
  if constexpr (K <= kT) {
    if (K <= N) {
      sort_step<K, K / 2>(key, tid, buf, pb);
      sort_stages<2 * K>(key, N, tid, buf, pb);
    }
  }
}

// Records 0..n) sit in s.box / s.cc in enumeration order; on return they are in SORTED order (confidence descending,
// ties by enumeration order), padded to a multiple of 64 with never-suppressed dummies. The 64-bit keys
// (confidence bits << 32 | ~enumeration position: all distinct, so the order is total and deterministic) are sorted
// by a bitonic network in LDS -- O(n log^2 n) work spread over the block; the earlier rank sort (every thread counts
// the keys above its own) was O(n^2) VALU work on one CU: 9.4 us at n = 700, 20.6 us at n = 1024 -- and each record
// is then fetched from the enumeration position its key carries.
__device__ void rank_sort_records(Smem& s, int n) {
  const int tid = threadIdx.x;
  int N = 64;
  while (N < n) N <<= 1;   // power of two >= n, <= kMaxDet == kT
  static_assert(kTriWords >= kT, "s.mask doubles as the second key buffer");
  // Every thread holds one key in a register (threads >= N hold 0 and only ever meet each other). Compare-exchange
  // partners tid ^ j with j < 64 sit in the same wave: those 45 of the 55 steps (N = 1024) are register exchanges with
  // no barrier; the 10 steps with j >= 64 go through LDS, ping-ponging between s.keys and s.mask (free at this point)
  // so that each costs ONE barrier.
  unsigned long long key = tid < n ? (((unsigned long long)__float_as_uint(s.cc[tid].x) << 32) | (0xFFFFFFFFu - (unsigned)tid)) : 0ull;
  unsigned long long* buf[2] = {s.keys, s.mask};
  int pb = 0;
  // The network is unrolled at compile time so that every partner distance is a constant: j = 1, 2, 8 are single DPP
  // moves (quad_perm / row_ror -- VALU latency instead of a trip through the LDS crossbar), j = 4 two of them and a
  // select, j = 16 / 32 a v_permlane16_swap / v_permlane32_swap and a select; 34 of the 55 steps at N = 1024 have j <= 8.
  sort_stages<2>(key, N, tid, buf, pb);
  __syncthreads();
  s.keys[tid] = key;
  __syncthreads();
  float4 mybox;
  float2 mycc;
  if (tid < n) {
    const int pos = (int)(0xFFFFFFFFu - (unsigned)s.keys[tid]);
    mybox = s.box[pos];
    mycc = s.cc[pos];
  }
  __syncthreads();
  if (tid < n) {
    s.box[tid] = mybox;
    s.cc[tid] = mycc;
  }
  // pad the sorted arrays up to the next multiple of 64 with records that can never be suppressed (class -1)
  const int n64 = (n + 63) & ~63;
  if (tid >= n && tid < n64) {
    s.box[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    s.cc[tid] = make_float2(0.f, __int_as_float(-1));
  }
  if (tid < kWords) s.rownz[tid] = 0ull;
  __syncthreads();
}

// One pair test of the greedy NMS, shared by the one-block and the tiled form (identical arithmetic and order).
__device__ __forceinline__ bool suppresses(const float4& a, const float2& ac, float area_a, const float4& bb, const float2& bc,
                                           bool ordered, float iou_thr) {
  const float ix1 = fmaxf(a.x, bb.x), iy1 = fmaxf(a.y, bb.y);
  const float ix2 = fminf(a.z, bb.z), iy2 = fminf(a.w, bb.w);
  const bool cand = ordered && __float_as_int(bc.y) == __float_as_int(ac.y) && ac.x > bc.x && !(ix1 >= ix2 || iy1 >= iy2);
  if (!cand) return false;
  const float inter = (ix2 - ix1) * (iy2 - iy1);
  const float area_b = (bb.z - bb.x) * (bb.w - bb.y);
  return inter / (area_a + area_b - inter + 1e-6f) > iou_thr;
}

// One-block form: sort, then s.removed = greedy-NMS suppression bits by sorted index.
//   masks: the upper-triangular 64x64 tiles are dealt round-robin to the 16 waves; lane = row, the 64 columns of
//          a tile are visited in lock-step so every LDS read is a broadcast; the division only runs for pairs
//          that really overlap
//   scan : wave 0 walks the chunks in order and visits only rows that suppress something
__device__ void sort_and_nms(Smem& s, int n, float iou_thr, long long* stamps = nullptr) {
#define STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[k] = wall_clock64(); } while (0)
  rank_sort_records(s, n);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  STAMP(3);
  const int nw = (n + 63) >> 6;
  const int ntiles = nw * (nw + 1) / 2;
  for (int t = wid; t < ntiles; t += kWaves) {
    int c = 0, rem = t;  // tile t -> (chunk row c, word w >= c)
    while (rem >= nw - c) {
      rem -= nw - c;
      ++c;
    }
    const int w = c + rem;
    const int i = c * 64 + lane;
    const bool row_ok = i < n;
    const float4 a = s.box[row_ok ? i : 0];
    const float2 ac = s.cc[row_ok ? i : 0];
    const float area_a = (a.z - a.x) * (a.w - a.y);
    unsigned long long bits = 0ull;
    for (int b0 = 0; b0 < 64; b0 += 8) {  // fixed trip count: 16 broadcast LDS reads in flight
      float4 bb[8];
      float2 bc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        bb[u] = s.box[w * 64 + b0 + u];  // same address in every lane: LDS broadcast
        bc[u] = s.cc[w * 64 + b0 + u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = w * 64 + b0 + u;
        if (suppresses(a, ac, area_a, bb[u], bc[u], row_ok && j > i, iou_thr)) bits |= 1ull << (b0 + u);
      }
    }
    s.mask[tri_off(c, nw) + lane * (nw - c) + (w - c)] = bits;
    const unsigned long long nz = __ballot(bits != 0ull);
    if (lane == 0 && nz) atomicOr(&s.rownz[c], nz);
  }
  __syncthreads();
  STAMP(4);

  if (tid < 64) greedy_scan(s.mask, s.rownz, s.removed, nw, lane);
  __syncthreads();
  STAMP(5);
#undef STAMP
}

}  // namespace

// Host hand-off without a stream synchronisation: when the outputs live in pinned host memory, the last block publishes
// them at system scope and then stores the caller's sequence number; the host spins on that word (engine.hip unina_infer).
__device__ __forceinline__ void signal_done(const PostParams& p, int tid) {
  if (!p.done_flag) return;
  // every storing wave waits for its own stores, the barrier collects the waves, ONE lane does the system-scope release in
  // front of the completion word (all 512 threads fencing cost 2-4x one lane's: MI355X_MICROARCH.md, fence prices)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_store(p.done_flag, p.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ================================================================================================ fused kernel
extern __shared__ __align__(16) unsigned char post_smem[];

__global__ __launch_bounds__(kPostBlock) void postprocess_kernel(const PostParams p) {
  Smem& s = *reinterpret_cast<Smem*>(post_smem);
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
  if (p.stamps && tid == 0 && blockIdx.x == 0) p.stamps[0] = wall_clock64();
  if (!arrive_and_check_last(p.ticket, &s.is_last)) return;
  if (p.stamps && tid == 0) p.stamps[1] = wall_clock64();

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

  if (p.stamps && tid == 0) p.stamps[2] = wall_clock64();
  sort_and_nms(s, n, p.iou_thr, p.stamps);

  // ---- compaction + output ----
  const bool kept = tid < n && !((s.removed[tid >> 6] >> (tid & 63)) & 1ull);
  int nkept;
  const int opos = block_rank(kept, s, &nkept);
  if (kept) p.out[opos] = get(s, tid, 1);
  if (tid == 0) {
    *p.out_count = nkept;
    if (p.out_candidates) *p.out_candidates = total;
    __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    if (p.stamps) p.stamps[6] = wall_clock64();
  }
  signal_done(p, tid);
}

constexpr int kMaxTiles = kWords * (kWords + 1) / 2;


// ================================================================================================ the frame's form: sort-free, folded
// Launch 1 (post_decode_kernel, 256-thread workgroups, a few hundred of them): every workgroup owns a run of consecutive
// cells of ONE head. A head is read from its fp32 planes, or -- fold -- its output convs are computed here: wave = 16-pixel
// subtile, B fragments (16 bytes of the NHWC hidden tensor per lane) and A fragments (the exporter's packed 1-KiB blocks)
// straight from global memory into v_mfma_f32_16x16x32_f16, k blocks in ascending order from a zero accumulator, `acc + bias`
// in fp32: the operations, operands and order of conv_glds / conv_epilogue, so the logits are the per-op table's bit for
// bit; they go through LDS to the thread that decodes the cell. Survivors are appended to ONE compact candidate list (a run per
// workgroup, reserved with one atomic; each record carries its enumeration index), their confidences counted into a 4096-bin
// histogram -- and that is all: no ticket, no gather, no sort.
// Launch 2 (post_nms_kernel, 512 threads): one workgroup per 64x64 tile of candidate pairs (I <= J). Each reads the list
// length and fetches its 64 rows and 64 columns straight from the list (when more than 1024 cells passed: the 1024 best by
// histogram-guided selection, see the overflow branch), and besides the overlap test compares
// the (confidence, ~enumeration index) keys of every pair, so a tile contributes (a) the DIRECTIONAL suppression bits of both
// orientations -- rows of tile (I,J) per lane, rows of tile (J,I) as wave ballots -- into a full n x n bit matrix
// indexed by list position, and (b) to every candidate's RANK = number of candidates with a larger key,
// accumulated with integer atomics: the stable sort order falls out of the pair loop that the NMS needs anyway.
// The last arriver (one ticket per TILE, not per grid slot) inverts the ranks into a permutation, and because a row only
// ever marks candidates of strictly lower confidence (= processed later) and of its own class, the greedy scan splits by
// class: wave v walks the candidates of class = v (mod 4) in rank order, skipping rows that mark nothing, 16 rows'
// masks fetched at a time. Output = unmarked candidates in rank order, every thread scattering its own two records.
// Measured inside a frame at ~490 candidates (profiles/r02): 20 us from the start of launch 1 to the completion word,
// against 31 us for the sorted two-launch form plus 8 us for the conv launch of the P3 | P4 output convs it absorbed.
namespace {

constexpr int kT2 = kPost2Block;
constexpr int kW2 = kT2 / 64;

struct SmemD {                     // launch 1
  float stage[32][kT2];            // fold: [output channel: 0..15 cls, 16..31 reg][cell of the workgroup]
  int scan[1024 + 8];
  int wave_cnt[kW2];
  int hist[256];
  int is_last;
  int misc[4];
};

constexpr int kTN = 512;             // threads per workgroup of launch 2
constexpr int kWN = kTN / 64;        // 8 waves: a 64 x 64 pair tile = 64 rows (lanes) x 8 columns per wave
constexpr int kColsPerWave = 64 / kWN;
struct SmemN {                     // launch 2
  float4 rbox[64], cbox[64];
  float2 rcc[64], ccc[64];
  unsigned char piece[64][kWN];
  unsigned char rcnt[64][kWN];
  unsigned long long mask[kMaxDet * kWords];   // last workgroup: row i, word w at [i * nw + w]
  unsigned short perm[kMaxDet];
  unsigned char cls4[kMaxDet];
  unsigned short list[4][kMaxDet];
  unsigned long long removed[4][kWords];
  unsigned long long rownz[kWords];
  int scan[1024 + 8];                // exclusive prefix of launch 1's per-workgroup candidate counts
  int hist[256];
  int misc[4];
  int wave_cnt[kWN];
  int is_last;
};

// block_rank for the kTN-thread workgroups of launch 2
__device__ __forceinline__ int block_rank_n(bool flag, int* wave_cnt /*LDS, kWN*/, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned long long b = __ballot(flag);
  const int rank = __popcll(b & ((1ull << lane) - 1ull));
  __syncthreads();
  if (lane == 0) wave_cnt[wid] = __popcll(b);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kWN; ++w) {
    const int c = wave_cnt[w];
    if (w < wid) off += c;
    tot += c;
  }
  *total = tot;
  return off + rank;
}

// block_rank for a 256-thread workgroup
__device__ __forceinline__ int block_rank2(bool flag, int* wave_cnt /*LDS, kW2*/, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned long long b = __ballot(flag);
  const int rank = __popcll(b & ((1ull << lane) - 1ull));
  __syncthreads();
  if (lane == 0) wave_cnt[wid] = __popcll(b);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kW2; ++w) {
    const int c = wave_cnt[w];
    if (w < wid) off += c;
    tot += c;
  }
  *total = tot;
  return off + rank;
}

// confidence -> bin of launch 1's histogram (4096 linear bins; x 4096 is exact in binary floating point, so bin T is
// exactly the confidences in [T / 4096, (T + 1) / 4096))
constexpr int kBins = 4096;
__device__ __forceinline__ int conf_bin(float c) {
  const int b = (int)(c * (float)kBins);
  return b > kBins - 1 ? kBins - 1 : b;
}
// (class id | enumeration index << 8) travels in the second float of a candidate's (confidence, class) pair
__device__ __forceinline__ float pack_ce(int class_id, int eidx) { return __int_as_float((class_id & 0xFF) | (eidx << 8)); }
__device__ __forceinline__ int ce_class(float y) { return (int)(signed char)(__float_as_int(y) & 0xFF); }
__device__ __forceinline__ unsigned ce_eidx(float y) { return (unsigned)__float_as_int(y) >> 8; }

}  // namespace

__global__ __launch_bounds__(kPost2Block) void post_decode_kernel(const PostParams p) {
  SmemD& s = *reinterpret_cast<SmemD*>(post_smem);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.x;
  const int h = b < p.bstart[1] ? 0 : (b < p.bstart[2] ? 1 : 2);
  const int cpb = p.cpb[h];
  const int cell0 = (b - p.bstart[h]) * cpb;
  const int gw = p.gw[h], ncell = gw * p.gh[h];
  const int idx = cell0 + tid;
  if (p.stamps && tid == 0 && b == 0) p.stamps[7] = wall_clock64();
  bool pass = false;
  GpuDetection d;
  if (p.h1[h] == nullptr) {
    if (idx < ncell) pass = decode_cell(p.cls[h], p.reg[h], gw, p.gh[h], p.stride[h], p.num_classes, p.conf_thr, p.conformal_q, idx, &d);
  } else {
    // ---- fold: the two output convs of the head on this workgroup's pixels (see the header comment) ----
    const int nsub = cpb >> 6;                       // pixel subtiles per wave: 1 or 4
    const int l15 = lane & 15, lq = lane >> 4;
    const int kblocks = p.h1_c[h] >> 5, ld = p.h1_ld[h];
    const _Float16* h1 = static_cast<const _Float16*>(p.h1[h]);
    const int slot = lane * 16;                      // this lane's 16 bytes of any weight block of the lane-order twin
    for (int si = 0; si < nsub; ++si) {
      const int sub = wv * nsub + si;
      int pix = cell0 + sub * 16 + l15;
      pix = pix < ncell ? pix : ncell - 1;           // tail subtiles: any valid pixel (never decoded)
      const _Float16* x0 = h1 + (size_t)pix * ld + 8 * lq;
      dev::floatx4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      const long long xlo = p.h1_lo[h];
      if (xlo) {   // STRICT engines (wave-uniform): split-fp16 hidden tensor and (hi | lo) weight block pairs, three MFMAs per k block
        for (int kb0 = 0; kb0 < kblocks; kb0 += 4) {
          dev::half8x2 av[2][4], bv[2][4];
#pragma unroll
          for (int br = 0; br < 2; ++br)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int kb = kb0 + k < kblocks ? kb0 + k : kblocks - 1;
              const unsigned char* wa = p.w2[h][br] + slot + (size_t)kb * 2048;
              const _Float16* xa = x0 + p.h1_coff[h][br] + kb * 32;
              av[br][k] = dev::half8x2{*reinterpret_cast<const dev::half8*>(wa), *reinterpret_cast<const dev::half8*>(wa + 1024)};
              bv[br][k] = dev::half8x2{*reinterpret_cast<const dev::half8*>(xa),
                                       *reinterpret_cast<const dev::half8*>(reinterpret_cast<const unsigned char*>(xa) + xlo)};
            }
#pragma unroll
          for (int br = 0; br < 2; ++br)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (kb0 + k < kblocks) acc[br] = dev::mfma_split(av[br][k], bv[br][k], acc[br]);
        }
      } else
      // every operand of up to 8 k blocks of BOTH branches is requested before the first MFMA waits for one: the hidden
      // tensor was written a launch ago by other CUs, each load is a trip to another XCD's side of the chip
      for (int kb0 = 0; kb0 < kblocks; kb0 += 8) {
        dev::half8 av[2][8], bv[2][8];
#pragma unroll
        for (int br = 0; br < 2; ++br)
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int kb = kb0 + k < kblocks ? kb0 + k : kblocks - 1;   // (past the end: a repeat, not used)
            av[br][k] = *reinterpret_cast<const dev::half8*>(p.w2[h][br] + slot + (size_t)kb * 1024);
            bv[br][k] = *reinterpret_cast<const dev::half8*>(x0 + p.h1_coff[h][br] + kb * 32);
          }
#pragma unroll
        for (int br = 0; br < 2; ++br)
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (kb0 + k < kblocks) acc[br] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[br][k], bv[br][k], acc[br], 0, 0, 0);
      }
#pragma unroll
      for (int br = 0; br < 2; ++br) {
        const dev::floatx4 bias = *reinterpret_cast<const dev::floatx4*>(p.b2[h][br] + 4 * lq);
        const dev::floatx4 v = acc[br] + bias;       // conv_epilogue: v = acc + bias, no activation (model.py:292,299)
#pragma unroll
        for (int r = 0; r < 4; ++r) s.stage[br * 16 + 4 * lq + r][sub * 16 + l15] = v[r];
      }
    }
    __syncthreads();
    if (tid < cpb && idx < ncell)
      pass = decode_core([&](int c) { return s.stage[c][tid]; }, [&](int k) { return s.stage[16 + k][tid]; }, gw, p.stride[h],
                         p.num_classes, p.conf_thr, p.conformal_q, idx, &d);
  }
  {
    // survivors go to ONE compact candidate list: the workgroup reserves a run of it with a single atomic (the runs land in
    // arrival order -- the order is immaterial, every later decision is keyed by (confidence, enumeration index), which each
    // record carries). No per-workgroup count table, no scan, no search in launch 2. Next to the records: the confidence bit
    // pattern and the enumeration index as dense arrays, and a 4096-bin histogram of the confidences -- what launch 2 needs to
    // find the kMaxDet best of MORE than kMaxDet candidates in one or two coalesced sweeps.
    int total;
    const int pos = block_rank2(pass, s.wave_cnt, &total);
    if (tid == 0) s.misc[0] = total ? __hip_atomic_fetch_add(p.ws_total, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    __syncthreads();
    if (pass) {
      const int at = s.misc[0] + pos;
      d._pad = p.eoff[h] + idx;                       // enumeration index (P2 -> P3 -> P4, row-major): the tie-break of equal confidences
      p.cand[at] = d;
      p.ws_ke[at] = make_uint2(__float_as_uint(d.confidence), (unsigned)d._pad);
      atomicAdd(&p.ws_hist[conf_bin(d.confidence)], 1);
    }
  }
  if (p.stamps && tid == 0 && b == 0) p.stamps[0] = wall_clock64();
}

__global__ __launch_bounds__(kTN) void post_nms_kernel(const PostParams p) {
  SmemN& s = *reinterpret_cast<SmemN*>(post_smem);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int t = blockIdx.x;
  if (p.stamps && tid == 0 && t == 0) p.stamps[3] = wall_clock64();
  // launch 1 left ONE compact candidate list and its length: no count table to scan, no search per row
  const int total = __hip_atomic_load(p.ws_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (p.stamps && tid == 0 && t == 0) p.stamps[1] = wall_clock64();
  const int n = total < kMaxDet ? total : kMaxDet;
  const int nw = (n + 63) >> 6;
  const int ntiles = nw * (nw + 1) / 2;
  if (t >= (ntiles > 0 ? ntiles : 1)) return;   // (the grid is sized for 1024 candidates; with none, workgroup 0 writes the empty result)
  if (ntiles > 0) {
    int c = 0, rem = t;  // tile t -> (row chunk c, column chunk w >= c)
    while (rem >= nw - c) {
      rem -= nw - c;
      ++c;
    }
    const int w = c + rem;
    if (tid < 128) {   // rows 64c.. (tid < 64), columns 64w.. : dummies (class -1, empty box, the largest enumeration index) past the end
      const int k = tid & 63;
      if (tid < 64) {
        s.rbox[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        s.rcc[k] = make_float2(0.f, __int_as_float(-1));
      } else {
        s.cbox[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        s.ccc[k] = make_float2(0.f, __int_as_float(-1));
      }
    }
    auto fetch = [&](int e, int pos) {   // list entry e is candidate `pos` of the (selected) list: into this tile's rows / columns
      const GpuDetection* cd = p.cand + e;
      const float4 bx = make_float4(cd->x1, cd->y1, cd->x2, cd->y2);
      const float2 cf = make_float2(cd->confidence, pack_ce(cd->class_id, cd->_pad));
      if ((pos >> 6) == c) { s.rbox[pos & 63] = bx; s.rcc[pos & 63] = cf; }
      if ((pos >> 6) == w) { s.cbox[pos & 63] = bx; s.ccc[pos & 63] = cf; }
    };
    if (total <= kMaxDet) {
      if (tid < 128) {
        const int e = (tid < 64 ? c : w) * 64 + (tid & 63);
        if (e < n) fetch(e, e);
      }
    } else {
      // ---- overflow: the kMaxDet largest keys (confidence, then the SMALLER enumeration index) of `total` candidates ----
      // Every tile workgroup finds the same cut element (cutkey, cuteidx) -- selected <=> key > cutkey || (key == cutkey &&
      // eidx <= cuteidx) -- and numbers the selected candidates in list order; it fetches those that fall into its rows / columns.
      // Launch 1's histogram gives the confidence bin T that holds the cut; as long as that bin's members are too many to rank
      // exactly (an untrained network puts half the frame into one bin) the bin is split again -- 4096 sub-ranges of the key,
      // and once all keys are equal, of the enumeration index -- one coalesced sweep over the dense key array per level
      // (typically none: ~10 members). Then: one sweep that collects the members and counts the candidates above the bin per
      // wave range, the exact ranking of the <= 1024 members in LDS, one sweep that numbers and fetches. Was: five passes with a
      // binary search per candidate (44 us at conf 0.3, 124 us at conf 0.05 on the synthetic weights).
      constexpr int kMemCap = 1024;
      int* hist = reinterpret_cast<int*>(s.mask);                       // (the suppression matrix's LDS is free until the scan)
      unsigned* mkey = reinterpret_cast<unsigned*>(hist + kBins);
      int* meidx = reinterpret_cast<int*>(mkey + kMemCap);
      int* mpos = meidx + kMemCap;
      int* part = mpos + kMemCap;                                       // [kTN] partial sums of 8 bins each
      const int R = (((total + kWN - 1) / kWN) + 63) & ~63;             // list entries per wave range (a multiple of 64)
      const int e_lo = wv * R, e_hi = (e_lo + R) < total ? (e_lo + R) : total;
      // a wave sweeps its range 1024 keys at a time: 16 coalesced loads in flight per lane before the first is used (with 4 a
      // sweep of 33 600 keys paid 17 dependent round trips through a loaded L2: ~20 us)
      constexpr int U = 16;
      auto for_chunks = [&](auto body) {   // body(e0, ke): ke[u] = (key, enumeration index) of list entry e0 + 64 u + lane, (0, 0) past the range
        uint2 cur[U], nxt[U];
        auto request = [&](uint2 (&d)[U], int e0) {
#pragma unroll
          for (int u = 0; u < U; ++u) d[u] = e0 + 64 * u + lane < e_hi ? p.ws_ke[e0 + 64 * u + lane] : make_uint2(0u, 0u);
        };
        request(cur, e_lo);
        for (int e0 = e_lo; e0 < e_hi; e0 += 64 * U) {
          request(nxt, e0 + 64 * U);          // (the next chunk is on its way while this one is processed)
          body(e0, cur);
#pragma unroll
          for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        }
      };
      // the bin that holds the `want`-th element counted from the top (ascending == false) or from the bottom of hist[0..kBins):
      // returns (bin, elements wanted from it, its count) through s.misc
      auto pick_bin = [&](int want, bool ascending) {
        int v8 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) v8 += hist[8 * tid + k];
        part[tid] = v8;
        __syncthreads();
        if (wv == 0) {
          int v = 0;
#pragma unroll
          for (int k = 0; k < 8; ++k) v += part[8 * lane + k];
          int incl = v;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
          }
          const int all = __shfl(incl, 63);
          const int before = ascending ? incl - v : all - incl;          // elements in front of this lane's 64 bins, in walking order
          // exactly one lane's 64 bins hold the element (all >= want); the bin inside them by a second wave-wide scan
          const unsigned long long hit = __ballot(before < want && before + v >= want);
          const int L = __ffsll((long long)hit) - 1;
          const int before_l = __shfl(before, L);
          const int bb = 64 * L + (ascending ? lane : 63 - lane);           // lane 0 = the first bin in walking order
          const int hv = hist[bb];
          int inc2 = hv;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(inc2, off);
            if (lane >= off) inc2 += u;
          }
          const int b2 = before_l + inc2 - hv;
          if (b2 < want && b2 + hv >= want) {
            s.misc[0] = bb;
            s.misc[1] = want - b2;
            s.misc[2] = hv;
          }
        }
        __syncthreads();
      };
      for (int i = tid; i < kBins; i += kTN) hist[i] = p.ws_hist[i];
      __syncthreads();
      pick_bin(kMaxDet, false);
      const int T = s.misc[0];
      int want = s.misc[1], cnt = s.misc[2];
      // the key range [klo, khi) of bin T (bit patterns of positive floats order like the values)
      unsigned klo = __float_as_uint((float)T / (float)kBins);
      unsigned khi = T == kBins - 1 ? 0x3F800001u : __float_as_uint((float)(T + 1) / (float)kBins);   // (a confidence is at most 1.0)
      __syncthreads();
      while (cnt > kMemCap && khi - klo > 1u) {                           // split the bin's key range 4096 ways
        const unsigned wdt = (khi - klo + (unsigned)kBins - 1u) / (unsigned)kBins;
        for (int i = tid; i < kBins; i += kTN) hist[i] = 0;
        __syncthreads();
        for_chunks([&](int e0, const uint2 (&ke)[U]) {
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (e0 + 64 * u + lane < e_hi && ke[u].x >= klo && ke[u].x < khi) atomicAdd(&hist[(ke[u].x - klo) / wdt], 1);
        });
        __syncthreads();
        pick_bin(want, false);
        const unsigned nlo = klo + (unsigned)s.misc[0] * wdt;
        khi = nlo + wdt < khi ? nlo + wdt : khi;
        klo = nlo;
        want = s.misc[1];
        cnt = s.misc[2];
        __syncthreads();
      }
      // all keys of the bin equal and still too many: split by enumeration index (the smaller index comes first)
      int esh = 0, ebin = -1;
      if (cnt > kMemCap) {
        const int emax = p.eoff[2] + p.gw[2] * p.gh[2];
        while ((emax >> esh) >= kBins) ++esh;
        for (int i = tid; i < kBins; i += kTN) hist[i] = 0;
        __syncthreads();
        for_chunks([&](int e0, const uint2 (&ke)[U]) {
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (e0 + 64 * u + lane < e_hi && ke[u].x == klo) atomicAdd(&hist[(int)ke[u].y >> esh], 1);
        });
        __syncthreads();
        pick_bin(want, true);
        ebin = s.misc[0];
        want = s.misc[1];
        cnt = s.misc[2];                                                  // <= 2^esh <= 64 members
        __syncthreads();
      }
      // 2 = selected for certain, 1 = member of the last bin (ranked exactly below), 0 = out
      auto classify = [&](unsigned key, int eidx) {
        if (key >= khi) return 2;
        if (key < klo) return 0;
        if (ebin < 0) return 1;
        const int eb = eidx >> esh;
        return eb < ebin ? 2 : (eb == ebin ? 1 : 0);
      };
      if (tid == 0) s.misc[3] = 0;
      if (tid < kWN) s.hist[tid] = 0;
      __syncthreads();
      // sweep: candidates selected for certain per wave range, members -> LDS
      int sure = 0;
      for_chunks([&](int e0, const uint2 (&ke)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int cl = e0 + 64 * u + lane < e_hi ? classify(ke[u].x, (int)ke[u].y) : 0;
          sure += __popcll(__ballot(cl == 2));
          if (cl == 1) {
            const int slot = atomicAdd(&s.misc[3], 1);
            if (slot < kMemCap) { mkey[slot] = ke[u].x; meidx[slot] = (int)ke[u].y; mpos[slot] = e0 + 64 * u + lane; }
          }
        }
      });
      if (lane == 0) s.wave_cnt[wv] = sure;
      __syncthreads();
      const int m = s.misc[3] < kMemCap ? s.misc[3] : kMemCap;           // (== cnt; the cap cannot bite: cnt <= kMemCap here)
      // exact ranking of the members; the member of rank want - 1 is the cut, the members up to it are selected
      for (int i = tid; i < m; i += kTN) {
        const unsigned ki = mkey[i];
        const int xi = meidx[i];
        int r = 0;
        for (int j = 0; j < m; ++j) r += (mkey[j] > ki || (mkey[j] == ki && meidx[j] < xi)) ? 1 : 0;
        if (r == want - 1) { s.misc[0] = (int)ki; s.misc[1] = xi; }
        if (r < want) atomicAdd(&s.hist[mpos[i] / R], 1);                 // selected members per wave range
      }
      __syncthreads();
      const unsigned cutkey = (unsigned)s.misc[0];
      const int cuteidx = s.misc[1];
      int run = 0;
      for (int v = 0; v < wv; ++v) run += s.wave_cnt[v] + s.hist[v];
      // sweep: number the selected candidates in list order, fetch this tile's
      for_chunks([&](int e0, const uint2 (&ke)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool sel = e0 + 64 * u + lane < e_hi && (ke[u].x > cutkey || (ke[u].x == cutkey && (int)ke[u].y <= cuteidx));
          const unsigned long long bal = __ballot(sel);
          const int pos = run + __popcll(bal & ((1ull << lane) - 1ull));
          // (only the list entry is noted here: a record load inside this loop would be one serialised round trip per hit)
          if (sel && (pos >> 6) == c) part[pos & 63] = e0 + 64 * u + lane;
          if (sel && (pos >> 6) == w) part[64 + (pos & 63)] = e0 + 64 * u + lane;
          run += __popcll(bal);
        }
      });
      __syncthreads();
      if (tid < 128) fetch(part[tid], (tid < 64 ? c : w) * 64 + (tid & 63));   // every slot is taken: 1024 candidates are selected
    }
    __syncthreads();
    if (p.stamps && tid == 0 && t == 0) p.stamps[2] = wall_clock64();
    if (c == w && tid < 64) {   // the diagonal tile publishes its chunk of the candidate list (output stage)
      st_sc1(p.ws_box + c * 64 + tid, s.rbox[tid]);     // (everything the last arriver reads is stored sc1: arrive_and_check_last_sc1)
      st_sc1(p.ws_cc + c * 64 + tid, s.rcc[tid]);
    }
    const int gi = c * 64 + lane;
    const float4 a = s.rbox[lane];
    const float2 ac = s.rcc[lane];
    const float area_a = (a.z - a.x) * (a.w - a.y);
    unsigned int rowbits = 0u;
    int rowcnt = 0, colcnt = 0;
    unsigned long long colword = 0ull;
#pragma unroll
    for (int u = 0; u < kColsPerWave; ++u) {
      const int col = wv * kColsPerWave + u;
      const int gj = w * 64 + col;
      const float4 bb = s.cbox[col];
      const float2 bc = s.ccc[col];
      // key_j > key_i with key = (confidence, ~enumeration index): the stable descending order
      const bool j_first = bc.x > ac.x || (bc.x == ac.x && ce_eidx(bc.y) < ce_eidx(ac.y));
      const bool i_first = gi != gj && !j_first;
      // the pair test of `suppresses` (same arithmetic; the IoU is symmetric bit for bit), once for both orientations
      const float ix1 = fmaxf(a.x, bb.x), iy1 = fmaxf(a.y, bb.y);
      const float ix2 = fminf(a.z, bb.z), iy2 = fminf(a.w, bb.w);
      bool over = gi != gj && ((__float_as_int(bc.y) ^ __float_as_int(ac.y)) & 0xFF) == 0 && !(ix1 >= ix2 || iy1 >= iy2);
      if (over) {
        const float inter = (ix2 - ix1) * (iy2 - iy1);
        const float area_b = (bb.z - bb.x) * (bb.w - bb.y);
        over = inter / (area_a + area_b - inter + 1e-6f) > p.iou_thr;
      }
      if (over && ac.x > bc.x) rowbits |= 1u << u;                       // i suppresses j (strictly higher confidence)
      rowcnt += j_first ? 1 : 0;
      const unsigned long long cw = __ballot(over && bc.x > ac.x);      // j suppresses i: row j of tile (w, c), bit = lane
      const int cc = __popcll(__ballot(i_first));
      if (lane == u) {
        colword = cw;
        colcnt = cc;
      }
    }
    s.piece[lane][wv] = (unsigned char)rowbits;
    s.rcnt[lane][wv] = (unsigned char)rowcnt;
    const unsigned long long colnz = __ballot(lane < kColsPerWave && colword != 0ull);
    if (w != c) {   // (the diagonal tile visits every ordered pair itself)
      if (lane < kColsPerWave) {
        const int gj = w * 64 + wv * kColsPerWave + lane;
        st_sc1(p.ws_full + (size_t)gj * kWords + c, colword);
        atomicAdd(&p.ws_rank[gj], colcnt);
      }
      if (lane == 0 && colnz) atomicOr(&p.ws_rownz[w], colnz << (kColsPerWave * wv));
    }
    __syncthreads();
    if (tid < 64) {
      unsigned long long word = 0ull;
      int cnt = 0;
#pragma unroll
      for (int v = 0; v < kWN; ++v) {
        word |= (unsigned long long)s.piece[tid][v] << (kColsPerWave * v);
        cnt += s.rcnt[tid][v];
      }
      st_sc1(p.ws_full + (size_t)gi * kWords + w, word);
      atomicAdd(&p.ws_rank[gi], cnt);
      const unsigned long long nz = __ballot(word != 0ull);
      if (tid == 0 && nz) atomicOr(&p.ws_rownz[c], nz);
    }
    if (!arrive_and_check_last_sc1(p.ticket2, &s.is_last, ntiles)) return;
  }
  if (p.stamps && tid == 0) p.stamps[4] = wall_clock64();

  // ---- last arriver: ranks -> permutation, masks -> LDS, greedy scan per class residue, output ----
  int rk[2];        // this thread's two candidates (enumeration positions tid, tid + kTN): rank, (confidence, class), box
  float2 cc2[2];
  float4 bx2[2];
  {
    // every load of this phase is requested before the first one is waited for (each is a trip to L2 / another XCD)
    constexpr int B = 8;
    const int total = n * nw;       // rows 0..n) x nw words
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + q * kTN;
      rk[q] = e < n ? __hip_atomic_load(p.ws_rank + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      cc2[q] = e < n ? ld_sc1(p.ws_cc + e) : make_float2(0.f, 0.f);
      bx2[q] = e < n ? ld_sc1(p.ws_box + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    unsigned long long rz = tid < kWords ? __hip_atomic_load(p.ws_rownz + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    for (int k0 = tid; k0 < total; k0 += kTN * B) {
      unsigned long long v[B];
#pragma unroll
      for (int q = 0; q < B; ++q) {
        const int k = k0 + q * kTN;
        const int row = k / nw, wd = k - row * nw;
        v[q] = k < total ? ld_sc1(p.ws_full + (size_t)row * kWords + wd) : 0ull;
      }
#pragma unroll
      for (int q = 0; q < B; ++q) {
        const int k = k0 + q * kTN;
        if (k < total) s.mask[k] = v[q];
      }
    }
    if (tid < kWords) s.rownz[tid] = rz;
    __syncthreads();
    // entry r of the rank-ordered list: candidate position | class residue << 10 | "its row marks something" << 12
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + q * kTN;
      if (e < n && (unsigned)rk[q] < (unsigned)n) {   // (ranks are a permutation of 0..n-1: keys are distinct)
        const unsigned nz = (unsigned)((s.rownz[e >> 6] >> (e & 63)) & 1ull);
        s.perm[rk[q]] = (unsigned short)(e | ((__float_as_int(cc2[q].y) & 3) << 10) | (nz << 12));
      }
    }
  }
  __syncthreads();
  if (p.stamps && tid == 0) p.stamps[8] = wall_clock64();
  unsigned long long removed_reg = 0ull;                // wave v < 4, lane q < nw: word q of the wave's suppressed set
  if (wv < 4) {
    // this wave's rows, in rank order: class = wv (mod 4), mask not empty
    int m = 0;
    for (int r0 = 0; r0 < n; r0 += 64) {
      const int r = r0 + lane;
      const int ent = r < n ? (int)s.perm[r] : 0;
      const bool take = r < n && (ent >> 12) && ((ent >> 10) & 3) == wv;
      const unsigned long long bal = __ballot(take);
      if (take) s.list[wv][m + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(ent & 1023);
      m += __popcll(bal);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the list was written by this wave itself
    if (p.stamps && tid == 0) p.stamps[9] = wall_clock64();
    if (p.stamps && lane == 0) p.stamps[10 + wv] = m;
    auto word_of = [&](int q) {
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)removed_reg, q);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(removed_reg >> 32), q);
      return ((unsigned long long)hi << 32) | lo;
    };
    for (int k0 = 0; k0 < m; k0 += 64) {
      // 64 list entries in a register (lane L: entry k0 + L); per group of 16: the 16 mask rows are fetched together, then
      // the 16 decisions run from registers -- branch-free, so that no LDS latency sits in the dependent chain (with a
      // conditional refill inside the loop the compiler waited lgkmcnt(0) per row: ~200 cycles a row)
      const int cnt = m - k0 < 64 ? m - k0 : 64;
      const int mine = k0 + lane < m ? (int)s.list[wv][k0 + lane] : 0;
      for (int kb = 0; kb < cnt; kb += 16) {
        unsigned long long pre[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int iq = __builtin_amdgcn_readlane(mine, (kb + q) & 63);     // (past the end: entry 0 of the register, unused)
          pre[q] = lane < nw ? s.mask[iq * nw + lane] : 0ull;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int i = __builtin_amdgcn_readlane(mine, (kb + q) & 63);
          const unsigned long long cur = word_of(i >> 6);
          const bool act = kb + q < cnt && !((cur >> (i & 63)) & 1ull);     // (a suppressed row suppresses nothing)
          removed_reg |= act ? pre[q] : 0ull;
        }
      }
    }
    if (lane < kWords) s.removed[wv][lane] = removed_reg;
    if (p.stamps && lane == 0) p.stamps[10 + wv] |= (long long)(wall_clock64() & 0xFFFFFFFFll) << 20;
  }
  __syncthreads();
  if (tid < kWords) s.removed[0][tid] |= s.removed[1][tid] | s.removed[2][tid] | s.removed[3][tid];
  __syncthreads();
  if (p.stamps && tid == 0) p.stamps[5] = wall_clock64();
  // output position of a kept candidate = number of kept candidates of lower rank: prefix over the rank order, then every
  // thread scatters ITS OWN records (enumeration order, already in registers) -- no second trip to memory
  int base = 0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = tid + q * kTN;
    if (q * kTN >= n) break;
    bool kept = false;
    if (r < n) {
      const int i = s.perm[r] & 1023;
      kept = !((s.removed[0][i >> 6] >> (i & 63)) & 1ull);
    }
    const unsigned long long bal = __ballot(kept);
    __syncthreads();
    if (lane == 0) s.wave_cnt[wv] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int v = 0; v < kWN; ++v) {
      const int cwv = s.wave_cnt[v];
      if (v < wv) off += cwv;
      tot += cwv;
    }
    if (r < n) s.list[0][r] = (unsigned short)(base + off + __popcll(bal & ((1ull << lane) - 1ull)));   // (the lists are done with)
    base += tot;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + q * kTN;
    if (e < n && (unsigned)rk[q] < (unsigned)n && !((s.removed[0][e >> 6] >> (e & 63)) & 1ull)) {
      GpuDetection d;
      d.x1 = bx2[q].x; d.y1 = bx2[q].y; d.x2 = bx2[q].z; d.y2 = bx2[q].w;
      d.confidence = cc2[q].x;
      d.class_id = ce_class(cc2[q].y);
      d.valid = 1;
      d._pad = 0;
      p.out[s.list[0][rk[q]]] = d;
    }
  }
  // the rank / row accumulators, the list length and the confidence histogram go back to zero for the next frame (they are
  // zero at rest, like the ticket)
  for (int e = tid; e < nw * 64; e += kTN) p.ws_rank[e] = 0;
  if (tid < kWords) p.ws_rownz[tid] = 0ull;
  for (int i = tid; i < kBins; i += kTN) p.ws_hist[i] = 0;
  if (tid == 0) {
    *p.out_count = base;
    if (p.out_candidates) *p.out_candidates = total;
    __hip_atomic_store(p.ws_total, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p.ticket2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    if (p.stamps) p.stamps[6] = wall_clock64();
  }
  signal_done(p, tid);
}

int post_num_blocks(const int gw[3], const int gh[3]) {
  const int cells = gw[0] * gh[0] + gw[1] * gh[1] + gw[2] * gh[2];
  return (cells + kPostBlock - 1) / kPostBlock;
}

constexpr size_t kListCap = 256 * 1024;   // candidates the compact list can hold (= cells of the largest supported frame: 1024 workgroups x 256)
size_t post_workspace_bytes() {
  return sizeof(float4) * kMaxDet + sizeof(float2) * kMaxDet + 256 + sizeof(int) * kBins + sizeof(uint2) * kListCap +
         sizeof(int) * kMaxDet + sizeof(unsigned long long) * ((size_t)kMaxDet * kWords + kWords);
}

void post_bind_workspace(PostParams* p, void* ws) {
  char* c = static_cast<char*>(ws);
  p->ws_box = reinterpret_cast<float4*>(c); c += sizeof(float4) * kMaxDet;
  p->ws_cc = reinterpret_cast<float2*>(c); c += sizeof(float2) * kMaxDet;
  p->ws_total = reinterpret_cast<int*>(c);
  p->ticket2 = reinterpret_cast<unsigned int*>(c + 64);
  c += 256;
  p->ws_hist = reinterpret_cast<int*>(c); c += sizeof(int) * kBins;
  p->ws_ke = reinterpret_cast<uint2*>(c); c += sizeof(uint2) * kListCap;
  p->ws_full = reinterpret_cast<unsigned long long*>(c); c += sizeof(unsigned long long) * (size_t)kMaxDet * kWords;
  p->ws_rownz = reinterpret_cast<unsigned long long*>(c); c += sizeof(unsigned long long) * kWords;
  p->ws_rank = reinterpret_cast<int*>(c);
}

// mode 2: workgroups per head. A head read from planes: 256 cells per workgroup. A folded head: 64 cells (one 16-pixel
// subtile per wave: its hidden tensor -- 64 px x 2C x 2 bytes per workgroup -- is then spread over many CUs), or 256 when
// the grid would not fit the 1024-entry block table.
bool post_plan_blocks(PostParams* p) {
  p->eoff[0] = 0;                                  // enumeration index of a head's first cell (P2 -> P3 -> P4, row-major)
  p->eoff[1] = p->gw[0] * p->gh[0];
  p->eoff[2] = p->eoff[1] + p->gw[1] * p->gh[1];
  if ((size_t)p->eoff[2] + (size_t)p->gw[2] * p->gh[2] > kListCap) return false;
  for (int per = 64; per <= 256; per *= 4) {
    int tot = 0;
    for (int h = 0; h < 3; ++h) {
      const int cells = p->gw[h] * p->gh[h];
      p->cpb[h] = p->h1[h] ? per : kT2;
      p->bstart[h] = tot;
      tot += (cells + p->cpb[h] - 1) / p->cpb[h];
    }
    p->bstart[3] = tot;
    if (tot >= 1 && tot <= 1024) return true;
  }
  return false;
}

int postprocess_desc(const PostParams& p, LaunchDesc out[2]) {
  if (p.mode == 2) {
    if (p.bstart[3] < 1 || p.bstart[3] > 1024 || !p.ws_full) return -1;
    out[0].func = reinterpret_cast<const void*>(&post_decode_kernel);
    out[0].grid = dim3(p.bstart[3]);
    out[0].block = dim3(kT2);
    out[0].shmem = (unsigned)sizeof(SmemD);
    out[1].func = reinterpret_cast<const void*>(&post_nms_kernel);
    out[1].grid = dim3(kMaxTiles);   // n is only known on the device: tiles past the triangle just draw their ticket
    out[1].block = dim3(kTN);
    out[1].shmem = (unsigned)sizeof(SmemN);
    return 2;
  }
  const int nb = post_num_blocks(p.gw, p.gh);
  if (nb < 1 || nb > kPostBlock) return -1;
  out[0].func = reinterpret_cast<const void*>(&postprocess_kernel);
  out[0].grid = dim3(nb);
  out[0].block = dim3(kPostBlock);
  out[0].shmem = (unsigned)kPostSmemBytes;
  return 1;
}

hipError_t postprocess_launch(const PostParams& p, hipStream_t stream) {
  LaunchDesc d[2];
  const int n = postprocess_desc(p, d);
  if (n < 1) return hipErrorInvalidValue;
  PostParams copy = p;
  void* args[] = {&copy};
  for (int k = 0; k < n; ++k) {
    hipError_t e = hipLaunchKernel(d[k].func, d[k].grid, d[k].block, args, d[k].shmem, stream);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
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
  Smem& s = *reinterpret_cast<Smem*>(post_smem);
  const int tid = threadIdx.x;
  const int idx = blockIdx.x * kT + tid;
  bool pass = false;
  GpuDetection d;
  if (idx < gw * gh) pass = decode_cell(cls, reg, gw, gh, stride, num_classes, conf_thr, q, idx, &d);
  int total;
  const int pos = block_rank(pass, s, &total);
  if (pass) cand[(size_t)blockIdx.x * kT + pos] = d;
  if (tid == 0) block_count[blockIdx.x] = total;
  if (!arrive_and_check_last(ticket, &s.is_last)) return;
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
  Smem& s = *reinterpret_cast<Smem*>(post_smem);
  const int tid = threadIdx.x;
  if (tid < n) put(s, tid, dets + tid);
  __syncthreads();
  sort_and_nms(s, n, iou_thr);
  if (tid < n) dets[tid] = get(s, tid, ((s.removed[tid >> 6] >> (tid & 63)) & 1ull) ? 0 : 1);
}

__global__ __launch_bounds__(kPostBlock) void compact_valid_kernel(const GpuDetection* dets, int n, GpuDetection* out,
                                                                   int* num_selected) {
  Smem& s = *reinterpret_cast<Smem*>(post_smem);
  const int tid = threadIdx.x;
  const bool v = tid < n && dets[tid].valid != 0;  // IsValidDetection (gpu_postprocess.cu:247-251)
  int tot;
  const int pos = block_rank(v, s, &tot);
  if (v) out[pos] = dets[tid];
  if (tid == 0) *num_selected = tot;
}


}  // namespace

// hipFuncSetAttribute applies to the CURRENT device: called by unina_load_engine after hipSetDevice (one handle per GPU)
// and by init_postprocess_resources.
hipError_t post_init() {
  const void* fns[] = {reinterpret_cast<const void*>(postprocess_kernel), reinterpret_cast<const void*>(decode_head_append_kernel),
                       reinterpret_cast<const void*>(nms_inplace_kernel), reinterpret_cast<const void*>(compact_valid_kernel)};
  for (const void* f : fns) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPostSmemBytes);
    if (e != hipSuccess) return e;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(post_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmemD));
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(post_nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmemN));
}

}  // namespace unina

using unina::g_ws;

extern "C" {

hipError_t init_postprocess_resources(void) {
  hipError_t err;
  if ((err = unina::post_init()) != hipSuccess) return err;
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
  unina::decode_head_append_kernel<<<nb, unina::kPostBlock, unina::kPostSmemBytes, stream>>>(
      d_cls, d_reg, d_detections, g_ws.d_count, g_ws.d_cand, g_ws.d_block_count, g_ws.d_ticket, grid_w, grid_h, stride,
      num_classes, conf_threshold, conformal_q);
  return hipGetLastError();
}

hipError_t run_gpu_nms(GpuDetection* d_detections, int num_detections, float iou_threshold, hipStream_t stream) {
  if (num_detections == 0) return hipSuccess;
  if (num_detections < 0 || num_detections > MAX_DETECTIONS) return hipErrorInvalidValue;
  unina::nms_inplace_kernel<<<1, unina::kPostBlock, unina::kPostSmemBytes, stream>>>(d_detections, num_detections, iou_threshold);
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
  unina::compact_valid_kernel<<<1, unina::kPostBlock, unina::kPostSmemBytes, stream>>>(d_detections, num_detections, g_ws.d_compact,
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
