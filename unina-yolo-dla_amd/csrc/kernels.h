// kernels.h -- device-kernel launch interface of the engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

#include "../../include/unina_mi355.h"

namespace unina {

typedef _Float16 half_t;
// kS16 ("split fp16", the STRICT precision mode): a value is an fp16 PAIR hi + lo (hi = fp16(v), lo = fp16(v - hi), ~22 mantissa
// bits). A tensor is two fp16 planes of identical NHWC layout: the hi plane at the tensor's address, the lo plane `*_lo` bytes
// behind it; weights are pairs of 1-KiB fragment blocks [hi | lo]. A conv is three v_mfma_f32_16x16x32_f16 per k block
// (lo*hi, hi*lo, hi*hi; the lo*lo term is below fp32 resolution) into one fp32 accumulator.
enum DType : int { kF16 = 0, kF32 = 1, kI8 = 2, kS16 = 3 };   // element types (accumulation: fp32 for f16/f32/s16 inputs, int32 for int8)
constexpr int kNumDTypes = 4;
struct s16_t { _Float16 v; };                        // one plane element of a kS16 tensor (pointer arithmetic in plane elements)

// ------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution  D[cout][pixel] = sum_k W[cout][k] * X[pixel][k]   (+bias, ReLU, +residual)
//   X : NHWC fp16 activation buffer, read through (ld, channel offset) so concat slices are free
//   W : fp16 [n_pad][K], K = (kh, kw, cin)  -- BatchNorm already folded in
// One launch covers up to two output-channel slices (merged sibling convs / two-group head layers).
// ------------------------------------------------------------------------------------------------
struct ConvSeg {
  const void* w;       // packed 1-KiB fragment blocks [n_pad/16][K/kb][64 slots][16 B], kb = 32 (fp16) / 16 (fp32)
  const void* w_lane;  // the same blocks with their 16-byte slots in LANE order (engine.hip appends the twin at load): what the
                       // register-path kernels (conv3x3_regq, conv3x3_ws*) read; nullptr if the slice has none
  const float* bias;   // [n_pad]
  void* dst;           // NHWC destination (engine dtype), channel offset already applied (unused when planar)
  float* dst_planar;   // fp32 [n][Ho*Wo] destination (head outputs), or nullptr
  int src_coff;        // channel offset of this slice's input
  int n_count;         // valid output channels
  int dst_ld;          // channels per pixel of the destination buffer
  int up2;             // 1: write every output pixel to its 2x2 block of a (2Ho x 2Wo) destination
  int tile0;           // first N-tile (blockIdx.y) that belongs to this slice
  int out_dtype;       // DType of dst (property of the destination buffer)
  long long dst_lo;    // kS16 destinations: byte distance hi plane -> lo plane
  float out_inv_scale; // int8 destinations: 1 / s_out
  const float* mult;   // int8 convs: per-channel s_in*s_w*bn_scale applied to the int32 accumulator; nullptr otherwise
};

struct ConvParams {
  int dtype;           // DType of src and weights (kernel instantiation)
  const void* src;
  long long src_lo;    // kS16: byte distance hi plane -> lo plane of the source buffer
  long long res_lo;    // kS16 residual: the same for the residual buffer
  int src_ld;          // channels per pixel of the source buffer
  int H, W, Cin;       // input spatial size, input channels of each slice
  int Ho, Wo, M;       // output spatial size, M = Ho*Wo
  int ksize, stride, pad;
  int relu;
  const void* res;     // residual (added after ReLU), channel offset applied; nullptr = none
  int res_ld;
  int res_dtype;       // DType of the residual buffer
  float res_scale;     // int8 residual: s_res
  int nseg;
  ConvSeg seg[2];
  const void* zeros;   // >= 16 bytes of zeros in HBM: source of out-of-image taps / tile tails
  int force_cfg;       // >= 0: use this tile configuration (autotuner / tests); -1: heuristic
  int grid_m;          // number of M tiles (filled at launch): the 1-D grid is re-mapped XCD-aware in the kernel
  int grid_n;          // number of N tiles (filled at launch)
  int xcd_m_major;     // 1: an XCD owns a range of pixel tiles (input-heavy ops), 0: a range of channel tiles (weight-heavy)
  unsigned gn_magic;
  unsigned gm_magic, wo_magic, spt_magic, tx_magic;  // ceil(2^32/d) for d = grid_m, Wo, K-steps per tap, spatial tiles per row (0: d == 1), filled at launch
  long long* stamps;   // debug: s_memtime stamps of workgroup (0,0) (nullptr = off): start, prologue issued, first data, loop end, end
  long long* wg_times; // debug (stamped dual kernels): [2 * blockIdx.x] = 100 MHz wall clock at the workgroup's start, [+1] at its end
};

// tile configurations of the conv kernel: block tile = BM pixels x BN output channels, K-step BK
enum ConvConfig : int {
  kCfg64x64k64 = 0, kCfg64x64k32, kCfg128x64k64, kCfg128x64k32, kCfg128x128k64,
  kCfg128x32k64, kCfg128x32k32, kCfg128x16k64, kCfg32x64k64, kCfg32x64k64s8, kCfg64x64k64s6,
  kCfgHalo8x8n64, kCfgHalo8x8n32, kCfgHalo8x16n64, kCfgHalo8x16n32, kCfgHalo8x8n64k32, kCfgHalo8x16n32k32,
  kCfg32x64k128, kCfg64x64k128,
  kCfgRegq8x16n64c128, kCfgRegq8x8n64c128, kCfgRegq8x8n64c256, kCfgRegq8x8n32c256, kCfgRegq8x16n64c64, kCfgRegq8x16n32c128,
  kCfgRegqS2_8x8n64c64, kCfgRegqS2_8x16n64c64, kCfgRegqS2_8x8n64c128, kCfgRegqS2_4x8n64c128, kCfgRegqS2_8x16n64c32, kCfgRegqS2_8x8n32c128,
  kCfgWs16x16n64c128, kCfgWs8x16n64c256,
  kCfgWsS8x16n64c64, kCfgWsS8x16n64c128, kCfgWsS8x16n64c256,   // split fp16 (STRICT engines): weights-stationary, row-walking, channel-chunked
  kCfgWs8x16n64c128, kCfgWs4x16n64c256,                        // fp16: the half-height tiles of the head pair on large frames (two workgroups per CU)
  kCfgCount
};
struct ConvLaunch {
  ConvConfig cfg;
  dim3 grid, block;
  const char* kernel_name;
};
hipError_t conv_init();                                  // once per process: raise the dynamic-LDS limit of every instantiation
bool conv_config_valid(const ConvParams& p, int cfg);
ConvLaunch conv_plan(const ConvParams& p);               // heuristic (or p.force_cfg)
ConvLaunch conv_plan_with(const ConvParams& p, int cfg);
const char* conv_config_name(int cfg, int dtype = kF16);
hipError_t conv_launch(const ConvParams& p, const ConvLaunch& l, hipStream_t stream);
// Two independent convs of one kernel family as ONE grid (conv_igemm.hip, dual launches): kind >= 0 if the pair fits.
int conv_dual_match(const ConvParams& a, const ConvParams& b);
const char* conv_dual_name(int kind);
hipError_t conv_dual_launch(int kind, const ConvParams& a, const ConvParams& b, hipStream_t stream, int* grid_out = nullptr);
int conv_dual_grid(int kind, const ConvParams& a, const ConvParams& b);   // workgroups that launch will have (-1: bad kind)

// ------------------------------------------------------------------------------------------------
// Fused C3k2 block (model.py:76-110): cv1|cv2 -> n x Bottleneck(1x1, 3x3 + shortcut) -> cv3 in ONE launch, all
// intermediates in LDS (c3k2_fused.hip). dtype kF16: fp16 blocks; kI8 (INT8 engines): blocks whose input, output and
// every intermediate tensor are int8 codes -- same per-tensor scales and epilogue arithmetic as the per-op table.
// ------------------------------------------------------------------------------------------------
struct C3k2Params {
  int dtype;                     // kF16, kI8 or kS16: element type of src, dst and every tensor in between
  const void* src;               // block input, channel offset applied
  long long src_lo, src2_lo, dst_lo, dst2_lo;   // kS16: byte distance hi plane -> lo plane of src / src2 / dst / dst2
  int lds_lo;                    // kS16 (filled by c3k2_layout): byte distance from an LDS image to its lo twin
  int src_ld, Cin;
  int H, W;                      // spatial size (input == output)
  void* dst;                     // block output (cv3), channel offset applied
  int dst_ld;
  const unsigned char* wstream;  // every conv's weight blocks in consumption order (c3k2_pack)
  const float* bias;             // per-step constants, same order: fp16 [bias(n)]; int8 [bias(n) | mult(n) | 1/s_out(n)]
  const void* zeros;             // >= 16 bytes of zeros in HBM
  int hid, nb;                   // hidden width h = Cout/2, number of bottlenecks
  int cpre;                      // != 0: the 3x3 / stride-2 ConvBlock (cpre -> cx channels) in front of the block runs as a first
                                 // step; src / src_ld then describe ITS input (preH x preW pixels), H x W stays the block's size
  int preH, preW;
  int cx;                        // channels of the block input the pre-conv produces (== Cin, or the first part of a concat)
  const void* src2;              // cx < Cin: the remaining Cin - cx input channels (block resolution), channel offset applied
  int src2_ld;
  float res_scale[2];            // int8: scale of each bottleneck's shortcut tensor
  signed char* dst_q;            // fp16 blocks in INT8 engines: int8 twin of the block output (the QUANT op that follows), or nullptr
  int dst_q_ld;
  float q_inv;                   // 1 / scale of the twin
  int tail;                      // 1: the lateral 1x1 conv (2h -> h) + nearest x2 upsample that follows runs as a last step;
                                 // 2: a plain 1x1 ConvBlock 2h -> h; 3: as 1, int8 block whose lateral writes an fp16 tensor
  void* dst2;                    // tail output (2H x 2W pixels), channel offset applied
  int dst2_ld;
  long long* stamps;             // debug (stamped twin kernels only): s_memtime of the mid workgroup after every step (nullptr = off)
  // filled by c3k2_layout():
  int n_bias;
  int tiles_x, tiles_y;
  unsigned tiles_x_magic;
  int off_bias, off_x, off_y, off_t, off_u1, off_u2, off_stage, off_tail, off_p, off_xr, smem_bytes;   // LDS layout (bytes)
};
struct C3k2Conv {                // one conv of the block as the exporter stored it (host pointers)
  const unsigned char* w[2];     // packed 1-KiB fragment blocks [n/16][K/32 | K/64] per output slice (slice 1 only for cv1|cv2)
  const float* bias[2];
  const float* mult[2];          // int8: per-channel multipliers (SegDesc::m_off)
  float out_inv[2];              // int8: 1 / scale of the slice's destination tensor
  int n[2];                      // output channels per slice (multiples of 16)
  int K;                         // ksize*ksize*cin
};
hipError_t c3k2_init();
bool c3k2_layout(C3k2Params* p);
bool c3k2_supported(int hid, int nb, int cin, int tail = 0, int dtype = kF16, int cpre = 0, int cx = 0);
bool c3k2_pack(int hid, int nb, int cin, int tail, const C3k2Conv* convs, std::vector<unsigned char>* stream, std::vector<float>* bias,
               int dtype = kF16, int cpre = 0, int cx = 0);   // with cpre: convs[0] is the pre-conv (cpre -> cx channels)
hipError_t c3k2_launch(const C3k2Params& p, hipStream_t stream);
hipError_t c3k2_launch_stamped(const C3k2Params& p, hipStream_t stream);   // debug twin with per-step stamps (p.stamps); InvalidValue if the class has none
const char* c3k2_kernel_name(int hid, int nb, int cin, int tail = 0, int dtype = kF16, int cpre = 0, int cx = 0);
int c3k2_block_threads(int hid, int nb, int cin, int tail = 0, int dtype = kF16, int cpre = 0, int cx = 0);

// One 1-KiB weight fragment block (16 rows x 4 chunks of 16 bytes) from the engine file's LDS-image order -- slot(r, c) =
// 4r + (c ^ G[r >> 2]), G = (0,2,3,1): what the LDS-DMA kernels (conv_glds, conv3x3_halo) copy into LDS and read
// conflict-free -- to LANE order, slot = 16c + r = the MFMA lane that holds (row r, k-chunk c): what every kernel that takes
// weights straight into registers reads, one contiguous KiB per wave instruction.
void weight_block_to_lane_order(const unsigned char* in, unsigned char* out);   // (c3k2_fused.hip)

// Generic packer of the block kernels' weight stream: per conv, k-block-major [K/32][N/16] 1-KiB blocks (slice 0's
// channel subtiles first), then the concatenated biases (n entries per slice).
void block_pack(const C3k2Conv* convs, int nconv, std::vector<unsigned char>* stream, std::vector<float>* bias, int dtype = kF16);

// ------------------------------------------------------------------------------------------------
// Fused DetectionHead (model.py:274-303): cls.0|reg.0 (3x3) -> cls.1|reg.1 (grouped 3x3) -> cls.2|reg.2 (grouped 1x1,
// raw fp32 planar outputs) in ONE launch (head_fused.hip). fp16 engines, heads whose weights a workgroup can stream.
// ------------------------------------------------------------------------------------------------
struct HeadParams {
  const half_t* src;             // feature map, channel offset applied
  int src_ld, C;
  int H, W;
  float* out_cls;                // fp32 planar [n_cls][H*W]
  float* out_reg;                // fp32 planar [n_reg][H*W]
  int n_cls, n_reg;              // valid output channels (<= 16 each; the weight blocks are zero-padded to 16 rows)
  const unsigned char* wstream;  // block_pack of the three launches' convs
  const float* bias;
  const void* zeros;
  // filled by head_layout():
  int n_bias;
  int tiles_x, tiles_y;
  unsigned tiles_x_magic;
  int off_bias, off_x, off_h0, off_h1, smem_bytes;
};
hipError_t head_init();
bool head_supported(int c);
bool head_layout(HeadParams* p);
hipError_t head_launch(const HeadParams& p, hipStream_t stream);
const char* head_kernel_name(int c);
bool head_is_ws(int c);                // the row-streaming / weights-stationary class is selected (head_ws_body)
int head_block_threads(int c);

// Two consecutive 1x1 ConvBlocks (SPPF cv2 -> FPN lateral, + x2 upsample store) in ONE launch (conv_pair.hip).
struct PairParams {
  int dtype;                     // kF16, kI8 or kS16: element type of src, dst and dst2
  const void* src;               // input, channel offset applied
  long long src_lo, dst_lo, dst2_lo;   // kS16: byte distance hi plane -> lo plane of src / dst / dst2
  int lds_lo;                    // kS16 (filled by pair_layout): byte distance from an LDS image to its lo twin
  int src_ld;
  int c0, c1, c2;                // channels: input, first conv's output, second conv's output
  int H, W;
  void* dst;                     // first conv's output (other ops may read it), channel offset applied
  int dst_ld;
  void* dst2;                    // second conv's output; up2: a (2H x 2W) destination, every pixel written to its 2x2 block
  int dst2_ld;
  int up2;
  int pool;                      // 1: src holds only x = the first c0/4 channels of the input; the kernel forms [x | p1 | p2 | p3]
                                 // (SPPF's three chained 5x5 max-pools) itself
  const unsigned char* wstream;  // block_pack of the two convs
  const float* bias;             // per-step constants (fp16 [bias]; int8 [bias | mult | 1/s_out])
  const void* zeros;
  // filled by pair_layout():
  int n_bias, tiles_x, tiles_y;
  unsigned tiles_x_magic;
  int off_bias, off_x, off_stage, off_out, off_r, off_v, smem_bytes;
};
hipError_t pair_init();
bool pair_supported(int dtype, int c0, int c1, int c2, int up2, int pool = 0);
bool pair_layout(PairParams* p);
hipError_t pair_launch(const PairParams& p, hipStream_t stream);
const char* pair_kernel_name(const PairParams& p);
int pair_block_threads(const PairParams& p);

// A fused C3k2 block and a fused head that do not depend on each other, side by side in one grid (block_dual.hip).
hipError_t block_dual_init();
bool block_dual_match(const C3k2Params& pc, const HeadParams& ph);
const char* block_dual_name(int dtype = kF16, int cpre = 0);
hipError_t block_dual_launch(const C3k2Params& pc, const HeadParams& ph, hipStream_t stream, int* grid_out = nullptr);
bool c3k2_tile_is(const C3k2Params& p, int th, int tw);    // the tile the layout of `p` was computed for
bool head_tile_is(const HeadParams& p, int th, int tw);
// Rows of the row-streaming head's (head_ws_body) th x 14 pixel strips. 13: at 640^2 the P2 map's 160 rows x 12 strips give 156
// workgroups, which with pan_c3k2_2's 100 (block_dual.hip) fill the 256 CUs in ONE round, and a strip's 18 row iterations take
// about as long as a block workgroup (16 rows: 120 + 100 workgroups, the head's the long pole: 22.2 us for the launch).
constexpr int kHeadWsTH = 13;

// ------------------------------------------------------------------------------------------------
// Stem: fp32 NCHW image -> 3x3/s2 conv (Cin=3) + bias + ReLU -> NHWC fp16
// ------------------------------------------------------------------------------------------------
struct StemParams {
  int dtype;           // DType of dst
  int src_kind;        // 0: fp32 planar tensor `src`; 1: BGRA u8 camera frame of the network's size; 2: BGRA u8 frame of
                       // cam_w x cam_h, bilinear-resized to W x H -- 1 / 2 compute the pre-process (preprocess.hip, same
                       // arithmetic) on the fly instead of reading a tensor it would have written (unina_infer_bgra)
  const unsigned char* cam;   // src_kind 1 / 2: pitched BGRA
  int cam_w, cam_h, cam_pitch;
  NormParams norm;
  const float* src;    // [3][H][W]
  const float* w;      // [Co][27], (c,kh,kw)
  const float* wt;     // the same weights transposed to [27][Co] (prepared at load): wave-uniform scalar loads in the stem kernel
  const float* bias;   // [Co]
  void* dst;           // [Ho][Wo][dst_ld]
  long long dst_lo;    // kS16: byte distance hi plane -> lo plane of dst
  int H, W, Ho, Wo, Co, dst_ld;
};
hipError_t stem_launch(const StemParams& p, hipStream_t stream, dim3* grid_out = nullptr, dim3* block_out = nullptr);
// Launch descriptor of a kernel whose only argument is its parameter struct: what hipGraphExecKernelNodeSetParams needs
// to re-point a captured node at new parameters (a different input frame, other thresholds / output buffers).
struct LaunchDesc {
  const void* func;
  dim3 grid, block;
  unsigned shmem;
};
hipError_t stem_desc(const StemParams& p, LaunchDesc* out);
hipError_t stem_init();   // per device: dynamic-LDS limit of the tiled stem kernels

// ------------------------------------------------------------------------------------------------
// SPPF pool pyramid: y1 = pool5(x), y2 = pool5(y1), y3 = pool5(y2) (== 5x5, 9x9, 13x13 clipped windows of x)
// x is channels [coff, coff+C) of buf; y1,y2,y3 are written at coff+C, coff+2C, coff+3C of the same buffer.
// ------------------------------------------------------------------------------------------------
struct PoolParams {
  int dtype;
  void* buf;
  long long lo;        // kS16: byte distance hi plane -> lo plane
  int H, W, C, ld, coff;
};
hipError_t sppf_pool_launch(const PoolParams& p, hipStream_t stream, dim3* grid_out = nullptr, dim3* block_out = nullptr);

// fp16 -> int8 re-quantisation of a whole NHWC buffer (int8 engines: tensors with both fp16 and int8 consumers)
struct QuantParams {
  const half_t* src;
  signed char* dst;
  size_t n;            // elements (multiple of 16)
  float inv_scale;
};
hipError_t quant_launch(const QuantParams& p, hipStream_t stream);

// Standalone nearest x2 upsample into a channel slice (the graph folds this into the producer conv; kept for
// op tables that cannot fold it and for tests).
struct UpsampleParams {
  const half_t* src;
  half_t* dst;
  int H, W, C, src_ld, dst_ld;
};
hipError_t upsample2x_launch(const UpsampleParams& p, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// Fused post-process (one launch): decode three heads -> candidates -> top-1024 -> stable sort -> greedy NMS
// ------------------------------------------------------------------------------------------------
struct PostParams {
  const float* cls[3];
  const float* reg[3];
  int gw[3], gh[3], stride[3];
  int num_classes;
  float conf_thr, iou_thr, conformal_q;
  // workspace (engine-owned)
  GpuDetection* cand;        // mode 0: [nblocks][kPostBlock] per-block candidate segments; mode 2: the compact candidate list
  int* block_count;          // mode 0: [nblocks]
  unsigned int* ticket;      // arrival counter (zero at rest)
  // outputs
  GpuDetection* out;         // [MAX_DETECTIONS]
  int* out_count;            // kept
  int* out_candidates;       // optional (may be nullptr): number of cells that passed the threshold
  long long* stamps;         // optional debug: 8 wall_clock64 stamps of the last block's phases (nullptr = off)
  unsigned int* done_flag;   // optional (pinned host memory): receives done_value, system-scope release, after every output store
  unsigned int done_value;
  // two-launch form (the engine's, mode 2; postprocess.hip): launch 1 = decode on many 256-thread workgroups, survivors appended to
  // ONE compact candidate list (`cand`; a run per workgroup reserved with one atomic on ws_total; a record's `_pad` carries its
  // enumeration index P2 -> P3 -> P4, row-major), confidences counted into ws_hist; launch 2 = one workgroup per 64x64 tile of
  // the candidate pairs computes the directional suppression bits (both directions) AND every candidate's rank (number of
  // candidates with a larger (confidence, ~enumeration index) key: a stable sort order without sorting), its last arriver runs
  // the greedy scan in rank order (one wave per class residue) and writes the compacted output. Same results as mode 0.
  int mode;                  // 0: everything in one launch (one workgroup gathers, sorts, scans); 2: the two-launch form
  float4* ws_box;            // [MAX_DETECTIONS] boxes of the (selected) candidates, by list position
  float2* ws_cc;             // [MAX_DETECTIONS] (confidence, class | enumeration index << 8)
  unsigned int* ticket2;     // arrival counter of launch 2 (zero at rest)
  int* ws_total;             // candidates in the list (zero at rest: launch 2's last arriver resets it)
  int* ws_hist;              // [4096] histogram of the candidates' confidences, bin = floor(conf * 4096) (zero at rest)
  uint2* ws_ke;              // [list] (confidence bit pattern, enumeration index), dense: what the > MAX_DETECTIONS selection sweeps
  int eoff[3];               // enumeration index of each head's first cell (post_plan_blocks)
  int* ws_rank;              // [MAX_DETECTIONS] rank of each candidate (zero at rest)
  unsigned long long* ws_full;    // [MAX_DETECTIONS][16] suppression bits, row = suppressor, bit = suppressed (list positions)
  unsigned long long* ws_rownz;   // [16] rows with a non-empty mask (zero at rest)
  // fold: the head's output convs (model.py:292,299: Conv2d 1x1 + bias on the hidden tensor) computed INSIDE launch 1 with
  // the same MFMA / K order / bias add as the conv kernels (bit-identical logits), straight into the decode -- the fp32
  // planes of such a head are neither written nor read. h1[h] == nullptr: head h is read from its planes cls[h] / reg[h].
  const void* h1[3];         // hidden tensor of head h: fp16 NHWC
  long long h1_lo[3];        // != 0: a split-fp16 tensor (STRICT engines): byte distance to its lo plane; w2 then holds (hi | lo) block pairs
  int h1_ld[3], h1_c[3];     // channels per pixel of that buffer; input channels of each output conv (multiple of 32)
  int h1_coff[3][2];         // channel offset of the cls / reg branch's input
  const unsigned char* w2[3][2];  // 1-KiB fragment blocks [1][C/32] (16 rows, LANE order: ConvSeg::w_lane) of the cls / reg output conv
  const float* b2[3][2];     // [16] biases
  int bstart[4];             // launch 1 of mode 2: first workgroup of each head, total (post_plan_blocks)
  int cpb[3];                // cells per workgroup of each head: 256 (planes, or 4 pixel subtiles per wave) or 64 (1 per wave)
};
size_t post_workspace_bytes();
void post_bind_workspace(PostParams* p, void* ws);   // ws: post_workspace_bytes() of device memory, zeroed once
bool post_plan_blocks(PostParams* p);                // mode 2: fills bstart / cpb from gw, gh, h1; false if the grid would exceed 1024 workgroups
constexpr int kPostBlock = 1024;
constexpr int kPost2Block = 256;                     // threads per workgroup of the mode-2 kernels (= candidate segment size)
int post_num_blocks(const int gw[3], const int gh[3]);
hipError_t post_init();   // per device: raise the dynamic-LDS limit of the post-process kernels
hipError_t postprocess_launch(const PostParams& p, hipStream_t stream);
int postprocess_desc(const PostParams& p, LaunchDesc out[2]);   // number of launches (1 or 2), or -1 on error

}  // namespace unina
