// c3k2_fused.hip -- one launch for a whole C3k2 block (reference: unina_yolo_dla/model.py:76-110, Bottleneck :53-73).
//
//     a | b = ReLU(BN(cv1(x))) | ReLU(BN(cv2(x)))                 1x1, Cin -> h | h          (step 0, one GEMM, N = 2h)
//     for each bottleneck:  t = ReLU(BN(b.cv1(a)))                1x1, h -> h                (odd steps)
//                           a = ReLU(BN(b.cv2(t))) + a            3x3, h -> h, pad 1         (even steps)
//     y = ReLU(BN(cv3(cat[a, b])))                                1x1, 2h -> 2h              (last step)
//
// The unfused op table runs this as 2 + 2n launches that each sit on the ~4 us launch / latency floor. Here a
// workgroup owns a TH x TW tile of the block's output and keeps every intermediate tensor of its tile IN LDS:
//   * the input patch (tile + n-pixel halo, all Cin channels) is DMA'd into LDS once (global_load_lds);
//   * every step is an MFMA GEMM whose B operand (activations, pixels as columns) is read from an LDS image;
//   * the A operand (weights) never touches LDS: a wave owns ONE 16-channel subtile of a step's output, so each
//     1-KiB weight fragment block is needed by exactly that wave (x the few waves that split the pixels) and is
//     loaded straight from L2 into its registers, 16 bytes per lane. All steps' blocks form one flat per-wave
//     sequence (packed by c3k2_pack in consumption order) that is prefetched D blocks ahead through a circular
//     register queue ACROSS step boundaries -- weights depend on nothing, so the L2 latency is paid once per launch
//     and the bytes in flight per CU are D KiB x waves, not an LDS ring (the first version streamed weights through
//     a 48 KiB LDS-DMA ring and was bound by DMA issue rate and ring depth: 33 us for the 917 KB of stage3's block);
//   * epilogues (bias, ReLU, zero outside the image = the 3x3's zero padding, residual) write fp16 LDS images;
//     only cv3's output goes to HBM (staged through LDS, full 16-byte row segments).
// The 1x1 convs in front of a 3x3 are recomputed on the halo (1.27-1.9x of their small cost). Every loop is
// unrolled at compile time (shapes, including Cin, are template parameters), so queue slots are plain registers and
// the compiler's own counted s_waitcnt vmcnt tracks the prefetches.
// Arithmetic is identical to the unfused kernels: same MFMA (v_mfma_f32_16x16x32_f16), same K order (tap-major, 32
// channels per block), same fp32 epilogue and the same fp16 rounding points -> results are bit-identical (tested).
//
#include "block_kernels.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace unina {

using namespace dev;

extern __shared__ __align__(16) unsigned char c3_smem[];

template <int H_, int TH, int TW, int NB, int CIN, int NW, int D, int TAIL = 0, typename E = EltH, int CPRE = 0, int CX = CIN>
__global__ __launch_bounds__(NW * 64) void c3k2_fused_kernel(const C3k2Params p) {
  c3k2_fused_body<H_, TH, TW, NB, CIN, NW, D, TAIL, E, CPRE, CX>(p, (int)blockIdx.x, c3_smem);
}

// debug twins with per-step stamps (unina_debug_block_stamps): the 40^2 blocks, whose dependent-step chain is what bounds them
__global__ __launch_bounds__(512) void c3k2_fused_stage3_stamped(const C3k2Params p) {
  c3k2_fused_body<128, 4, 4, 2, 256, 8, 16, 2, EltH, 0, 256, true>(p, (int)blockIdx.x, c3_smem);
}
__global__ __launch_bounds__(512) void c3k2_fused_pan2_stamped(const C3k2Params p) {
  c3k2_fused_body<128, 4, 4, 1, 384, 8, 16, 0, EltH, 128, 128, true>(p, (int)blockIdx.x, c3_smem);
}

// ------------------------------------------------------------------------------------------------- host side
namespace {

struct Class {
  int dtype, hid, nb, cin, tail, th, tw, nw;
  const char* name;
  void (*fn)(const C3k2Params);
  int cpre;   // 0, or the pre-conv's input channels
  int cx;     // channels the pre-conv produces (0 without one)
};
#define C3K2(H_, TH, TW, NB, CIN, NW, D) \
  {kF16, H_, NB, CIN, 0, TH, TW, NW, "c3k2_fused<" #H_ "," #TH "x" #TW "," #NB "," #CIN "," #NW "w>", c3k2_fused_kernel<H_, TH, TW, NB, CIN, NW, D, 0>, 0, 0}
#define C3K2S(H_, TH, TW, NB, CIN, NW, D, CPRE, CX) \
  {kF16, H_, NB, CIN, 0, TH, TW, NW, "c3k2_fused<" #H_ "," #TH "x" #TW "," #NB "," #CIN "," #NW "w,s2conv " #CPRE ">", c3k2_fused_kernel<H_, TH, TW, NB, CIN, NW, D, 0, EltH, CPRE, CX>, CPRE, CX}
#define C3K2IS(H_, TH, TW, NB, CIN, NW, D, CPRE, CX) \
  {kI8, H_, NB, CIN, 0, TH, TW, NW, "c3k2_fused<i8," #H_ "," #TH "x" #TW "," #NB "," #CIN "," #NW "w,s2conv " #CPRE ">", c3k2_fused_kernel<H_, TH, TW, NB, CIN, NW, D, 0, EltI8, CPRE, CX>, CPRE, CX}
#define C3K2T(H_, TH, TW, NB, CIN, NW, D) \
  {kF16, H_, NB, CIN, 1, TH, TW, NW, "c3k2_fused<" #H_ "," #TH "x" #TW "," #NB "," #CIN "," #NW "w,lat>", c3k2_fused_kernel<H_, TH, TW, NB, CIN, NW, D, 1>, 0, 0}
#define C3K2P(H_, TH, TW, NB, CIN, NW, D) \
  {kF16, H_, NB, CIN, 2, TH, TW, NW, "c3k2_fused<" #H_ "," #TH "x" #TW "," #NB "," #CIN "," #NW "w,+1x1>", c3k2_fused_kernel<H_, TH, TW, NB, CIN, NW, D, 2>, 0, 0}
#define C3K2IP(H_, TH, TW, NB, CIN, NW, D) \
  {kI8, H_, NB, CIN, 2, TH, TW, NW, "c3k2_fused<i8," #H_ "," #TH "x" #TW "," #NB "," #CIN "," #NW "w,+1x1>", c3k2_fused_kernel<H_, TH, TW, NB, CIN, NW, D, 2, EltI8>, 0, 0}
#define C3K2IL(H_, TH, TW, NB, CIN, NW, D) \
  {kI8, H_, NB, CIN, 3, TH, TW, NW, "c3k2_fused<i8," #H_ "," #TH "x" #TW "," #NB "," #CIN "," #NW "w,lat f16>", c3k2_fused_kernel<H_, TH, TW, NB, CIN, NW, D, 3, EltI8>, 0, 0}
// STRICT engines (split fp16: hi / lo images, 2-KiB weight block pairs, D pairs in flight per wave)
#define C3K2X(H_, TH, TW, NB, CIN, NW, D, TAIL, TN) \
  {kS16, H_, NB, CIN, TAIL, TH, TW, NW, "c3k2_fused<s16," #H_ "," #TH "x" #TW "," #NB "," #CIN "," #NW "w" TN ">", c3k2_fused_kernel<H_, TH, TW, NB, CIN, NW, D, TAIL, EltS>, 0, 0}
#define C3K2XS(H_, TH, TW, NB, CIN, NW, D, CPRE, CX) \
  {kS16, H_, NB, CIN, 0, TH, TW, NW, "c3k2_fused<s16," #H_ "," #TH "x" #TW "," #NB "," #CIN "," #NW "w,s2conv " #CPRE ">", c3k2_fused_kernel<H_, TH, TW, NB, CIN, NW, D, 0, EltS, CPRE, CX>, CPRE, CX}
#define C3K2I(H_, TH, TW, NB, CIN, NW, D) \
  {kI8, H_, NB, CIN, 0, TH, TW, NW, "c3k2_fused<i8," #H_ "," #TH "x" #TW "," #NB "," #CIN "," #NW "w>", c3k2_fused_kernel<H_, TH, TW, NB, CIN, NW, D, 0, EltI8>, 0, 0}
const Class kClasses[] = {
    // Tiles chosen by end-to-end A/B at 640^2: twice the workgroups of the first choice (8x16 / 8x8 / 4x8) cost more
    // halo recompute but cut the serial frame by ~10 us at equal throughput; one step smaller still (4x8 / 4x4 / 4x4)
    // lost 4-6 % throughput; deeper weight queues (D x2, x1.5) changed nothing.
    C3K2(32, 8, 8, 1, 64, 8, 4),       // backbone.stage1_block        160^2 at 640: 400 workgroups
    C3K2S(32, 8, 8, 1, 64, 8, 4, 32, 64),     // backbone.stage1_conv (3x3/s2, 32 -> 64) + backbone.stage1_block
    // (stage2_conv + stage2_c3k2, n = 2: the 2-pixel halo recomputes the conv 3x -- round 2: 18.3 vs 9.8 + 12.5 us as launches,
    //  round 3 with lane-order weights: 17.0 vs 8.4 + 11.7 us; both times the serial frame and frames/s did not move: not instantiated)
    C3K2S(64, 4, 8, 1, 192, 8, 8, 64, 64),    // neck.down1 (3x3/s2, 64 -> 64 = the first half of [p2_down | p3_fused]) + neck.pan_c3k2_1
    C3K2S(128, 4, 4, 1, 384, 8, 16, 128, 128),  // neck.down2 + neck.pan_c3k2_2
    C3K2(32, 8, 8, 1, 128, 8, 4),      // neck.fpn_c3k2_2
    C3K2(64, 4, 8, 2, 128, 8, 8),      // backbone.stage2_c3k2          80^2: 200
    C3K2(64, 4, 8, 1, 256, 8, 8),      // neck.fpn_c3k2_1
    C3K2T(64, 4, 8, 1, 256, 8, 8),     // neck.fpn_c3k2_1 + neck.lateral_p2 (+ x2 upsample)
    C3K2(64, 4, 8, 1, 192, 8, 8),      // neck.pan_c3k2_1
    C3K2(128, 4, 4, 2, 256, 8, 16),    // backbone.stage3_c3k2          40^2: 100  (16 waves: 27 vs 20 us)
    C3K2P(128, 4, 4, 2, 256, 8, 16),   // backbone.stage3_c3k2 + backbone.sppf.cv1 (model.py:215-216)
    C3K2(128, 4, 4, 1, 384, 8, 16),    // neck.pan_c3k2_2
    C3K2(128, 4, 4, 1, 512, 8, 16),    // graph (B) fpn_c3k2_1 (qat.py:397)
    C3K2T(128, 4, 4, 1, 512, 8, 16),   // graph (B) fpn_c3k2_1 + lateral_p3
    // INT8 engines: the blocks whose tensors are all int8 (h = 32 blocks touch the fp16 carve-outs: train.py:779)
    C3K2I(64, 4, 8, 2, 128, 8, 8),     // backbone.stage2_c3k2
    C3K2I(64, 4, 8, 1, 256, 8, 8),     // neck.fpn_c3k2_1
    C3K2IL(64, 4, 8, 1, 256, 8, 8),    // neck.fpn_c3k2_1 + neck.lateral_p2 (+ x2 upsample) into the fp16 concat of fpn_c3k2_2
    C3K2I(64, 4, 8, 1, 192, 8, 8),     // neck.pan_c3k2_1
    C3K2IS(64, 4, 8, 1, 192, 8, 8, 64, 64),     // neck.down1 + neck.pan_c3k2_1
    C3K2IS(128, 4, 4, 1, 384, 8, 8, 128, 128),  // neck.down2 + neck.pan_c3k2_2
    C3K2I(128, 4, 4, 2, 256, 8, 8),    // backbone.stage3_c3k2
    C3K2IP(128, 4, 4, 2, 256, 8, 8),   // backbone.stage3_c3k2 + backbone.sppf.cv1
    C3K2I(128, 4, 4, 1, 384, 8, 8),    // neck.pan_c3k2_2
    C3K2I(128, 4, 4, 1, 512, 8, 8),    // graph (B) fpn_c3k2_1
    // STRICT engines: the same seven blocks of graph (A) on split-fp16 tensors
    C3K2X(32, 8, 8, 1, 64, 8, 4, 0, ""),            // backbone.stage1_block
    C3K2XS(32, 8, 8, 1, 64, 8, 4, 32, 64),          // backbone.stage1_conv + backbone.stage1_block
    C3K2XS(64, 4, 8, 1, 192, 8, 4, 64, 64),         // neck.down1 + neck.pan_c3k2_1
    C3K2XS(128, 4, 4, 1, 384, 8, 8, 128, 128),      // neck.down2 + neck.pan_c3k2_2
    C3K2X(32, 8, 8, 1, 128, 8, 4, 0, ""),           // neck.fpn_c3k2_2
    C3K2X(64, 4, 8, 2, 128, 8, 4, 0, ""),           // backbone.stage2_c3k2
    C3K2X(64, 4, 8, 1, 256, 8, 4, 0, ""),           // neck.fpn_c3k2_1
    C3K2X(64, 4, 8, 1, 256, 8, 4, 1, ",lat"),       // neck.fpn_c3k2_1 + neck.lateral_p2 (+ x2 upsample)
    C3K2X(64, 4, 8, 1, 192, 8, 4, 0, ""),           // neck.pan_c3k2_1
    C3K2X(128, 4, 4, 2, 256, 8, 8, 0, ""),          // backbone.stage3_c3k2
    C3K2X(128, 4, 4, 2, 256, 8, 8, 2, ",+1x1"),     // backbone.stage3_c3k2 + backbone.sppf.cv1
    C3K2X(128, 4, 4, 1, 384, 8, 8, 0, ""),          // neck.pan_c3k2_2
};
#undef C3K2
#undef C3K2T
#undef C3K2S
#undef C3K2IS
#undef C3K2I
#undef C3K2P
#undef C3K2IL
#undef C3K2IP
#undef C3K2X
#undef C3K2XS
const Class* find_class(int hid, int nb, int cin, int tail, int dtype, int cpre = 0, int cx = 0) {
  for (const Class& c : kClasses)
    if (c.dtype == dtype && c.hid == hid && c.nb == nb && c.cin == cin && c.tail == tail && c.cpre == cpre && (!cpre || c.cx == cx)) return &c;
  return nullptr;
}
constexpr int kMaxLds = 160 * 1024;
int align_up(int v, int a) { return (v + a - 1) / a * a; }

}  // namespace

hipError_t c3k2_launch_stamped(const C3k2Params& p, hipStream_t stream) {
  void (*fn)(const C3k2Params) = nullptr;
  if (p.dtype == kF16 && p.hid == 128 && p.nb == 2 && p.Cin == 256 && p.tail == 2 && !p.cpre) fn = c3k2_fused_stage3_stamped;
  if (p.dtype == kF16 && p.hid == 128 && p.nb == 1 && p.Cin == 384 && p.tail == 0 && p.cpre == 128 && p.cx == 128) fn = c3k2_fused_pan2_stamped;
  if (!fn) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3(p.tiles_x * p.tiles_y, 1, 1), dim3(512, 1, 1), p.smem_bytes, stream, p);
  return hipGetLastError();
}

hipError_t c3k2_init() {
  for (const void* f : {reinterpret_cast<const void*>(c3k2_fused_stage3_stamped), reinterpret_cast<const void*>(c3k2_fused_pan2_stamped)}) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) return e;
  }
  for (const Class& c : kClasses) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

bool c3k2_supported(int hid, int nb, int cin, int tail, int dtype, int cpre, int cx) {
  C3k2Params p;
  memset(&p, 0, sizeof p);
  p.dtype = dtype;
  p.cpre = cpre;
  p.cx = cx;
  p.hid = hid; p.nb = nb; p.Cin = cin; p.tail = tail; p.H = p.W = 64;
  return c3k2_layout(&p);
}

// Fills tile geometry and the LDS layout of `p` (needs hid, nb, Cin, H, W). False = no such class / no fit.
bool c3k2_layout(C3k2Params* p) {
  const Class* c = find_class(p->hid, p->nb, p->Cin, p->tail, p->dtype, p->cpre, p->cx);
  if (!c) return false;
  const int h = p->hid, nb = p->nb;
  const int esz = p->dtype == kI8 ? 1 : 2, cm = p->dtype == kI8 ? 3 : 1;
  const int p0 = (c->th + 2 * nb) * (c->tw + 2 * nb), p1 = (c->th + 2) * (c->tw + 2), pt = c->th * c->tw;
  p->tiles_x = (p->W + c->tw - 1) / c->tw;
  p->tiles_y = (p->H + c->th - 1) / c->th;
  p->tiles_x_magic = div_magic((unsigned)p->tiles_x);
  p->n_bias = cm * (2 * h * (2 + nb) + (p->tail ? h : 0) + (p->cpre ? p->cx : 0));
  // input image(s): DMA'd whole; or the pre-conv's output (written by its epilogue) followed by the DMA'd rest of a concat
  const int xa_bytes = p->cpre ? align_up(p0 * p->cx * esz, 1024) : 0;
  const int x_bytes = p->cpre ? xa_bytes + (p->Cin > p->cx ? align_up(p0 * (p->Cin - p->cx) * esz, 1024) + 1024 : 0)
                              : align_up(p0 * p->Cin * esz, 1024) + 1024;  // the last patch DMA instruction may overrun by < 1 KiB
  const int t_bytes = p0 * h * esz, u1_bytes = nb == 2 ? p1 * h * esz : 0, u2_bytes = pt * h * esz;
  const int stage_bytes = pt * (2 * h * esz + 16);
  // region A: the input patch; once step 0 has consumed it, t (and later the output staging tile), u1 and u2 live there
  const int head = t_bytes > stage_bytes ? t_bytes : stage_bytes;
  const int a_need = align_up(head, 16) + align_up(u1_bytes, 16) + align_up(u2_bytes, 16);
  const int a_bytes = x_bytes > a_need ? x_bytes : a_need;
  int off = 0;
  p->off_bias = off; off += align_up(p->n_bias * 4, 1024);
  p->off_x = off;
  p->off_xr = off + xa_bytes;
  p->off_t = off;
  p->off_stage = off;
  p->off_u1 = off + align_up(head, 16);
  p->off_u2 = p->off_u1 + align_up(u1_bytes, 16);
  off += align_up(a_bytes, 1024);
  p->off_y = off;
  p->off_tail = off;                                          // the tail's output tile replaces a | b (dead after cv3)
  p->off_p = off;                                             // the pre-conv's input patch lives where a | b will (dead after the pre-step)
  const int y_bytes = p0 * 2 * h * esz, tail_bytes = p->tail ? pt * (h * (p->tail == 3 ? 2 : esz) + 16) : 0;
  const int ph = 2 * (c->th + 2 * nb - 1) + 3, pw = 2 * (c->tw + 2 * nb - 1) + 3;
  const int p_bytes = p->cpre ? align_up(ph * pw * p->cpre * esz, 1024) + 1024 : 0;   // (+ DMA overrun of the last instruction)
  int yr = y_bytes > tail_bytes ? y_bytes : tail_bytes;
  yr = yr > p_bytes ? yr : p_bytes;
  off += align_up(yr, 1024);
  p->lds_lo = 0;
  if (p->dtype == kS16) {   // every image has a lo twin: the whole image area once more behind itself
    p->lds_lo = off - p->off_x;
    off += p->lds_lo;
  }
  p->smem_bytes = off;
  return off <= kMaxLds;
}

void weight_block_to_lane_order(const unsigned char* in, unsigned char* out) {
  static const int G[4] = {0, 2, 3, 1};
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 4; ++c) memcpy(out + (16 * c + r) * 16, in + (4 * r + (c ^ G[r >> 2])) * 16, 16);
}

// Packs the weights of a block's convs (each given as the exporter's [n/16][K/32] 1-KiB fragment blocks, up to two
// output slices) into the stream the block kernels read: per conv, k-block-major [K/32][N/16] blocks (the wave that
// owns subtile j of step s reads blocks blk(s) + kb*ns + j, kb = 0..), and concatenates the biases.
// Inside a 1-KiB block the 16-byte slots go from the file's LDS-image order to LANE order (weight_block_to_lane_order,
// kernels.h): the kernels take weights straight into registers, lane l loading slot l -- one contiguous KiB per wave
// instruction (tools/probes/ingest_probe: 107 against 55 bytes per clock and CU for the permuted form).
void block_pack(const C3k2Conv* convs, int nconv, std::vector<unsigned char>* stream, std::vector<float>* bias, int dtype) {
  stream->clear();
  bias->clear();
  for (int ci = 0; ci < nconv; ++ci) {
    const C3k2Conv& cv = convs[ci];
    const int ns = (cv.n[0] + cv.n[1]) / 16, kbn = cv.K / (dtype == kI8 ? 64 : 32);
    const size_t wblk = dtype == kS16 ? 2048 : 1024;   // split fp16: a block is the (hi | lo) pair
    const size_t base = stream->size();
    stream->resize(base + (size_t)kbn * ns * wblk, 0);
    for (int kb = 0; kb < kbn; ++kb)
      for (int s = 0; s < ns; ++s) {
        const int seg = s * 16 < cv.n[0] ? 0 : 1;
        const int ls = seg ? s - cv.n[0] / 16 : s;
        for (size_t h = 0; h < wblk; h += 1024)
          weight_block_to_lane_order(cv.w[seg] + ((size_t)ls * kbn + kb) * wblk + h, stream->data() + base + ((size_t)kb * ns + s) * wblk + h);
      }
    for (int seg = 0; seg < 2; ++seg)
      for (int i = 0; i < cv.n[seg]; ++i) bias->push_back(cv.bias[seg][i]);
    if (dtype == kI8) {   // [bias | mult | 1/s_out] per conv (block_pipeline.h act_relu / store4)
      for (int seg = 0; seg < 2; ++seg)
        for (int i = 0; i < cv.n[seg]; ++i) bias->push_back(cv.mult[seg][i]);
      for (int seg = 0; seg < 2; ++seg)
        for (int i = 0; i < cv.n[seg]; ++i) bias->push_back(cv.out_inv[seg]);
    }
  }
}

// C3k2 convs in execution order: cv1|cv2, {b.cv1, b.cv2} x nb, cv3.
bool c3k2_pack(int hid, int nb, int cin, int tail, const C3k2Conv* convs, std::vector<unsigned char>* stream, std::vector<float>* bias,
               int dtype, int cpre, int cx) {
  if (!find_class(hid, nb, cin, tail, dtype, cpre, cx)) return false;
  const int pre = cpre ? 1 : 0;
  if (pre && (convs[0].n[0] != cx || convs[0].n[1] || convs[0].K != 9 * cpre)) return false;
  const int ncv3 = 1 + 2 * nb, nconv = 2 + 2 * nb + (tail ? 1 : 0);
  for (int ci = 0; ci < nconv; ++ci) {
    const C3k2Conv& cv = convs[pre + ci];
    const int want_n = (ci == 0 || ci == ncv3) ? 2 * hid : hid;
    const int want_k = ci == 0 ? cin : (ci >= ncv3 ? 2 * hid : ((ci & 1) ? hid : 9 * hid));
    if (cv.n[0] + cv.n[1] != want_n || cv.K != want_k || cv.n[0] % 16 || cv.n[1] % 16) return false;
    if (dtype == kI8 && (!cv.mult[0] || (cv.n[1] && !cv.mult[1]))) return false;
  }
  block_pack(convs, pre + nconv, stream, bias, dtype);
  return true;
}

hipError_t c3k2_launch(const C3k2Params& p, hipStream_t stream) {
  const Class* c = find_class(p.hid, p.nb, p.Cin, p.tail, p.dtype, p.cpre, p.cx);
  if (!c) return hipErrorInvalidValue;
  hipLaunchKernelGGL(c->fn, dim3(p.tiles_x * p.tiles_y, 1, 1), dim3(c->nw * 64, 1, 1), p.smem_bytes, stream, p);
  return hipGetLastError();
}

bool c3k2_tile_is(const C3k2Params& p, int th, int tw) {
  const Class* c = find_class(p.hid, p.nb, p.Cin, p.tail, p.dtype, p.cpre, p.cx);
  return c && c->th == th && c->tw == tw;
}

const char* c3k2_kernel_name(int hid, int nb, int cin, int tail, int dtype, int cpre, int cx) {
  const Class* c = find_class(hid, nb, cin, tail, dtype, cpre, cx);
  return c ? c->name : "c3k2_fused<?>";
}

int c3k2_block_threads(int hid, int nb, int cin, int tail, int dtype, int cpre, int cx) {
  const Class* c = find_class(hid, nb, cin, tail, dtype, cpre, cx);
  return c ? c->nw * 64 : 0;
}

}  // namespace unina
