// c3k2_fused.hip -- one launch for a whole C3k2 block (reference: unina_yolo_dla/model.py:76-110, Bottleneck :53-73).
//
//     a | b = ReLU(BN(cv1(x))) | ReLU(BN(cv2(x)))                 1x1, Cin -> h | h          (step S1, one GEMM, N = 2h)
//     for each bottleneck:  t = ReLU(BN(b.cv1(a)))                1x1, h -> h                (S2)
//                           a = ReLU(BN(b.cv2(t))) + a            3x3, h -> h, pad 1         (S3)
//     y = ReLU(BN(cv3(cat[a, b])))                                1x1, 2h -> 2h              (S4)
//
// The unfused op table runs this as 2 + 2n launches that each sit on the ~4 us launch / latency floor. Here a
// workgroup owns a TH x TW tile of the block's output and keeps every intermediate tensor of its tile IN LDS:
//   * the input patch (tile + n-pixel halo, all Cin channels) is DMA'd into LDS once;
//   * every step is an MFMA GEMM whose B operand (activations, pixels as columns) is read from an LDS image and whose
//     A operand (weights) streams from L2 through an LDS-DMA ring that never stops between steps: the weights of all
//     steps are ONE flat stream of fixed-size stages (packed by the host in consumption order, c3k2_pack), so the
//     next step's weights are already landing while the current step's epilogue runs;
//   * epilogues (bias, ReLU, zero outside the image = the 3x3's zero padding, residual) write fp16 LDS images;
//     only cv3's output goes to HBM (staged through LDS, full 16-byte row segments).
// The 1x1 convs in front of a 3x3 are recomputed on the halo (1.27-1.9x of their small cost).
// Arithmetic is identical to the unfused kernels: same MFMA (v_mfma_f32_16x16x32_f16), same K order (tap-major, 32
// channels per block), same fp32 epilogue and the same fp16 rounding points -> results are bit-identical (tested).
//
// LDS image: pixel row r owns nch 16-byte chunks (8 channels each); chunk c lives at slot c ^ ((r >> sh) & mask) of
// its row, (sh, mask) chosen from the row pitch so that the 16 pixels of a fragment read hit 16 different bank slots.
#include "kernels.h"
#include "mfma_common.h"

#include <cstring>
#include <vector>

namespace unina {

using namespace dev;

namespace {

struct Img {
  int base, nch, sh, mask;
  __device__ __forceinline__ int key(int row) const { return (row >> sh) & mask; }
  __device__ __forceinline__ int addr(int row, int chunk) const { return base + ((row * nch + (chunk ^ key(row))) << 4); }
};
__host__ __device__ constexpr Img make_img(int base, int nch) {
  // pitch = nch 16-byte slots; a ds_read_b128 group covers 16 slots' worth of banks
  return (nch % 16 == 0) ? Img{base, nch, 0, 15} : ((nch % 8 == 0) ? Img{base, nch, 1, 7} : Img{base, nch, 2, 3});
}

template <int SB, int RING>
struct Pipe {
  static constexpr int LPT = SB / 4, STAGE_BYTES = SB * 1024;
  const unsigned char* gsrc;  // this lane's source address inside stage 0 (block wid, slot lane)
  unsigned char* ring;        // LDS: RING stage buffers
  int total, issued, cur, ibuf, cbuf, wid;

  __device__ __forceinline__ void issue() {
    const unsigned char* g = gsrc + (size_t)issued * STAGE_BYTES;
    unsigned char* l = ring + ibuf * STAGE_BYTES + wid * 1024;
#pragma unroll
    for (int q = 0; q < LPT; ++q) glds16(g + q * 4096, l + q * 4096);
    ++issued;
    ibuf = ibuf + 1 == RING ? 0 : ibuf + 1;
  }
  // Stage `cur` is usable after this: its DMA has landed in every wave's view (counted vmcnt + barrier); the same
  // barrier retires every wave's reads of stage cur-1 (its buffer is refilled here) and publishes the LDS writes of
  // the previous step's epilogue (lgkmcnt(0) first).
  __device__ __forceinline__ void acquire() {
    wait_stages<LPT>(issued - cur - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (issued < total) issue();
  }
  __device__ __forceinline__ const unsigned char* stage() const { return ring + cbuf * STAGE_BYTES; }
  __device__ __forceinline__ void release() {
    ++cur;
    cbuf = cbuf + 1 == RING ? 0 : cbuf + 1;
  }
};

constexpr int waves_m_for(int P) { return ((P + 15) / 16 >= 8) ? 4 : 2; }
constexpr int stages_of(int kb, int n, int sb) { return (kb + sb / (n / 16) - 1) / (sb / (n / 16)); }

// One GEMM step: D[N channels][P pixels] = sum over KB k-blocks of W-block x X-block.
//   baddr(sub, kb) : LDS byte address of this lane's 16-byte B fragment of pixel subtile `sub`, k-block kb
//   epi(sub, n, acc): consumes the 4 channels n..n+3 of pixel sub*16 + (lane & 15)
template <int P, int N, int SB, int RING, typename BAddr, typename Epi>
__device__ __forceinline__ void gemm_step(Pipe<SB, RING>& pipe, const unsigned char* smem, int KB, int lane, int wid,
                                          BAddr baddr, Epi epi) {
  constexpr int WAVES_M = waves_m_for(P), WAVES_N = 4 / WAVES_M;
  constexpr int MS = (P + 15) / 16, NS = N / 16;
  constexpr int WM_T = (MS + WAVES_M - 1) / WAVES_M, WN_T = NS / WAVES_N, KPS = SB / NS;
  static_assert(NS % WAVES_N == 0 && SB % NS == 0 && KPS >= 1, "step tiling");
  const int wm = wid % WAVES_M, wn = wid / WAVES_M;
  const int l15 = lane & 15, lq = lane >> 4;
  const int rd_off = (4 * l15 + (lq ^ swz_g(l15))) * 16;
  floatx4 acc[WN_T][WM_T];
#pragma unroll
  for (int j = 0; j < WN_T; ++j)
#pragma unroll
    for (int i = 0; i < WM_T; ++i) acc[j][i] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int nst = (KB + KPS - 1) / KPS;
  for (int s = 0; s < nst; ++s) {
    pipe.acquire();
    const unsigned char* sb = pipe.stage() + rd_off;
#pragma unroll
    for (int jk = 0; jk < KPS; ++jk) {
      int kb = s * KPS + jk;
      kb = kb < KB ? kb : KB - 1;  // zero-padded weight blocks past the end: multiply finite data by 0
      half8 a[WN_T], b[WM_T];
#pragma unroll
      for (int i = 0; i < WM_T; ++i) b[i] = *reinterpret_cast<const half8*>(smem + baddr(wm * WM_T + i, kb));
#pragma unroll
      for (int j = 0; j < WN_T; ++j) a[j] = *reinterpret_cast<const half8*>(sb + ((jk * NS + wn * WN_T + j) << 10));
#pragma unroll
      for (int j = 0; j < WN_T; ++j)
#pragma unroll
        for (int i = 0; i < WM_T; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[i], acc[j][i], 0, 0, 0);
    }
    pipe.release();
  }
#pragma unroll
  for (int j = 0; j < WN_T; ++j)
#pragma unroll
    for (int i = 0; i < WM_T; ++i)
      if (wm * WM_T + i < MS) epi(wm * WM_T + i, (wn * WN_T + j) * 16 + lq * 4, acc[j][i]);
}

__device__ __forceinline__ floatx4 bias_relu(const floatx4& acc, const float* bias_lds, int n) {
  floatx4 v = acc + *reinterpret_cast<const floatx4*>(bias_lds + n);
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
  return v;
}
__device__ __forceinline__ void store_h4(unsigned char* smem, const Img& im, int row, int n, const floatx4& v) {
  half4 hv;
#pragma unroll
  for (int r = 0; r < 4; ++r) hv[r] = (half_t)v[r];
  *reinterpret_cast<half4*>(smem + im.addr(row, n >> 3) + (n & 4) * 2) = hv;
}
__device__ __forceinline__ floatx4 load_h4(const unsigned char* smem, const Img& im, int row, int n) {
  const half4 hv = *reinterpret_cast<const half4*>(smem + im.addr(row, n >> 3) + (n & 4) * 2);
  return floatx4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
}

}  // namespace

extern __shared__ __align__(16) unsigned char c3_smem[];

template <int H_, int TH, int TW, int NB>
__global__ __launch_bounds__(256) void c3k2_fused_kernel(const C3k2Params p) {
  static_assert(NB == 1 || NB == 2, "bottleneck count");
  constexpr int SB = H_ >= 128 ? 16 : 8, RING = H_ >= 128 ? 3 : 4;
  constexpr int R0W = TW + 2 * NB, P0 = (TH + 2 * NB) * R0W;   // input / first-level region (tile + NB-pixel halo)
  constexpr int R1W = TW + 2, P1 = (TH + 2) * R1W;              // NB == 2: region of the first bottleneck's output
  constexpr int PT = TH * TW;
  constexpr int HB = H_ / 32;                                   // k-blocks per tap of the hidden width
  typedef Pipe<SB, RING> PipeT;

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  unsigned char* smem = c3_smem;
  const int tyi = fast_div((int)blockIdx.x, p.tiles_x_magic), txi = (int)blockIdx.x - tyi * p.tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const half_t* zeros = reinterpret_cast<const half_t*>(p.zeros);

  // biases of every step -> LDS (read by the epilogues without touching the vmcnt bookkeeping of the ring)
  float* bias_lds = reinterpret_cast<float*>(smem + p.off_bias);
  for (int i = threadIdx.x; i < p.n_bias; i += 256) bias_lds[i] = p.bias[i];

  // input patch: 16-byte slot s = (region pixel r, chunk cs); out-of-image pixels read the zero page
  const Img X = make_img(p.off_x, p.Cin >> 3);
  {
    const int nchx = p.Cin >> 3, nslots = P0 * nchx;
    for (int s0 = wid * 64; s0 < nslots; s0 += 256) {
      const int s = s0 + lane;
      const half_t* g = zeros;
      if (s < nslots) {
        const int r = fast_div(s, p.nchx_magic), cs = s - r * nchx;
        const int ry = r / R0W, rx = r - ry * R0W;
        const int iy = ty0 - NB + ry, ix = tx0 - NB + rx;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          g = p.src + (size_t)(iy * p.W + ix) * p.src_ld + ((cs ^ X.key(r)) << 3);
      }
      glds16(g, smem + p.off_x + s0 * 16);
    }
  }

  PipeT pipe;
  pipe.gsrc = p.wstream + wid * 1024 + lane * 16;
  pipe.ring = smem + p.off_ring;
  pipe.total = p.total_stages;
  pipe.issued = pipe.cur = pipe.ibuf = pipe.cbuf = 0;
  pipe.wid = wid;
  for (int s = 0; s < RING - 1 && s < pipe.total; ++s) pipe.issue();

  const Img Y = make_img(p.off_y, 2 * H_ / 8);   // a | b on R0
  const Img T = make_img(p.off_t, H_ / 8);       // t of the current bottleneck (R0, then R1)
  const Img U1 = make_img(p.off_u1, H_ / 8);     // NB == 2: first bottleneck's output on R1
  const Img U2 = make_img(p.off_u2, H_ / 8);     // last bottleneck's output on the tile
  auto in_image = [&](int iy, int ix) { return (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W; };

  // ---- S1: a | b = ReLU(W12 x + b12) on R0 ------------------------------------------------------------------------
  gemm_step<P0, 2 * H_, SB, RING>(
      pipe, smem, p.Cin >> 5, lane, wid,
      [&](int sub, int kb) {
        const int r = sub * 16 + l15;
        return X.addr(r < P0 ? r : P0 - 1, kb * 4 + lq);
      },
      [&](int sub, int n, const floatx4& acc) {
        const int r = sub * 16 + l15;
        if (r < P0) store_h4(smem, Y, r, n, bias_relu(acc, bias_lds, n));
      });
  const float* bias_b = bias_lds + 2 * H_;

  // ---- bottleneck 0 -----------------------------------------------------------------------------------------------
  // S2: t = ReLU(Wb1 a + b) on R0, forced to 0 outside the image (zero padding of the 3x3 that follows)
  gemm_step<P0, H_, SB, RING>(
      pipe, smem, HB, lane, wid,
      [&](int sub, int kb) {
        const int r = sub * 16 + l15;
        return Y.addr(r < P0 ? r : P0 - 1, kb * 4 + lq);
      },
      [&](int sub, int n, const floatx4& acc) {
        const int r = sub * 16 + l15;
        if (r >= P0) return;
        const int ry = r / R0W, rx = r - ry * R0W;
        floatx4 v = bias_relu(acc, bias_b, n);
        if (!in_image(ty0 - NB + ry, tx0 - NB + rx)) v = floatx4{0.f, 0.f, 0.f, 0.f};
        store_h4(smem, T, r, n, v);
      });
  if constexpr (NB == 1) {
    // S3: u = ReLU(3x3(t) + b) + a on the tile
    gemm_step<PT, H_, SB, RING>(
        pipe, smem, 9 * HB, lane, wid,
        [&](int sub, int kb) {
          int pp = sub * 16 + l15;
          pp = pp < PT ? pp : PT - 1;
          const int py = pp / TW, px = pp - py * TW;
          const int tap = kb / HB, cb = kb - tap * HB, th3 = tap / 3;
          return T.addr((py + th3) * R0W + px + (tap - th3 * 3), cb * 4 + lq);
        },
        [&](int sub, int n, const floatx4& acc) {
          const int pp = sub * 16 + l15;
          if (pp >= PT) return;
          const int py = pp / TW, px = pp - py * TW;
          const floatx4 v = bias_relu(acc, bias_b + H_, n) + load_h4(smem, Y, (py + 1) * R0W + px + 1, n);
          store_h4(smem, U2, pp, n, v);
        });
  } else {
    // S3a: u1 = ReLU(3x3(t1) + b) + a on R1
    gemm_step<P1, H_, SB, RING>(
        pipe, smem, 9 * HB, lane, wid,
        [&](int sub, int kb) {
          int pp = sub * 16 + l15;
          pp = pp < P1 ? pp : P1 - 1;
          const int py = pp / R1W, px = pp - py * R1W;
          const int tap = kb / HB, cb = kb - tap * HB, th3 = tap / 3;
          return T.addr((py + th3) * R0W + px + (tap - th3 * 3), cb * 4 + lq);
        },
        [&](int sub, int n, const floatx4& acc) {
          const int pp = sub * 16 + l15;
          if (pp >= P1) return;
          const int py = pp / R1W, px = pp - py * R1W;
          const floatx4 v = bias_relu(acc, bias_b + H_, n) + load_h4(smem, Y, (py + 1) * R0W + px + 1, n);
          store_h4(smem, U1, pp, n, v);
        });
    // ---- bottleneck 1 ---------------------------------------------------------------------------------------------
    const float* bias_c = bias_b + 2 * H_;
    // S2b: t2 = ReLU(Wb1' u1 + b) on R1, 0 outside the image
    gemm_step<P1, H_, SB, RING>(
        pipe, smem, HB, lane, wid,
        [&](int sub, int kb) {
          const int r = sub * 16 + l15;
          return U1.addr(r < P1 ? r : P1 - 1, kb * 4 + lq);
        },
        [&](int sub, int n, const floatx4& acc) {
          const int r = sub * 16 + l15;
          if (r >= P1) return;
          const int ry = r / R1W, rx = r - ry * R1W;
          floatx4 v = bias_relu(acc, bias_c, n);
          if (!in_image(ty0 - 1 + ry, tx0 - 1 + rx)) v = floatx4{0.f, 0.f, 0.f, 0.f};
          store_h4(smem, T, r, n, v);
        });
    // S3b: u2 = ReLU(3x3(t2) + b) + u1 on the tile
    gemm_step<PT, H_, SB, RING>(
        pipe, smem, 9 * HB, lane, wid,
        [&](int sub, int kb) {
          int pp = sub * 16 + l15;
          pp = pp < PT ? pp : PT - 1;
          const int py = pp / TW, px = pp - py * TW;
          const int tap = kb / HB, cb = kb - tap * HB, th3 = tap / 3;
          return T.addr((py + th3) * R1W + px + (tap - th3 * 3), cb * 4 + lq);
        },
        [&](int sub, int n, const floatx4& acc) {
          const int pp = sub * 16 + l15;
          if (pp >= PT) return;
          const int py = pp / TW, px = pp - py * TW;
          const floatx4 v = bias_relu(acc, bias_c + H_, n) + load_h4(smem, U1, (py + 1) * R1W + px + 1, n);
          store_h4(smem, U2, pp, n, v);
        });
  }

  // ---- S4: y = ReLU(W3 [u | b] + b3) on the tile -> staging image (linear rows) -> HBM ----------------------------
  constexpr int ROWB = 2 * H_ * 2 + 16;  // staged output row: 2h halfs + 16 bytes of padding (bank spread)
  unsigned char* stage = smem + p.off_stage;
  const float* bias_3 = bias_lds + 2 * H_ * (1 + NB);
  gemm_step<PT, 2 * H_, SB, RING>(
      pipe, smem, 2 * HB, lane, wid,
      [&](int sub, int kb) {
        int pp = sub * 16 + l15;
        pp = pp < PT ? pp : PT - 1;
        const int py = pp / TW, px = pp - py * TW;
        const int a0 = U2.addr(pp, kb * 4 + lq);                                            // k-blocks [0, h/32): u
        const int a1 = Y.addr((py + NB) * R0W + px + NB, H_ / 8 + (kb - HB) * 4 + lq);      // k-blocks [h/32, 2h/32): b
        return kb < HB ? a0 : a1;
      },
      [&](int sub, int n, const floatx4& acc) {
        const int pp = sub * 16 + l15;
        if (pp >= PT) return;
        const floatx4 v = bias_relu(acc, bias_3, n);
        half4 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = (half_t)v[r];
        *reinterpret_cast<half4*>(stage + pp * ROWB + n * 2) = hv;
      });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  typedef float vec16 __attribute__((ext_vector_type(4)));  // 16 opaque bytes
  constexpr int CPR = 2 * H_ * 2 / 16;                      // 16-byte chunks per output pixel
  for (int c = threadIdx.x; c < PT * CPR; c += 256) {
    const int pp = c / CPR, ch = c - pp * CPR;
    const int oy = ty0 + pp / TW, ox = tx0 + pp % TW;
    if (oy < p.H && ox < p.W)
      *reinterpret_cast<vec16*>(p.dst + (size_t)(oy * p.W + ox) * p.dst_ld + ch * 8) =
          *reinterpret_cast<const vec16*>(stage + pp * ROWB + ch * 16);
  }
}

// ------------------------------------------------------------------------------------------------- host side
namespace {

struct Class {
  int hid, nb, th, tw, sb, ring;
  void (*fn)(const C3k2Params);
};
const Class kClasses[] = {
    {32, 1, 8, 16, 8, 4, c3k2_fused_kernel<32, 8, 16, 1>},
    {64, 1, 8, 8, 8, 4, c3k2_fused_kernel<64, 8, 8, 1>},
    {64, 2, 8, 8, 8, 4, c3k2_fused_kernel<64, 8, 8, 2>},
    {128, 1, 4, 8, 16, 3, c3k2_fused_kernel<128, 4, 8, 1>},
    {128, 2, 4, 8, 16, 3, c3k2_fused_kernel<128, 4, 8, 2>},
};
const Class* find_class(int hid, int nb) {
  for (const Class& c : kClasses)
    if (c.hid == hid && c.nb == nb) return &c;
  return nullptr;
}
constexpr int kMaxLds = 160 * 1024;
int align_up(int v, int a) { return (v + a - 1) / a * a; }

}  // namespace

hipError_t c3k2_init() {
  for (const Class& c : kClasses) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

bool c3k2_supported(int hid, int nb, int cin) {
  C3k2Params p;
  memset(&p, 0, sizeof p);
  p.hid = hid; p.nb = nb; p.Cin = cin; p.H = p.W = 64;
  return cin % 64 == 0 && c3k2_layout(&p);
}

// Fills tile geometry, stage counts and the LDS layout of `p` (needs hid, nb, Cin, H, W). False = no such class / no fit.
bool c3k2_layout(C3k2Params* p) {
  const Class* c = find_class(p->hid, p->nb);
  if (!c || p->Cin % 32) return false;
  const int h = p->hid, nb = p->nb;
  const int p0 = (c->th + 2 * nb) * (c->tw + 2 * nb), p1 = (c->th + 2) * (c->tw + 2), pt = c->th * c->tw;
  p->tiles_x = (p->W + c->tw - 1) / c->tw;
  p->tiles_y = (p->H + c->th - 1) / c->th;
  p->tiles_x_magic = div_magic((unsigned)p->tiles_x);
  p->nchx_magic = div_magic((unsigned)(p->Cin / 8));
  p->n_bias = 2 * h * (2 + nb);
  p->total_stages = stages_of(p->Cin / 32, 2 * h, c->sb) + nb * (stages_of(h / 32, h, c->sb) + stages_of(9 * h / 32, h, c->sb)) +
                    stages_of(2 * h / 32, 2 * h, c->sb);
  const int x_bytes = align_up(p0 * p->Cin * 2, 1024) + 1024;  // the last patch DMA instruction may overrun by < 1 KiB
  const int t_bytes = p0 * h * 2, u1_bytes = nb == 2 ? p1 * h * 2 : 0, u2_bytes = pt * h * 2;
  const int stage_bytes = pt * (2 * h * 2 + 16);
  // region A: the input patch; once S1 has consumed it, t (and later the output staging tile), u1 and u2 live there
  const int head = t_bytes > stage_bytes ? t_bytes : stage_bytes;
  const int a_need = align_up(head, 16) + align_up(u1_bytes, 16) + align_up(u2_bytes, 16);
  const int a_bytes = x_bytes > a_need ? x_bytes : a_need;
  int off = 0;
  p->off_bias = off; off += align_up(p->n_bias * 4, 1024);
  p->off_ring = off; off += c->ring * c->sb * 1024;
  p->off_x = off;
  p->off_t = off;
  p->off_stage = off;
  p->off_u1 = off + align_up(head, 16);
  p->off_u2 = p->off_u1 + align_up(u1_bytes, 16);
  off += align_up(a_bytes, 1024);
  p->off_y = off; off += align_up(p0 * 2 * h * 2, 1024);
  p->smem_bytes = off;
  return off <= kMaxLds;
}

// Packs the weights of the block's convs (execution order: cv1|cv2, {b.cv1, b.cv2} x nb, cv3; each given as the
// exporter's [n/16][K/32] 1-KiB fragment blocks) into the stage stream the kernel's ring consumes, and concatenates
// the biases. Stage of a step with N output channels: SB blocks = (SB / (N/16)) k-blocks x (N/16) channel subtiles,
// k-block-major; the tail of a step's last stage is zero blocks.
bool c3k2_pack(int hid, int nb, int cin, const C3k2Conv* convs, std::vector<unsigned char>* stream, std::vector<float>* bias) {
  const Class* c = find_class(hid, nb);
  if (!c) return false;
  stream->clear();
  bias->clear();
  const int nconv = 2 + 2 * nb;
  for (int ci = 0; ci < nconv; ++ci) {
    const C3k2Conv& cv = convs[ci];
    const int n = cv.n[0] + cv.n[1];
    const int want_n = (ci == 0 || ci == nconv - 1) ? 2 * hid : hid;
    const int want_k = ci == 0 ? cin : (ci == nconv - 1 ? 2 * hid : ((ci & 1) ? hid : 9 * hid));
    if (n != want_n || cv.K != want_k || cv.n[0] % 16 || cv.n[1] % 16) return false;
    const int ns = n / 16, kbn = cv.K / 32, kps = c->sb / ns;
    const int nst = (kbn + kps - 1) / kps;
    const size_t base = stream->size();
    stream->resize(base + (size_t)nst * c->sb * 1024, 0);
    for (int kb = 0; kb < kbn; ++kb)
      for (int s = 0; s < ns; ++s) {
        const int seg = s * 16 < cv.n[0] ? 0 : 1;
        const int ls = seg ? s - cv.n[0] / 16 : s;
        const unsigned char* src = cv.w[seg] + ((size_t)ls * kbn + kb) * 1024;
        unsigned char* dst = stream->data() + base + ((size_t)(kb / kps) * c->sb + (size_t)(kb % kps) * ns + s) * 1024;
        memcpy(dst, src, 1024);
      }
    for (int seg = 0; seg < 2; ++seg)
      for (int i = 0; i < cv.n[seg]; ++i) bias->push_back(cv.bias[seg][i]);
  }
  return true;
}

hipError_t c3k2_launch(const C3k2Params& p, hipStream_t stream) {
  const Class* c = find_class(p.hid, p.nb);
  if (!c) return hipErrorInvalidValue;
  hipLaunchKernelGGL(c->fn, dim3(p.tiles_x * p.tiles_y, 1, 1), dim3(256, 1, 1), p.smem_bytes, stream, p);
  return hipGetLastError();
}

const char* c3k2_kernel_name(int hid, int nb) {
  static const char* names[] = {"c3k2_fused<32,8x16,1>", "c3k2_fused<64,8x8,1>", "c3k2_fused<64,8x8,2>",
                                "c3k2_fused<128,4x8,1>", "c3k2_fused<128,4x8,2>"};
  const Class* c = find_class(hid, nb);
  return c ? names[c - kClasses] : "c3k2_fused<?>";
}

}  // namespace unina
