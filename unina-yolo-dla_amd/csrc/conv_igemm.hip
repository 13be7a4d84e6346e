// conv_igemm.hip -- implicit-GEMM convolution on CDNA4 matrix cores (gfx950), LDS-DMA pipelined.
//
// Computes, for every output pixel m and output channel n of a slice,
//     y[m][n] = act( bias[n] + sum_{kh,kw,c} W[n][kh][kw][c] * x[pix(m,kh,kw)][c] )  (+ residual[m][n])
// which is ConvBlock = Conv2d(bias=False, pad=k//2) -> BatchNorm(eval, folded) -> ReLU of the reference
// (unina_yolo_dla/model.py:23-50), the Bottleneck shortcut add (model.py:73), the head output convs
// (model.py:292,299: bias, no activation, fp32 out) and the folded nearest-x2 Upsample (model.py:145-147).
//
// MFMA mapping (v_mfma_f32_16x16x32_f16): WEIGHTS are the A operand (rows = output channels), ACTIVATIONS the
// B operand (columns = pixels): D = W * X^T. Both are K-contiguous in memory, so a lane's fragment is 16 bytes,
// and the accumulator holds 4 consecutive output CHANNELS of one pixel per lane -> 8-byte NHWC stores.
//   A frag: lane l = W[n0 + (l&15)][k0 + 8*(l>>4) .. +8)      B frag: lane l = X[m0 + (l&15)][k0 + 8*(l>>4) .. +8)
//   D     : lane l, reg r  ->  channel n0 + 4*(l>>4) + r, pixel m0 + (l&15)
//
// Data movement: the K loop runs over (tap, 32*KSUB input channels). Every K-step's operands are staged in LDS as
// 1-KiB "fragment blocks" (16 rows x 32 k, fp16) filled by ONE global_load_lds_dwordx4 wave-instruction each
// (LDS-DMA: no VGPR round trip). Inside a block the 16-byte slot of (row r, k-chunk c) is
//     slot(r,c) = 4*r + (c ^ G[r>>2]),  G = {0,2,3,1}
// so that (a) the loading lane s = 4*r + c' fetches chunk c'^G[r>>2] of row r: the 4 lanes of a row still read one
// contiguous 64-byte segment (quad-coalesced), and (b) each of ds_read_b128's four 16-lane groups touches 16
// distinct 16-byte bank slots: conflict-free fragment reads. Weights are stored by the exporter already in this
// block image (export.py: pack_weights), so their loads are fully linear. Out-of-image taps and tile tails read a
// zero page instead of branching.
// Pipeline: STAGES LDS buffers, STAGES-1 K-steps in flight, one s_barrier per K-step, counted s_waitcnt vmcnt.
#include "kernels.h"
#include "block_pipeline.h"

#include <cstdlib>

namespace unina {

using namespace dev;

namespace {

typedef float floatx4_t __attribute__((ext_vector_type(4)));

// Workgroup -> (M-tile bx, N-tile by). The grid is launched 1-D and re-mapped so that the 8 XCDs (which receive
// consecutive workgroup ids round-robin, each with its own non-coherent 4 MiB L2) own CONTIGUOUS ranges of the
// by-major tile order: every XCD then streams only its own N-tiles' weights through its L2 instead of all of them
// (measured before the remap: 23 MB of fabric reads per P4 head conv for 4.8 MB of algorithmic bytes = the 2.4 MB
// weight set fetched once per XCD). Bijective form of the CDNA guide's T1 remap; placement affects speed only.
__device__ __forceinline__ void tile_of_block(const ConvParams& p, int orig, int nwg, int* bx, int* by) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  // xcd_m_major: an XCD owns a contiguous range of PIXEL tiles (all channel tiles): its L2 holds its share of the input;
  // otherwise of CHANNEL tiles (all pixel tiles): its L2 holds its share of the weights.
  // (Selects, not two branches storing through the pointers in swapped order: the compiler turned those into a
  // dynamically indexed private array -- a scratch store + load in every conv kernel's prologue, and tile indices in VGPRs.)
  const bool mm = p.xcd_m_major != 0;
  const int major = fast_div(v, mm ? p.gn_magic : p.gm_magic);
  const int minor = v - major * (mm ? p.grid_n : p.grid_m);
  *bx = mm ? major : minor;
  *by = mm ? minor : major;
}

__device__ __forceinline__ void stamp_entry(const ConvParams& p, long long t) {
  if (p.stamps && blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0) p.stamps[7] = t;
}

__device__ __forceinline__ void stamp(const ConvParams& p, int k) {
  if (p.stamps && blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0) {
    p.stamps[k] = __builtin_amdgcn_s_memtime();
    if (k == 0 || k == 4) p.stamps[5 + (k >> 2)] = wall_clock64();  // 100 MHz reference: effective shader clock
  }
}

// Same for a body that runs as workgroup `bid` of `nwg` inside a larger grid (dual launches): the mid workgroup of ITS conv.
__device__ __forceinline__ void stamp_b(const ConvParams& p, int k, int bid, int nwg) {
  if (p.stamps && bid == (nwg >> 1) && threadIdx.x == 0) {
    p.stamps[k] = __builtin_amdgcn_s_memtime();
    if (k == 0 || k == 4) p.stamps[5 + (k >> 2)] = wall_clock64();
  }
}

// debug: every workgroup's start / end on the 100 MHz wall clock (dispatch ramp and tail of a launch)
__device__ __forceinline__ void stamp_wg(const ConvParams& p, int which) {
  if (p.wg_times && threadIdx.x == 0) p.wg_times[2 * blockIdx.x + which] = wall_clock64();
}

// Element-type traits. A 1-KiB fragment block is always 16 rows x 4 chunks of 16 bytes:
//   fp16: chunk = 8 k  -> block = 32 k, one v_mfma_f32_16x16x32_f16 per (A block, B block)
//   fp32: chunk = 4 k  -> block = 16 k, four v_mfma_f32_16x16x4_f32 (exact fp32 products and accumulation);
//         MFMA e takes element e of every lane's chunk, i.e. k = {4*lq + e}: A and B use the same k permutation.
//   int8: chunk = 16 k -> block = 64 k, one v_mfma_i32_16x16x64_i8 (exact int32 accumulation)
typedef int intx4 __attribute__((ext_vector_type(4)));
template <typename T> struct Elem;
template <> struct Elem<half_t> {
  static constexpr int kChunk = 8, kBlockK = 32, kPlanes = 1;
  typedef half8 frag;
  typedef floatx4 acc_t;
  static __device__ __forceinline__ acc_t mma(const frag& a, const frag& b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ floatx4 to_float(const acc_t& c) { return c; }
  static __device__ __forceinline__ frag lds(const unsigned char* at, int) { return *reinterpret_cast<const frag*>(at); }
};
// split fp16 (kS16): a fragment is the pair (hi, lo) of 16-byte fragments; its LDS / block image keeps the lo plane `lo_off`
// bytes behind the hi plane (the next 1-KiB block of a staged block pair). Three MFMAs per k block, small terms first.
template <> struct Elem<s16_t> {
  static constexpr int kChunk = 8, kBlockK = 32, kPlanes = 2;
  typedef half8x2 frag;
  typedef floatx4 acc_t;
  static __device__ __forceinline__ acc_t mma(const frag& a, const frag& b, acc_t c) { return mfma_split(a, b, c); }
  static __device__ __forceinline__ floatx4 to_float(const acc_t& c) { return c; }
  static __device__ __forceinline__ frag lds(const unsigned char* at, int lo_off) {
    return frag{*reinterpret_cast<const half8*>(at), *reinterpret_cast<const half8*>(at + lo_off)};
  }
};
template <> struct Elem<float> {
  static constexpr int kChunk = 4, kBlockK = 16, kPlanes = 1;
  static __device__ __forceinline__ floatx4_t lds(const unsigned char* at, int) { return *reinterpret_cast<const floatx4_t*>(at); }
  typedef floatx4_t frag;
  typedef floatx4 acc_t;
  static __device__ __forceinline__ acc_t mma(const frag& a, const frag& b, acc_t c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
    return c;
  }
  static __device__ __forceinline__ floatx4 to_float(const acc_t& c) { return c; }
};
template <> struct Elem<signed char> {
  static constexpr int kChunk = 16, kBlockK = 64, kPlanes = 1;
  static __device__ __forceinline__ intx4 lds(const unsigned char* at, int) { return *reinterpret_cast<const intx4*>(at); }
  typedef intx4 frag;
  typedef intx4 acc_t;
  static __device__ __forceinline__ acc_t mma(const frag& a, const frag& b, acc_t c) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ floatx4 to_float(const acc_t& c) {
    return floatx4{(float)c[0], (float)c[1], (float)c[2], (float)c[3]};
  }
};


// ---- epilogue ------------------------------------------------------------------------------------------------
// y = act( acc * mult[n] + bias[n] ) (+ residual, after the ReLU: model.py:72-73), all in fp32 registers:
//   fp16 / fp32 engines : mult == nullptr (BatchNorm is folded into the weights)
//   int8 convs          : acc is the exact int32 dot product, mult[n] = s_in * s_w * gamma/sqrt(var+eps)
// The output type is a property of the DESTINATION buffer (per slice, run time): fp16, fp32 or int8 with
// q = clamp(rne(y / s_out), -127, 127). Then:
//  * NHWC outputs: the workgroup's BM x BN tile is staged through LDS ([pixel][BN] rows, 16-byte row padding against
//    bank conflicts) and written back as FULL contiguous rows, 16 bytes per lane -- an accumulator fragment only
//    holds 4 channels of a pixel per lane. The folded nearest-x2 Upsample (model.py:145-147) writes each row to its
//    2x2 block.
//  * planar fp32 head outputs: lane = pixel already gives 64-byte contiguous runs per channel: stored directly.
// pix_to_m(pl) maps a workgroup-local pixel to the row-major output pixel index, or -1 outside the image.
// Per-lane bias / multiplier registers, loaded BEFORE the K loop so their global-load latency is hidden behind it.
template <int WN_T>
struct EpiConsts {
  floatx4 bias[WN_T], mult[WN_T];
};
template <int WN_T>
__device__ __forceinline__ void load_epi_consts(const ConvSeg& sg, int n_first, int lq, EpiConsts<WN_T>& c) {
#pragma unroll
  for (int j = 0; j < WN_T; ++j) {
    const int n = n_first + j * 16 + lq * 4;
    const bool ok = n < sg.n_count;  // bias/mult arrays are n_pad long (multiple of 16): n + 3 stays inside when ok
    c.bias[j] = ok ? *reinterpret_cast<const floatx4*>(sg.bias + n) : floatx4{0.f, 0.f, 0.f, 0.f};
    c.mult[j] = (ok && sg.mult) ? *reinterpret_cast<const floatx4*>(sg.mult + n) : floatx4{1.f, 1.f, 1.f, 1.f};
  }
}

#ifndef UNINA_GLDS_DIRECT_STORE
#define UNINA_GLDS_DIRECT_STORE 1   // conv_glds too: same-box A/B +1-2 % frames/s (8 400-8 570 vs 8 300-8 400), latency unchanged
#endif
#ifndef UNINA_REGQ_DIRECT_STORE
#define UNINA_REGQ_DIRECT_STORE 1   // same-box A/B: 18.3 vs 19.1 us per head pair, -5 us serial latency, +0.5-1.5 % frames/s
#endif
// DIRECT (register-queue 3x3 kernels, fp16 destinations without the x2 upsample): every lane stores its 4 channels (8 bytes)
// straight from the accumulators -- 32-byte runs per pixel and wave instead of full rows, but no LDS staging and NO
// workgroup barrier: a wave that has finished its K loop stores and retires without waiting for the others.
template <typename T, int BM, int BN, int WM_T, int WN_T, typename PixToM, bool DIRECT = false>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, const ConvSeg& sg,
                                              typename Elem<T>::acc_t (&acc)[WN_T][WM_T], const EpiConsts<WN_T>& ec,
                                              int wm, int wn, int nb0, int l15, int lq, PixToM pix_to_m,
                                              unsigned char* stage, int nthreads = 256) {
  typedef Elem<T> E;
  const int od = sg.out_dtype;
  const int esz = od == kF32 ? 4 : (od == kF16 ? 2 : 1);
  const int rowb = BN * esz + 16;              // padded LDS row (bytes)
  const int n_w0 = wn * (WN_T * 16);           // tile-local first channel of this wave
  if constexpr (E::kPlanes == 2) {
    // split fp16: 8 bytes of the hi plane and 8 of the lo plane per lane straight from the accumulators (no staging pass);
    // the folded x2 upsample writes the pair to its 2x2 block
#pragma unroll
    for (int j = 0; j < WN_T; ++j) {
      const int n = nb0 + n_w0 + j * 16 + lq * 4;
      if (n >= sg.n_count) continue;
      const floatx4 bias = ec.bias[j];
#pragma unroll
      for (int i = 0; i < WM_T; ++i) {
        const int pl = (wm * WM_T + i) * 16 + l15;
        const int m = pix_to_m(pl);
        if (m < 0) continue;
        floatx4 v = acc[j][i] + bias;
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
        }
        if (p.res) {
          const half_t* rp = static_cast<const half_t*>(p.res) + (size_t)m * p.res_ld + n;
          const half4 rh = *reinterpret_cast<const half4*>(rp);
          const half4 rl = *reinterpret_cast<const half4*>(reinterpret_cast<const unsigned char*>(rp) + p.res_lo);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rh[r] + (float)rl[r];
        }
        if (sg.dst_planar) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < sg.n_count) sg.dst_planar[(size_t)(n + r) * p.M + m] = v[r];
          continue;
        }
        half4 hv, lv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          hv[r] = (half_t)v[r];
          lv[r] = (half_t)(v[r] - (float)hv[r]);
        }
        unsigned char* dst = static_cast<unsigned char*>(sg.dst);
        if (sg.up2) {
          const int oy = m / p.Wo, ox = m - oy * p.Wo;
          unsigned char* d = dst + (((size_t)(2 * oy) * (2 * p.Wo) + 2 * ox) * sg.dst_ld + n) * 2;
          const size_t px = (size_t)sg.dst_ld * 2, row = (size_t)(2 * p.Wo) * px;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            unsigned char* dq = d + (q & 1) * px + (q >> 1) * row;
            *reinterpret_cast<half4*>(dq) = hv;
            *reinterpret_cast<half4*>(dq + sg.dst_lo) = lv;
          }
        } else {
          unsigned char* d = dst + ((size_t)m * sg.dst_ld + n) * 2;
          *reinterpret_cast<half4*>(d) = hv;
          *reinterpret_cast<half4*>(d + sg.dst_lo) = lv;
        }
      }
    }
    return;
  }
  const bool direct = DIRECT && od == kF16 && !sg.up2 && sg.dst_planar == nullptr;
  const bool planar = sg.dst_planar != nullptr || direct;
  if (!planar) __syncthreads();                // every wave is done reading the operand buffers: reuse them
#pragma unroll
  for (int j = 0; j < WN_T; ++j) {
    const int nl = n_w0 + j * 16 + lq * 4;     // tile-local channel of this lane's 4 values
    const int n = nb0 + nl;                    // slice-relative
    if (n >= sg.n_count) continue;
    const floatx4 bias = ec.bias[j], mult = ec.mult[j];
#pragma unroll
    for (int i = 0; i < WM_T; ++i) {
      const int pl = (wm * WM_T + i) * 16 + l15;
      const int m = pix_to_m(pl);
      if (m < 0) continue;
      floatx4 v = E::to_float(acc[j][i]);
      if (sg.mult) {  // one rounding: fma(acc, mult, bias) -- the integer emulation in tests/emulate.py does the same
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(v[r], mult[r], bias[r]);
      } else {
        v = v + bias;
      }
      if (p.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      }
      if (p.res) {
        const size_t ro = (size_t)m * p.res_ld + n;
        if (p.res_dtype == kF16) {
          const half4 rv = *reinterpret_cast<const half4*>(static_cast<const half_t*>(p.res) + ro);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
        } else if (p.res_dtype == kF32) {
          v = v + *reinterpret_cast<const floatx4*>(static_cast<const float*>(p.res) + ro);
        } else {
          const int rv = *reinterpret_cast<const int*>(static_cast<const signed char*>(p.res) + ro);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf((float)(signed char)(rv >> (8 * r)), p.res_scale, v[r]);
        }
      }
      if (direct) {
        half4 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = (half_t)v[r];
        *reinterpret_cast<half4*>(static_cast<half_t*>(sg.dst) + (size_t)m * sg.dst_ld + n) = hv;
      } else if (planar) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < sg.n_count) sg.dst_planar[(size_t)(n + r) * p.M + m] = v[r];
      } else if (od == kF16) {
        half4 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = (half_t)v[r];
        *reinterpret_cast<half4*>(stage + pl * rowb + nl * 2) = hv;
      } else if (od == kF32) {
        *reinterpret_cast<floatx4*>(stage + pl * rowb + nl * 4) = v;
      } else {
        unsigned int q = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = __builtin_rintf(v[r] * sg.out_inv_scale);  // round half to even
          t = t > 127.f ? 127.f : (t < -127.f ? -127.f : t);
          q |= ((unsigned int)(int)t & 0xFFu) << (8 * r);
        }
        *reinterpret_cast<unsigned int*>(stage + pl * rowb + nl) = q;
      }
    }
  }
  if (planar) return;
  __syncthreads();
  typedef float vec16 __attribute__((ext_vector_type(4)));  // 16 opaque bytes
  unsigned char* dst = static_cast<unsigned char*>(sg.dst);
  const int cpr = BN * esz / 16, epc = 16 / esz;  // 16-byte chunks per pixel row, elements per chunk
  for (int c = threadIdx.x; c < BM * cpr; c += nthreads) {
    const int pl = c / cpr, ch = c - pl * cpr;
    const int n = nb0 + ch * epc;
    const int m = pix_to_m(pl);
    if (m < 0 || n >= sg.n_count) continue;
    const vec16 v = *reinterpret_cast<const vec16*>(stage + pl * rowb + ch * 16);
    if (sg.up2) {
      const int oy = m / p.Wo, ox = m - oy * p.Wo;
      unsigned char* d = dst + (((size_t)(2 * oy) * (2 * p.Wo) + 2 * ox) * sg.dst_ld + n) * esz;
      const size_t px = (size_t)sg.dst_ld * esz, row = (size_t)(2 * p.Wo) * px;
      *reinterpret_cast<vec16*>(d) = v;
      *reinterpret_cast<vec16*>(d + px) = v;
      *reinterpret_cast<vec16*>(d + row) = v;
      *reinterpret_cast<vec16*>(d + row + px) = v;
    } else {
      *reinterpret_cast<vec16*>(dst + ((size_t)m * sg.dst_ld + n) * esz) = v;
    }
  }
}

}  // namespace

extern __shared__ __align__(16) unsigned char conv_smem[];

// BK is counted in fragment blocks' worth of k: KSUB = BK/32 blocks per row-subtile per K-step, i.e. a K-step covers
// KSUB*32 input channels in fp16 and KSUB*16 in fp32.
// conv_glds_body: workgroup `bid` of `nwg` (a plain launch passes blockIdx.x / gridDim.x; conv_dual runs two convs'
// workgroups side by side in one grid).
template <typename T, int BM, int BN, int BK, int WAVES_M, int WAVES_N, int STAGES>
__device__ __forceinline__ void conv_glds_body(const ConvParams& p, int bid, int nwg) {
  static_assert(WAVES_M * WAVES_N == 4, "256-thread blocks");
  typedef Elem<T> E;
  typedef typename E::frag frag_t;
  constexpr int WM_T = BM / (WAVES_M * 16), WN_T = BN / (WAVES_N * 16), KSUB = BK / 32;
  constexpr int KSTEP = KSUB * E::kBlockK;  // input channels per K-step
  constexpr int NP = E::kPlanes;            // split fp16: every block comes as a (hi, lo) pair of adjacent 1-KiB blocks
  constexpr int ABLK = (BM / 16) * KSUB * NP, WBLK = (BN / 16) * KSUB * NP, NBLK = ABLK + WBLK;
  constexpr int LPT = (NBLK + 3) / 4;            // LDS-DMA instructions per wave per stage (same for every wave)
  constexpr int STAGE_BYTES = LPT * 4 * 1024;
  static_assert(WM_T >= 1 && WN_T >= 1 && KSUB >= 1, "tile");

  const long long t_entry = p.stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wid % WAVES_M, wn = wid / WAVES_M;
  const int l15 = lane & 15, lq = lane >> 4;

  int bx, by;
  tile_of_block(p, bid, nwg, &bx, &by);
  const int sidx = (p.nseg > 1 && by >= p.seg[1].tile0) ? 1 : 0;
  const ConvSeg& sg = p.seg[sidx];
  const int n_pad = (sg.n_count + 15) & ~15;
  const int nb0 = (by - sg.tile0) * BN;  // first channel of this block's tile (slice-relative)
  const int m_blk = bx * BM;
  const int kblocks = p.ksize * p.ksize * (p.Cin / E::kBlockK);  // k blocks per weight row

  // ---- per-thread description of the LPT blocks this wave loads every stage (block b = q*4 + wid) ----
  // activation block (i,j): rows = pixels m_blk + 16*i .. +16, k = 32*j .. +32 of the K-step
  // weight block     (n,j): rows = channels nb0 + 16*n .. +16
  const int ld_row = lane >> 2;                                  // row (pixel / channel) this lane fetches
  const int ld_chunk = (lane & 3) ^ swz_g(ld_row);              // 8-element k-chunk this lane fetches
  int a_iy0[LPT], a_ix0[LPT];                                    // input coords of tap (0,0), or big negative if row invalid
  int a_off0[LPT];                                               // element offset of tap (0,0) from a_base (may be negative)
  const T* a_base[LPT];                                          // src + channel offset of this lane's chunk
  const T* w_base[LPT];                                          // weight block address for k block 0 (nullptr = zero rows)
  int kind[LPT];                                                 // 0 activation, 1 weight, 2 padding
#pragma unroll
  for (int q = 0; q < LPT; ++q) {
    const int b = q * 4 + wid;
    a_iy0[q] = a_ix0[q] = -(1 << 20);
    a_off0[q] = 0;
    a_base[q] = nullptr;
    w_base[q] = nullptr;
    if (b < ABLK) {
      kind[q] = 0;
      const int i = b / (KSUB * NP), jp = b - i * (KSUB * NP), j = jp / NP, pl = jp - j * NP;
      const int m = m_blk + i * 16 + ld_row;
      if (m < p.M) {
        const int oy = fast_div(m, p.wo_magic), ox = m - oy * p.Wo;
        a_iy0[q] = oy * p.stride - p.pad;
        a_ix0[q] = ox * p.stride - p.pad;
        a_off0[q] = (a_iy0[q] * p.W + a_ix0[q]) * p.src_ld;
      }
      a_base[q] = static_cast<const T*>(p.src) + sg.src_coff + j * E::kBlockK + ld_chunk * E::kChunk + pl * (p.src_lo / (long long)sizeof(T));
    } else if (b < NBLK) {
      kind[q] = 1;
      const int bb = b - ABLK;
      const int n = bb / (KSUB * NP), jp = bb - n * (KSUB * NP), j = jp / NP, pl = jp - j * NP;
      const int nsub = (nb0 >> 4) + n;
      if (nsub * 16 < n_pad) w_base[q] = static_cast<const T*>(sg.w) + (((size_t)nsub * kblocks + j) * NP + pl) * (1024 / sizeof(T)) + lane * E::kChunk;
    } else {
      kind[q] = 2;
    }
  }

  // K-step iterator for the NEXT stage to issue (all wave-uniform): tap (kh,kw), first channel c0, weight k block
  int i_kh = 0, i_kw = 0, i_c0 = 0, i_k32 = 0;
  const int nk = p.ksize * p.ksize * (p.Cin / KSTEP);
  const T* zeros = reinterpret_cast<const T*>(p.zeros);

  auto issue = [&](int buf) {
    unsigned char* sb = conv_smem + buf * STAGE_BYTES;
    const int tap_off = (i_kh * p.W + i_kw) * p.src_ld + i_c0;   // scalar: the same for every pixel of the tile
#pragma unroll
    for (int q = 0; q < LPT; ++q) {
      const T* g = zeros;
      if (kind[q] == 0) {
        const bool ok = (unsigned)(a_iy0[q] + i_kh) < (unsigned)p.H && (unsigned)(a_ix0[q] + i_kw) < (unsigned)p.W;
        if (ok) g = a_base[q] + (a_off0[q] + tap_off);
      } else if (kind[q] == 1) {
        if (w_base[q]) g = w_base[q] + (size_t)i_k32 * NP * (1024 / sizeof(T));
      }
      glds16(g, sb + (q * 4 + wid) * 1024);
    }
    i_c0 += KSTEP;
    i_k32 += KSUB;
    if (i_c0 >= p.Cin) {
      i_c0 = 0;
      if (++i_kw == p.ksize) {
        i_kw = 0;
        ++i_kh;
      }
    }
  };

  typename E::acc_t acc[WN_T][WM_T];
#pragma unroll
  for (int j = 0; j < WN_T; ++j)
#pragma unroll
    for (int i = 0; i < WM_T; ++i) acc[j][i] = typename E::acc_t{0, 0, 0, 0};

  // ---- software pipeline ----------------------------------------------------------------------------------
  // LDS-DMA ring: STAGES buffers, stage kt lives in buffer kt % STAGES, up to STAGES-1 steps in flight; every wave
  // issues exactly LPT DMA instructions per stage, so "stage X has landed" is a counted s_waitcnt vmcnt.
  // Register pipeline: the fragments of step kt+1 are read from LDS into the OTHER register set while the MFMAs of
  // step kt issue, so LDS latency overlaps matrix work (the compiler alone serialises read -> lgkmcnt(0) -> MFMA).
  // Step kt: wait until stage kt+1 has landed, barrier, refill buffer (kt-1) % STAGES -- its fragments were consumed
  // by MFMAs every wave finished before this barrier --, read stage kt+1, run the MFMAs of stage kt.
  // The steady-state loop has no conditionals (a branch around the reads makes hipcc fall back to lgkmcnt(0) before
  // the MFMAs); head and tail, where fewer stages are in flight, use run-time counts.
  static_assert(STAGES >= 3, "ring depth");
  struct Frags {
    frag_t a[KSUB][WN_T], b[KSUB][WM_T];
  };
  const int rd_off = (4 * l15 + (lq ^ swz_g(l15))) * 16;  // this lane's fragment slot inside any block
  auto read_frags = [&](Frags& f, int kt) {
    const unsigned char* sb = conv_smem + (kt % STAGES) * STAGE_BYTES + rd_off;
#pragma unroll
    for (int j = 0; j < KSUB; ++j) {
#pragma unroll
      for (int i = 0; i < WM_T; ++i) f.b[j][i] = E::lds(sb + ((((wm * WM_T + i) * KSUB + j) * NP) << 10), 1024);
#pragma unroll
      for (int n = 0; n < WN_T; ++n) f.a[j][n] = E::lds(sb + ((ABLK + ((wn * WN_T + n) * KSUB + j) * NP) << 10), 1024);
    }
  };
  auto mma_frags = [&](const Frags& f) {
#pragma unroll
    for (int j = 0; j < KSUB; ++j)
#pragma unroll
      for (int n = 0; n < WN_T; ++n)
#pragma unroll
        for (int i = 0; i < WM_T; ++i) acc[n][i] = E::mma(f.a[j][n], f.b[j][i], acc[n][i]);
  };
  stamp_entry(p, t_entry);
  stamp(p, 0);
  const int n_pro = nk < STAGES - 1 ? nk : STAGES - 1;
  for (int s = 0; s < n_pro; ++s) issue(s);
  stamp(p, 1);
  EpiConsts<WN_T> ec;
  load_epi_consts<WN_T>(sg, nb0 + wn * (WN_T * 16), lq, ec);
  Frags fa, fb;
  wait_stages<LPT>(n_pro - 1);  // stage 0 landed (the epilogue-constant loads above only make this stricter)
  __builtin_amdgcn_s_barrier();
  read_frags(fa, 0);
  stamp(p, 2);
  int kt = 0;
  for (; kt + STAGES < nk; kt += 2) {  // steady state: two steps, every stage they touch exists
    wait_vmcnt<(STAGES - 3) * LPT>();
    __builtin_amdgcn_s_barrier();
    issue((kt + STAGES - 1) % STAGES);
    read_frags(fb, kt + 1);
    mma_frags(fa);
    wait_vmcnt<(STAGES - 3) * LPT>();
    __builtin_amdgcn_s_barrier();
    issue((kt + STAGES) % STAGES);
    read_frags(fa, kt + 2);
    mma_frags(fb);
  }
  bool cur_a = true;
  for (; kt < nk; ++kt) {  // tail: run-time counts, nothing is issued past the last stage
    if (kt + 1 < nk) {
      const int issued = kt + STAGES - 1 < nk ? kt + STAGES - 1 : nk;
      wait_stages<LPT>(issued - (kt + 2));
      __builtin_amdgcn_s_barrier();
      if (kt + STAGES - 1 < nk) issue((kt + STAGES - 1) % STAGES);
      if (cur_a) {
        read_frags(fb, kt + 1);
        mma_frags(fa);
      } else {
        read_frags(fa, kt + 1);
        mma_frags(fb);
      }
    } else if (cur_a) {
      mma_frags(fa);
    } else {
      mma_frags(fb);
    }
    cur_a = !cur_a;
  }
  wait_vmcnt<0>();  // drain the dummy tail before the wave retires

  // ---- epilogue ----
  auto pix_to_m = [&](int pl) { const int m = m_blk + pl; return m < p.M ? m : -1; };
  conv_epilogue<T, BM, BN, WM_T, WN_T, decltype(pix_to_m), (UNINA_GLDS_DIRECT_STORE != 0 && sizeof(T) == 2)>(p, sg, acc, ec, wm, wn, nb0, l15, lq, pix_to_m, conv_smem);
}

template <typename T, int BM, int BN, int BK, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(256) void conv_glds(const ConvParams p) {
  conv_glds_body<T, BM, BN, BK, WAVES_M, WAVES_N, STAGES>(p, (int)blockIdx.x, (int)gridDim.x);
}


// ================================================================================================ 3x3 halo kernel
// 3x3 / stride 1 / pad 1 convolution with the INPUT PATCH RESIDENT IN LDS. A workgroup owns a TH x TW tile of output
// pixels and BN output channels. Its (TH+2) x (TW+2) x Cin input patch is DMA'd into LDS once; all nine taps then read
// their B fragments from that patch at shifted pixel addresses, and only the WEIGHTS stream through the STAGES-deep
// LDS-DMA ring (same 1-KiB packed fragment blocks, same counted-vmcnt pipeline as conv_glds). Compared with the
// im2col K-steps of conv_glds this removes the nine-fold re-fetch of every input pixel: L2->LDS traffic per
// workgroup drops from (BM + BN)*K to (1.27..1.56)*BM*Cin + BN*K elements.
// Patch image: pixel h = hy*(TW+2) + hx owns Cin elements = NCH 16-byte chunks; chunk c sits at slot c ^ (h & SWZ)
// (SWZ = min(NCH,16)-1): a fragment read touches 16 different pixels at a (nearly) common chunk index, and the XOR
// spreads them over the 16 bank slots; the loading lane fetches the matching source chunk, so every pixel's row is
// still one contiguous global segment.
template <typename T, int TH, int TW, int BN, int BK, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(256) void conv3x3_halo(const ConvParams p) {
  static_assert(WAVES_M * WAVES_N == 4, "256-thread blocks");
  typedef Elem<T> E;
  typedef typename E::frag frag_t;
  constexpr int BM = TH * TW, HW_ = TW + 2, NPIX = (TH + 2) * (TW + 2);
  constexpr int WM_T = BM / (WAVES_M * 16), WN_T = BN / (WAVES_N * 16), KSUB = BK / 32;
  constexpr int KSTEP = KSUB * E::kBlockK;
  constexpr int WBLK = (BN / 16) * KSUB;
  constexpr int LPT = (WBLK + 3) / 4;
  constexpr int STAGE_BYTES = LPT * 4 * 1024;
  constexpr int RING_BYTES = STAGES * STAGE_BYTES;
  static_assert(WM_T >= 1 && WN_T >= 1 && (TW == 8 || TW == 16), "tile");

  const long long t_entry = p.stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wid % WAVES_M, wn = wid / WAVES_M;
  const int l15 = lane & 15, lq = lane >> 4;

  int bx, by;
  tile_of_block(p, (int)blockIdx.x, (int)gridDim.x, &bx, &by);
  const int sidx = (p.nseg > 1 && by >= p.seg[1].tile0) ? 1 : 0;
  const ConvSeg& sg = p.seg[sidx];
  const int n_pad = (sg.n_count + 15) & ~15;
  const int nb0 = (by - sg.tile0) * BN;
  const int tiles_x = (p.Wo + TW - 1) / TW;
  const int ty0 = (bx / tiles_x) * TH, tx0 = (bx % tiles_x) * TW;
  const int nch = p.Cin / E::kChunk;              // 16-byte chunks per pixel (power of two)
  const int nch_log = 31 - __builtin_clz(nch);
  const int swz = (nch < 16 ? nch : 16) - 1;
  const int kblocks = 9 * (p.Cin / E::kBlockK);
  unsigned char* patch = conv_smem + RING_BYTES;
  const T* zeros = reinterpret_cast<const T*>(p.zeros);

  // ---- 1. patch DMA: slot s (16 B) of the image = pixel s / nch, slot-in-pixel s % nch ----
  stamp_entry(p, t_entry);
  stamp(p, 0);
  {
    const int nslots = NPIX * nch;
    const T* src = static_cast<const T*>(p.src) + sg.src_coff;
    for (int s0 = wid * 64; s0 < nslots; s0 += 256) {
      const int s = s0 + lane;
      const T* g = zeros;
      if (s < nslots) {
        const int h = s >> nch_log, cs = s & (nch - 1);
        const int hy = h / HW_, hx = h - hy * HW_;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
          g = src + (size_t)(iy * p.W + ix) * p.src_ld + ((cs ^ (h & swz)) * E::kChunk);
      }
      glds16(g, patch + (size_t)s0 * 16);
    }
  }

  // ---- 2. weight ring (identical to conv_glds, activations removed) ----
  const T* w_base[LPT];
#pragma unroll
  for (int q = 0; q < LPT; ++q) {
    const int b = q * 4 + wid;
    w_base[q] = nullptr;
    if (b < WBLK) {
      const int n = b / KSUB, j = b - n * KSUB;
      const int nsub = (nb0 >> 4) + n;
      if (nsub * 16 < n_pad) w_base[q] = static_cast<const T*>(sg.w) + ((size_t)nsub * kblocks + j) * (1024 / sizeof(T)) + lane * E::kChunk;
    }
  }
  const int steps_per_tap = p.Cin / KSTEP;
  const int nk = 9 * steps_per_tap;
  int i_kt = 0;
  auto issue = [&](int buf) {
    unsigned char* sb = conv_smem + buf * STAGE_BYTES;
#pragma unroll
    for (int q = 0; q < LPT; ++q) {
      const T* g = w_base[q] ? w_base[q] + (size_t)i_kt * KSUB * (1024 / sizeof(T)) : zeros;
      glds16(g, sb + (q * 4 + wid) * 1024);
    }
    ++i_kt;
  };
  const int n_pro = nk < STAGES - 1 ? nk : STAGES - 1;
  for (int s = 0; s < n_pro; ++s) issue(s);
  stamp(p, 1);

  // ---- 3. per-lane patch addressing: pixel of every subtile, tap (0,0) ----
  int h00[WM_T];
#pragma unroll
  for (int i = 0; i < WM_T; ++i) {
    const int pix = (wm * WM_T + i) * 16 + l15;
    h00[i] = (pix / TW) * HW_ + (pix % TW);
  }
  typename E::acc_t acc[WN_T][WM_T];
#pragma unroll
  for (int j = 0; j < WN_T; ++j)
#pragma unroll
    for (int i = 0; i < WM_T; ++i) acc[j][i] = typename E::acc_t{0, 0, 0, 0};

  // ---- software pipeline (see conv_glds): weights ring + register double buffer; B fragments come from the patch ----
  static_assert(STAGES >= 3, "ring depth");
  struct Frags {
    frag_t a[KSUB][WN_T], b[KSUB][WM_T];
  };
  const int rd_off = (4 * l15 + (lq ^ swz_g(l15))) * 16;
  auto read_frags = [&](Frags& f, int kt) {
    const int tap = fast_div(kt, p.spt_magic), cs = kt - tap * steps_per_tap;  // wave-uniform
    const int tap_off = (tap / 3) * HW_ + (tap - (tap / 3) * 3);
    const unsigned char* sb = conv_smem + (kt % STAGES) * STAGE_BYTES + rd_off;
    int hrow[WM_T], hsw[WM_T];  // byte offset of the pixel row, swizzle term
#pragma unroll
    for (int i = 0; i < WM_T; ++i) {
      const int h = h00[i] + tap_off;
      hrow[i] = h << (nch_log + 4);
      hsw[i] = h & swz;
    }
#pragma unroll
    for (int j = 0; j < KSUB; ++j) {
      const int ch = (cs * KSUB + j) * 4 + lq;  // this lane's 16-byte chunk index inside the pixel
#pragma unroll
      for (int i = 0; i < WM_T; ++i) f.b[j][i] = *reinterpret_cast<const frag_t*>(patch + hrow[i] + ((ch ^ hsw[i]) << 4));
#pragma unroll
      for (int n = 0; n < WN_T; ++n) f.a[j][n] = *reinterpret_cast<const frag_t*>(sb + (((wn * WN_T + n) * KSUB + j) << 10));
    }
  };
  auto mma_frags = [&](const Frags& f) {
#pragma unroll
    for (int j = 0; j < KSUB; ++j)
#pragma unroll
      for (int n = 0; n < WN_T; ++n)
#pragma unroll
        for (int i = 0; i < WM_T; ++i) acc[n][i] = E::mma(f.a[j][n], f.b[j][i], acc[n][i]);
  };
  EpiConsts<WN_T> ec;
  load_epi_consts<WN_T>(sg, nb0 + wn * (WN_T * 16), lq, ec);
  Frags fa, fb;
  wait_stages<LPT>(n_pro - 1);  // patch + stage 0 landed
  __builtin_amdgcn_s_barrier();
  read_frags(fa, 0);
  stamp(p, 2);
  int kt = 0;
  for (; kt + STAGES < nk; kt += 2) {
    wait_vmcnt<(STAGES - 3) * LPT>();
    __builtin_amdgcn_s_barrier();
    issue((kt + STAGES - 1) % STAGES);
    read_frags(fb, kt + 1);
    mma_frags(fa);
    wait_vmcnt<(STAGES - 3) * LPT>();
    __builtin_amdgcn_s_barrier();
    issue((kt + STAGES) % STAGES);
    read_frags(fa, kt + 2);
    mma_frags(fb);
  }
  bool cur_a = true;
  for (; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      const int issued = kt + STAGES - 1 < nk ? kt + STAGES - 1 : nk;
      wait_stages<LPT>(issued - (kt + 2));
      __builtin_amdgcn_s_barrier();
      if (kt + STAGES - 1 < nk) issue((kt + STAGES - 1) % STAGES);
      if (cur_a) {
        read_frags(fb, kt + 1);
        mma_frags(fa);
      } else {
        read_frags(fa, kt + 1);
        mma_frags(fb);
      }
    } else if (cur_a) {
      mma_frags(fa);
    } else {
      mma_frags(fb);
    }
    cur_a = !cur_a;
  }
  wait_vmcnt<0>();
  stamp(p, 3);

  conv_epilogue<T, BM, BN, WM_T, WN_T>(p, sg, acc, ec, wm, wn, nb0, l15, lq,
                                       [&](int pl) {
                                         const int oy = ty0 + pl / TW, ox = tx0 + pl % TW;
                                         return (oy < p.Ho && ox < p.Wo) ? oy * p.Wo + ox : -1;
                                       },
                                       conv_smem);
  if (p.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(p, 4);
  }
}

// ============================================================================================ 3x3 register-queue kernel
// 3x3 / stride 1 / pad 1 with the input patch resident in LDS (as conv3x3_halo) and the WEIGHTS STREAMED L2 -> REGISTERS:
// a wave owns one 16-channel subtile of the workgroup's BN channels, so its weight blocks are one contiguous run of the
// exporter's [n/16][K/32] fragment-block array, prefetched D blocks ahead through a circular register queue
// (block_pipeline.h). Compared with the LDS-DMA ring this puts D KiB x waves of weights in flight per CU instead of
// (STAGES-1) K-steps (24 KB), needs no barrier inside the K loop, and leaves LDS bandwidth to the activation fragments:
// the weight-heavy head layers at 80^2 / 40^2 (590 KB / 2.4 MB of weights per launch) are bound by exactly that.
// fp16 only; Cin is a template parameter (the K loop is unrolled at compile time so queue slots are registers).
// Same MFMA, same K order (tap-major, 32 channels per block), same epilogue as the other kernels: bit-identical.
// WN = 16-channel subtiles per wave (each activation fragment then feeds WN MFMAs: halves the LDS reads per MFMA at 2).
// T = half_t, or signed char (INT8 engines: int8 patch image, 64-k weight blocks, v_mfma_i32_16x16x64_i8).
#ifndef UNINA_PATCH_VIA_REGS
#define UNINA_PATCH_VIA_REGS 1
#endif
constexpr bool kPatchViaRegs = UNINA_PATCH_VIA_REGS != 0;   // input patch: 16-byte loads to registers + ds_write (1) or LDS-DMA (0)
// STAMPS (debug instantiations only: a branch around the loads would change the schedule of the product kernels): phase
// stamps of the conv's mid workgroup -- 0 start, 1 patch DMA + first weight blocks issued, 2 patch landed (barrier passed),
// 3 K loop done, 4 stores issued.
template <int TH, int TW, int BN, int CIN, int NW, int D, int S = 1, int WN = 1, typename T = half_t, bool STAMPS = false>
__device__ __forceinline__ void conv3x3_regq_body(const ConvParams& p, int bid, int nwg) {
  typedef Elem<T> E;
  typedef typename E::frag frag;
  constexpr bool SPLIT = E::kPlanes == 2;
  typedef typename std::conditional<sizeof(T) == 1, EltI8, typename std::conditional<SPLIT, EltS, EltH>::type>::type PE;   // block_pipeline.h element traits of T
  // S = stride (1 or 2): the patch is the (S*TH + 2 or S*TH + 1) x (...) input footprint of the tile
  constexpr int BM = TH * TW, R0W = S * (TW - 1) + 3, R0H = S * (TH - 1) + 3, NT = NW * 64, CB = CIN / E::kBlockK, KB = 9 * CB;
  static_assert(CIN % E::kBlockK == 0, "a weight block must not straddle a tap");
  constexpr int NS = BN / 16 / WN, WVM = NW / NS, MS = (BM + 15) / 16, WM_T = (MS + WVM - 1) / WVM;   // NS = waves along the channels
  static_assert((BN / 16) % WN == 0 && NW % NS == 0 && WM_T >= 1 && KB * WN >= D, "tile");

  if constexpr (STAMPS) { stamp_b(p, 0, bid, nwg); stamp_wg(p, 0); }
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wid / NS, wn = wid % NS;
  const int l15 = lane & 15, lq = lane >> 4;
  int bx, by;
  tile_of_block(p, bid, nwg, &bx, &by);
  const int sidx = (p.nseg > 1 && by >= p.seg[1].tile0) ? 1 : 0;
  const ConvSeg& sg = p.seg[sidx];
  const int n_pad = (sg.n_count + 15) & ~15;
  const int nb0 = (by - sg.tile0) * BN;
  const int tiles_x = (p.Wo + TW - 1) / TW;
  const int tyi = fast_div(bx, p.tx_magic), txi = bx - tyi * tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TW;

  // this wave's channel subtiles wn*WN .. +WN (tail tiles: clamp, never stored); queue element g = (k-block g / WN, subtile g % WN)
  const unsigned char* wptr[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    int nsub = (nb0 >> 4) + wn * WN + j;
    nsub = nsub * 16 < n_pad ? nsub : (n_pad >> 4) - 1;
    wptr[j] = static_cast<const unsigned char*>(sg.w_lane) + (size_t)nsub * KB * PE::WBLK + lane * 16;   // lane-order twin: one contiguous KiB per load
  }
  frag q[D];
  auto fetch = [&](auto gc) {
    constexpr int g = decltype(gc)::value;
    if constexpr (g < KB * WN) q[g % D] = PE::ldw(wptr[g % WN] + (g / WN) * PE::WBLK);
  };
  // split fp16: the lo image of the patch lies LDS_LO bytes behind the hi image
  constexpr int LDS_LO = SPLIT ? ((R0H * R0W * CIN * 2 + 1023) / 1024) * 1024 + 1024 : 0;

  constexpr Img X = make_img(0, CIN / E::kChunk);
  if constexpr (SPLIT) {   // both planes by LDS-DMA (twice the registers of a register-staged patch would not fit)
    load_patch<R0H, R0W, CIN, NT, PE>(conv_smem, static_cast<const T*>(p.src) + sg.src_coff, p.src_ld, p.H, p.W, S * ty0 - 1,
                                      S * tx0 - 1, p.zeros, wid, lane, p.src_lo, LDS_LO);
    static_for<0, D>(fetch);
  } else if constexpr (kPatchViaRegs) {
    PatchRegs<R0H, R0W, CIN, NT, PE> pr;
    patch_issue<R0H, R0W, CIN, NT, PE>(pr, static_cast<const T*>(p.src) + sg.src_coff, p.src_ld, p.H, p.W, S * ty0 - 1, S * tx0 - 1, wid, lane);
    static_for<0, D>(fetch);
    if constexpr (STAMPS) stamp_b(p, 1, bid, nwg);
    patch_commit<R0H, R0W, CIN, NT, PE>(pr, conv_smem, wid, lane);
  } else {
    load_patch<R0H, R0W, CIN, NT, PE>(conv_smem, static_cast<const T*>(p.src) + sg.src_coff, p.src_ld, p.H, p.W, S * ty0 - 1,
                                      S * tx0 - 1, p.zeros, wid, lane);
    static_for<0, D>(fetch);
  }
  EpiConsts<WN> ec;
  load_epi_consts<WN>(sg, nb0 + wn * (WN * 16), lq, ec);
  if constexpr (STAMPS && (SPLIT || !kPatchViaRegs)) stamp_b(p, 1, bid, nwg);
  if constexpr (SPLIT || !kPatchViaRegs) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the patch has landed
  lds_barrier();
  if constexpr (STAMPS) stamp_b(p, 2, bid, nwg);

  int row0[WM_T];
#pragma unroll
  for (int i = 0; i < WM_T; ++i) {
    int pp = (wm * WM_T + i) * 16 + l15;
    pp = pp < BM ? pp : BM - 1;                     // subtiles past the tile: any valid pixel (never stored)
    row0[i] = S * ((pp / TW) * R0W + pp % TW);
  }
  auto baddr = [&](int i, auto kc) {
    constexpr int kb = decltype(kc)::value, tap = kb / CB, cb = kb - tap * CB, th3 = tap / 3;
    return X.addr(row0[i] + th3 * R0W + (tap - th3 * 3), cb * 4 + lq);
  };
  typename E::acc_t acc[WN][WM_T];
  frag b[2][WM_T];
#pragma unroll
  for (int i = 0; i < WM_T; ++i) {
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[j][i] = typename E::acc_t{0, 0, 0, 0};
    b[0][i] = PE::ld(conv_smem + baddr(i, std::integral_constant<int, 0>{}), LDS_LO);
  }
  static_for<0, KB>([&](auto kc) {
    constexpr int kb = decltype(kc)::value;
    frag a[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) a[j] = q[(kb * WN + j) % D];
    static_for<0, WN>([&](auto jc) { fetch(std::integral_constant<int, kb * WN + decltype(jc)::value + D>{}); });
    if constexpr (kb + 1 < KB) {
#pragma unroll
      for (int i = 0; i < WM_T; ++i) b[(kb + 1) & 1][i] = PE::ld(conv_smem + baddr(i, std::integral_constant<int, kb + 1>{}), LDS_LO);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int i = 0; i < WM_T; ++i) acc[j][i] = E::mma(a[j], b[kb & 1][i], acc[j][i]);
  });

  if constexpr (STAMPS) stamp_b(p, 3, bid, nwg);
  auto pix_to_m = [&](int pl) {
    const int oy = ty0 + pl / TW, ox = tx0 + pl % TW;
    return (pl < BM && oy < p.Ho && ox < p.Wo) ? oy * p.Wo + ox : -1;
  };
  conv_epilogue<T, BM, BN, WM_T, WN, decltype(pix_to_m), (UNINA_REGQ_DIRECT_STORE != 0 && sizeof(T) == 2)>(p, sg, acc, ec, wm, wn, nb0, l15, lq, pix_to_m,
                                                                                                       conv_smem, NT);
  if constexpr (STAMPS) { stamp_b(p, 4, bid, nwg); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp_wg(p, 1); }
}

template <int TH, int TW, int BN, int CIN, int NW, int D, int S = 1, int WN = 1, typename T = half_t>
__global__ __launch_bounds__(NW * 64) void conv3x3_regq(const ConvParams p) {
  conv3x3_regq_body<TH, TW, BN, CIN, NW, D, S, WN, T>(p, (int)blockIdx.x, (int)gridDim.x);
}

// ============================================================================================ 3x3 weights-stationary kernel
// 3x3 / stride 1 / pad 1, ReLU, NHWC destination; fp16, int8 (INT8 engines: int8 patch, 64-k weight blocks, exact int32
// accumulators, the per-op kernels' int8 epilogue) and split fp16 (STRICT engines, kS16: (hi, lo) pairs, 3 MFMAs per product).
// The register-queue kernel above reads one activation fragment from LDS per MFMA and is bound by exactly that
// (profiles/r02/pmc_mfma.json: LDS busy = the whole K loop), and two of its waves fetch every weight block. Here a wave owns
// ONE 16-channel subtile and a TH x 16 pixel tile, keeps its weight blocks in registers and walks DOWN the patch rows: the
// fragment of patch row rho (shifted by kx, channel block cb) feeds the three output rows rho, rho-1, rho-2 (taps ky = 0, 1, 2),
// so LDS is read once per ~3 MFMAs (padded pixel pitch, no xor swizzle: the address of (rho, kx, cb) is lane base + an
// immediate -- the K loop is issue-bound, tools/probes/mix_probe.hip) and every weight block is fetched once per workgroup.
// The input channels run in NCHUNK chunks of CC channels. What the chunks buy: a wave's vector-memory requests cost it ~66
// cycles of ISSUE each when they come back to back (tools/probes/ingest_probe: one 1-KiB load per 66 cycles and wave, whatever
// is in flight; ~20-40 with MFMAs in between, tools/probes/vmem_mfma_probe), and an in-order wave issues no MFMA behind a blocked
// request -- with ALL of a subtile's weights stationary (round 2's form: 288 VGPRs at Cin 256) the P4 conv spent 7 700-8 700
// cycles requesting its 72 + 24 KiB before its first MFMA. Here only chunk 0 (its weights and its slice of the patch: 24
// requests at Cin 256) is requested up front; the requests of chunk c + 1 -- weights into a SECOND register set, patch slice
// into staging registers -- are spread over the steps of chunk c's K loop. The patch slices alternate between two LDS buffers;
// the TH accumulators live across the chunks.
// The sum order is chunk-major: (chunk, ky, kx, cb-in-chunk) instead of the (ky, kx, cb) of every other conv kernel, so this
// kernel is NOT bit-identical to them in fp16 / split fp16 (fp32 accumulation: the difference is the last bit of an fp16 output
// now and then; the tests hold it to that against the register-queue pair, and to the fp32 oracle like everything else). int8
// accumulates exactly: there the codes are the same as every other kernel's.
template <typename T, int TH, int CIN, int NCHUNK, int NW, bool STAMPS = false>
__device__ __forceinline__ void conv3x3_wsc_body(const ConvParams& p, int bid, int nwg) {
  typedef Elem<T> E;
  typedef typename E::frag frag;
  constexpr bool SP = E::kPlanes == 2;
  constexpr bool I8 = sizeof(T) == 1;                       // INT8 engines: int8 patch, 64-k weight blocks, exact int32 accumulators
  constexpr int ESZ = I8 ? 1 : 2, KBLK = E::kBlockK;
  constexpr int WB = 1024 * E::kPlanes;                     // bytes of a weight block (split: the (hi | lo) pair)
  if constexpr (STAMPS) { stamp_b(p, 0, bid, nwg); stamp_wg(p, 0); }
  typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
  typedef float floatx2 __attribute__((ext_vector_type(2)));
  constexpr int TW = 16, R0W = TW + 2, R0H = TH + 2, NT = NW * 64, BN = NW * 16;
  constexpr int CB = CIN / KBLK, CC = CIN / NCHUNK, CBC = CC / KBLK, KBC = 9 * CBC;
  constexpr int STEPS = R0H * 3 * CBC, PF = 4;
  constexpr int PITCH = CC * ESZ + 32;
  constexpr int PLANE = R0H * R0W * PITCH;                   // one plane of a patch chunk; the lo image lies right behind the hi image
  constexpr int BUF = E::kPlanes * PLANE;                    // one LDS buffer; chunks alternate between two
  static_assert(CIN % (KBLK * NCHUNK) == 0 && KBC * 4 * E::kPlanes <= 160 && PLANE % 16 == 0, "chunking: two weight sets must fit the registers");
  static_assert((R0H * R0W + 2) * PITCH < 65536, "ds_read immediate offsets");
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  int bx, by;
  tile_of_block(p, bid, nwg, &bx, &by);
  const int sidx = (p.nseg > 1 && by >= p.seg[1].tile0) ? 1 : 0;
  const ConvSeg& sg = p.seg[sidx];
  const int n_pad = (sg.n_count + 15) & ~15;
  const int nb0 = (by - sg.tile0) * BN;
  const int tiles_x = (p.Wo + TW - 1) / TW;
  const int tyi = fast_div(bx, p.tx_magic), txi = bx - tyi * tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TW;

  int nsub = (nb0 >> 4) + wid;
  nsub = nsub * 16 < n_pad ? nsub : (n_pad >> 4) - 1;   // tail subtile: clamp (never stored)
  const unsigned char* wptr = static_cast<const unsigned char*>(sg.w_lane) + (size_t)nsub * (9 * CB) * WB + lane * 16;   // lane-order twin
  const int n = nb0 + wid * 16 + lq * 4;                // slice-relative first channel of this lane's 4 outputs
  const bool n_ok = n < sg.n_count;
  const floatx4 bias = *reinterpret_cast<const floatx4*>(sg.bias + (n_ok ? n : 0));

  // source planes through buffer descriptors: a slot outside the image (the conv's zero padding) or past the patch gets an
  // out-of-range offset and reads zeros -- no branch around a load
  const int src_bytes = p.H * p.W * p.src_ld * ESZ;
  const __amdgpu_buffer_rsrc_t srs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srs_l =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(static_cast<const unsigned char*>(p.src) + (SP ? p.src_lo : 0)), 0, src_bytes, 0x00020000);
  constexpr int nchx = CC * ESZ / 16, nslots = R0H * R0W * nchx, PITER = (nslots + NT - 1) / NT;

  // destination planes: lanes outside the image / past the slice get an out-of-range offset and their stores are dropped
  const bool out16 = !I8 || sg.out_dtype == kF16;   // (int8 convs may feed an fp16 buffer: the layer in front of the heads' fp16 output convs)
  const int DSZ = out16 ? 2 : 1;
  const unsigned rowb = (unsigned)(p.Wo * sg.dst_ld * DSZ);
  const __amdgpu_buffer_rsrc_t drs_h = __builtin_amdgcn_make_buffer_rsrc(sg.dst, 0, (int)((unsigned)p.Ho * rowb), 0x00020000);
  const __amdgpu_buffer_rsrc_t drs_l =
      __builtin_amdgcn_make_buffer_rsrc(static_cast<unsigned char*>(sg.dst) + (SP ? sg.dst_lo : 0), 0, (int)((unsigned)p.Ho * rowb), 0x00020000);
  const bool lane_ok = n_ok && tx0 + l15 < p.Wo;
  const unsigned voff0 = lane_ok ? (unsigned)(((ty0 * p.Wo + tx0 + l15) * sg.dst_ld + n) * DSZ) : 0x40000000u;
  floatx4 mult = {1.f, 1.f, 1.f, 1.f};
  if constexpr (I8) mult = *reinterpret_cast<const floatx4*>(sg.mult + (n_ok ? n : 0));   // int8: s_in * s_w * bn_scale per channel
  const float out_inv = sg.out_inv_scale;

  typename E::acc_t acc[TH];
  auto store_row = [&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if constexpr (!I8) {
      const floatx4 v = acc[r] + bias;
      half4 hv, lv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = v[e] > 0.f ? v[e] : 0.f;
        hv[e] = (half_t)a;
        lv[e] = (half_t)(a - (float)hv[e]);
      }
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(floatx2, hv), drs_h, voff0 + (unsigned)r * rowb, 0, 0);
      if constexpr (SP) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(floatx2, lv), drs_l, voff0 + (unsigned)r * rowb, 0, 0);
    } else {   // the per-op kernels' int8 epilogue operation for operation (conv_epilogue): fma, ReLU, then fp16 or rint(y / s_out), clamp
      const floatx4 c = E::to_float(acc[r]);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = __builtin_fmaf(c[e], mult[e], bias[e]);
        v[e] = v[e] > 0.f ? v[e] : 0.f;
      }
      if (out16) {   // (wave-uniform)
        half4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = (half_t)v[e];
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(floatx2, hv), drs_h, voff0 + (unsigned)r * rowb, 0, 0);
      } else {
        unsigned q = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = __builtin_rintf(v[e] * out_inv);
          t = t > 127.f ? 127.f : (t < -127.f ? -127.f : t);
          q |= ((unsigned)(int)t & 0xFFu) << (8 * e);
        }
        __builtin_amdgcn_raw_buffer_store_b32(q, drs_h, voff0 + (unsigned)r * rowb, 0, 0);
      }
    }
  };

  frag wA[KBC], wB[KBC];
  uintx4 pvh[PITER], pvl[SP ? PITER : 1];
  auto ldw = [&](const unsigned char* at) {
    if constexpr (SP) return EltS::ldw(at);
    else return *reinterpret_cast<const frag*>(at);
  };
  // block g of chunk ch: tap g / CBC, channel block ch * CBC + g % CBC of the tap's CB blocks
  auto w_request = [&](frag (&w)[KBC], auto chc, auto g0c, auto g1c) {
    static_for<decltype(g0c)::value, decltype(g1c)::value>([&](auto gc) {
      constexpr int g = decltype(gc)::value, ch = decltype(chc)::value;
      w[g] = ldw(wptr + (size_t)((g / CBC) * CB + ch * CBC + g % CBC) * WB);
      __builtin_amdgcn_sched_barrier(0);   // (issue order = program order: the counted waits are exact)
    });
  };
  auto patch_request_one = [&](auto chc, auto itc) {
    constexpr int ch = decltype(chc)::value, it = decltype(itc)::value;
    const int sl = it * NT + (int)threadIdx.x;
    const int r = sl / nchx, cs = sl - r * nchx;
    const int ry = r / R0W, rx = r - ry * R0W;
    const int iy = ty0 - 1 + ry, ix = tx0 - 1 + rx;
    const bool in = sl < nslots && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const unsigned off = in ? (unsigned)(((iy * p.W + ix) * p.src_ld + sg.src_coff + ch * CC) * ESZ + (cs << 4)) : 0x40000000u;
    pvh[it] = __builtin_amdgcn_raw_buffer_load_b128(srs_h, off, 0, 0);
    if constexpr (SP) pvl[it] = __builtin_amdgcn_raw_buffer_load_b128(srs_l, off, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto patch_commit = [&](int buf) {
#pragma unroll
    for (int it = 0; it < PITER; ++it) {
      const int sl = it * NT + (int)threadIdx.x;
      const int r = sl / nchx, cs = sl - r * nchx;
      if (sl < nslots) {
        *reinterpret_cast<uintx4*>(conv_smem + buf * BUF + r * PITCH + cs * 16) = pvh[it];
        if constexpr (SP) *reinterpret_cast<uintx4*>(conv_smem + buf * BUF + PLANE + r * PITCH + cs * 16) = pvl[it];
      }
    }
  };
  typedef std::integral_constant<int, 0> I0;
  // chunk 0 -- request order = order of first use: tap row ky = 0, the patch slice, tap rows ky = 1, 2
  w_request(wA, I0{}, I0{}, std::integral_constant<int, 3 * CBC>{});
  static_for<0, PITER>([&](auto itc) { patch_request_one(I0{}, itc); });
  w_request(wA, I0{}, std::integral_constant<int, 3 * CBC>{}, std::integral_constant<int, KBC>{});
  if constexpr (STAMPS) stamp_b(p, 1, bid, nwg);
  patch_commit(0);
  lds_barrier();
  if constexpr (STAMPS) stamp_b(p, 2, bid, nwg);

  // the next chunk's requests, one per K-loop step from step 0 on: its patch slice first (committed at the end of this
  // chunk), then its weight blocks in the order of first use (all issued well before this chunk's last steps)
  constexpr int NREQ = PITER + KBC;
  constexpr int RSTRIDE = (STEPS * 7 / 8) / NREQ > 0 ? (STEPS * 7 / 8) / NREQ : 1;   // a request every RSTRIDE steps, all issued by ~7/8 of the loop
  static_assert(NREQ * RSTRIDE <= STEPS, "the next chunk's requests fit this chunk's steps");
  static_for<0, NCHUNK>([&](auto chc) {
    constexpr int ch = decltype(chc)::value;
    auto& w = (ch & 1) ? wB : wA;
    auto& wn = (ch & 1) ? wA : wB;
    const unsigned lo0 = (unsigned)((ch & 1) * BUF + l15 * PITCH + lq * 16);
    unsigned lo1 = lo0 + PLANE;                     // second base register: the lo plane (ds offsets are 16 bits)
    asm volatile("" : "+v"(lo1));
    auto bfrag = [&](auto sc) {
      constexpr int s = decltype(sc)::value, rho = s / (3 * CBC), kx = (s / CBC) % 3, cbl = s % CBC;
      constexpr int imm = (rho * R0W + kx) * PITCH + cbl * 64;
      if constexpr (SP) return frag{*reinterpret_cast<const half8*>(conv_smem + lo0 + imm), *reinterpret_cast<const half8*>(conv_smem + lo1 + imm)};
      else return *reinterpret_cast<const frag*>(conv_smem + lo0 + imm);
    };
    frag b[PF + 1];
    static_for<0, PF>([&](auto sc) { b[decltype(sc)::value] = bfrag(sc); });
    static_for<0, STEPS>([&](auto sc) {
      constexpr int s = decltype(sc)::value, rho = s / (3 * CBC), kx = (s / CBC) % 3, cbl = s % CBC;
      if constexpr (ch + 1 < NCHUNK && s % RSTRIDE == 0 && s / RSTRIDE < NREQ) {
        constexpr int q = s / RSTRIDE;
        if constexpr (q < PITER) patch_request_one(std::integral_constant<int, ch + 1>{}, std::integral_constant<int, (q < PITER ? q : 0)>{});
        else w_request(wn, std::integral_constant<int, ch + 1>{}, std::integral_constant<int, q - PITER>{}, std::integral_constant<int, q - PITER + 1>{});
      }
      if constexpr (s + PF < STEPS) b[(s + PF) % (PF + 1)] = bfrag(std::integral_constant<int, s + PF>{});
      if constexpr (ch == 0 && kx == 0 && cbl == 0 && rho < TH) acc[rho] = typename E::acc_t{0, 0, 0, 0};
      if constexpr (SP) {
        // term-major: the three products of a split MFMA go to the same accumulator, so the rows' MFMAs are interleaved
        // (lo*hi of rows rho, rho-1, rho-2, then hi*lo, then hi*hi): a dependent MFMA is three issues behind its predecessor
        static_for<0, 3>([&](auto tc) {
          constexpr int term = decltype(tc)::value;
          static_for<0, 3>([&](auto kyc) {
            constexpr int ky = decltype(kyc)::value, r = rho - ky;
            if constexpr (r >= 0 && r < TH) {
              const auto& wv = w[(ky * 3 + kx) * CBC + cbl];
              const auto& bv = b[s % (PF + 1)];
              acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(term == 0 ? wv.l : wv.h, term == 1 ? bv.l : bv.h, acc[r], 0, 0, 0);
            }
          });
        });
      } else {
        static_for<0, 3>([&](auto kyc) {
          constexpr int ky = decltype(kyc)::value, r = rho - ky;
          if constexpr (r >= 0 && r < TH) acc[r] = E::mma(w[(ky * 3 + kx) * CBC + cbl], b[s % (PF + 1)], acc[r]);
        });
      }
      // last chunk: row rho-3 was completed by the previous patch row: convert and store it in the shadow of this row's MFMAs
      if constexpr (ch == NCHUNK - 1 && kx == 1 && cbl == 0 && rho >= 3) store_row(std::integral_constant<int, rho - 3>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (ch + 1 < NCHUNK) {
      patch_commit((ch + 1) & 1);   // (that buffer's last readers finished before the barrier that ended the previous chunk)
      lds_barrier();
    }
  });
  if constexpr (STAMPS) stamp_b(p, 3, bid, nwg);
  store_row(std::integral_constant<int, TH - 1>{});
  if constexpr (STAMPS) { stamp_b(p, 4, bid, nwg); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp_wg(p, 1); }
}

template <typename T, int TH, int CIN, int NCHUNK, int NW>
__global__ __launch_bounds__(NW * 64) void conv3x3_wsc(const ConvParams p) {
  conv3x3_wsc_body<T, TH, CIN, NCHUNK, NW>(p, (int)blockIdx.x, (int)gridDim.x);
}
// fp16 engines: the P3 | P4 head pair (P4's workgroups take the first block ids). P3: 16 x 16 pixel tiles, two chunks of 64
// channels; P4: 8 x 16 tiles, four chunks. One wave per SIMD. (Two waves per SIMD -- 512-thread workgroups whose wave halves
// split the chunks by parity and meet in LDS -- were built and measured: the K loop reaches the MFMA pipe's rate, the launch
// does not get shorter: 14.3 against 13.4 us fp16, 30.7 against 28.5 us split; profiles/r03/head_pair_experiments.txt.)
__global__ __launch_bounds__(256) void conv_dual_head3x3_ws(const ConvParams pa, const ConvParams pb, int nb) {
  if ((int)blockIdx.x < nb) conv3x3_wsc_body<half_t, 8, 256, 4, 4>(pb, (int)blockIdx.x, nb);
  else conv3x3_wsc_body<half_t, 16, 128, 2, 4>(pa, (int)blockIdx.x - nb, (int)gridDim.x - nb);
}
// The same pair on LARGE frames (more workgroups than CUs even with the tiles above: 1280^2, or several such launches in flight):
// half-height tiles and at most 256 registers, so that TWO workgroups share a CU -- a SIMD's two waves together run at ~19 cycles
// per MFMA against 25 for a lone wave. At 640^2 (220 workgroups: one round either way) this form starts 440 workgroups that each
// load the same weights for half the pixels: launch 14.2 against 13.2 us; at 1280^2 38.2 -> 32.3 us, 0.32 -> 0.37 of the peak.
__global__ __launch_bounds__(256, 2) void conv_dual_head3x3_ws_small(const ConvParams pa, const ConvParams pb, int nb) {
  if ((int)blockIdx.x < nb) conv3x3_wsc_body<half_t, 4, 256, 4, 4>(pb, (int)blockIdx.x, nb);
  else conv3x3_wsc_body<half_t, 8, 128, 2, 4>(pa, (int)blockIdx.x - nb, (int)gridDim.x - nb);
}
__global__ __launch_bounds__(256) void conv_dual_head3x3_ws_stamped(const ConvParams pa, const ConvParams pb, int nb) {
  if ((int)blockIdx.x < nb) conv3x3_wsc_body<half_t, 8, 256, 4, 4, true>(pb, (int)blockIdx.x, nb);
  else conv3x3_wsc_body<half_t, 16, 128, 2, 4, true>(pa, (int)blockIdx.x - nb, (int)gridDim.x - nb);
}
__global__ __launch_bounds__(256) void conv_dual_head3x3_ws_s16_stamped(const ConvParams pa, const ConvParams pb, int nb) {
  if ((int)blockIdx.x < nb) conv3x3_wsc_body<s16_t, 8, 256, 4, 4, true>(pb, (int)blockIdx.x, nb);
  else conv3x3_wsc_body<s16_t, 16, 128, 4, 4, true>(pa, (int)blockIdx.x - nb, (int)gridDim.x - nb);
}
// STRICT engines: the P3 | P4 head pair on it (P4's workgroups take the first block ids)
__global__ __launch_bounds__(256) void conv_dual_head3x3_ws_s16(const ConvParams pa, const ConvParams pb, int nb) {
  if ((int)blockIdx.x < nb) conv3x3_wsc_body<s16_t, 8, 256, 4, 4>(pb, (int)blockIdx.x, nb);
  else conv3x3_wsc_body<s16_t, 16, 128, 4, 4>(pa, (int)blockIdx.x - nb, (int)gridDim.x - nb);
}

// INT8 engines: the same pair on int8 tensors (18 / 36 weight blocks of 64 k per wave: 72 / 144 registers)
__global__ __launch_bounds__(256) void conv_dual_head3x3_ws_i8(const ConvParams pa, const ConvParams pb, int nb) {
  if ((int)blockIdx.x < nb) conv3x3_wsc_body<signed char, 8, 256, 4, 4>(pb, (int)blockIdx.x, nb);
  else conv3x3_wsc_body<signed char, 16, 128, 2, 4>(pa, (int)blockIdx.x - nb, (int)gridDim.x - nb);
}

#ifndef UNINA_CONV_PROBE   // (ISA probe builds stop here: tools/isa_probe.sh compiles only the kernels above)
// ================================================================================================ dual launches
// Two INDEPENDENT convs of the same kernel family in ONE grid: workgroups [0, na) run conv A, the rest conv B. The P3
// and P4 head layers (model.py:361-365) are such pairs: each alone half-fills the chip (200 workgroups) and sits on the
// ~11-14 us cold-weights + patch-wait + launch plateau; side by side they share one launch and fill
// the CUs the other leaves idle. Same bodies, so results are bit-identical to the separate launches.
// (launch bound 4 waves per SIMD = two 512-thread workgroups per CU: both convs' workgroups must be co-resident for the
// side-by-side launch to overlap them -- with the 207 VGPRs of the stand-alone 8x16 body only one fits and the two halves
// simply run one after the other; hence the shallower queue, D = 8)
__global__ __launch_bounds__(512, 4) void conv_dual_head3x3(const ConvParams pa, const ConvParams pb, int na) {
  if ((int)blockIdx.x < na) conv3x3_regq_body<8, 16, 64, 128, 8, 8, 1>(pa, (int)blockIdx.x, na);
  else conv3x3_regq_body<8, 8, 64, 256, 8, 8, 1>(pb, (int)blockIdx.x - na, (int)gridDim.x - na);
}
// INT8 engines: the same pair on int8 inputs (P3 | P4 head layers .0 and .1 are int8 convs there)
__global__ __launch_bounds__(512, 4) void conv_dual_head3x3_i8(const ConvParams pa, const ConvParams pb, int na) {
  if ((int)blockIdx.x < na) conv3x3_regq_body<8, 16, 64, 128, 8, 8, 1, 1, signed char>(pa, (int)blockIdx.x, na);
  else conv3x3_regq_body<8, 8, 64, 256, 8, 8, 1, 1, signed char>(pb, (int)blockIdx.x - na, (int)gridDim.x - na);
}
// STRICT engines: the same pair on split-fp16 tensors (hi / lo patch images: one workgroup per CU)
__global__ __launch_bounds__(512) void conv_dual_head3x3_s16(const ConvParams pa, const ConvParams pb, int na) {
  if ((int)blockIdx.x < na) conv3x3_regq_body<8, 16, 64, 128, 8, 8, 1, 1, s16_t>(pa, (int)blockIdx.x, na);
  else conv3x3_regq_body<8, 8, 64, 256, 8, 8, 1, 1, s16_t>(pb, (int)blockIdx.x - na, (int)gridDim.x - na);
}
__global__ __launch_bounds__(256) void conv_dual_head1x1(const ConvParams pa, const ConvParams pb, int na) {
  if ((int)blockIdx.x < na) conv_glds_body<half_t, 128, 16, 64, 4, 1, 4>(pa, (int)blockIdx.x, na);
  else conv_glds_body<half_t, 128, 16, 64, 4, 1, 4>(pb, (int)blockIdx.x - na, (int)gridDim.x - na);
}

// ---------------------------------------------------------------------------------------------- launch side
namespace {

struct CfgInfo {
  int bm, bn, bk, stages;
  const char* name;
  void (*fn)(const ConvParams);
  size_t smem;         // im2col kernel: total dynamic LDS; halo kernel: weight ring only (the patch is added per op)
  int th, tw;          // halo / register-queue kernels: spatial tile (0 = im2col kernel)
  int cin = 0;         // register-queue kernel: the input channel count it is instantiated for (0 = any)
  int nthreads = 256;
  int stride = 1;      // register-queue kernel: conv stride it is instantiated for
  bool ws = false;     // weights-stationary 3x3 kernel: fp16 NHWC destination, ReLU, no residual / upsample / planar output
  int chunks = 0;      // chunked weights-stationary kernel (conv3x3_wsc_body): channel chunks; 0 = another kernel
};

constexpr size_t stage_bytes(int bm, int bn) { return (size_t)bm * (bn * 4 + 16); }  // epilogue staging tile (fp32 worst case)
constexpr size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

template <int BM, int BN, int BK, int WM, int WN, int ST, int NP = 1>
constexpr size_t smem_of() {
  return max_sz((size_t)ST * (((BM / 16 + BN / 16) * (BK / 32) * NP + 3) / 4) * 4 * 1024, stage_bytes(BM, BN));
}

#define CFG(T, TN, BM, BN, BK, WM, WN, ST)                                                         \
  {BM, BN, BK, ST, "conv_glds<" TN "," #BM "," #BN "," #BK "," #WM "," #WN "," #ST ">",            \
   conv_glds<T, BM, BN, BK, WM, WN, ST>, smem_of<BM, BN, BK, WM, WN, ST, Elem<T>::kPlanes>(), 0, 0}
#define HALO(T, TN, TH, TW, BN, BK, WM, WN, ST)                                                     \
  {(TH) * (TW), BN, BK, ST, "conv3x3_halo<" TN "," #TH "x" #TW "," #BN "," #BK "," #WM "," #WN "," #ST ">", \
   conv3x3_halo<T, TH, TW, BN, BK, WM, WN, ST>, (size_t)ST * ((((BN) / 16) * ((BK) / 32) + 3) / 4) * 4 * 1024, TH, TW}

#define REGQ(TH, TW, BN, CIN, NW, D)                                                                 \
  {(TH) * (TW), BN, 32, 3, "conv3x3_regq<f16," #TH "x" #TW "," #BN "," #CIN "," #NW "w>",                \
   conv3x3_regq<TH, TW, BN, CIN, NW, D>, 0, TH, TW, CIN, (NW) * 64, 1}
#define REGQ2(TH, TW, BN, CIN, NW, D)                                                                \
  {(TH) * (TW), BN, 32, 3, "conv3x3_regq<f16," #TH "x" #TW "," #BN "," #CIN "," #NW "w,s2>",             \
   conv3x3_regq<TH, TW, BN, CIN, NW, D, 2>, 0, TH, TW, CIN, (NW) * 64, 2}
#define REGQI(TH, TW, BN, CIN, NW, D)                                                                \
  {(TH) * (TW), BN, 32, 3, "conv3x3_regq<i8," #TH "x" #TW "," #BN "," #CIN "," #NW "w>",                 \
   conv3x3_regq<TH, TW, BN, CIN, NW, D, 1, 1, signed char>, 0, TH, TW, CIN, (NW) * 64, 1}
#define REGQI2(TH, TW, BN, CIN, NW, D)                                                               \
  {(TH) * (TW), BN, 32, 3, "conv3x3_regq<i8," #TH "x" #TW "," #BN "," #CIN "," #NW "w,s2>",              \
   conv3x3_regq<TH, TW, BN, CIN, NW, D, 2, 1, signed char>, 0, TH, TW, CIN, (NW) * 64, 2}
#define REGQS(TH, TW, BN, CIN, NW, D)                                                                \
  {(TH) * (TW), BN, 32, 3, "conv3x3_regq<s16," #TH "x" #TW "," #BN "," #CIN "," #NW "w>",                \
   conv3x3_regq<TH, TW, BN, CIN, NW, D, 1, 1, s16_t>, 0, TH, TW, CIN, (NW) * 64, 1}
#define REGQS2(TH, TW, BN, CIN, NW, D)                                                               \
  {(TH) * (TW), BN, 32, 3, "conv3x3_regq<s16," #TH "x" #TW "," #BN "," #CIN "," #NW "w,s2>",             \
   conv3x3_regq<TH, TW, BN, CIN, NW, D, 2, 1, s16_t>, 0, TH, TW, CIN, (NW) * 64, 2}
#define WS(TH, CIN, NCH, NW)                                                                         \
  {(TH) * 16, (NW) * 16, 32, 3, "conv3x3_ws<f16," #TH "x16," #CIN "/" #NCH "," #NW "w>",                 \
   conv3x3_wsc<half_t, TH, CIN, NCH, NW>, 0, TH, 16, CIN, (NW) * 64, 1, true, NCH}
#define WSI(TH, CIN, NCH, NW)                                                                        \
  {(TH) * 16, (NW) * 16, 32, 3, "conv3x3_ws<i8," #TH "x16," #CIN "/" #NCH "," #NW "w>",                  \
   conv3x3_wsc<signed char, TH, CIN, NCH, NW>, 0, TH, 16, CIN, (NW) * 64, 1, true, NCH}
#define WSS(TH, CIN, NCH, NW)                                                                        \
  {(TH) * 16, (NW) * 16, 32, 3, "conv3x3_ws<s16," #TH "x16," #CIN "/" #NCH "," #NW "w>",                 \
   conv3x3_wsc<s16_t, TH, CIN, NCH, NW>, 0, TH, 16, CIN, (NW) * 64, 1, true, NCH}
#define NOCFG {0, 0, 0, 0, "n/a", nullptr, 0, 0, 0, -1, 0, 0}

// [dtype][config]; BK is in fp16-equivalent k (KSUB = BK/32 fragment blocks): a K-step covers BK channels in fp16
// and BK/2 channels in fp32.
const CfgInfo kCfg[kNumDTypes][kCfgCount] = {
    {
        CFG(half_t, "f16", 64, 64, 64, 2, 2, 4),    // kCfg64x64k64
        CFG(half_t, "f16", 64, 64, 32, 2, 2, 4),    // kCfg64x64k32
        CFG(half_t, "f16", 128, 64, 64, 2, 2, 4),    // kCfg128x64k64
        CFG(half_t, "f16", 128, 64, 32, 2, 2, 4),    // kCfg128x64k32
        CFG(half_t, "f16", 128, 128, 64, 2, 2, 4),    // kCfg128x128k64
        CFG(half_t, "f16", 128, 32, 64, 4, 1, 4),    // kCfg128x32k64
        CFG(half_t, "f16", 128, 32, 32, 4, 1, 4),    // kCfg128x32k32
        CFG(half_t, "f16", 128, 16, 64, 4, 1, 4),    // kCfg128x16k64
        CFG(half_t, "f16", 32, 64, 64, 1, 4, 4),    // kCfg32x64k64
        CFG(half_t, "f16", 32, 64, 64, 1, 4, 8),    // kCfg32x64k64s8  (deep pipeline for latency-bound small grids)
        CFG(half_t, "f16", 64, 64, 64, 2, 2, 6),    // kCfg64x64k64s6
        HALO(half_t, "f16", 8, 8, 64, 64, 2, 2, 4),   // kCfgHalo8x8n64
        HALO(half_t, "f16", 8, 8, 32, 64, 4, 1, 4),   // kCfgHalo8x8n32
        HALO(half_t, "f16", 8, 16, 64, 64, 2, 2, 4),   // kCfgHalo8x16n64
        HALO(half_t, "f16", 8, 16, 32, 64, 4, 1, 4),   // kCfgHalo8x16n32
        HALO(half_t, "f16", 8, 8, 64, 32, 2, 2, 4),   // kCfgHalo8x8n64k32
        HALO(half_t, "f16", 8, 16, 32, 32, 4, 1, 4),   // kCfgHalo8x16n32k32
        CFG(half_t, "f16", 32, 64, 128, 1, 4, 4),     // kCfg32x64k128
        CFG(half_t, "f16", 64, 64, 128, 2, 2, 4),     // kCfg64x64k128
        REGQ(8, 16, 64, 128, 8, 16),                  // kCfgRegq8x16n64c128   (P3 head layers at 640^2)
        REGQ(8, 8, 64, 128, 8, 16),                   // kCfgRegq8x8n64c128
        REGQ(8, 8, 64, 256, 8, 16),                   // kCfgRegq8x8n64c256    (P4 head layers)
        REGQ(8, 8, 32, 256, 8, 16),                   // kCfgRegq8x8n32c256
        REGQ(8, 16, 64, 64, 8, 16),                   // kCfgRegq8x16n64c64    (P2 head layers when not fused)
        REGQ(8, 16, 32, 128, 8, 16),                  // kCfgRegq8x16n32c128
        // stride 2 (stage convs, PAN down-sampling convs)
        REGQ2(8, 8, 64, 64, 8, 16),                   // kCfgRegqS2_8x8n64c64     (stage2_conv, down1)
        REGQ2(8, 16, 64, 64, 8, 16),                  // kCfgRegqS2_8x16n64c64
        REGQ2(8, 8, 64, 128, 8, 16),                  // kCfgRegqS2_8x8n64c128    (stage3_conv, down2)
        REGQ2(4, 8, 64, 128, 8, 16),                  // kCfgRegqS2_4x8n64c128
        REGQ2(8, 16, 64, 32, 8, 8),                   // kCfgRegqS2_8x16n64c32    (stage1_conv)
        REGQ2(8, 8, 32, 128, 8, 16),                  // kCfgRegqS2_8x8n32c128
        WS(16, 128, 2, 4),                            // kCfgWs16x16n64c128    (P3 head layers, weights-stationary, two chunks of 64 channels)
        WS(8, 256, 4, 4),                             // kCfgWs8x16n64c256     (P4 head layers, weights-stationary, four chunks)
        NOCFG, NOCFG, NOCFG,                          // (split-fp16 weights-stationary kernels)
        WS(8, 128, 2, 4),                             // kCfgWs8x16n64c128     (the pair on large frames: half-height tiles, 230 VGPRs, two workgroups per CU)
        WS(4, 256, 4, 4),                             // kCfgWs4x16n64c256
    },
    {
        CFG(float, "f32", 64, 64, 64, 2, 2, 4),    // kCfg64x64k64
        CFG(float, "f32", 64, 64, 32, 2, 2, 4),    // kCfg64x64k32
        CFG(float, "f32", 128, 64, 64, 2, 2, 4),    // kCfg128x64k64
        CFG(float, "f32", 128, 64, 32, 2, 2, 4),    // kCfg128x64k32
        CFG(float, "f32", 128, 128, 64, 2, 2, 4),    // kCfg128x128k64
        CFG(float, "f32", 128, 32, 64, 4, 1, 4),    // kCfg128x32k64
        CFG(float, "f32", 128, 32, 32, 4, 1, 4),    // kCfg128x32k32
        CFG(float, "f32", 128, 16, 64, 4, 1, 4),    // kCfg128x16k64
        CFG(float, "f32", 32, 64, 64, 1, 4, 4),    // kCfg32x64k64
        CFG(float, "f32", 32, 64, 64, 1, 4, 8),    // kCfg32x64k64s8  (deep pipeline for latency-bound small grids)
        CFG(float, "f32", 64, 64, 64, 2, 2, 6),    // kCfg64x64k64s6
        HALO(float, "f32", 8, 8, 64, 64, 2, 2, 4),   // kCfgHalo8x8n64
        HALO(float, "f32", 8, 8, 32, 64, 4, 1, 4),   // kCfgHalo8x8n32
        HALO(float, "f32", 8, 16, 64, 64, 2, 2, 4),   // kCfgHalo8x16n64
        HALO(float, "f32", 8, 16, 32, 64, 4, 1, 4),   // kCfgHalo8x16n32
        HALO(float, "f32", 8, 8, 64, 32, 2, 2, 4),   // kCfgHalo8x8n64k32
        HALO(float, "f32", 8, 16, 32, 32, 4, 1, 4),   // kCfgHalo8x16n32k32
        CFG(float, "f32", 32, 64, 128, 1, 4, 4),     // kCfg32x64k128
        CFG(float, "f32", 64, 64, 128, 2, 2, 4),     // kCfg64x64k128
        NOCFG, NOCFG, NOCFG, NOCFG, NOCFG, NOCFG,     // register-queue / weights-stationary kernels: fp16 / int8 / split fp16 only
        NOCFG, NOCFG, NOCFG, NOCFG, NOCFG, NOCFG,
        NOCFG, NOCFG,
        NOCFG, NOCFG, NOCFG,
        NOCFG, NOCFG,                                 // (fp16 half-height weights-stationary tiles)
    },
    {
        CFG(signed char, "i8", 64, 64, 64, 2, 2, 4),    // kCfg64x64k64
        CFG(signed char, "i8", 64, 64, 32, 2, 2, 4),    // kCfg64x64k32
        CFG(signed char, "i8", 128, 64, 64, 2, 2, 4),    // kCfg128x64k64
        CFG(signed char, "i8", 128, 64, 32, 2, 2, 4),    // kCfg128x64k32
        CFG(signed char, "i8", 128, 128, 64, 2, 2, 4),    // kCfg128x128k64
        CFG(signed char, "i8", 128, 32, 64, 4, 1, 4),    // kCfg128x32k64
        CFG(signed char, "i8", 128, 32, 32, 4, 1, 4),    // kCfg128x32k32
        CFG(signed char, "i8", 128, 16, 64, 4, 1, 4),    // kCfg128x16k64
        CFG(signed char, "i8", 32, 64, 64, 1, 4, 4),    // kCfg32x64k64
        CFG(signed char, "i8", 32, 64, 64, 1, 4, 8),    // kCfg32x64k64s8  (deep pipeline for latency-bound small grids)
        CFG(signed char, "i8", 64, 64, 64, 2, 2, 6),    // kCfg64x64k64s6
        HALO(signed char, "i8", 8, 8, 64, 64, 2, 2, 4),   // kCfgHalo8x8n64
        HALO(signed char, "i8", 8, 8, 32, 64, 4, 1, 4),   // kCfgHalo8x8n32
        HALO(signed char, "i8", 8, 16, 64, 64, 2, 2, 4),   // kCfgHalo8x16n64
        HALO(signed char, "i8", 8, 16, 32, 64, 4, 1, 4),   // kCfgHalo8x16n32
        HALO(signed char, "i8", 8, 8, 64, 32, 2, 2, 4),   // kCfgHalo8x8n64k32
        HALO(signed char, "i8", 8, 16, 32, 32, 4, 1, 4),   // kCfgHalo8x16n32k32
        CFG(signed char, "i8", 32, 64, 128, 1, 4, 4),     // kCfg32x64k128
        CFG(signed char, "i8", 64, 64, 128, 2, 2, 4),     // kCfg64x64k128
        REGQI(8, 16, 64, 128, 8, 16),                 // kCfgRegq8x16n64c128   (P3 head layers)
        REGQI(8, 8, 64, 128, 8, 16),                  // kCfgRegq8x8n64c128
        REGQI(8, 8, 64, 256, 8, 16),                  // kCfgRegq8x8n64c256    (P4 head layers)
        REGQI(8, 8, 32, 256, 8, 16),                  // kCfgRegq8x8n32c256
        NOCFG,                                        // kCfgRegq8x16n64c64    (the P2 head is an fp16 carve-out)
        REGQI(8, 16, 32, 128, 8, 16),                 // kCfgRegq8x16n32c128
        REGQI2(8, 8, 64, 64, 8, 8),                   // kCfgRegqS2_8x8n64c64     (down1)
        REGQI2(8, 16, 64, 64, 8, 8),                  // kCfgRegqS2_8x16n64c64
        REGQI2(8, 8, 64, 128, 8, 16),                 // kCfgRegqS2_8x8n64c128    (stage3_conv, down2)
        REGQI2(4, 8, 64, 128, 8, 16),                 // kCfgRegqS2_4x8n64c128
        NOCFG,                                        // kCfgRegqS2_8x16n64c32    (Cin 32 < one int8 block)
        REGQI2(8, 8, 32, 128, 8, 16),                 // kCfgRegqS2_8x8n32c128
        WSI(16, 128, 2, 4),                           // kCfgWs16x16n64c128
        WSI(8, 256, 4, 4),                            // kCfgWs8x16n64c256
        NOCFG, NOCFG, NOCFG,
        NOCFG, NOCFG,                                 // (fp16 half-height weights-stationary tiles)
    },
    {   // split fp16 (kS16): a K-step stages (hi, lo) block pairs -- twice the LDS per stage, hence shallower rings on the wide tiles
        CFG(s16_t, "s16", 64, 64, 64, 2, 2, 3),     // kCfg64x64k64
        CFG(s16_t, "s16", 64, 64, 32, 2, 2, 4),     // kCfg64x64k32
        CFG(s16_t, "s16", 128, 64, 64, 2, 2, 3),    // kCfg128x64k64
        CFG(s16_t, "s16", 128, 64, 32, 2, 2, 4),    // kCfg128x64k32
        NOCFG,                                      // kCfg128x128k64 (a stage would be 64 KB)
        CFG(s16_t, "s16", 128, 32, 64, 4, 1, 3),    // kCfg128x32k64
        CFG(s16_t, "s16", 128, 32, 32, 4, 1, 4),    // kCfg128x32k32
        CFG(s16_t, "s16", 128, 16, 64, 4, 1, 3),    // kCfg128x16k64
        CFG(s16_t, "s16", 32, 64, 64, 1, 4, 4),     // kCfg32x64k64
        NOCFG,                                      // kCfg32x64k64s8
        NOCFG,                                      // kCfg64x64k64s6
        NOCFG, NOCFG, NOCFG, NOCFG, NOCFG, NOCFG,   // halo kernels
        NOCFG, NOCFG,                               // kCfg32x64k128, kCfg64x64k128
        // register-queue 3x3: hi / lo patch images in LDS, (hi | lo) weight block pairs through the queue (D pairs = 2 D KiB in flight)
        REGQS(8, 16, 64, 128, 8, 8),                // kCfgRegq8x16n64c128   (P3 head layers)
        REGQS(8, 8, 64, 128, 8, 8),                 // kCfgRegq8x8n64c128
        REGQS(8, 8, 64, 256, 8, 8),                 // kCfgRegq8x8n64c256    (P4 head layers)
        NOCFG,                                      // kCfgRegq8x8n32c256
        REGQS(8, 16, 64, 64, 8, 8),                 // kCfgRegq8x16n64c64    (P2 head layers)
        NOCFG,                                      // kCfgRegq8x16n32c128
        REGQS2(8, 8, 64, 64, 8, 8),                 // kCfgRegqS2_8x8n64c64     (stage2_conv, down1)
        NOCFG,                                      // kCfgRegqS2_8x16n64c64
        REGQS2(8, 8, 64, 128, 8, 8),                // kCfgRegqS2_8x8n64c128    (stage3_conv, down2)
        REGQS2(4, 8, 64, 128, 8, 8),                // kCfgRegqS2_4x8n64c128
        REGQS2(8, 16, 64, 32, 8, 8),                // kCfgRegqS2_8x16n64c32    (stage1_conv)
        NOCFG,                                      // kCfgRegqS2_8x8n32c128
        WSS(16, 128, 4, 4),                         // kCfgWs16x16n64c128    (P3 head layers in the pair: 16-row tiles, four chunks of 32 channels --
                                                    //  100 + 120 workgroups run in ONE round; with 8-row tiles the pair's 320 needed two)
        NOCFG,                                      // kCfgWs8x16n64c256     (fp16 / int8 only)
        WSS(16, 64, 2, 4),                          // kCfgWsS8x16n64c64     (P2 head layers: 16-row tiles, two chunks of 32 channels -- 200 workgroups, ONE round; the 8-row
                                                    //  single-chunk form ran 400 workgroups at one per CU in two rounds: 18 us per layer)
        WSS(8, 128, 2, 4),                          // kCfgWsS8x16n64c128    (P3 head layers: two chunks of 64 channels)
        WSS(8, 256, 4, 4),                          // kCfgWsS8x16n64c256    (P4 head layers: four chunks. 4-row tiles -- 240 workgroups of the P3 conv's size -- ran 34 us per pair against 30.6)
        NOCFG, NOCFG,                               // (fp16 half-height weights-stationary tiles)
    },
};
#undef CFG
#undef HALO
#undef REGQ
#undef REGQ2
#undef REGQI
#undef REGQI2
#undef NOCFG
#undef WS
#undef WSI
#undef WSS
#undef REGQS
#undef REGQS2

inline int block_k(int dtype) { return dtype == kF32 ? 16 : (dtype == kI8 ? 64 : 32); }   // (kS16: 32 per plane)
inline int kstep_of(const ConvParams& p, const CfgInfo& c) { return (c.bk / 32) * block_k(p.dtype); }
inline size_t esize(const ConvParams& p) { return p.dtype == kF32 ? 4 : (p.dtype == kI8 ? 1 : 2); }   // (kS16: per plane)
inline size_t smem_for(const ConvParams& p, const CfgInfo& c) {
  if (!c.th) return c.smem;
  if (c.cin) {  // register-queue kernel: patch (+ < 1 KiB overrun of its last DMA instruction) or the epilogue staging tile
    const size_t ph = c.stride * (c.th - 1) + 3, pw = c.stride * (c.tw - 1) + 3;
    const size_t patch = ((ph * pw * c.cin * esize(p) + 1023) & ~(size_t)1023) + 1024;
    if (c.ws && c.chunks)   // image(s) of a channel chunk, two buffers (conv3x3_wsc_body)
      return (c.chunks > 1 ? 2 : 1) * (p.dtype == kS16 ? 2 : 1) * ph * pw * ((c.cin / c.chunks) * (p.dtype == kI8 ? 1 : 2) + 32);
    if (p.dtype == kS16) return 2 * patch;                // hi and lo images; accumulators are stored straight from registers
    return max_sz(patch, stage_bytes((c.bm + 15) & ~15, c.bn));
  }
  const size_t patch = (size_t)(c.th + 2) * (c.tw + 2) * p.Cin * esize(p);
  return max_sz(c.smem + ((patch + 1023) & ~(size_t)1023), stage_bytes(c.bm, c.bn));
}
constexpr size_t kMaxLds = 160 * 1024;

int n_tiles(const ConvParams& p, int bn) {
  int t = 0;
  for (int s = 0; s < p.nseg; ++s) t += (((p.seg[s].n_count + 15) & ~15) + bn - 1) / bn;
  return t;
}

}  // namespace

hipError_t conv_init() {
  for (const void* f : {reinterpret_cast<const void*>(conv_dual_head3x3), reinterpret_cast<const void*>(conv_dual_head1x1),
                        reinterpret_cast<const void*>(conv_dual_head3x3_i8), reinterpret_cast<const void*>(conv_dual_head3x3_s16),
                        reinterpret_cast<const void*>(conv_dual_head3x3_ws_s16), reinterpret_cast<const void*>(conv_dual_head3x3_ws_s16_stamped),
                        reinterpret_cast<const void*>(conv_dual_head3x3_ws), reinterpret_cast<const void*>(conv_dual_head3x3_ws_stamped),
                        reinterpret_cast<const void*>(conv_dual_head3x3_ws_i8),
                        }) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (e != hipSuccess) return e;
  }
  for (int d = 0; d < kNumDTypes; ++d)
    for (int c = 0; c < kCfgCount; ++c) {
      if (!kCfg[d][c].fn) continue;
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kCfg[d][c].fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(kCfg[d][c].th ? kMaxLds : kCfg[d][c].smem));
      if (e != hipSuccess) return e;
    }
  return hipSuccess;
}

bool conv_config_valid(const ConvParams& p, int cfg) {
  if (cfg < 0 || cfg >= kCfgCount) return false;
  const CfgInfo& c = kCfg[p.dtype][cfg];
  if (!c.fn) return false;
  if (c.cin) {  // register-queue kernel: 3x3 on exactly its Cin and stride (fp16 / int8 rows of the table); every slice at least one tile wide
    if (p.Cin != c.cin || p.ksize != 3 || p.stride != c.stride || p.pad != 1 || smem_for(p, c) > kMaxLds) return false;
    for (int s = 0; s < p.nseg; ++s)
      if (!p.seg[s].w_lane) return false;   // these kernels read the lane-order twin of the weights
    if (c.ws) {
      if (!p.relu || p.res || !p.zeros) return false;
      for (int s = 0; s < p.nseg; ++s)
        if ((p.seg[s].out_dtype != p.dtype && !(p.dtype == kI8 && p.seg[s].out_dtype == kF16)) || p.seg[s].up2 || p.seg[s].dst_planar ||
            (p.dtype == kI8) != (p.seg[s].mult != nullptr)) return false;
    }
    for (int s = 0; s < p.nseg; ++s)
      if (((p.seg[s].n_count + 15) & ~15) < c.bn) return false;
    return true;
  }
  if (p.Cin % kstep_of(p, c)) return false;
  if (c.th) {  // halo kernel: 3x3, stride 1, pad 1, power-of-two chunk count, patch + ring must fit the CU's LDS
    const int nch = (int)(p.Cin * esize(p) / 16);
    if (p.ksize != 3 || p.stride != 1 || p.pad != 1 || (nch & (nch - 1)) || smem_for(p, c) > kMaxLds) return false;
  }
  int min_npad = 1 << 30;
  for (int s = 0; s < p.nseg; ++s) {
    const int np = (p.seg[s].n_count + 15) & ~15;
    if (np < min_npad) min_npad = np;
  }
  return c.bn <= min_npad || (c.bn == 16);
}

ConvLaunch conv_plan_with(const ConvParams& p, int cfg) {
  const CfgInfo& c = kCfg[p.dtype][cfg];
  ConvLaunch l;
  l.cfg = (ConvConfig)cfg;
  if (c.th) l.grid = dim3(((p.Ho + c.th - 1) / c.th) * ((p.Wo + c.tw - 1) / c.tw), n_tiles(p, c.bn), 1);
  else l.grid = dim3((p.M + c.bm - 1) / c.bm, n_tiles(p, c.bn), 1);
  l.block = dim3(c.nthreads, 1, 1);
  l.kernel_name = c.name;
  return l;
}

// Heuristic: the widest tile that still yields >= ~1.5 workgroups per CU (256 CUs), K-step 64 when Cin allows.
ConvLaunch conv_plan(const ConvParams& p) {
  const int override_cfg = p.force_cfg;
  if (override_cfg >= 0 && conv_config_valid(p, override_cfg)) return conv_plan_with(p, override_cfg);
  if (p.dtype == kS16 && p.ksize == 3) {   // STRICT engines: 3x3 convs on the patch-resident kernels even without autotuning
    for (int c : {(int)kCfgWsS8x16n64c64, (int)kCfgWsS8x16n64c128, (int)kCfgWsS8x16n64c256, (int)kCfgRegqS2_8x16n64c32, (int)kCfgRegqS2_8x8n64c64,
                  (int)kCfgRegqS2_4x8n64c128, (int)kCfgRegq8x16n64c64, (int)kCfgRegq8x16n64c128, (int)kCfgRegq8x8n64c256})
      if (conv_config_valid(p, c)) return conv_plan_with(p, c);
  }
  const bool k64 = (p.Cin % (2 * block_k(p.dtype))) == 0;
  int min_npad = 1 << 30;
  for (int s = 0; s < p.nseg; ++s) {
    const int np = (p.seg[s].n_count + 15) & ~15;
    if (np < min_npad) min_npad = np;
  }
  int cfg;
  if (min_npad < 32) cfg = kCfg128x16k64;
  else if (min_npad < 64) cfg = k64 ? kCfg128x32k64 : kCfg128x32k32;
  else {
    auto blocks = [&](int c) { return ((p.M + kCfg[0][c].bm - 1) / kCfg[0][c].bm) * n_tiles(p, kCfg[0][c].bn); };
    const int want = 384;
    if (k64 && min_npad >= 128 && blocks(kCfg128x128k64) >= want) cfg = kCfg128x128k64;
    else if (blocks(k64 ? kCfg128x64k64 : kCfg128x64k32) >= want) cfg = k64 ? kCfg128x64k64 : kCfg128x64k32;
    else if (!k64 || blocks(kCfg64x64k64) >= want || p.M % 64) cfg = k64 ? kCfg64x64k64 : kCfg64x64k32;
    else cfg = kCfg32x64k64;
  }
  if (!conv_config_valid(p, cfg)) cfg = k64 ? kCfg64x64k64 : kCfg64x64k32;
  return conv_plan_with(p, cfg);
}

namespace {
// launch-time fields of a ConvParams for configuration `cfg` (tile counts, division magics, slice -> tile ranges)
dim3 conv_prepare(ConvParams& p, int cfg) {
  const CfgInfo& c = kCfg[p.dtype][cfg];
  const ConvLaunch l = conv_plan_with(p, cfg);
  p.grid_m = (int)l.grid.x;
  p.gm_magic = div_magic(l.grid.x);
  p.grid_n = (int)l.grid.y;
  p.gn_magic = div_magic(l.grid.y);
  {  // which operand should each XCD's L2 see only its share of? The other one is fetched by all 8 XCDs.
    const double in_bytes = (double)esize(p) * p.H * p.W * p.Cin * ((p.nseg > 1 && p.seg[0].src_coff != p.seg[1].src_coff) ? p.nseg : 1);
    double w_bytes = 0;
    for (int s = 0; s < p.nseg; ++s) w_bytes += (double)esize(p) * ((p.seg[s].n_count + 15) & ~15) * p.ksize * p.ksize * p.Cin;
    p.xcd_m_major = (l.grid.x >= 8 && in_bytes + 8.0 * w_bytes < 8.0 * in_bytes + w_bytes) ? 1 : 0;
  }
  p.wo_magic = div_magic((unsigned)p.Wo);
  p.spt_magic = div_magic((unsigned)(p.Cin / kstep_of(p, c)));
  if (c.th) p.tx_magic = div_magic((unsigned)((p.Wo + c.tw - 1) / c.tw));
  int t = 0;
  for (int s = 0; s < p.nseg; ++s) {
    p.seg[s].tile0 = t;
    t += (((p.seg[s].n_count + 15) & ~15) + c.bn - 1) / c.bn;
  }
  return l.grid;
}
}  // namespace

hipError_t conv_launch(const ConvParams& pin, const ConvLaunch& l, hipStream_t stream) {
  ConvParams p = pin;
  const CfgInfo& c = kCfg[p.dtype][l.cfg];
  const dim3 g = conv_prepare(p, l.cfg);
  hipLaunchKernelGGL(c.fn, dim3(g.x * g.y, 1, 1), l.block, smem_for(p, c), stream, p);  // 1-D: see tile_of_block
  return hipGetLastError();
}

// ---- dual launches (conv_dual_head3x3 / conv_dual_head1x1) ----
namespace {
struct DualKind {
  int cfg_a, cfg_b, threads;
  const char* name;
  void (*fn)(const ConvParams, const ConvParams, int);
};
enum { kDualRegq = 0, kDual1x1, kDualRegqI8, kDualWs, kDualWsI8, kDualRegqS16, kDualWsS16, kDualWsSmall, kDualKinds };
const DualKind kDual[kDualKinds] = {
    {kCfgRegq8x16n64c128, kCfgRegq8x8n64c256, 512, "conv_dual_head3x3<regq 8x16,64,128 | regq 8x8,64,256>", conv_dual_head3x3},
    {kCfg128x16k64, kCfg128x16k64, 256, "conv_dual_head1x1<glds 128,16,64 x2>", conv_dual_head1x1},
    {kCfgRegq8x16n64c128, kCfgRegq8x8n64c256, 512, "conv_dual_head3x3_i8<regq i8,8x16,64,128 | regq i8,8x8,64,256>", conv_dual_head3x3_i8},
    {kCfgWs16x16n64c128, kCfgWs8x16n64c256, 256, "conv_dual_head3x3_ws<ws 16x16,64,128/2 | ws 8x16,64,256/4>", conv_dual_head3x3_ws},
    {kCfgWs16x16n64c128, kCfgWs8x16n64c256, 256, "conv_dual_head3x3_ws_i8<ws i8,16x16,64,128/2 | ws i8,8x16,64,256/4>", conv_dual_head3x3_ws_i8},
    {kCfgRegq8x16n64c128, kCfgRegq8x8n64c256, 512, "conv_dual_head3x3_s16<regq s16,8x16,64,128 | regq s16,8x8,64,256>", conv_dual_head3x3_s16},
    {kCfgWs16x16n64c128, kCfgWsS8x16n64c256, 256, "conv_dual_head3x3_ws_s16<ws s16,16x16,64,128/4 | ws s16,8x16,64,256/4>", conv_dual_head3x3_ws_s16},
    {kCfgWs8x16n64c128, kCfgWs4x16n64c256, 256, "conv_dual_head3x3_ws_small<ws 8x16,64,128/2 | ws 4x16,64,256/4, 2 per CU>", conv_dual_head3x3_ws_small},
};
}  // namespace

int conv_dual_match(const ConvParams& a, const ConvParams& b) {
  if (a.stamps || b.stamps) return -1;   // (stamped launches go through conv_dual_launch with an explicit kind)
  auto fits = [&](int k) { return a.ksize == 3 && b.ksize == 3 && conv_config_valid(a, kDual[k].cfg_a) && conv_config_valid(b, kDual[k].cfg_b); };
  // the weights-stationary pairs are the default; UNINA_DUAL_WS=0 selects the register-queue pairs (the tile-kernel family: an
  // independent implementation of the same convs, tests/test_gpu_pipeline.py compares the two byte for byte)
  const char* wsenv = getenv("UNINA_DUAL_WS");
  const bool ws = !(wsenv && wsenv[0] == '0');
  if (a.dtype == kI8 && b.dtype == kI8) {
    if (ws && fits(kDualWsI8)) return kDualWsI8;
    return fits(kDualRegqI8) ? kDualRegqI8 : -1;
  }
  if (a.dtype == kS16 && b.dtype == kS16) {   // STRICT engines: the weights-stationary pair (UNINA_DUAL_WS=0: the register-queue pair)
    if (ws && fits(kDualWsS16)) return kDualWsS16;
    return fits(kDualRegqS16) ? kDualRegqS16 : -1;
  }
  if (a.dtype != kF16 || b.dtype != kF16) return -1;
  // the weights-stationary pair: default for fp16 (same-box A/B against the register-queue pair: +2-3 % frames/s at 2 frames in
  // flight, serial latency equal within noise; workgroup lives 9-11 us against 12-15). UNINA_DUAL_WS=0 falls back.
  if (ws && fits(kDualWs)) {
    // more workgroups than CUs with the full-height tiles: the half-height form, two workgroups per CU
    if (fits(kDualWsSmall) && conv_dual_grid(kDualWs, a, b) > 256) return kDualWsSmall;
    return kDualWs;
  }
  if (fits(kDualRegq)) return kDualRegq;
  auto tiny = [](const ConvParams& p) {
    for (int s = 0; s < p.nseg; ++s)
      if (p.seg[s].n_count > 16) return false;
    return p.ksize == 1 && p.stride == 1;
  };
  if (tiny(a) && tiny(b) && a.Cin != b.Cin && conv_config_valid(a, kDual[kDual1x1].cfg_a) && conv_config_valid(b, kDual[kDual1x1].cfg_b)) return kDual1x1;
  return -1;
}

const char* conv_dual_name(int kind) { return kind >= 0 && kind < kDualKinds ? kDual[kind].name : "?"; }

int conv_dual_grid(int kind, const ConvParams& pa_in, const ConvParams& pb_in) {
  if (kind < 0 || kind >= kDualKinds) return -1;
  ConvParams pa = pa_in, pb = pb_in;
  const dim3 ga = conv_prepare(pa, kDual[kind].cfg_a), gb = conv_prepare(pb, kDual[kind].cfg_b);
  return (int)(ga.x * ga.y + gb.x * gb.y);
}

hipError_t conv_dual_launch(int kind, const ConvParams& pa_in, const ConvParams& pb_in, hipStream_t stream, int* grid_out) {
  if (kind < 0 || kind >= kDualKinds) return hipErrorInvalidValue;
  const DualKind& k = kDual[kind];
  ConvParams pa = pa_in, pb = pb_in;
  const dim3 ga = conv_prepare(pa, k.cfg_a), gb = conv_prepare(pb, k.cfg_b);
  const int na = (int)(ga.x * ga.y), nb = (int)(gb.x * gb.y);
  const size_t sa = smem_for(pa, kCfg[pa.dtype][k.cfg_a]), sb = smem_for(pb, kCfg[pb.dtype][k.cfg_b]);
  if (grid_out) *grid_out = na + nb;
  auto fn = k.fn;
  if (pa.stamps || pb.stamps) {   // debug: the stamped twin (only the default fp16 pair has one)
    if (kind != kDualWs && kind != kDualWsS16) return hipErrorInvalidValue;
    fn = kind == kDualWs ? conv_dual_head3x3_ws_stamped : conv_dual_head3x3_ws_s16_stamped;
  }
  hipLaunchKernelGGL(fn, dim3(na + nb, 1, 1), dim3(k.threads, 1, 1), max_sz(sa, sb), stream, pa, pb, (kind == kDualWs || kind == kDualWsI8 || kind == kDualWsS16 || kind == kDualWsSmall) ? nb : na);   // (the weights-stationary pairs put conv B first)
  return hipGetLastError();
}

const char* conv_config_name(int cfg, int dtype) {
  return (cfg >= 0 && cfg < kCfgCount && dtype >= 0 && dtype < kNumDTypes) ? kCfg[dtype][cfg].name : "?";
}

}  // namespace unina
#else
}  // namespace unina
#endif
