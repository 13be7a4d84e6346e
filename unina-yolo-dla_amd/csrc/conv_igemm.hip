// conv_igemm.hip -- implicit-GEMM convolution on CDNA4 matrix cores (gfx950).
//
// Computes, for every output pixel m and output channel n of a slice,
//     y[m][n] = act( bias[n] + sum_{kh,kw,c} W[n][kh][kw][c] * x[pix(m,kh,kw)][c] )  (+ residual[m][n])
// which is ConvBlock = Conv2d(bias=False, pad=k//2) -> BatchNorm(eval, folded) -> ReLU of the reference
// (unina_yolo_dla/model.py:23-50), the Bottleneck shortcut add (model.py:73), the head output convs
// (model.py:292,299: bias, no activation, fp32 out) and the folded nearest-x2 Upsample (model.py:145-147).
//
// Mapping to MFMA (v_mfma_f32_16x16x32_f16): the WEIGHTS are the A operand (rows = output channels) and
// the ACTIVATIONS the B operand (columns = pixels), i.e. D = W * X^T. Both operands are K-contiguous in
// memory (weights [n][K], activations NHWC), so every lane's fragment is one 16-byte load, and the
// accumulator comes out with 4 consecutive output CHANNELS of one pixel per lane -> 8-byte NHWC stores.
//   A frag: lane l holds W[n0 + (l&15)][k0 + 8*(l>>4) .. +8)
//   B frag: lane l holds X[m0 + (l&15)][k0 + 8*(l>>4) .. +8)
//   D     : lane l, reg r  ->  channel n0 + 4*(l>>4) + r, pixel m0 + (l&15)
#include "kernels.h"

namespace unina {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int WAVES_M, int WAVES_N, int WM_T, int WN_T>
__global__ __launch_bounds__(256) void conv_igemm_f16(const ConvParams p) {
  constexpr int BM = WAVES_M * WM_T * 16;
  constexpr int BN = WAVES_N * WN_T * 16;
  static_assert(WAVES_M * WAVES_N == 4, "256-thread blocks");
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int wm = wid % WAVES_M, wn = wid / WAVES_M;
  const int l15 = lane & 15, lq = lane >> 4;

  const int sidx = (p.nseg > 1 && (int)blockIdx.y >= p.seg[1].tile0) ? 1 : 0;
  const ConvSeg& sg = p.seg[sidx];
  const int n_pad = (sg.n_count + 15) & ~15;
  const int n0 = ((int)blockIdx.y - sg.tile0) * BN + wn * (WN_T * 16);
  if (n0 >= n_pad) return;  // no barriers in this kernel: a wave with no channels may leave

  const int K = p.ksize * p.ksize * p.Cin;
  const int m_base = blockIdx.x * BM + wm * (WM_T * 16);

  int oy[WM_T], ox[WM_T];
  bool mvalid[WM_T];
#pragma unroll
  for (int i = 0; i < WM_T; ++i) {
    const int m = m_base + i * 16 + l15;
    mvalid[i] = m < p.M;
    const int mm = mvalid[i] ? m : 0;
    oy[i] = mm / p.Wo;
    ox[i] = mm - oy[i] * p.Wo;
  }

  floatx4 acc[WN_T][WM_T];
#pragma unroll
  for (int j = 0; j < WN_T; ++j)
#pragma unroll
    for (int i = 0; i < WM_T; ++i) acc[j][i] = floatx4{0.f, 0.f, 0.f, 0.f};

  const half_t* wrow[WN_T];
  bool nvalid[WN_T];
#pragma unroll
  for (int j = 0; j < WN_T; ++j) {
    nvalid[j] = (n0 + j * 16) < n_pad;  // wave-uniform
    wrow[j] = sg.w + (size_t)(n0 + j * 16 + l15) * K + lq * 8;
  }
  const half_t* src = p.src + sg.src_coff + lq * 8;
  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  int kbase = 0;
  for (int kh = 0; kh < p.ksize; ++kh) {
    for (int kw = 0; kw < p.ksize; ++kw) {
      const half_t* px[WM_T];
      bool pvalid[WM_T];
#pragma unroll
      for (int i = 0; i < WM_T; ++i) {
        const int iy = oy[i] * p.stride + kh - p.pad;
        const int ix = ox[i] * p.stride + kw - p.pad;
        pvalid[i] = mvalid[i] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        px[i] = src + (size_t)(pvalid[i] ? (iy * p.W + ix) : 0) * p.src_ld;
      }
      for (int c0 = 0; c0 < p.Cin; c0 += 32, kbase += 32) {
        half8 a[WN_T], b[WM_T];
#pragma unroll
        for (int j = 0; j < WN_T; ++j) a[j] = nvalid[j] ? *reinterpret_cast<const half8*>(wrow[j] + kbase) : zero8;
#pragma unroll
        for (int i = 0; i < WM_T; ++i) b[i] = pvalid[i] ? *reinterpret_cast<const half8*>(px[i] + c0) : zero8;
#pragma unroll
        for (int j = 0; j < WN_T; ++j)
#pragma unroll
          for (int i = 0; i < WM_T; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[i], acc[j][i], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: bias, ReLU, residual (after the ReLU: model.py:72-73), store ----
#pragma unroll
  for (int j = 0; j < WN_T; ++j) {
    const int n = n0 + j * 16 + lq * 4;  // first of this lane's 4 consecutive channels (slice-relative)
    if (!nvalid[j] || n >= sg.n_count) continue;
    const floatx4 bias = *reinterpret_cast<const floatx4*>(sg.bias + n);
#pragma unroll
    for (int i = 0; i < WM_T; ++i) {
      if (!mvalid[i]) continue;
      const int m = m_base + i * 16 + l15;
      floatx4 v = acc[j][i] + bias;
      if (p.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      }
      if (p.res) {
        const half4 rv = *reinterpret_cast<const half4*>(p.res + (size_t)m * p.res_ld + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
      }
      if (sg.dst_planar) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < sg.n_count) sg.dst_planar[(size_t)(n + r) * p.M + m] = v[r];
      } else {
        half4 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = (half_t)v[r];
        if (sg.up2) {
          const size_t row = (size_t)(2 * oy[i]) * (2 * p.Wo) + 2 * ox[i];
          half_t* d = sg.dst + row * sg.dst_ld + n;
          *reinterpret_cast<half4*>(d) = hv;
          *reinterpret_cast<half4*>(d + sg.dst_ld) = hv;
          *reinterpret_cast<half4*>(d + (size_t)(2 * p.Wo) * sg.dst_ld) = hv;
          *reinterpret_cast<half4*>(d + (size_t)(2 * p.Wo + 1) * sg.dst_ld) = hv;
        } else {
          *reinterpret_cast<half4*>(sg.dst + (size_t)m * sg.dst_ld + n) = hv;
        }
      }
    }
  }
}

namespace {
struct CfgInfo {
  int bm, bn;
  const char* name;
};
const CfgInfo kCfg[kCfgCount] = {
    {64, 64, "conv_igemm_f16<2,2,2,2>"},
    {128, 32, "conv_igemm_f16<4,1,2,2>"},
    {128, 16, "conv_igemm_f16<4,1,2,1>"},
};
}  // namespace

ConvLaunch conv_plan(const ConvParams& p) {
  int min_npad = 1 << 30;
  for (int s = 0; s < p.nseg; ++s) {
    const int np = (p.seg[s].n_count + 15) & ~15;
    if (np < min_npad) min_npad = np;
  }
  ConvLaunch l;
  l.cfg = min_npad >= 64 ? kCfg64x64 : (min_npad >= 32 ? kCfg128x32 : kCfg128x16);
  const CfgInfo& c = kCfg[l.cfg];
  int ntiles = 0;
  for (int s = 0; s < p.nseg; ++s) ntiles += (((p.seg[s].n_count + 15) & ~15) + c.bn - 1) / c.bn;
  l.grid = dim3((p.M + c.bm - 1) / c.bm, ntiles, 1);
  l.block = dim3(256, 1, 1);
  l.kernel_name = c.name;
  return l;
}

// fills seg[].tile0 for the chosen config (the caller's params are const: work on a copy)
hipError_t conv_launch(const ConvParams& pin, const ConvLaunch& l, hipStream_t stream) {
  ConvParams p = pin;
  const CfgInfo& c = kCfg[l.cfg];
  int t = 0;
  for (int s = 0; s < p.nseg; ++s) {
    p.seg[s].tile0 = t;
    t += (((p.seg[s].n_count + 15) & ~15) + c.bn - 1) / c.bn;
  }
  switch (l.cfg) {
    case kCfg64x64:
      conv_igemm_f16<2, 2, 2, 2><<<l.grid, l.block, 0, stream>>>(p);
      break;
    case kCfg128x32:
      conv_igemm_f16<4, 1, 2, 2><<<l.grid, l.block, 0, stream>>>(p);
      break;
    case kCfg128x16:
      conv_igemm_f16<4, 1, 2, 1><<<l.grid, l.block, 0, stream>>>(p);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace unina
