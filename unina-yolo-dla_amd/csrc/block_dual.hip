// block_dual.hip -- two INDEPENDENT block kernels side by side in one grid (same idea as conv_igemm.hip's dual launches).
//
// After the FPN the graph forks (model.py:262-269, 361-365): the P2 head needs only p2_fused, while the PAN path
// down1 -> pan_c3k2_1 -> down2 -> pan_c3k2_2 leads to the P3 / P4 heads. head_fused (200 workgroups at 640^2, MFMA-heavy)
// and c3k2_fused<128, ..., 384> = pan_c3k2_2 (100 workgroups, bound by its per-CU weight stream) are independent and
// complementary: the block kernel's workgroups take the first block ids (it is on the critical path), the head's
// follow; both bodies are the stand-alone kernels' (block_kernels.h), so results are bit-identical. Bounded to 4 waves
// per SIMD so that two 512-thread workgroups share a CU.
#include "block_kernels.h"

namespace unina {

using namespace dev;

extern __shared__ __align__(16) unsigned char bd_smem[];

__global__ __launch_bounds__(512, 4) void block_dual_c3k2_128x384_head64(const C3k2Params pc, const HeadParams ph, int nc) {
  if ((int)blockIdx.x < nc) c3k2_fused_body<128, 4, 4, 1, 384, 8, 8, 0>(pc, (int)blockIdx.x, bd_smem);
  else head_fused_body<64, 8, 16, 8, 4>(ph, (int)blockIdx.x - nc, bd_smem);
}

// INT8 engines: the same pair with pan_c3k2_2 as an int8 block (the P2 head stays fp16: train.py:779 carve-out)
__global__ __launch_bounds__(512, 4) void block_dual_c3k2i8_128x384_head64(const C3k2Params pc, const HeadParams ph, int nc) {
  if ((int)blockIdx.x < nc) c3k2_fused_body<128, 4, 4, 1, 384, 8, 8, 0, EltI8>(pc, (int)blockIdx.x, bd_smem);
  else head_fused_body<64, 8, 16, 8, 4>(ph, (int)blockIdx.x - nc, bd_smem);
}

// ... with neck.down2 (3x3/s2, the first 128 channels of the block's input) as the block's first step
__global__ __launch_bounds__(512, 4) void block_dual_s2c3k2_128x384_head64(const C3k2Params pc, const HeadParams ph, int nc) {
  if ((int)blockIdx.x < nc) c3k2_fused_body<128, 4, 4, 1, 384, 8, 8, 0, EltH, 128, 128>(pc, (int)blockIdx.x, bd_smem);
  else head_fused_body<64, 8, 16, 8, 4>(ph, (int)blockIdx.x - nc, bd_smem);
}
__global__ __launch_bounds__(512, 4) void block_dual_s2c3k2i8_128x384_head64(const C3k2Params pc, const HeadParams ph, int nc) {
  if ((int)blockIdx.x < nc) c3k2_fused_body<128, 4, 4, 1, 384, 8, 8, 0, EltI8, 128, 128>(pc, (int)blockIdx.x, bd_smem);
  else head_fused_body<64, 8, 16, 8, 4>(ph, (int)blockIdx.x - nc, bd_smem);
}

// The head as the row-streaming / weights-stationary body (head_ws_body: kHeadWsTH x 14 output pixels per workgroup, 156 workgroups at
// 640^2) next to the block's 100: ONE 512-thread workgroup per CU (234 VGPRs), so the block's weight queue can be as deep as
// the stand-alone kernel's. 30.5 -> 23.5 us for the launch, -7 us serial latency, frames/s unchanged (same-box A/B).
#define BLOCK_DUAL_WS(NAME, ...)                                                                                     \
  __global__ __launch_bounds__(512) void NAME(const C3k2Params pc, const HeadParams ph, int nc) {                    \
    if ((int)blockIdx.x < nc) c3k2_fused_body<__VA_ARGS__>(pc, (int)blockIdx.x, bd_smem);                           \
    else head_ws_body<64, kHeadWsTH, 8>(ph, (int)blockIdx.x - nc, bd_smem);                                                 \
  }
BLOCK_DUAL_WS(block_dual_s2c3k2_128x384_head64ws, 128, 4, 4, 1, 384, 8, 16, 0, EltH, 128, 128)
BLOCK_DUAL_WS(block_dual_s2c3k2i8_128x384_head64ws, 128, 4, 4, 1, 384, 8, 8, 0, EltI8, 128, 128)
BLOCK_DUAL_WS(block_dual_c3k2_128x384_head64ws, 128, 4, 4, 1, 384, 8, 16, 0)
BLOCK_DUAL_WS(block_dual_c3k2i8_128x384_head64ws, 128, 4, 4, 1, 384, 8, 8, 0, EltI8)
#undef BLOCK_DUAL_WS

hipError_t block_dual_init() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(block_dual_c3k2_128x384_head64),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  for (const void* f : {reinterpret_cast<const void*>(block_dual_s2c3k2_128x384_head64),
                        reinterpret_cast<const void*>(block_dual_s2c3k2_128x384_head64ws), reinterpret_cast<const void*>(block_dual_s2c3k2i8_128x384_head64ws),
                        reinterpret_cast<const void*>(block_dual_c3k2_128x384_head64ws), reinterpret_cast<const void*>(block_dual_c3k2i8_128x384_head64ws),
                        reinterpret_cast<const void*>(block_dual_s2c3k2i8_128x384_head64)}) {
    e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
  }
  return hipFuncSetAttribute(reinterpret_cast<const void*>(block_dual_c3k2i8_128x384_head64),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

bool block_dual_match(const C3k2Params& pc, const HeadParams& ph) {
  if (head_tile_is(ph, kHeadWsTH, 14))   // the row-streaming head: next to either block form, fp16 or int8
    return (pc.dtype == kF16 || pc.dtype == kI8) && pc.hid == 128 && pc.nb == 1 && pc.Cin == 384 && pc.tail == 0 && ph.C == 64 &&
           (pc.cpre == 0 || (pc.cpre == 128 && pc.cx == 128)) && c3k2_tile_is(pc, 4, 4);
  const bool pre_ok = pc.cpre == 0 || (pc.cpre == 128 && pc.cx == 128 && head_tile_is(ph, 8, 16));
  return (pc.dtype == kF16 || pc.dtype == kI8) && pc.hid == 128 && pc.nb == 1 && pc.Cin == 384 && pc.tail == 0 && ph.C == 64 && pre_ok &&
         c3k2_tile_is(pc, 4, 4) && head_tile_is(ph, 8, 16);
}

const char* block_dual_name(int dtype, int cpre) {
  if (head_is_ws(64)) {
    if (cpre) return dtype == kI8 ? "block_dual_s2c3k2i8_128x384_head64ws<s2conv 128 + c3k2 i8,128,4x4,1,384 | head_ws 64,13x14>"
                                  : "block_dual_s2c3k2_128x384_head64ws<s2conv 128 + c3k2 128,4x4,1,384 | head_ws 64,13x14>";
    return dtype == kI8 ? "block_dual_c3k2i8_128x384_head64ws<c3k2 i8,128,4x4,1,384 | head_ws 64,13x14>"
                        : "block_dual_c3k2_128x384_head64ws<c3k2 128,4x4,1,384 | head_ws 64,13x14>";
  }
  if (cpre) return dtype == kI8 ? "block_dual_s2c3k2i8_128x384_head64<s2conv 128 + c3k2 i8,128,4x4,1,384 | head 64,8x16>"
                                : "block_dual_s2c3k2_128x384_head64<s2conv 128 + c3k2 128,4x4,1,384 | head 64,8x16>";
  return dtype == kI8 ? "block_dual_c3k2i8_128x384_head64<c3k2 i8,128,4x4,1,384 | head 64,8x16>"
                      : "block_dual_c3k2_128x384_head64<c3k2 128,4x4,1,384 | head 64,8x16>";
}

hipError_t block_dual_launch(const C3k2Params& pc, const HeadParams& ph, hipStream_t stream, int* grid_out) {
  const int nc = pc.tiles_x * pc.tiles_y, nh = ph.tiles_x * ph.tiles_y;
  const int smem = pc.smem_bytes > ph.smem_bytes ? pc.smem_bytes : ph.smem_bytes;
  if (grid_out) *grid_out = nc + nh;
  if (head_tile_is(ph, kHeadWsTH, 14)) {
    auto fn = pc.cpre ? (pc.dtype == kI8 ? block_dual_s2c3k2i8_128x384_head64ws : block_dual_s2c3k2_128x384_head64ws)
                      : (pc.dtype == kI8 ? block_dual_c3k2i8_128x384_head64ws : block_dual_c3k2_128x384_head64ws);
    hipLaunchKernelGGL(fn, dim3(nc + nh, 1, 1), dim3(512, 1, 1), smem, stream, pc, ph, nc);
  } else if (pc.cpre && pc.dtype == kI8)
    hipLaunchKernelGGL(block_dual_s2c3k2i8_128x384_head64, dim3(nc + nh, 1, 1), dim3(512, 1, 1), smem, stream, pc, ph, nc);
  else if (pc.cpre)
    hipLaunchKernelGGL(block_dual_s2c3k2_128x384_head64, dim3(nc + nh, 1, 1), dim3(512, 1, 1), smem, stream, pc, ph, nc);
  else if (pc.dtype == kI8)
    hipLaunchKernelGGL(block_dual_c3k2i8_128x384_head64, dim3(nc + nh, 1, 1), dim3(512, 1, 1), smem, stream, pc, ph, nc);
  else
    hipLaunchKernelGGL(block_dual_c3k2_128x384_head64, dim3(nc + nh, 1, 1), dim3(512, 1, 1), smem, stream, pc, ph, nc);
  return hipGetLastError();
}

}  // namespace unina
