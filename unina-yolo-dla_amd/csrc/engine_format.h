// engine_format.h -- on-disk layout of a .une engine file (written by unina-yolo-dla_amd/export.py).
//
// Plays the role of the reference's serialized TensorRT plan (export_trt.py:466-468, loaded at
// perception_node.cpp:230-249), but is an open, versioned table: header, activation-buffer table,
// fused-op table, then one blob of folded weights / biases. All little endian, packed.
#pragma once
#include <cstdint>

namespace unina {

constexpr char kMagic[8] = {'U', 'N', 'I', 'N', 'A', 'E', 'N', 'G'};
constexpr uint32_t kVersion = 3;  // 2: packed fragment-block weights; 3: buffer scales, per-channel multipliers, QUANT op (int8)

enum Precision : uint32_t { kFp16 = 0, kInt8 = 1, kFp32 = 2, kSplit16 = 3 };   // kSplit16: fp16 hi/lo pairs ("strict" mode)
enum BufDtype : uint32_t { kBufF16Nhwc = 0, kBufF32Planar = 1, kBufF32NchwInput = 2, kBufI8Nhwc = 3, kBufF32Nhwc = 4,
                           kBufS16Nhwc = 5 };   // kBufS16Nhwc: two fp16 NHWC planes, hi then lo (h*w*c*2 bytes each)
enum BufFlags : uint32_t { kBufInput = 1, kBufOutput = 2 };
enum OpKind : uint32_t { kOpConv = 1, kOpStem = 2, kOpSppfPool = 3, kOpUpsample = 4, kOpQuant = 5 };
enum SegFlags : uint32_t { kSegUp2 = 1, kSegPlanarF32 = 2 };

#pragma pack(push, 1)
struct FileHeader {          // 128 bytes
  char magic[8];
  uint32_t version;
  uint32_t precision;
  uint32_t in_c, in_h, in_w;
  uint32_t num_classes;
  uint32_t n_buffers, n_ops;
  uint32_t n_heads;
  uint32_t strides[3];
  uint64_t blob_bytes;
  uint64_t macs;
  uint8_t reserved[56];
};
static_assert(sizeof(FileHeader) == 128, "FileHeader");

struct BufferDesc {          // 64 bytes
  uint32_t h, w, c;
  uint32_t dtype;
  uint32_t flags;
  char name[40];
  float scale;               // int8 buffers: real value = code * scale (per-tensor symmetric); 1.0 otherwise
};
static_assert(sizeof(BufferDesc) == 64, "BufferDesc");

struct SegDesc {             // 64 bytes: one slice of an op's output-channel (N) dimension
  uint32_t src_coff;         // channel offset of this slice's INPUT inside src_buf (differs per slice for grouped convs)
  uint32_t n_count;          // real output channels
  uint32_t n_pad;            // rows stored in the weight matrix (n_count rounded up to 16, zero rows)
  uint32_t dst_buf, dst_coff;
  uint32_t flags;            // SegFlags
  uint64_t w_off;            // blob offset: conv: fp16 fragment blocks [n_pad/16][K/32][64][8], K = (kh,kw,cin) (export.py pack_weights);
                             //              kSplit16 engines: [n_pad/16][K/32][2 = hi, lo][64][8];
                             //              stem: fp32 [n][27] ordered (c,kh,kw)
  uint64_t b_off;            // blob offset: fp32 [n_pad] folded bias
  float w_scale, out_scale;  // int8 engines only (informational; the kernels use m_off / the buffer scale)
  uint64_t m_off;            // int8 convs: blob offset of fp32 [n_pad] multipliers s_in*s_w*bn_scale; 0 = none
  uint8_t reserved[8];
};
static_assert(sizeof(SegDesc) == 64, "SegDesc");

struct OpDesc {              // 256 bytes
  uint32_t kind;
  uint32_t ksize, stride, relu;
  uint32_t src_buf, cin;
  int32_t res_buf, res_coff; // residual added AFTER the ReLU (model.py:72-73); -1 = none
  uint32_t nseg;
  uint32_t in_h, in_w, out_h, out_w;
  float in_scale;            // int8 engines only
  SegDesc seg[2];
  char name[72];
};
static_assert(sizeof(OpDesc) == 256, "OpDesc");
#pragma pack(pop)

}  // namespace unina
