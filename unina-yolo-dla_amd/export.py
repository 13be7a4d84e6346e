"""Engine builder: state_dict of graph (A)  ->  .une engine file for libunina_mi355.so.

Plays the role of the reference's ``export_trt.py::export_pipeline`` (export_trt.py:497-566:
ONNX export -> TensorRT build -> serialized plan) for the MI355X engine. Instead of an opaque plan
the output is an explicit table of FUSED ops over NHWC fp16 activation buffers:

* BatchNorm (eval) is folded into the conv:  W' = W*g/sqrt(var+eps),  b' = beta - mean*g/sqrt(var+eps)
  (model.py:41-50; eps = 1e-5); ReLU is an epilogue flag.
* every ``torch.cat`` (model.py:110,132,257,260,264,267) becomes a shared buffer: producers write
  channel slices, so concat costs nothing.
* C3k2 ``cv1``/``cv2`` (same input, model.py:107-108) and the head's ``cls_branch.0``/``reg_branch.0``
  (same input, model.py:303) are merged into one implicit GEMM with two output slices; the head's
  ``.1`` and ``.2`` layers run as one two-group launch.
* the Bottleneck residual (model.py:73) is added in the 3x3 conv's epilogue, after its ReLU.
* ``Upsample`` (model.py:145-147) is folded into the lateral 1x1 conv's store (each output pixel is
  written to its 2x2 block of the FPN concat buffer).
* the SPPF pool pyramid (model.py:129-131) is ONE op producing y1,y2,y3 into the 4-way concat buffer.

67 reference convs -> 52 launches + 1 pool.
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .graph import Graph, OUTPUT_NAMES, STRIDES

MAGIC = b"UNINAENG"
VERSION = 3
FP16, INT8, FP32, SPLIT = 0, 1, 2, 3
STRICT = SPLIT            # the precision mode that meets the north-star tolerance on every detection at fp16-MFMA speed
BUF_F16, BUF_F32_PLANAR, BUF_F32_NCHW_IN, BUF_I8, BUF_F32_NHWC, BUF_S16 = 0, 1, 2, 3, 4, 5
BUF_INPUT, BUF_OUTPUT = 1, 2
OP_CONV, OP_STEM, OP_SPPF_POOL, OP_UPSAMPLE, OP_QUANT = 1, 2, 3, 4, 5
SEG_UP2, SEG_PLANAR_F32 = 1, 2
BN_EPS = 1e-5

_HDR = struct.Struct("<8sII3II2II3IQQ56x")
_BUF = struct.Struct("<3III40sf")
_SEG = struct.Struct("<6IQQffQ8x")
_OP_HEAD = struct.Struct("<4I2I2iI4If")
assert _HDR.size == 128 and _BUF.size == 64 and _SEG.size == 64 and _OP_HEAD.size == 56


@dataclass
class View:
    buf: int
    coff: int
    c: int


@dataclass
class Seg:
    module: str               # reference module path whose weights this slice carries
    src_coff: int
    n_count: int
    dst: View
    flags: int = 0
    w_off: int = 0
    b_off: int = 0
    n_pad: int = 0
    # raw parameters (filled by the emitters, serialised by EngineBuilder._finalize)
    w_raw: Optional[np.ndarray] = None      # [n, K] fp64, K = (kh,kw,cin), UNFOLDED conv weight
    fold: Optional[np.ndarray] = None       # [n] fp64: gamma/sqrt(var+eps) (1 for convs without BN)
    bias: Optional[np.ndarray] = None       # [n] fp64: beta - mean*fold (or the conv bias)
    bn: bool = True
    m_off: int = 0                          # int8 convs: blob offset of the per-channel multiplier
    w_scale: float = 1.0


@dataclass
class Op:
    kind: int
    name: str
    src_buf: int
    cin: int
    k: int = 1
    s: int = 1
    relu: int = 1
    res: Optional[View] = None
    in_hw: Tuple[int, int] = (0, 0)
    out_hw: Tuple[int, int] = (0, 0)
    segs: List[Seg] = field(default_factory=list)


_SWZ_G = (0, 2, 3, 1)


def pack_weights(wk: np.ndarray) -> np.ndarray:
    """[n_pad][K] fp16 or fp32 (K = (kh,kw,cin), n_pad % 16 == 0) -> the LDS image the conv kernel DMA-loads:
    1-KiB blocks [n_pad/16][K/kb], each 64 slots x 16 bytes with slot(r, c) = 4*r + (c ^ G[r>>2]), G = (0,2,3,1)
    (conv_igemm.hip: conflict-free ds_read_b128 fragment reads). A 16-byte chunk is 8 halfs / 4 floats, so a block
    spans kb = 32 (fp16) or 16 (fp32) values of k."""
    n_pad, K = wk.shape
    ce = 16 // wk.dtype.itemsize
    kb = 4 * ce
    assert n_pad % 16 == 0 and K % kb == 0, (n_pad, K)
    blk = wk.reshape(n_pad // 16, 16, K // kb, 4, ce).transpose(0, 2, 1, 3, 4)       # [nsub][kblk][r][c][ce]
    out = np.empty_like(blk)
    for r in range(16):
        for c in range(4):
            out[:, :, r, c ^ _SWZ_G[r >> 2]] = blk[:, :, r, c]
    return np.ascontiguousarray(out).reshape(-1)


def pack_weights_split(hi: np.ndarray, lo: np.ndarray) -> np.ndarray:
    """SPLIT engines: two [n_pad][K] fp16 matrices (w = hi + lo) -> pairs of 1-KiB fragment blocks [n_pad/16][K/32][hi | lo],
    each block in pack_weights' image: the kernels fetch a k block's hi and lo fragments from adjacent blocks."""
    n_pad, K = hi.shape
    nblk = (n_pad // 16) * (K // 32)
    out = np.empty((nblk, 2, 512), dtype=np.float16)
    out[:, 0] = pack_weights(hi).reshape(nblk, 512)
    out[:, 1] = pack_weights(lo).reshape(nblk, 512)
    return out.reshape(-1)


def unpack_weights_split(packed: np.ndarray, n_pad: int, K: int) -> np.ndarray:
    """Inverse of pack_weights_split: the fp32 values hi + lo, [n_pad][K] (tests / emulator)."""
    blk = packed.reshape(-1, 2, 512)
    hi = unpack_weights(np.ascontiguousarray(blk[:, 0]).reshape(-1), n_pad, K)
    lo = unpack_weights(np.ascontiguousarray(blk[:, 1]).reshape(-1), n_pad, K)
    return hi.astype(np.float32) + lo.astype(np.float32)


def unpack_weights(packed: np.ndarray, n_pad: int, K: int) -> np.ndarray:
    """Inverse of pack_weights (tests / emulator)."""
    ce = 16 // packed.dtype.itemsize
    blk = packed.reshape(n_pad // 16, K // (4 * ce), 16, 4, ce)
    out = np.empty_like(blk)
    for r in range(16):
        for c in range(4):
            out[:, :, r, c] = blk[:, :, r, c ^ _SWZ_G[r >> 2]]
    return np.ascontiguousarray(out.transpose(0, 2, 1, 3, 4)).reshape(n_pad, K)


def bn_terms(sd: Dict[str, np.ndarray], module: str):
    """ConvBlock -> (W [O,C,k,k] fp64 unfolded, f [O] = gamma/sqrt(var+eps), b' [O] = beta - mean*f)."""
    w = sd[f"{module}.conv.weight"].astype(np.float64)
    g = sd[f"{module}.bn.weight"].astype(np.float64)
    b = sd[f"{module}.bn.bias"].astype(np.float64)
    m = sd[f"{module}.bn.running_mean"].astype(np.float64)
    v = sd[f"{module}.bn.running_var"].astype(np.float64)
    f = g / np.sqrt(v + BN_EPS)
    return w, f, b - m * f


def fold_bn(sd: Dict[str, np.ndarray], module: str):
    """ConvBlock -> (W' [O,C,k,k] fp64, b' [O] fp64)."""
    w, f, b = bn_terms(sd, module)
    return w * f[:, None, None, None], b


# FP16 carve-outs of the reference's QAT recipe (train.py:779, qat.py:700-753): module-path prefixes kept in float
INT8_CARVE_OUT = ("backbone.stem", "backbone.stage1_conv", "head_p2",      # graph (A) module paths
                  "stem", "stage1_conv")                                  # graph (B) (qat.py) module paths


def quantize_sym(x: np.ndarray, scale: float) -> np.ndarray:
    """Per-tensor symmetric int8: clamp(round_half_even(x / scale), -127, 127)   (qat.py:91-126: num_bits=8, axis=None)."""
    return np.clip(np.rint(x / scale), -127, 127).astype(np.int8)


class EngineBuilder:
    def __init__(self, sd: Dict[str, np.ndarray], graph: Optional[Graph] = None, precision: int = FP16,
                 amax: Optional[Dict[str, float]] = None, weight_amax: Optional[Dict[str, float]] = None):
        """precision: FP16 (fp16 weights + activations, v_mfma_f32_16x16x32_f16), FP32 (fp32 everywhere,
        v_mfma_f32_16x16x4_f32: meets the north-star tolerance outright at 1/16 of the fp16 matrix rate), SPLIT (= STRICT:
        every folded weight and every stored activation is an fp16 PAIR hi + lo -- ~22 mantissa bits -- and a conv is three
        fp16 MFMAs per k block, hi*hi + hi*lo + lo*hi, into one fp32 accumulator: the tolerance of FP32 at the fp16 rate), or INT8
        (per-tensor symmetric int8 activations AND weights as in qat.py:91-126, v_mfma_i32_16x16x64_i8, with the
        reference's FP16 carve-outs; needs `amax`: buffer name -> calibrated |activation| maximum, see calibrate() or,
        for a QAT checkpoint, amax_from_qat()). `weight_amax`: optional conv module -> weight range from a checkpoint's
        `_weight_quantizer._amax` (default: max |W| of the conv, which is what a max calibrator stores)."""
        if precision not in (FP16, FP32, INT8, SPLIT):
            raise NotImplementedError("precision must be FP16, FP32, INT8 or SPLIT")
        if precision == INT8 and amax is None:
            raise ValueError("INT8 needs calibrated activation ranges (amax): run export.calibrate() first")
        self.sd = sd
        self.precision = precision
        self.amax = amax
        self.weight_amax = weight_amax or {}
        self.wdtype = np.float32 if precision == FP32 else np.float16
        self.act_dtype = BUF_F32_NHWC if precision == FP32 else (BUF_S16 if precision == SPLIT else BUF_F16)
        self.g = graph or Graph()
        self.narrow_base_channels = 0
        if self.g.base_channels % 32:
            # the kernels' K block is 32 channels: a narrower model (model.py:331-333 base_channels=16) is embedded, exactly,
            # in the same topology at the next supported width (statedict.widen_state_dict: zero channels)
            from .statedict import widen_state_dict
            if self.g.base_channels <= 0:
                raise NotImplementedError("base_channels must be positive")
            self.narrow_base_channels = self.g.base_channels
            wide = self.g.base_channels * 32 // math.gcd(self.g.base_channels, 32)
            self.sd, self.g = widen_state_dict(sd, self.g, wide)
        self.buffers: List[list] = []   # [name, h, w, c, dtype, flags, scale]
        self.ops: List[Op] = []
        self.blob = bytearray()
        if self.g.variant == "A":
            self._lower()
        else:
            self._lower_b()
        if precision == INT8:
            self._quantize_pass()
        self._finalize()

    # ---- tables -------------------------------------------------------------------------------
    def buf(self, name: str, h: int, w: int, c: int, dtype: Optional[int] = None, flags: int = 0) -> int:
        self.buffers.append([name, h, w, c, self.act_dtype if dtype is None else dtype, flags, 1.0])
        return len(self.buffers) - 1

    def view(self, name: str, h: int, w: int, c: int) -> View:
        return View(self.buf(name, h, w, c), 0, c)

    def _hw(self, buf: int) -> Tuple[int, int]:
        return self.buffers[buf][1], self.buffers[buf][2]

    def _blob_add(self, arr: np.ndarray) -> int:
        pad = (-len(self.blob)) % 256
        self.blob += b"\0" * pad
        off = len(self.blob)
        self.blob += arr.tobytes()
        return off

    # ---- op emitters ---------------------------------------------------------------------------
    def conv(self, items, src_buf: int, cin: int, k: int, s: int = 1, relu: bool = True,
             res: Optional[View] = None, bn: bool = True):
        """items: list of (module, src_coff, dst View, flags). One launch, len(items) output slices."""
        ih, iw = self._hw(src_buf)
        p = k // 2
        oh, ow = (ih + 2 * p - k) // s + 1, (iw + 2 * p - k) // s + 1
        op = Op(OP_CONV, "+".join(m for m, *_ in items), src_buf, cin, k, s, int(relu), res, (ih, iw), (oh, ow))
        for module, src_coff, dst, flags in items:
            if bn:
                w, f, b = bn_terms(self.sd, module)
            else:
                w = self.sd[f"{module}.weight"].astype(np.float64)
                b = self.sd[f"{module}.bias"].astype(np.float64)
                f = np.ones_like(b)
            n = w.shape[0]
            assert w.shape[1] == cin and w.shape[2] == k, (module, w.shape, cin, k)
            dh, dw = self._hw(dst.buf)
            want = (2 * oh, 2 * ow) if flags & SEG_UP2 else (oh, ow)
            assert (dh, dw) == want, (module, (dh, dw), want)
            assert dst.c == n, (module, dst.c, n)
            seg = Seg(module, src_coff, n, dst, flags, n_pad=-(-n // 16) * 16)
            seg.w_raw = np.transpose(w, (0, 2, 3, 1)).reshape(n, k * k * cin)      # [O][(kh,kw,C)]
            seg.fold, seg.bias, seg.bn = f, b, bn
            op.segs.append(seg)
        self.ops.append(op)

    def c3k2(self, name: str, src: View, dst: View, n: int):
        """C3k2 (model.py:76-110) -> 2 + 2n launches."""
        h, w = self._hw(src.buf)
        cout = dst.c
        hid = cout // 2
        cat = self.buf(f"{name}.cat", h, w, 2 * hid)
        cur = self.view(f"{name}.cv1", h, w, hid)
        self.conv([(f"{name}.cv1", src.coff, cur, 0), (f"{name}.cv2", src.coff, View(cat, hid, hid), 0)],
                  src.buf, src.c, 1)
        for i in range(n):
            t = self.view(f"{name}.bottlenecks.{i}.cv1", h, w, hid)
            self.conv([(f"{name}.bottlenecks.{i}.cv1", cur.coff, t, 0)], cur.buf, hid, 1)
            out = View(cat, 0, hid) if i == n - 1 else self.view(f"{name}.bottlenecks.{i}", h, w, hid)
            self.conv([(f"{name}.bottlenecks.{i}.cv2", t.coff, out, 0)], t.buf, hid, 3, res=cur)
            cur = out
        self.conv([(f"{name}.cv3", 0, dst, 0)], cat, 2 * hid, 1)

    def head(self, name: str, feat: View, out_cls: int, out_reg: int):
        """DetectionHead (model.py:274-303; qat.py:411-440) -> 3 launches (both branches per launch)."""
        h, w = self._hw(feat.buf)
        c = feat.c
        h0 = self.buf(f"{name}.h0", h, w, 2 * c)
        h1 = self.buf(f"{name}.h1", h, w, 2 * c)
        cls, reg = (f"{name}.cls_branch", f"{name}.reg_branch") if self.g.variant == "A" else (f"{name}_cls", f"{name}_reg")
        self.conv([(f"{cls}.0", feat.coff, View(h0, 0, c), 0),
                   (f"{reg}.0", feat.coff, View(h0, c, c), 0)], feat.buf, c, 3)
        self.conv([(f"{cls}.1", 0, View(h1, 0, c), 0),
                   (f"{reg}.1", c, View(h1, c, c), 0)], h0, c, 3)
        self.conv([(f"{cls}.2", 0, View(out_cls, 0, self.g.num_classes), SEG_PLANAR_F32),
                   (f"{reg}.2", c, View(out_reg, 0, 4), SEG_PLANAR_F32)], h1, c, 1,
                  relu=False, bn=False)

    def sppf(self, name: str, src: View, dst: View):
        """SPPF_DLA (model.py:113-132; qat.py:328-345): cv1 -> ONE pool op (y1,y2,y3) -> cv2 over the 4-way concat."""
        h, w = self._hw(src.buf)
        hid = src.c // 2
        sp = self.buf(f"{name}.cat", h, w, 4 * hid)
        self.conv([(f"{name}.cv1", src.coff, View(sp, 0, hid), 0)], src.buf, src.c, 1)
        pool = Op(OP_SPPF_POOL, f"{name}.pool1+pool2+pool3", sp, hid, 5, 1, 0, None, (h, w), (h, w))
        pool.segs.append(Seg(f"{name}.pool", 0, 3 * hid, View(sp, hid, 3 * hid)))
        self.ops.append(pool)
        self.conv([(f"{name}.cv2", 0, dst, 0)], sp, 4 * hid, 1)

    def stem(self, name: str, images: int, dst: View):
        H, W = self._hw(images)
        w, b = fold_bn(self.sd, name)
        c1 = dst.c
        op = Op(OP_STEM, name, images, 3, 3, 2, 1, None, (H, W), (H // 2, W // 2))
        seg = Seg(name, 0, c1, dst, 0, n_pad=c1)
        seg.w_raw, seg.fold, seg.bias = w.reshape(c1, 27), np.ones(c1), b      # stem: folded fp32 [O][(c,kh,kw)]
        op.segs.append(seg)
        self.ops.append(op)

    # ---- graph (B): UNINA_YOLO_DLA_QAT.forward (qat.py:443-491) -------------------------------------------
    def _lower_b(self):
        g = self.g
        bc = g.base_channels
        c1, c2, c3, c4, c5 = bc, 2 * bc, 4 * bc, 8 * bc, 16 * bc
        H, W = g.in_h, g.in_w
        images = self.buf("images", H, W, 3, BUF_F32_NCHW_IN, BUF_INPUT)
        outs = {}
        for name, s in zip(OUTPUT_NAMES, (4, 4, 8, 8, 16, 16)):
            c = g.num_classes if name.endswith("cls") else 4
            outs[name] = self.buf(name, H // s, W // s, c, BUF_F32_PLANAR, BUF_OUTPUT)
        # concat buffers (orders: qat.py:464,467,470,473,476)
        fpn1 = self.buf("cat_fpn1", H // 16, W // 16, c4 + c4)     # [p5_up | p4]
        fpn2 = self.buf("cat_fpn2", H // 8, W // 8, c3 + c3)       # [p4_up | p3]
        fpn3 = self.buf("cat_fpn3", H // 4, W // 4, c2 + c2)       # [p3_up | p2]
        pan1 = self.buf("cat_pan1", H // 8, W // 8, c2 + c3)       # [p2_down | p3_fused]
        pan2 = self.buf("cat_pan2", H // 16, W // 16, c3 + c4)     # [p3_down | p4_fused]
        p2, p3, p4 = View(fpn3, c2, c2), View(fpn2, c3, c3), View(fpn1, c4, c4)
        p3_fused, p4_fused = View(pan1, c2, c3), View(pan2, c3, c4)

        stem = self.view("stem", H // 2, W // 2, c1)
        self.stem("stem", images, stem)
        s1 = self.view("stage1_conv", H // 4, W // 4, c2)
        self.conv([("stage1_conv", 0, s1, 0)], stem.buf, c1, 3, 2)
        self.c3k2("stage1_c3k2", s1, p2, 1)
        s2 = self.view("stage2_conv", H // 8, W // 8, c3)
        self.conv([("stage2_conv", p2.coff, s2, 0)], p2.buf, c2, 3, 2)
        self.c3k2("stage2_c3k2", s2, p3, 2)
        s3 = self.view("stage3_conv", H // 16, W // 16, c4)
        self.conv([("stage3_conv", p3.coff, s3, 0)], p3.buf, c3, 3, 2)
        self.c3k2("stage3_c3k2", s3, p4, 2)
        s4 = self.view("stage4_conv", H // 32, W // 32, c5)
        self.conv([("stage4_conv", p4.coff, s4, 0)], p4.buf, c4, 3, 2)
        p5 = self.view("stage4_sppf", H // 32, W // 32, c5)
        self.sppf("stage4_sppf", s4, p5)

        self.conv([("lateral_p4", 0, View(fpn1, 0, c4), SEG_UP2)], p5.buf, c5, 1)
        self.c3k2("fpn_c3k2_1", View(fpn1, 0, 2 * c4), p4_fused, 1)
        self.conv([("lateral_p3", p4_fused.coff, View(fpn2, 0, c3), SEG_UP2)], p4_fused.buf, c4, 1)
        self.c3k2("fpn_c3k2_2", View(fpn2, 0, 2 * c3), p3_fused, 1)
        self.conv([("lateral_p2", p3_fused.coff, View(fpn3, 0, c2), SEG_UP2)], p3_fused.buf, c3, 1)
        p2_fused = self.view("p2_fused", H // 4, W // 4, c2)
        self.c3k2("fpn_c3k2_3", View(fpn3, 0, 2 * c2), p2_fused, 1)
        self.conv([("down1", 0, View(pan1, 0, c2), 0)], p2_fused.buf, c2, 3, 2)
        p3_out = self.view("p3_out", H // 8, W // 8, c3)
        self.c3k2("pan_c3k2_1", View(pan1, 0, c2 + c3), p3_out, 1)
        self.conv([("down2", 0, View(pan2, 0, c3), 0)], p3_out.buf, c3, 3, 2)
        p4_out = self.view("p4_out", H // 16, W // 16, c4)
        self.c3k2("pan_c3k2_2", View(pan2, 0, c3 + c4), p4_out, 1)

        self.head("head_p2", p2_fused, outs["p2_cls"], outs["p2_reg"])
        self.head("head_p3", p3_out, outs["p3_cls"], outs["p3_reg"])
        self.head("head_p4", p4_out, outs["p4_cls"], outs["p4_reg"])

    # ---- the network ---------------------------------------------------------------------------
    def _lower(self):
        g = self.g
        bc = g.base_channels
        c1, c2, c3, c4 = bc, 2 * bc, 4 * bc, 8 * bc
        H, W = g.in_h, g.in_w
        images = self.buf("images", H, W, 3, BUF_F32_NCHW_IN, BUF_INPUT)
        outs = {}
        for name, s in zip(OUTPUT_NAMES, (4, 4, 8, 8, 16, 16)):
            c = g.num_classes if name.endswith("cls") else 4
            outs[name] = self.buf(name, H // s, W // s, c, BUF_F32_PLANAR, BUF_OUTPUT)

        # concat buffers of the neck (orders: model.py:257,260,264,267)
        fpn1 = self.buf("neck.cat_fpn1", H // 8, W // 8, c3 + c3)      # [p4_up | p3]
        fpn2 = self.buf("neck.cat_fpn2", H // 4, W // 4, c2 + c2)      # [p3_up | p2]
        pan1 = self.buf("neck.cat_pan1", H // 8, W // 8, c2 + c3)      # [p2_down | p3_fused]
        pan2 = self.buf("neck.cat_pan2", H // 16, W // 16, c3 + c4)    # [p3_down | p4 (pre-SPPF)]
        p2, p3, p4 = View(fpn2, c2, c2), View(fpn1, c3, c3), View(pan2, c3, c4)
        p3_fused = View(pan1, c2, c3)

        # Backbone (model.py:205-219)
        stem = self.view("backbone.stem", H // 2, W // 2, c1)
        w, b = fold_bn(self.sd, "backbone.stem")
        op = Op(OP_STEM, "backbone.stem", images, 3, 3, 2, 1, None, (H, W), (H // 2, W // 2))
        seg = Seg("backbone.stem", 0, c1, stem, 0, n_pad=c1)
        seg.w_raw, seg.fold, seg.bias = w.reshape(c1, 27), np.ones(c1), b      # stem: folded fp32 [O][(c,kh,kw)]
        op.segs.append(seg)
        self.ops.append(op)
        s1 = self.view("backbone.stage1_conv", H // 4, W // 4, c2)
        self.conv([("backbone.stage1_conv", 0, s1, 0)], stem.buf, c1, 3, 2)
        if g.lite_p2:
            self.conv([("backbone.stage1_block", 0, p2, 0)], s1.buf, c2, 3)
        else:
            self.c3k2("backbone.stage1_block", s1, p2, 1)
        s2 = self.view("backbone.stage2_conv", H // 8, W // 8, c3)
        self.conv([("backbone.stage2_conv", p2.coff, s2, 0)], p2.buf, c2, 3, 2)
        self.c3k2("backbone.stage2_c3k2", s2, p3, 2)
        s3 = self.view("backbone.stage3_conv", H // 16, W // 16, c4)
        self.conv([("backbone.stage3_conv", p3.coff, s3, 0)], p3.buf, c3, 3, 2)
        self.c3k2("backbone.stage3_c3k2", s3, p4, 2)
        # SPPF_DLA (model.py:113-132)
        hid = c4 // 2
        sp = self.buf("backbone.sppf.cat", H // 16, W // 16, 4 * hid)
        self.conv([("backbone.sppf.cv1", p4.coff, View(sp, 0, hid), 0)], p4.buf, c4, 1)
        pool = Op(OP_SPPF_POOL, "backbone.sppf.pool1+pool2+pool3", sp, hid, 5, 1, 0, None,
                  (H // 16, W // 16), (H // 16, W // 16))
        pool.segs.append(Seg("backbone.sppf.pool", 0, 3 * hid, View(sp, hid, 3 * hid)))
        self.ops.append(pool)
        p4_sppf = self.view("backbone.sppf", H // 16, W // 16, c4)
        self.conv([("backbone.sppf.cv2", 0, p4_sppf, 0)], sp, 4 * hid, 1)

        # Neck (model.py:252-269)
        self.conv([("neck.lateral_p3", 0, View(fpn1, 0, c3), SEG_UP2)], p4_sppf.buf, c4, 1)
        self.c3k2("neck.fpn_c3k2_1", View(fpn1, 0, 2 * c3), p3_fused, 1)
        self.conv([("neck.lateral_p2", p3_fused.coff, View(fpn2, 0, c2), SEG_UP2)], p3_fused.buf, c3, 1)
        p2_fused = self.view("p2_fused", H // 4, W // 4, c2)
        self.c3k2("neck.fpn_c3k2_2", View(fpn2, 0, 2 * c2), p2_fused, 1)
        self.conv([("neck.down1", 0, View(pan1, 0, c2), 0)], p2_fused.buf, c2, 3, 2)
        p3_out = self.view("p3_out", H // 8, W // 8, c3)
        self.c3k2("neck.pan_c3k2_1", View(pan1, 0, c2 + c3), p3_out, 1)
        self.conv([("neck.down2", 0, View(pan2, 0, c3), 0)], p3_out.buf, c3, 3, 2)
        p4_out = self.view("p4_out", H // 16, W // 16, c4)
        self.c3k2("neck.pan_c3k2_2", View(pan2, 0, c3 + c4), p4_out, 1)

        # Heads (model.py:361-365)
        self.head("head_p2", p2_fused, outs["p2_cls"], outs["p2_reg"])
        self.head("head_p3", p3_out, outs["p3_cls"], outs["p3_reg"])
        self.head("head_p4", p4_out, outs["p4_cls"], outs["p4_reg"])

    # ---- INT8 pass (generic over the op table) ----------------------------------------------------
    def _quantize_pass(self):
        """Decides, per op, int8 or fp16 execution and, per buffer, int8 or fp16 storage.

        * an op runs in int8 iff it is a BN conv of a module outside the reference's FP16 carve-outs
          (train.py:779) whose Cin is a multiple of the int8 MFMA block (64). The head's output convs are plain
          nn.Conv2d in the reference's QAT graph (qat.py:416,421: unquantised) -> fp16.
        * precision is a per-layer builder choice, as in the reference's TensorRT build (export_trt.py:432-441 sets
          the INT8 flag without precision constraints, so the builder keeps a layer in float where that is faster):
          a C3k2 block whose bottlenecks cannot be int8 (hidden width 32 < one int8 MFMA block) runs in fp16 as a
          whole -- alternating types inside it costs more than int8 saves and loses the one-launch block kernel --
          and so does a conv that shares a multi-writer concat buffer with fp16 readers (an int8 twin of a buffer
          that is written in several places would need several QUANT passes).
        * a buffer is int8 iff every op that reads it as an INPUT is int8 (the SPPF pool is type-agnostic); a
          buffer with mixed readers stays fp16 and gets an int8 twin filled by one QUANT op.
        * scales: activations s = amax/127 per BUFFER (a concat buffer has one scale: its consumer conv has one
          input quantizer, qat.py:245-248); weights s_w = max|W|/127 per conv on the UNFOLDED weight (BN stays a
          float per-channel multiplier, qat.py:225-260)."""
        def seg_q(seg):
            return seg.bn and not any(seg.module.startswith(c) for c in INT8_CARVE_OUT)
        self.op_int8 = [op.kind == OP_CONV and op.cin % 64 == 0 and all(seg_q(sg) for sg in op.segs) for op in self.ops]
        nbuf = len(self.buffers)
        readers = {b: [] for b in range(nbuf)}
        writers = {b: set() for b in range(nbuf)}
        for i, op in enumerate(self.ops):
            if op.kind == OP_CONV:
                readers[op.src_buf].append(i)
            for sg in op.segs:
                writers[sg.dst.buf].add(i)
        narrow = {sg.module.split(".bottlenecks.")[0] + "." for op in self.ops if op.kind == OP_CONV and op.cin % 64
                  for sg in op.segs if ".bottlenecks." in sg.module and seg_q(sg)}
        for i, op in enumerate(self.ops):
            if self.op_int8[i] and any(sg.module.startswith(n) for sg in op.segs for n in narrow):
                self.op_int8[i] = False
        changed = True
        while changed:
            changed = False
            for b in range(nbuf):
                qs = [self.op_int8[i] for i in readers[b]]
                if len(writers[b]) > 1 and any(qs) and not all(qs):
                    for i in readers[b]:
                        self.op_int8[i] = False
                    changed = True
        new_ops: List[Op] = []
        twin_of: Dict[int, int] = {}
        for b in range(nbuf):
            name, h, w, c, dtype, flags, _ = self.buffers[b]
            if dtype != BUF_F16 or not readers[b]:
                continue
            qs = [self.op_int8[i] for i in readers[b]]
            if all(qs):
                self.buffers[b][4] = BUF_I8
                self.buffers[b][6] = self._scale_of(name)
            elif any(qs):                                   # mixed readers: int8 twin
                t = self.buf(name + ".q8", h, w, c, BUF_I8)
                self.buffers[t][6] = self._scale_of(name)
                twin_of[b] = t
        # re-point int8 readers to the twins and schedule the QUANT ops after the last writer of the original
        for b, t in twin_of.items():
            for i in readers[b]:
                if self.op_int8[i]:
                    self.ops[i].src_buf = t
        last_writer = {}
        for i, op in enumerate(self.ops):
            for sg in op.segs:
                if sg.dst.buf in twin_of:
                    last_writer[sg.dst.buf] = i
        out_ops, out_q = [], []
        for i, op in enumerate(self.ops):
            out_ops.append(op)
            out_q.append(self.op_int8[i])
            for b, t in twin_of.items():
                if last_writer.get(b) == i:
                    name, h, w, c = self.buffers[b][:4]
                    q = Op(OP_QUANT, f"quant({name})", b, c, 1, 1, 0, None, (h, w), (h, w))
                    q.segs.append(Seg("quant", 0, c, View(t, 0, c)))
                    out_ops.append(q)
                    out_q.append(False)
        self.ops, self.op_int8 = out_ops, out_q

    def _scale_of(self, buffer_name: str) -> float:
        if buffer_name not in self.amax:
            raise KeyError(f"no calibrated range for buffer '{buffer_name}'")
        return max(float(self.amax[buffer_name]), 1e-8) / 127.0

    # ---- blob -----------------------------------------------------------------------------------
    def _finalize(self):
        for i, op in enumerate(self.ops):
            if op.kind == OP_STEM:
                sg = op.segs[0]
                sg.w_off = self._blob_add(sg.w_raw.astype(np.float32))
                sg.b_off = self._blob_add(sg.bias.astype(np.float32))
            if op.kind != OP_CONV:
                continue
            int8 = self.precision == INT8 and self.op_int8[i]
            in_scale = self.buffers[op.src_buf][6]
            for sg in op.segs:
                K = sg.w_raw.shape[1]
                bk = np.zeros((sg.n_pad,), dtype=np.float32)
                bk[:sg.n_count] = sg.bias.astype(np.float32)
                if int8:
                    sg.w_scale = max(float(self.weight_amax.get(sg.module, np.abs(sg.w_raw).max())), 1e-12) / 127.0
                    wk = np.zeros((sg.n_pad, K), dtype=np.int8)
                    wk[:sg.n_count] = quantize_sym(sg.w_raw, sg.w_scale)
                    mk = np.zeros((sg.n_pad,), dtype=np.float32)
                    mk[:sg.n_count] = (in_scale * sg.w_scale * sg.fold).astype(np.float32)
                    sg.w_off = self._blob_add(pack_weights(wk))
                    sg.m_off = self._blob_add(mk)
                elif self.precision == SPLIT:
                    wf = sg.w_raw * sg.fold[:, None]                                     # fp64 folded weights
                    hi = np.zeros((sg.n_pad, K), dtype=np.float16)
                    lo = np.zeros((sg.n_pad, K), dtype=np.float16)
                    hi[:sg.n_count] = wf.astype(np.float16)
                    lo[:sg.n_count] = (wf - hi[:sg.n_count].astype(np.float64)).astype(np.float16)
                    sg.w_off = self._blob_add(pack_weights_split(hi, lo))
                else:
                    wk = np.zeros((sg.n_pad, K), dtype=self.wdtype)
                    wk[:sg.n_count] = (sg.w_raw * sg.fold[:, None]).astype(self.wdtype)
                    sg.w_off = self._blob_add(pack_weights(wk))
                sg.b_off = self._blob_add(bk)

    # ---- serialisation --------------------------------------------------------------------------
    def tobytes(self) -> bytes:
        g = self.g
        out = bytearray()
        out += _HDR.pack(MAGIC, VERSION, self.precision, 3, g.in_h, g.in_w, g.num_classes, len(self.buffers), len(self.ops),
                         3, *STRIDES, len(self.blob), g.macs())
        for name, h, w, c, dtype, flags, scale in self.buffers:
            out += _BUF.pack(h, w, c, dtype, flags, name.encode()[:39], scale)
        for op in self.ops:
            rec = bytearray(_OP_HEAD.pack(op.kind, op.k, op.s, op.relu, op.src_buf, op.cin,
                                          op.res.buf if op.res else -1, op.res.coff if op.res else 0,
                                          len(op.segs), op.in_hw[0], op.in_hw[1], op.out_hw[0], op.out_hw[1],
                                          float(self.buffers[op.src_buf][6])))
            for i in range(2):
                if i < len(op.segs):
                    s = op.segs[i]
                    rec += _SEG.pack(s.src_coff, s.n_count, s.n_pad, s.dst.buf, s.dst.coff, s.flags,
                                     s.w_off, s.b_off, s.w_scale, float(self.buffers[s.dst.buf][6]), s.m_off)
                else:
                    rec += b"\0" * _SEG.size
            rec += op.name.encode()[:71].ljust(72, b"\0")
            assert len(rec) == 256
            out += rec
        out += self.blob
        return bytes(out)

    def save(self, path: str) -> None:
        with open(path, "wb") as f:
            f.write(self.tobytes())


class HistogramCalibrator:
    """Per-tensor |x| histogram + range selection, the role of pytorch-quantization's ``HistogramCalibrator`` that the
    reference configures for activations AND weights (qat.py:91-126: ``QuantDescriptor(num_bits=8, calib_method="histogram",
    axis=None)``; its docstring calls it "Histogram (Entropy) calibration"). That library is not available here and the
    reference holds no calibrated value, so this is the build's OWN restatement of the published algorithm (parity
    unpinned): 2048 bins of |x| whose range grows by whole bins when a later batch exceeds it; then one of
      * ``entropy``    -- the threshold (from bin 128 on) whose 128-level re-quantised distribution has the smallest
                          KL divergence from the clipped original (the TensorRT-style entropy calibration);
      * ``mse``        -- the threshold whose int8 fake-quantisation of the bin centres has the smallest count-weighted
                          squared error;
      * ``percentile`` -- the |x| below which `percentile` percent of the values lie."""

    def __init__(self, num_bins: int = 2048):
        self.num_bins = num_bins
        self.hist: Optional[np.ndarray] = None
        self.edges: Optional[np.ndarray] = None

    def collect(self, x: np.ndarray) -> None:
        a = np.abs(np.asarray(x, dtype=np.float32)).reshape(-1)
        amax = float(a.max()) if a.size else 0.0
        if self.hist is not None and self.edges[-1] <= 1e-6 and amax > self.edges[-1]:
            # everything seen so far was (numerically) zero -- a dead buffer on the first frames: its range (0, 1e-8) must not be
            # GROWN by whole bins (1e12 of them for an ordinary activation); start over from this batch, the zeros go to bin 0
            zeros = float(self.hist.sum())
            self.hist = None
        else:
            zeros = 0.0
        if self.hist is None:
            amax = max(amax, 1e-8)
            self.hist, self.edges = np.histogram(a, bins=self.num_bins, range=(0.0, amax))
            self.hist = self.hist.astype(np.float64)
            self.hist[0] += zeros
            return
        if amax > self.edges[-1]:                      # grow the range by whole bins of the same width
            width = self.edges[1] - self.edges[0]
            extra = int(np.ceil((amax - self.edges[-1]) / width))
            self.edges = np.concatenate([self.edges, self.edges[-1] + width * np.arange(1, extra + 1)])
            self.hist = np.concatenate([self.hist, np.zeros(extra)])
        h, _ = np.histogram(a, bins=self.edges)
        self.hist += h

    # -- range selection ------------------------------------------------------------------------------
    def amax(self, method: str = "entropy", percentile: float = 99.99, start_bin: int = 128, stride: int = 1) -> float:
        if self.hist is None:
            raise ValueError("no data collected")
        if method == "percentile":
            cdf = np.cumsum(self.hist) / self.hist.sum()
            idx = int(np.searchsorted(cdf, percentile / 100.0))
            return float(self.edges[min(idx + 1, len(self.edges) - 1)])
        if method == "mse":
            return self._mse(start_bin, stride)
        if method == "entropy":
            return self._entropy(start_bin, stride)
        raise ValueError(f"unknown calibration method {method!r}")

    def _mse(self, start_bin: int, stride: int) -> float:
        centers = (self.edges[1:] + self.edges[:-1]) / 2
        counts = self.hist
        best, arg = None, len(centers) - 1
        for i in range(min(start_bin, len(centers) - 1), len(centers), stride):
            amax = centers[i]
            q = np.clip(np.rint(centers * (127.0 / amax)), -127, 127) * (amax / 127.0)
            mse = float((((q - centers) ** 2) * counts).mean())
            if best is None or mse < best:
                best, arg = mse, i
        return float(centers[arg])

    def _entropy(self, start_bin: int, stride: int) -> float:
        bins = self.hist.copy()
        bins[0] = bins[1]
        nbins = 128                                                  # 1 << (num_bits - 1): positive int8 levels
        stop = len(bins)
        if stop <= start_bin:
            return float(self.edges[-1])
        csum = np.concatenate([[0.0], np.cumsum(bins)])
        total = csum[-1]
        best, arg = None, stop
        for i in range(start_bin, stop + 1, stride):
            # reference: the first i bins with everything beyond folded into the last one
            ref = bins[:i].copy()
            ref[-1] += total - csum[i]
            # candidate: the first i bins merged into 128 levels; a level's mass is spread evenly over its NON-EMPTY bins
            level = (np.arange(i) * nbins) // i
            nz = bins[:i] > 0
            mass = np.bincount(level, weights=bins[:i], minlength=nbins)
            cnt = np.bincount(level, weights=nz.astype(np.float64), minlength=nbins)
            per = np.divide(mass, cnt, out=np.zeros(nbins), where=cnt > 0)
            cand = np.where(nz, per[level], 0.0)
            p = ref / ref.sum()
            qsum = cand.sum()
            if qsum <= 0:
                continue
            q = cand / qsum
            m = p > 0
            if np.any(m & (q <= 0)):
                continue                                             # infinite divergence
            kl = float(np.sum(p[m] * np.log(p[m] / q[m])))
            if best is None or kl <= best:                           # ties: the LAST (largest) threshold, as the library's argmin over the reversed list
                best, arg = kl, i
        return float(self.edges[arg])


def calibrate(named_buffers_per_frame, percentile: Optional[float] = None, method: Optional[str] = None) -> Dict[str, float]:
    """The build's own activation calibrator (the reference's lives in NVIDIA pytorch-quantization, qat.py:129-220,
    which is not available: parity unpinned). Input: an iterable of {buffer name: ndarray} (one dict per
    calibration frame, e.g. Engine.read_buffer() of an fp16/fp32 engine, or tests/emulate.py on CPU). Output:
    {buffer name: amax}.
      method None / "max" : max |x| over all frames (with `percentile`: the max over frames of that per-frame percentile --
                            round 1's calibrator);
      "entropy" | "mse" | "percentile" : per-buffer |x| histogram over ALL frames (HistogramCalibrator above -- what the
                            reference configures, qat.py:91-126), then that range selection (`percentile` defaults to 99.99)."""
    if method in (None, "max"):
        amax: Dict[str, float] = {}
        for named in named_buffers_per_frame:
            for name, arr in named.items():
                a = np.abs(np.asarray(arr, dtype=np.float32))
                v = float(a.max()) if percentile is None else float(np.percentile(a, percentile))
                amax[name] = max(amax.get(name, 0.0), v)
        return amax
    cals: Dict[str, HistogramCalibrator] = {}
    for named in named_buffers_per_frame:
        for name, arr in named.items():
            cals.setdefault(name, HistogramCalibrator()).collect(arr)
    return {name: c.amax(method, 99.99 if percentile is None else percentile) for name, c in cals.items()}


def calibrate_all(named_buffers_per_frame, specs) -> Dict[str, Dict[str, float]]:
    """One pass over the calibration frames, several range selections: specs = {label: (method, percentile | None)}.
    Returns {label: {buffer name: amax}} (tools/int8_drift.py compares them)."""
    cals: Dict[str, HistogramCalibrator] = {}
    mx: Dict[str, float] = {}
    for named in named_buffers_per_frame:
        for name, arr in named.items():
            cals.setdefault(name, HistogramCalibrator()).collect(arr)
            mx[name] = max(mx.get(name, 0.0), float(np.abs(np.asarray(arr)).max()))
    out = {}
    for label, (method, pct) in specs.items():
        out[label] = dict(mx) if method == "max" else {n: c.amax(method, 99.99 if pct is None else pct) for n, c in cals.items()}
    return out


def export_engine(sd: Dict[str, np.ndarray], path: str, graph: Optional[Graph] = None,
                  precision: int = FP16, amax: Optional[Dict[str, float]] = None,
                  weight_amax: Optional[Dict[str, float]] = None) -> EngineBuilder:
    """state_dict (reference key names) -> engine file. Returns the builder (op table for inspection)."""
    b = EngineBuilder(sd, graph, precision, amax, weight_amax)
    b.save(path)
    return b


def amax_from_qat(sd: Dict[str, np.ndarray], graph: Graph, quant: dict,
                  fallback: Optional[Dict[str, float]] = None) -> Dict[str, float]:
    """Activation ranges for an INT8 engine taken from a QAT checkpoint's own quantizers (statedict.from_qat_checkpoint):
    the reference puts ONE input quantizer on every QuantConv2d (qat.py:109-124, 245-248), the engine keeps ONE scale per
    activation buffer (a concat buffer feeds one conv, sibling convs such as C3k2's cv1 / cv2 share their input), so a
    buffer takes the LARGEST range of the convs that read it (no clipping that the checkpoint's quantizers would not do).
    Buffers no checkpoint quantizer speaks for (e.g. a residual-only tensor) come from `fallback` (a calibrate() result)."""
    b = EngineBuilder(sd, graph)                         # fp16 table: which conv module reads which buffer
    out: Dict[str, float] = dict(fallback or {})
    seen = set()
    for op in b.ops:
        if op.kind != OP_CONV:
            continue
        name = b.buffers[op.src_buf][0]
        for sg in op.segs:
            a = quant.get("input_amax", {}).get(sg.module)
            if a is not None:
                out[name] = max(a, out[name]) if name in seen else a
                seen.add(name)
    return out


def export_qat_checkpoint(ck: Dict[str, np.ndarray], path: str, in_h: int = 640, in_w: int = 640, num_classes: int = 4,
                          base_channels: int = 32, fallback_amax: Optional[Dict[str, float]] = None) -> EngineBuilder:
    """A `UNINA_YOLO_DLA_QAT` checkpoint (qat.py key names, quantizer `_amax` entries included) -> INT8 engine file of
    graph (B), using the checkpoint's own activation / weight ranges; without quantizer entries (a float checkpoint of
    the QAT topology) -> fp16 engine."""
    from . import statedict
    weights, quant = statedict.from_qat_checkpoint(ck)
    g = Graph(num_classes=num_classes, base_channels=base_channels, in_h=in_h, in_w=in_w, variant="B")
    if not quant["input_amax"]:
        return export_engine(weights, path, g, FP16)
    amax = amax_from_qat(weights, g, quant, fallback_amax)
    return export_engine(weights, path, g, INT8, amax, quant["weight_amax"])


def read_engine_header(path: str) -> dict:
    with open(path, "rb") as f:
        raw = f.read(_HDR.size)
    (magic, version, precision, in_c, in_h, in_w, nc, n_buf, n_ops, n_heads, s0, s1, s2, blob, macs) = _HDR.unpack(raw)
    if magic != MAGIC:
        raise ValueError("not a UNINAENG file")
    return dict(version=version, precision=precision, in_c=in_c, in_h=in_h, in_w=in_w, num_classes=nc,
                n_buffers=n_buf, n_ops=n_ops, n_heads=n_heads, strides=(s0, s1, s2), blob_bytes=blob, macs=macs)
