"""Test-only emulator of the ENGINE's arithmetic: executes the exporter's fused op table with torch CPU math,
rounding exactly where the HIP kernels do (folded fp16 weights, every NHWC buffer write; int8 engines: integer dot
products, per-channel float multiplier, round-half-even requantisation). It checks, independently of the kernels:
(1) the op table (fusions, concat slices, residual/upsample folding, int8 pass) computes graph (A); (2) the fp16
engine's deviation from the fp32 oracle is the fp16 FORMAT's rounding noise, not a kernel error; (3) the int8
engine's integer arithmetic bit for bit."""
import numpy as np
import torch
import torch.nn.functional as F

from unina_yolo_dla_amd import export


def split16(t):
    """fp32 tensor -> (hi, lo) fp16 pair as fp32 tensors: hi = fp16(t), lo = fp16(t - hi) (the SPLIT engine's storage format)."""
    hi = t.float().half().float()
    lo = (t.float() - hi).half().float()
    return hi, lo


def run_op_table(builder: "export.EngineBuilder", x: np.ndarray, fp16: bool = True, teacher: dict = None,
                 precise_w=None, precise_a=None, builder32: "export.EngineBuilder" = None, split: bool = False):
    """x: [1,3,H,W] fp32. Returns ({output name: [C,H,W] fp32}, {buffer name: [C,H,W] fp32}).
    int8 buffers are returned as their integer codes (multiply by the buffer scale to dequantise).

    teacher = {buffer name: stored values (codes for int8 buffers) read back from the ENGINE after a per-op forward}:
    every op then reads the engine's own buffers and writes into a separate set, so the returned buffers hold each
    op's emulated output GIVEN THE ENGINE'S INPUTS -- a per-op comparison that rounding flips cannot snowball through
    (deep in an int8 network a handful of +-1 input codes moves a third of the outputs by one code).

    Error-budget switches (tools/fp16_error_budget.py): ops whose index is in `precise_w` take their weights from
    `builder32` (an FP32 builder of the same state_dict: same op order, fp32 folded weights) instead of the fp16 blob;
    ops in `precise_a` store their output without the fp16 rounding. Everything else is unchanged, so the head error
    of such a run against the fp32 oracle is the contribution of the roundings left switched on.

    A SPLIT builder (or split=True with an FP32 builder): the SPLIT precision mode's arithmetic -- every folded weight and every stored
    activation is an fp16 pair hi + lo, a conv is the three fp16 products hi*hi + lo*hi + hi*lo accumulated in fp32
    (the lo*lo term is dropped), the stem and the epilogues are fp32."""
    precise_w = precise_w or ()
    precise_a = precise_a or ()
    blob32 = bytes(builder32.blob) if builder32 is not None else None
    cur_op = [0]
    prec = builder.precision
    split = split or prec == export.SPLIT
    wdt = {export.FP16: "<f2", export.FP32: "<f4", export.INT8: "<f2", export.SPLIT: "<f2"}[prec]
    blob = bytes(builder.blob)
    bdtype = [b[4] for b in builder.buffers]
    bscale = [b[6] for b in builder.buffers]
    bufs = {i: torch.zeros((c, h, w), dtype=torch.float64 if prec == export.INT8 else torch.float32)
            for i, (name, h, w, c, *_rest) in enumerate(builder.buffers)}
    img = next(i for i, b in enumerate(builder.buffers) if b[5] & export.BUF_INPUT)
    bufs[img] = torch.from_numpy(np.ascontiguousarray(x[0])).to(bufs[img].dtype)
    wr = bufs
    if teacher is not None:
        for i, b in enumerate(builder.buffers):
            if i != img and b[0] in teacher:
                bufs[i] = torch.from_numpy(np.ascontiguousarray(teacher[b[0]])).to(bufs[i].dtype)
        wr = {i: t.clone() for i, t in bufs.items()}

    def store(dst_buf, y):
        """round y (real values) the way a kernel writing into buffer `dst_buf` does"""
        d = bdtype[dst_buf]
        if d == export.BUF_I8:
            return torch.clamp(torch.round(y.float() * np.float32(1.0 / bscale[dst_buf])), -127, 127).to(y.dtype)
        if d == export.BUF_F16 and fp16 and cur_op[0] not in precise_a:
            return y.half().to(y.dtype)
        if split and d in (export.BUF_F32_NHWC, export.BUF_S16):
            hi, lo = split16(y)
            return (hi + lo).to(y.dtype)
        return y.float().to(y.dtype)

    def real(buf_idx, t):
        """real values held by (a slice of) buffer buf_idx"""
        return t * bscale[buf_idx] if bdtype[buf_idx] == export.BUF_I8 else t

    for oi, op in enumerate(builder.ops):
        src = bufs[op.src_buf]
        cur_op[0] = oi
        if op.kind == export.OP_STEM:
            s = op.segs[0]
            w = torch.from_numpy(np.frombuffer(blob, dtype="<f4", count=s.n_count * 27, offset=s.w_off).reshape(s.n_count, 3, 3, 3).copy())
            b = torch.from_numpy(np.frombuffer(blob, dtype="<f4", count=s.n_count, offset=s.b_off).copy())
            y = F.relu(F.conv2d(src[None].float(), w, b, stride=2, padding=1))[0].to(src.dtype)
            wr[s.dst.buf][s.dst.coff:s.dst.coff + s.n_count] = store(s.dst.buf, y)
        elif op.kind == export.OP_SPPF_POOL:
            s = op.segs[0]
            c = op.cin
            t = src[s.src_coff:s.src_coff + c][None]
            for i in range(1, 4):
                t = F.max_pool2d(t, 5, 1, 2)
                wr[op.src_buf][s.src_coff + i * c:s.src_coff + (i + 1) * c] = t[0]
        elif op.kind == export.OP_QUANT:
            s = op.segs[0]
            wr[s.dst.buf][s.dst.coff:s.dst.coff + s.n_count] = store(s.dst.buf, src[:s.n_count])
        elif op.kind == export.OP_CONV:
            k = op.k
            int8 = prec == export.INT8 and builder.op_int8[oi]
            for s in op.segs:
                K = k * k * op.cin
                xin = src[s.src_coff:s.src_coff + op.cin][None]
                b = torch.from_numpy(np.frombuffer(blob, dtype="<f4", count=s.n_count, offset=s.b_off).copy())
                if int8:
                    w = np.frombuffer(blob, dtype="i1", count=s.n_pad * K, offset=s.w_off)
                    w = export.unpack_weights(w, s.n_pad, K).reshape(s.n_pad, k, k, op.cin)[:s.n_count]
                    w = torch.from_numpy(w.astype(np.float64)).permute(0, 3, 1, 2).contiguous()
                    mult = torch.from_numpy(np.frombuffer(blob, dtype="<f4", count=s.n_count, offset=s.m_off).copy())
                    acc = F.conv2d(xin.double(), w, None, stride=op.s, padding=k // 2)[0]          # exact integers
                    # kernel: fmaf(acc, mult, bias) -> one fp32 rounding; the product is exact in fp64 (|acc| < 2^26)
                    y = (acc * mult.double()[:, None, None] + b.double()[:, None, None]).float().double()
                elif oi in precise_w:
                    s32 = builder32.ops[oi].segs[op.segs.index(s)]
                    w = np.frombuffer(blob32, dtype="<f4", count=s.n_pad * K, offset=s32.w_off)
                    w = export.unpack_weights(w, s.n_pad, K).reshape(s.n_pad, k, k, op.cin)[:s.n_count]
                    w = torch.from_numpy(w.astype(np.float32)).permute(0, 3, 1, 2).contiguous()
                    y = F.conv2d(real(op.src_buf, xin).float(), w, b, stride=op.s, padding=k // 2)[0].to(src.dtype)
                elif split:
                    if prec == export.SPLIT:
                        w = np.frombuffer(blob, dtype="<f2", count=2 * s.n_pad * K, offset=s.w_off)
                        w = export.unpack_weights_split(w, s.n_pad, K).reshape(s.n_pad, k, k, op.cin)[:s.n_count]
                    else:
                        w = np.frombuffer(blob, dtype="<f4", count=s.n_pad * K, offset=s.w_off)
                        w = export.unpack_weights(w, s.n_pad, K).reshape(s.n_pad, k, k, op.cin)[:s.n_count]
                    w = torch.from_numpy(w.astype(np.float32)).permute(0, 3, 1, 2).contiguous()
                    wh, wl = split16(w)
                    xh, xl = split16(xin)
                    y = (F.conv2d(xh, wl, None, stride=op.s, padding=k // 2) + F.conv2d(xl, wh, None, stride=op.s, padding=k // 2)
                         + F.conv2d(xh, wh, b, stride=op.s, padding=k // 2))[0].to(src.dtype)
                else:
                    w = np.frombuffer(blob, dtype=wdt, count=s.n_pad * K, offset=s.w_off)
                    w = export.unpack_weights(w, s.n_pad, K).reshape(s.n_pad, k, k, op.cin)[:s.n_count]
                    w = torch.from_numpy(w.astype(np.float32)).permute(0, 3, 1, 2).contiguous()
                    y = F.conv2d(real(op.src_buf, xin).float(), w, b, stride=op.s, padding=k // 2)[0].to(src.dtype)
                if op.relu:
                    y = F.relu(y)
                if op.res is not None:
                    r = bufs[op.res.buf][op.res.coff:op.res.coff + s.n_count]
                    if bdtype[op.res.buf] == export.BUF_I8:      # kernel: fmaf(code, s_res, y)
                        y = (y.double() + r.double() * float(np.float32(bscale[op.res.buf]))).float().to(y.dtype)
                    else:
                        y = (y.float() + r.float()).to(y.dtype)
                if s.flags & export.SEG_PLANAR_F32:
                    wr[s.dst.buf][s.dst.coff:s.dst.coff + s.n_count] = y.float().to(y.dtype)
                    continue
                y = store(s.dst.buf, y)
                if s.flags & export.SEG_UP2:
                    y = y.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
                wr[s.dst.buf][s.dst.coff:s.dst.coff + s.n_count] = y
        else:
            raise NotImplementedError(op.kind)
    named = {builder.buffers[i][0]: t.float().numpy() for i, t in wr.items()}
    outs = {n: named[n] for n in ("p2_cls", "p2_reg", "p3_cls", "p3_reg", "p4_cls", "p4_reg")}
    return outs, named


def dequantised(builder: "export.EngineBuilder", named: dict) -> dict:
    """{buffer name: real-valued ndarray} (int8 buffers multiplied by their scale)."""
    out = {}
    for name, h, w, c, dtype, flags, scale in builder.buffers:
        out[name] = named[name] * scale if dtype == export.BUF_I8 else named[name]
    return out


def engine_buffers(builder: "export.EngineBuilder", read_buffer) -> dict:
    """{buffer name: stored values} of an engine after a per-op (unfused) forward, int8 buffers as codes: the `teacher`
    of run_op_table. read_buffer(name) returns real values (code * scale), as Engine.read_buffer does."""
    out = {}
    for name, h, w, c, dtype, flags, scale in builder.buffers:
        if dtype in (export.BUF_F16, export.BUF_I8) and not (flags & export.BUF_INPUT):
            v = read_buffer(name)
            out[name] = np.rint(v / np.float32(scale)) if dtype == export.BUF_I8 else v
    return out


def per_op_mismatch(builder: "export.EngineBuilder", teacher: dict, named: dict) -> dict:
    """Per activation buffer: (fraction of int8 codes that differ | fraction of fp16 values off by more than 2 fp16
    ulp, worst difference in codes | in units of the 2-ulp tolerance) between the engine and the teacher-forced emulation."""
    out = {}
    for name, h, w, c, dtype, flags, scale in builder.buffers:
        if name not in teacher:
            continue
        a, b = np.asarray(teacher[name], np.float64), np.asarray(named[name], np.float64)
        if dtype == export.BUF_I8:
            d = np.abs(a - b)
            out[name] = (float((d > 0.5).mean()), float(d.max()))
        else:
            tol = 2.0 ** -9 * np.maximum(np.abs(a), np.abs(b)) + 1e-4
            d = np.abs(a - b) / tol
            out[name] = (float((d > 1.0).mean()), float(d.max()))
    return out
