"""Test-only emulator of the ENGINE's arithmetic: executes the exporter's fused op table with torch CPU fp32
math, rounding to fp16 exactly where the HIP kernels do (folded weights, every NHWC buffer write). It checks two
things independently of the kernels: (1) the op table (fusions, concat slices, residual/upsample folding) computes
graph (A); (2) the fp16 engine's deviation from the fp32 oracle is the fp16 FORMAT's rounding noise, not a kernel
error -- the GPU must match this emulator far more tightly than it matches fp32."""
import numpy as np
import torch
import torch.nn.functional as F

from unina_yolo_dla_amd import export


def run_op_table(builder: "export.EngineBuilder", x: np.ndarray, fp16: bool = True):
    """x: [1,3,H,W] fp32. Returns ({output name: [C,H,W] fp32}, {buffer name: [C,H,W] fp32})."""
    fp32_engine = builder.precision == export.FP32
    q = (lambda t: t.half().float()) if (fp16 and not fp32_engine) else (lambda t: t)
    wdt = "<f4" if fp32_engine else "<f2"
    blob = bytes(builder.blob)
    bufs = {}
    for i, (name, h, w, c, dtype, flags) in enumerate(builder.buffers):
        bufs[i] = torch.zeros((c, h, w), dtype=torch.float32)
    img = next(i for i, b in enumerate(builder.buffers) if b[5] & export.BUF_INPUT)
    bufs[img] = torch.from_numpy(np.ascontiguousarray(x[0]))
    for op in builder.ops:
        src = bufs[op.src_buf]
        if op.kind == export.OP_STEM:
            s = op.segs[0]
            w = torch.from_numpy(np.frombuffer(blob, dtype="<f4", count=s.n_count * 27, offset=s.w_off).reshape(s.n_count, 3, 3, 3).copy())
            b = torch.from_numpy(np.frombuffer(blob, dtype="<f4", count=s.n_count, offset=s.b_off).copy())
            y = F.relu(F.conv2d(src[None], w, b, stride=2, padding=1))[0]
            bufs[s.dst.buf][s.dst.coff:s.dst.coff + s.n_count] = q(y)
        elif op.kind == export.OP_SPPF_POOL:
            s = op.segs[0]
            c = op.cin
            t = src[s.src_coff:s.src_coff + c][None]
            for i in range(1, 4):
                t = F.max_pool2d(t, 5, 1, 2)
                src[s.src_coff + i * c:s.src_coff + (i + 1) * c] = t[0]
        elif op.kind == export.OP_CONV:
            k = op.k
            for s in op.segs:
                w = np.frombuffer(blob, dtype=wdt, count=s.n_pad * k * k * op.cin, offset=s.w_off)
                w = export.unpack_weights(w, s.n_pad, k * k * op.cin)
                w = torch.from_numpy(w.reshape(s.n_pad, k, k, op.cin)[:s.n_count].astype(np.float32)).permute(0, 3, 1, 2).contiguous()
                b = torch.from_numpy(np.frombuffer(blob, dtype="<f4", count=s.n_count, offset=s.b_off).copy())
                y = F.conv2d(src[s.src_coff:s.src_coff + op.cin][None], w, b, stride=op.s, padding=k // 2)[0]
                if op.relu:
                    y = F.relu(y)
                if op.res is not None:
                    y = y + bufs[op.res.buf][op.res.coff:op.res.coff + s.n_count]
                if s.flags & export.SEG_PLANAR_F32:
                    bufs[s.dst.buf][s.dst.coff:s.dst.coff + s.n_count] = y
                    continue
                y = q(y)
                if s.flags & export.SEG_UP2:
                    y = y.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
                bufs[s.dst.buf][s.dst.coff:s.dst.coff + s.n_count] = y
        else:
            raise NotImplementedError(op.kind)
    named = {builder.buffers[i][0]: t.numpy() for i, t in bufs.items()}
    outs = {n: named[n] for n in ("p2_cls", "p2_reg", "p3_cls", "p3_reg", "p4_cls", "p4_reg")}
    return outs, named
