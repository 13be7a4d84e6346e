"""UNSD: a minimal name -> fp32 tensor container.

Stands in for the reference's ``.pt`` checkpoints (train.py:648) without pickles:
the exporter, the CPU oracle (C) and the tests all read the same file.

Layout (little endian):  b"UNSD0001" | u32 count | count x { u16 name_len | name |
u32 ndim | u32 dims[ndim] | f32 data[prod(dims)] }
"""
from __future__ import annotations

import struct
from typing import Dict

import numpy as np

MAGIC = b"UNSD0001"


def save(path: str, tensors: Dict[str, np.ndarray]) -> None:
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(tensors)))
        for name, arr in tensors.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            nb = name.encode("utf-8")
            f.write(struct.pack("<H", len(nb)))
            f.write(nb)
            f.write(struct.pack("<I", a.ndim))
            f.write(struct.pack(f"<{a.ndim}I", *a.shape))
            f.write(a.tobytes())


def load(path: str) -> Dict[str, np.ndarray]:
    out: Dict[str, np.ndarray] = {}
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not a UNSD file")
        (count,) = struct.unpack("<I", f.read(4))
        for _ in range(count):
            (nl,) = struct.unpack("<H", f.read(2))
            name = f.read(nl).decode("utf-8")
            (nd,) = struct.unpack("<I", f.read(4))
            dims = struct.unpack(f"<{nd}I", f.read(4 * nd)) if nd else ()
            n = int(np.prod(dims)) if nd else 1
            out[name] = np.frombuffer(f.read(4 * n), dtype="<f4").reshape(dims).copy()
    return out


# ---- checkpoints of the reference's QAT model (graph (B), unina_yolo_dla/qat.py) ----------------------------
_QUANT_MARKS = ("._input_quantizer.", "._weight_quantizer.", "._output_quantizer.", ".residual_quantizer.")


def from_qat_checkpoint(ck: Dict[str, np.ndarray]):
    """{key: array} of a `UNINA_YOLO_DLA_QAT` state_dict (qat.py:350-491; loaded elsewhere with a non-executing loader,
    e.g. ``torch.load(p, weights_only=True)`` -> ``{k: v.numpy()}``)  ->  (weights, quant).

    `weights` keeps exactly the parameter keys graph (B) needs (conv / BN / head conv tensors, fp32): an optional
    ``module.`` prefix is dropped, BatchNorm ``num_batches_tracked`` and every quantizer entry are removed (the name
    rules of qat.py:554-563, 657-673: pytorch-quantization nests `_input_quantizer` / `_weight_quantizer` under the
    conv, `residual_quantizer` under the bottleneck). `quant` collects the calibrated ranges those entries carry:
    ``{"input_amax": {conv module: amax}, "weight_amax": {conv module: amax}, "residual_amax": {bottleneck: amax}}``
    (per-tensor: QuantDescriptor(axis=None), qat.py:109-124)."""
    weights: Dict[str, np.ndarray] = {}
    quant = {"input_amax": {}, "weight_amax": {}, "residual_amax": {}}
    for key, val in ck.items():
        k = key[7:] if key.startswith("module.") else key
        if k.endswith("num_batches_tracked"):
            continue
        if any(m in k + "." for m in _QUANT_MARKS):
            if k.endswith("._amax"):
                a = float(np.max(np.abs(np.asarray(val, dtype=np.float64))))
                if "._input_quantizer." in k:
                    quant["input_amax"][k.split(".conv._input_quantizer.")[0]] = a
                elif "._weight_quantizer." in k:
                    quant["weight_amax"][k.split(".conv._weight_quantizer.")[0]] = a
                elif ".residual_quantizer." in k:
                    quant["residual_amax"][k.split(".residual_quantizer.")[0]] = a
            continue
        weights[k] = np.asarray(val, dtype=np.float32)
    return weights, quant


def detect_variant(sd: Dict[str, np.ndarray]) -> str:
    """'A' for model.py key names (``backbone.stem.conv.weight``), 'B' for qat.py's (``stem.conv.weight``)."""
    if "backbone.stem.conv.weight" in sd:
        return "A"
    if "stem.conv.weight" in sd and "stage4_conv.conv.weight" in sd:
        return "B"
    raise ValueError("state_dict matches neither graph (A) (model.py) nor graph (B) (qat.py)")


def widen_state_dict(sd: Dict[str, np.ndarray], g, base_channels: int):
    """Embeds a narrow model (``g.base_channels`` not a multiple of the kernels' 32-channel K block, e.g. the reference's
    ``base_channels=16`` option, model.py:331-333) in the SAME topology at ``base_channels``: every ConvBlock keeps its
    real output channels first and gains zero ones (zero weights, BN = identity with zero shift -> ReLU(0) = 0), and
    every conv's input weights are scattered to where its producers' real channels sit (a concat of widened tensors
    has gaps). The widened network computes the narrow one's outputs exactly (only zeros are added to the sums).
    Returns (state_dict, graph) of the wide model."""
    from .graph import Graph
    big = Graph(num_classes=g.num_classes, base_channels=base_channels, lite_p2=g.lite_p2, in_h=g.in_h, in_w=g.in_w,
                variant=g.variant)
    if len(big.nodes) != len(g.nodes):
        raise ValueError("widening changes the topology")
    pos: Dict[int, np.ndarray] = {}

    def positions(i: int) -> np.ndarray:
        """positions of node i's (narrow) channels inside the wide graph's node i"""
        if i < 0:
            return np.arange(3)
        if i not in pos:
            n = g.nodes[i]
            if n.kind in ("conv", "convout"):
                pos[i] = np.arange(n.c)
            elif n.kind == "cat":
                parts, off = [], 0
                for s, sb in zip(n.srcs, big.nodes[i].srcs):
                    parts.append(positions(s) + off)
                    off += big.nodes[sb].c
                pos[i] = np.concatenate(parts)
            else:                                   # add / pool5 / up2: channel-wise
                pos[i] = positions(n.srcs[0])
        return pos[i]

    out: Dict[str, np.ndarray] = {}
    for n, nb in zip(g.nodes, big.nodes):
        if n.kind not in ("conv", "convout"):
            continue
        if nb.name != n.name or nb.kind != n.kind:
            raise ValueError("widening changes the topology")
        ip = positions(n.srcs[0])
        if n.kind == "conv":
            w = np.zeros((nb.c, nb.cin, n.k, n.k), np.float32)
            w[:n.c, ip] = sd[f"{n.name}.conv.weight"]
            out[f"{n.name}.conv.weight"] = w
            for key, fill in (("weight", 1.0), ("bias", 0.0), ("running_mean", 0.0), ("running_var", 1.0)):
                v = np.full((nb.c,), fill, np.float32)
                v[:n.c] = sd[f"{n.name}.bn.{key}"]
                out[f"{n.name}.bn.{key}"] = v
        else:
            w = np.zeros((n.c, nb.cin, 1, 1), np.float32)
            w[:, ip] = sd[f"{n.name}.weight"]
            out[f"{n.name}.weight"] = w
            out[f"{n.name}.bias"] = np.asarray(sd[f"{n.name}.bias"], np.float32).copy()
    return out, big
