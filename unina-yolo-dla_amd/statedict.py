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
