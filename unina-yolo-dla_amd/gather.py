"""Multi-GPU plumbing: frame sharding and the detection-slot gather.

The reference runs one camera stream on one device (perception_node.cpp:472,802); frames are independent
(SURVEY.md section 8e), so N GPUs = N replicas and frame i goes to rank i % N. The only exchange is the gather of the
fixed-size detection slots to every rank (rank 0 publishes): RCCL over xGMI on the GPU box (backend "nccl"),
gloo in the CPU tests. A slot is ``8 + 8*MAX_DETECTIONS`` int32 words: word 0 = count, words 8.. = 32-byte
GpuDetection records (exactly what unina_infer_async writes).
"""
from __future__ import annotations

from typing import List

import numpy as np

MAX_DETECTIONS = 1024
SLOT_WORDS = 8 + 8 * MAX_DETECTIONS
DET_DTYPE = np.dtype([("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("confidence", "<f4"),
                      ("class_id", "<i4"), ("valid", "<i4"), ("_pad", "<i4")])


def frames_of_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """Global frame indices processed by `rank` (round-robin: frame i -> rank i % world)."""
    return list(range(rank, n_frames, world))


def pack_slot(dets: np.ndarray) -> np.ndarray:
    """Structured detections -> one int32 slot (host-side twin of the device layout; tests / CPU plumbing)."""
    assert dets.dtype == DET_DTYPE and len(dets) <= MAX_DETECTIONS
    slot = np.zeros(SLOT_WORDS, dtype=np.int32)
    slot[0] = len(dets)
    slot[8:8 + 8 * len(dets)] = dets.view(np.int32).reshape(-1)
    return slot


def unpack_slot(slot: np.ndarray) -> np.ndarray:
    n = int(slot[0])
    if not 0 <= n <= MAX_DETECTIONS:
        raise ValueError(f"corrupt detection slot: count {n}")
    return np.ascontiguousarray(slot[8:8 + 8 * n]).view(DET_DTYPE).copy()


def gather_slots(local, out=None, group=None):
    """all-gather a [k, SLOT_WORDS] int32 tensor of this rank's last k frames into [world, k, SLOT_WORDS].
    Works on CPU tensors (gloo) and GPU tensors (nccl == RCCL). Returns `out`."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view(-1), local.reshape(-1), group=group)   # flat concat: accepted by gloo and nccl
    return out


def interleave(gathered) -> list:
    """[world, k, SLOT_WORDS] -> slots in global frame order (frame i was on rank i % world)."""
    world, k = gathered.shape[0], gathered.shape[1]
    return [gathered[i % world, i // world] for i in range(world * k)]
