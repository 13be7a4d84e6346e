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


# ---------------------------------------------------------------------------------------------------------------------
# The per-GPU frame loop with the gather OFF the critical path (dedicated comm stream, K frames per
# collective, overlapped with the next frames' inference). bench.py (RCCL) and tests/test_distributed_cpu.py (gloo) run
# the same code; only the runtime underneath differs.

class CudaRuntime:
    """Streams / events of the GPU box (torch.cuda; backend "nccl" == RCCL over xGMI)."""

    def __init__(self, device):
        import torch
        self.torch, self.device = torch, device
        self.comm = torch.cuda.Stream(device=device)

    def record(self, stream, tag):
        ev = self.torch.cuda.Event()
        ev.record(stream)
        return ev

    def wait(self, stream, ev, why):
        stream.wait_event(ev)

    def all_gather(self, out, local, group, after):
        """Issue the collective on the comm stream once `after` (events) have fired; returns its completion event."""
        import torch.distributed as dist
        for ev in after:
            self.comm.wait_event(ev)
        with self.torch.cuda.stream(self.comm):
            dist.all_gather_into_tensor(out.view(-1), local.reshape(-1), group=group)
            done = self.torch.cuda.Event()
            done.record(self.comm)
        return done

    def host_sync(self, ev):
        ev.synchronize()


class HostRuntime:
    """CPU stand-in (gloo): 'streams' are names, work runs where it is issued, the collective is asynchronous
    (async_op=True) and every wait is LOGGED as (waiting stream, what it waited for) so that a test can assert which
    stream ever waited on a collective."""

    class Ev:
        def __init__(self, tag, work=None):
            self.tag, self.work = tag, work

    def __init__(self):
        self.comm = "comm"
        self.log = []

    def record(self, stream, tag):
        return HostRuntime.Ev(tag)

    def wait(self, stream, ev, why):
        self.log.append((stream, ev.tag, why))
        if ev.work is not None:
            ev.work.wait()
            ev.work = None

    def all_gather(self, out, local, group, after):
        import torch.distributed as dist
        for ev in after:
            self.wait(self.comm, ev, "gather input")
        work = dist.all_gather_into_tensor(out.view(-1), local.reshape(-1), group=group, async_op=True)
        return HostRuntime.Ev(("gather",), work)

    def host_sync(self, ev):
        if ev.work is not None:
            ev.work.wait()
            ev.work = None


class SlotRing:
    """`banks` x `k` detection slots per rank. Frame i is written into bank (i // k) % banks, slot i % k, by whichever
    inference stream runs it. When the last frame of a bank has been enqueued, the bank's all-gather is issued on the
    runtime's COMM stream behind the completion events of exactly those k frames; inference goes straight on into the
    next bank. The only time an inference stream waits for a collective is when it is about to overwrite a slot of a
    bank whose previous gather (issued (banks-1)*k frames earlier) has not finished: with banks >= 2 that wait is
    normally already satisfied."""

    def __init__(self, runtime, local, gathered, k, group=None, on_gathered=None):
        """local: [banks, k, SLOT_WORDS] int32 tensor; gathered: [banks, world, k, SLOT_WORDS]."""
        assert local.shape[0] >= 2 and local.shape[1] == k and gathered.shape[0] == local.shape[0]
        self.rt, self.local, self.gathered, self.k, self.group = runtime, local, gathered, k, group
        self.banks = local.shape[0]
        self.on_gathered = on_gathered
        self.frame_events = [[] for _ in range(self.banks)]
        self.gather_done = [None] * self.banks
        self.pending = [None] * self.banks          # (first frame index of the bank) awaiting consumption
        self.gathers_issued = 0
        self._last_commit = -1

    def slot(self, i, stream):
        """Slot tensor for frame i (rank-local index); `stream` will write it."""
        b, s = (i // self.k) % self.banks, i % self.k
        ev = self.gather_done[b]
        if ev is not None:                              # the bank's previous gather still reads these slots
            self.rt.wait(stream, ev, ("bank reuse", b))
        return self.local[b, s]

    def commit(self, i, stream):
        """Frame i has been enqueued on `stream`. Issues the bank's gather when the bank is complete."""
        b, s = (i // self.k) % self.banks, i % self.k
        self._last_commit = i
        self.frame_events[b].append(self.rt.record(stream, ("frame", i)))
        if s == self.k - 1:
            self._consume(b)
            ev = self.rt.all_gather(self.gathered[b], self.local[b], self.group, self.frame_events[b])
            ev.tag = ("gather", b, i - self.k + 1)
            self.gather_done[b] = ev
            self.pending[b] = i - self.k + 1
            self.frame_events[b] = []
            self.gathers_issued += 1

    def _consume(self, b):
        """Hand a finished bank to the consumer (host side) before the bank is gathered into again."""
        if self.pending[b] is not None:
            self.rt.host_sync(self.gather_done[b])
            if self.on_gathered is not None:
                self.on_gathered(self.pending[b], self.gathered[b])
            self.pending[b] = None

    def flush(self):
        """Complete every outstanding gather and hand it over (end of a stream of frames). A bank that is only partly
        filled (the frame count was not a multiple of k) is gathered as it stands -- every rank has the same number of
        frames, so every rank takes part; the slots past the last frame hold older frames and are the consumer's to ignore.
        Returns the index the next frame must continue at (the start of the next bank)."""
        nxt = None
        for b in range(self.banks):
            if self.frame_events[b]:
                first = self._last_commit - len(self.frame_events[b]) + 1
                self._consume(b)
                ev = self.rt.all_gather(self.gathered[b], self.local[b], self.group, self.frame_events[b])
                ev.tag = ("gather", b, first)
                self.gather_done[b] = ev
                self.pending[b] = first
                self.frame_events[b] = []
                self.gathers_issued += 1
                nxt = (self._last_commit // self.k + 1) * self.k
        order = sorted((f, b) for b, f in enumerate(self.pending) if f is not None)
        for _f, b in order:
            self._consume(b)
        return nxt


def run_frames(n_frames, n_streams, ring, infer, streams, start=0):
    """The loop of one rank: frame i on stream i % n_streams, result into the ring, gathers overlapped.
    infer(i, stream_index, slot_tensor) enqueues frame i (unina_infer_async on the GPU box). `start`: index of the first
    frame (a multiple of the ring's k), so that consecutive calls continue the same ring."""
    assert start % ring.k == 0
    for i in range(start, start + n_frames):
        k = i % n_streams
        slot = ring.slot(i, streams[k])
        infer(i, k, slot)
        ring.commit(i, streams[k])
