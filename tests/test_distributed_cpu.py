"""N > 1 path on CPU: two gloo ranks shard frames round-robin, fill detection slots, all-gather them, and rank 0
reassembles the global frame order -- the same code bench.py runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_dets(frame: int):
    from unina_yolo_dla_amd import gather
    rng = np.random.default_rng(frame)
    n = int(rng.integers(0, 40))
    d = np.zeros(n, dtype=gather.DET_DTYPE)
    d["x1"], d["y1"] = rng.uniform(0, 600, n), rng.uniform(0, 600, n)
    d["x2"], d["y2"] = d["x1"] + 20, d["y1"] + 30
    d["confidence"] = np.sort(rng.uniform(0.5, 1, n))[::-1]
    d["class_id"] = rng.integers(0, 4, n)
    d["valid"] = 1
    return d


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unina_yolo_dla_amd import gather
    mine = gather.frames_of_rank(n_frames, rank, world)
    local = torch.from_numpy(np.stack([gather.pack_slot(_fake_dets(f)) for f in mine]))
    out = gather.gather_slots(local)
    if rank == 0:
        slots = gather.interleave(out)
        ok = True
        for f, s in enumerate(slots):
            got = gather.unpack_slot(s.numpy())
            ok = ok and got.tobytes() == _fake_dets(f).tobytes()
        q.put(ok and len(slots) == n_frames)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_restores_frame_order():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 8, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=60)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok


def _ring_worker(rank, world, port, n_local, k, banks, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unina_yolo_dla_amd import gather
    rt = gather.HostRuntime()
    local = torch.zeros((banks, k, gather.SLOT_WORDS), dtype=torch.int32)
    gathered = torch.zeros((banks, world, k, gather.SLOT_WORDS), dtype=torch.int32)
    seen = {}

    def on_gathered(first_local_frame, g):
        # g: [world, k, SLOT_WORDS]; local frame j of rank r is global frame j * world + r
        for r in range(world):
            for s in range(k):
                seen[(first_local_frame + s) * world + r] = gather.unpack_slot(g[r, s].numpy().copy()).tobytes()

    ring = gather.SlotRing(rt, local, gathered, k, on_gathered=on_gathered)

    def infer(i, stream_index, slot):
        slot.copy_(torch.from_numpy(gather.pack_slot(_fake_dets(i * world + rank))))

    streams = ["infer0", "infer1"]
    half = (n_local // 2 // k) * k
    gather.run_frames(half, 2, ring, infer, streams)                  # two calls: the ring continues across them
    gather.run_frames(n_local - half, 2, ring, infer, streams, start=half)
    assert ring.flush() is None                          # n_local is a multiple of k: no partial bank
    ok = len(seen) == n_local * world and all(v == _fake_dets(f).tobytes() for f, v in seen.items())
    ok = ok and ring.gathers_issued == n_local // k
    # who waited for what: the comm stream waits for frame events only; an inference stream waits for a collective only
    # to reuse a bank, and then for the gather issued (banks - 1) * k frames before the frame it is about to enqueue --
    # never for the gather of the bank just completed
    infer_waits = [(st, tag, why) for st, tag, why in rt.log if st != "comm"]
    comm_waits = [(st, tag, why) for st, tag, why in rt.log if st == "comm"]
    ok = ok and all(tag[0] == "frame" and why == "gather input" for _st, tag, why in comm_waits)
    ok = ok and len(comm_waits) == n_local
    ok = ok and all(tag[0] == "gather" and why[0] == "bank reuse" for _st, tag, why in infer_waits)
    ok = ok and len(infer_waits) == n_local - banks * k                # none while the ring fills for the first time
    # a run that is NOT a multiple of k: the partial bank is gathered by flush(), the next run continues at a bank boundary
    seen.clear()
    start = n_local
    gather.run_frames(k + 2, 2, ring, infer, streams, start=start)
    nxt = ring.flush()
    ok = ok and nxt == start + 2 * k
    got = {f: v for f, v in seen.items() if start * world <= f < (start + k + 2) * world}
    ok = ok and len(got) == (k + 2) * world and all(v == _fake_dets(f).tobytes() for f, v in got.items())
    q.put((rank, ok, len(infer_waits), len(comm_waits)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_slot_ring_overlaps_the_gather():
    """gather.SlotRing / run_frames (the loop bench.py runs per rank over RCCL) on two gloo ranks: 3 banks x 4 slots, 40
    frames per rank on two 'inference streams'; every gathered slot is the right frame's, every gather was issued on the
    comm stream behind its own bank's frames, and no inference stream waited on the collective of the bank it just
    completed (only on bank reuse, (banks-1)*k frames later)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ring_worker, args=(r, 2, port, 40, 4, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _r, ok, _a, _b in res), res


def test_slot_roundtrip_and_sharding(pkg):
    from unina_yolo_dla_amd import gather
    d = _fake_dets(3)
    assert gather.unpack_slot(gather.pack_slot(d)).tobytes() == d.tobytes()
    assert gather.unpack_slot(gather.pack_slot(d[:0])).size == 0
    assert gather.frames_of_rank(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((gather.frames_of_rank(10, r, 4) for r in range(4)), [])) == list(range(10))
    bad = gather.pack_slot(d)
    bad[0] = 5000
    with pytest.raises(ValueError):
        gather.unpack_slot(bad)
