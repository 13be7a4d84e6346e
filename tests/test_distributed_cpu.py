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
