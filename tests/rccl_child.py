"""Child process of tests/test_gpu_rccl.py: ONE rank of the multi-GPU loop over the real RCCL backend (world size 1 on the
one-GPU box): SlotRing + CudaRuntime + unina_infer_async, every gathered slot byte-equal to a serial unina_infer."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    import torch
    import torch.distributed as dist
    import unina_yolo_dla_amd as u
    from unina_yolo_dla_amd import gather
    from unina_yolo_dla_amd.engine import Engine, MAX_DETECTIONS

    n_frames, k, banks, in_flight = int(sys.argv[1]), 4, 3, 2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    world = dist.get_world_size()
    assert world == int(os.environ["WORLD_SIZE"]) and dist.get_backend() == "nccl"
    S = 320
    g = u.graph.Graph(in_h=S, in_w=S)
    sd = u.synth.make_state_dict(7, u.graph.Graph())
    engines = [Engine.from_state_dict(sd, g) for _ in range(in_flight)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(in_flight)]
    frames = [torch.from_numpy(u.rng.frame(1234 + i, S, S)).to(dev) for i in range(8)]
    # serial reference: unina_infer of every distinct frame
    want = [engines[0].infer(f, 0.4, 0.45, 0.1) for f in frames]
    assert sum(len(w) for w in want) > 50
    slot_words = 8 + 8 * MAX_DETECTIONS
    local = torch.zeros((banks, k, slot_words), dtype=torch.int32, device=dev)
    gathered = torch.zeros((banks, world, k, slot_words), dtype=torch.int32, device=dev)
    seen = {}

    def on_gathered(first, bank):                       # bank: [world, k, slot_words]
        host = bank.cpu().numpy()
        for r in range(world):
            for s in range(k):
                seen[(first + s) * world + r] = gather.unpack_slot(host[r, s]).tobytes()

    ring = gather.SlotRing(gather.CudaRuntime(dev), local, gathered, k, on_gathered=on_gathered)

    def infer_into(i, kk, slot):
        with torch.cuda.stream(streams[kk]):
            engines[kk].infer_async(frames[i % len(frames)], 0.4, 0.45, 0.1, out=slot, stream=streams[kk])

    gather.run_frames(n_frames, in_flight, ring, infer_into, streams)
    ring.flush()
    dist.barrier()
    torch.cuda.synchronize()
    assert len(seen) >= n_frames, (len(seen), n_frames)
    bad = [i for i in range(n_frames) if seen[i] != want[i % len(frames)].tobytes()]
    assert not bad, bad[:8]
    assert ring.gathers_issued == (n_frames + k - 1) // k
    for e in engines:
        e.close()
    dist.destroy_process_group()
    print(f"RCCL_OK frames={n_frames} gathers={ring.gathers_issued} world={world}")


if __name__ == "__main__":
    main()
