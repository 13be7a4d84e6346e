#!/usr/bin/env python3
"""End-to-end A/B of tile configurations: serial latency (p50 of N frames, submit -> detections on host) and
2-in-flight throughput with the given per-op overrides on top of the autotuned choice.
Usage: python tools/try_cfg.py [--ops 46,47] [--match regq,halo]   (tries every valid config whose name matches)"""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import unina_yolo_dla_amd as u
from unina_yolo_dla_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--ops", default="46,47")
ap.add_argument("--match", default="regq,halo")
ap.add_argument("--frames", type=int, default=300)
a = ap.parse_args()
ops = [int(x) for x in a.ops.split(",")]
g = u.graph.Graph()
sd = u.synth.make_state_dict(7, g)
engs = [Engine.from_state_dict(sd, g) for _ in range(2)]
frames = [torch.from_numpy(u.rng.frame(1234 + i, 640, 640)).cuda() for i in range(8)]
for e in engs:
    e.autotune(frames[0], iters=5, cache="/tmp/try_cfg_tune.json")
names = engs[0].conv_configs()
streams = [torch.cuda.Stream() for _ in range(2)]


def measure():
    lat = []
    for i in range(20 + a.frames):
        torch.cuda.synchronize()
        t = time.perf_counter()
        engs[0].infer(frames[i % 8])
        if i >= 20:
            lat.append(time.perf_counter() - t)
    outs = [torch.zeros(8 + 8 * 1024, dtype=torch.int32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    n = 1500
    t = time.perf_counter()
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(streams[k]):
            engs[k].infer_async(frames[i % 8], out=outs[k], stream=streams[k])
    torch.cuda.synchronize()
    return 1e6 * float(np.median(lat)), n / (time.perf_counter() - t)


base = [o["kernel"] for o in engs[0].op_infos()]
print("autotuned:", {i: base[i] for i in ops}, "-> p50 %.1f us, %.0f fps" % measure())
for ci, nm in enumerate(names):
    if not any(m in nm for m in a.match.split(",")):
        continue
    ok = all(e.set_op_config(i, ci) for e in engs for i in ops)
    if ok:
        print("%-44s p50 %.1f us, %.0f fps" % ((nm,) + measure()))
    for e in engs:
        for i in ops:
            e.set_op_config(i, -1)
for e in engs:
    e.close()
