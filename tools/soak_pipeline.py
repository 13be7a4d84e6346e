#!/usr/bin/env python3
"""Concurrency soak: N frames through 2 handles on 2 streams (the bench's throughput path: rotating result slots, no sync
between frames), in rounds of 256; every slot must hold, byte for byte, what a serial unina_infer of that frame returns."""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import unina_yolo_dla_amd as u
from unina_yolo_dla_amd import export, gather
from unina_yolo_dla_amd.engine import Engine, calibrate_amax

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40000)
ap.add_argument("--precision", default="fp16")
a = ap.parse_args()
g = u.graph.Graph()
sd = u.synth.make_state_dict(7, g)
path = os.path.join(tempfile.mkdtemp(), "m.une")
if a.precision == "int8":
    amax = calibrate_amax(sd, g, [u.rng.frame(5000 + i, 640, 640) for i in range(8)])
    export.export_engine(sd, path, g, precision=export.INT8, amax=amax)
else:
    export.export_engine(sd, path, g)
engines = [Engine(path) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in engines]
frames = [torch.from_numpy(u.rng.frame(1234 + i, 640, 640)).cuda() for i in range(6)]
confs = (0.5, 0.45, 0.55, 0.6)
want = {(f, c): engines[0].infer(frames[f], c, 0.45, 0.1).tobytes() for f in range(len(frames)) for c in confs}
torch.cuda.synchronize()
R = 256
slots = torch.zeros((R, gather.SLOT_WORDS), dtype=torch.int32, device="cuda")
bad, done, t0 = 0, 0, time.perf_counter()
while done < a.n:
    for i in range(R):
        k = i % 2
        with torch.cuda.stream(streams[k]):
            engines[k].infer_async(frames[(done + i) % len(frames)], confs[(done + i) % len(confs)], 0.45, 0.1, out=slots[i], stream=streams[k])
    torch.cuda.synchronize()
    host = slots.cpu()
    for i in range(R):
        bad += Engine.unpack(host[i]).tobytes() != want[((done + i) % len(frames), confs[(done + i) % len(confs)])]
    done += R
print(f"{a.precision}: {done} pipelined frames on 2 handles, mismatches {bad}, {done / (time.perf_counter() - t0):.0f} frames/s incl. the checks")
for e in engines:
    e.close()
sys.exit(1 if bad else 0)
