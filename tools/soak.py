#!/usr/bin/env python3
"""Soak: N serial unina_infer calls over a ring of frames (fp16 or int8); every result must equal the first one of its
frame bit for bit, and no call may stall (the host waits on the completion word the post-process writes)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import unina_yolo_dla_amd as u
from unina_yolo_dla_amd import export
from unina_yolo_dla_amd.engine import Engine, calibrate_amax

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20000)
ap.add_argument("--precision", default="fp16")
a = ap.parse_args()
g = u.graph.Graph()
sd = u.synth.make_state_dict(7, g)
if a.precision == "int8":
    amax = calibrate_amax(sd, g, [u.rng.frame(5000 + i, 640, 640) for i in range(8)])
    e = Engine.from_state_dict(sd, g, precision=export.INT8, amax=amax)
elif a.precision == "strict":
    e = Engine.from_state_dict(sd, g, precision=export.STRICT)
else:
    e = Engine.from_state_dict(sd, g)
xs = [torch.from_numpy(u.rng.frame(1234 + i, 640, 640)).cuda() for i in range(8)]
ref = [e.infer(x, 0.5, 0.45, 0.1).tobytes() for x in xs]
lat = np.empty(a.n)
bad = 0
for i in range(a.n):
    t = time.perf_counter()
    d = e.infer(xs[i % 8], 0.5, 0.45, 0.1)
    lat[i] = time.perf_counter() - t
    bad += d.tobytes() != ref[i % 8]
print(f"{a.precision}: {a.n} frames, mismatches {bad}, latency ms p50 {np.percentile(lat, 50) * 1e3:.4f} p99 {np.percentile(lat, 99) * 1e3:.4f} "
      f"p99.9 {np.percentile(lat, 99.9) * 1e3:.4f} max {lat.max() * 1e3:.3f}")
e.close()
sys.exit(1 if bad else 0)
