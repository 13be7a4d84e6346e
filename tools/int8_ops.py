#!/usr/bin/env python3
"""In-sequence per-op table of the INT8 engine (mse calibration), like tools/profile_ops.py for fp16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import unina_yolo_dla_amd as u
from unina_yolo_dla_amd import export
from unina_yolo_dla_amd.engine import Engine, calibrate_amax
g = u.graph.Graph()
sd = u.synth.make_state_dict(7, g)
amax = calibrate_amax(sd, g, [u.rng.frame(5000 + i, 640, 640) for i in range(8)], method="mse")
e = Engine.from_state_dict(sd, g, precision=export.INT8, amax=amax)
x = torch.from_numpy(u.rng.frame(1234, 640, 640)).cuda()
e.forward(x)
ops = e.profile_ops(20)
tot = 0
for i, o in enumerate(ops):
    if o["ms"] > 0:
        print(f"{i:3d} {o['ms']*1e3:7.2f}  {o['kernel'][:70]:70s} {o['name'][:50]}")
        tot += o["ms"]
print("sum", tot * 1e3, "post", e.profile_post(20, 0.5, 0.45, 0.1))
e.close()
