#!/usr/bin/env python3
"""Per-op timing table of the forward (HIP events, eager launches) and, with --sweep, every valid conv tile
configuration per op. Usage on the GPU box:  python tools/profile_ops.py [--sweep] [--size 640] [--iters 30]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import unina_yolo_dla_amd as u
from unina_yolo_dla_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--sweep", action="store_true")
ap.add_argument("--size", type=int, default=640)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--autotune", action="store_true")
ap.add_argument("--precision", choices=["fp16", "fp32", "int8", "strict"], default="fp16")
a = ap.parse_args()

from unina_yolo_dla_amd import export
from unina_yolo_dla_amd.engine import calibrate_amax
g = u.graph.Graph(in_h=a.size, in_w=a.size)
sd = u.synth.make_state_dict(7, g)
prec = {"fp32": export.FP32, "int8": export.INT8, "strict": export.STRICT}.get(a.precision, export.FP16)
amax = calibrate_amax(sd, g, [u.rng.frame(5000 + i, a.size, a.size) for i in range(8)]) if a.precision == "int8" else None
e = Engine.from_state_dict(sd, g, precision=prec, amax=amax)
x = torch.from_numpy(u.rng.frame(1234, a.size, a.size)).cuda()
e.bind_images(x)
if a.autotune:
    e.autotune(x, iters=10)
ops = e.profile_ops(a.iters)
tot = sum(o["ms"] for o in ops)
print(f"{'#':>2} {'us':>7} {'TF/s':>6} {'GB/s':>6} {'grid':>5}  {'M':>6} {'N':>4} {'K':>5}  kernel / name")
for i, o in enumerate(ops):
    us = o["ms"] * 1e3
    ms_ = max(o["ms"], 1e-9)
    print(f"{i:2d} {us:7.2f} {o['flops'] / ms_ / 1e9:6.1f} {o['bytes'] / ms_ / 1e6:6.0f} {o['grid']:5d}  "
          f"{o['m']:6d} {o['n']:4d} {o['k']:5d}  {o['kernel']:34s} {o['name'][:60]}")
print(f"sum of ops: {tot * 1e3:.1f} us  ({g.macs() * 2 / tot / 1e9:.1f} TFLOP/s over the forward)")
pd, pn = e.profile_post(a.iters)
print(f"post-process: decode launch (+ folded head output convs) {pd * 1e3:.2f} us, pair tiles + scan + output {pn * 1e3:.2f} us")

if a.sweep:
    cfgs = e.conv_configs()
    print("\nsweep (us per config; * = best):")
    print(" " * 48 + " ".join(f"{c.split('<')[1].rstrip('>'):>16s}" for c in cfgs))
    best_total = 0.0
    choice = {}
    for i, o in enumerate(ops):
        if o["kind"] != 1:
            best_total += o["ms"]
            continue
        row = []
        for c in range(len(cfgs)):
            if not e.set_op_config(i, c):
                row.append(None)
                continue
            row.append(e.profile_ops(a.iters)[i]["ms"] * 1e3)
        e.set_op_config(i, -1)
        valid = [(t, c) for c, t in enumerate(row) if t is not None]
        tb, cb = min(valid)
        best_total += tb / 1e3
        choice[(o["m"], o["n"], o["k"])] = cfgs[cb]
        cells = " ".join(f"{'-':>16s}" if t is None else f"{t:15.2f}{'*' if c == cb else ' '}" for c, t in enumerate(row))
        print(f"{i:2d} M{o['m']:6d} N{o['n']:4d} K{o['k']:5d} {o['name'][:22]:22s} {cells}")
    print(f"sum with best configs: {best_total * 1e3:.1f} us")
e.close()
