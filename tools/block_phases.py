#!/usr/bin/env python3
"""Where a 40x40-level fused C3k2 block's time goes: per-step s_memtime stamps of its mid workgroup (unina_debug_block_stamps).
    python tools/block_phases.py            # stage3_c3k2 (+ sppf.cv1) and down2 + pan_c3k2_2 of the fp16 engine at 640^2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import unina_yolo_dla_amd as u
from unina_yolo_dla_amd.engine import Engine

NAMES = ["entry", "patch landed", "pre-conv", "cv1|cv2", "b0.cv1", "b0.cv2", "b1.cv1", "b1.cv2 / last", "cv3", "output stored", "tail conv", "drained"]
g = u.graph.Graph(in_h=640, in_w=640)
e = Engine.from_state_dict(u.synth.make_state_dict(7, g), g)
x = torch.from_numpy(u.rng.frame(1234, 640, 640)).cuda()
e.bind_images(x)
e.forward(x)
os.environ.setdefault("UNINA_NO_DUAL", "0")
for i, o in enumerate(e.op_infos()):
    if "c3k2" not in o["kernel"] and "block_dual" not in o["kernel"]:
        continue
    try:
        runs = np.array([e.block_stamps(i) for _ in range(7)], dtype=np.float64)
    except RuntimeError:
        continue
    st = np.median(runs, axis=0)
    clk = (st[11] - st[0]) / max((st[14] - st[15]) * 10.0, 1.0)     # shader ticks per ns (s_memtime runs at 100 MHz x ?)
    print(f"op {i}: {o['name']}  [{o['kernel']}]  mid workgroup, median of 7: {(st[14] - st[15]) * 0.01:.2f} us entry -> drained")
    prev = st[0]
    for k in range(1, 12):
        if st[k] == 0:
            continue
        print(f"   {NAMES[k]:16s} +{(st[k] - prev):9.0f} ticks   (t = {(st[k] - st[0]):9.0f})")
        prev = st[k]
    if st[12] and st[13]:     # sub-stamps of the b0.cv1 step (wave 0): its K loop done, its epilogue's LDS stores landed; the rest is the barrier
        print(f"   inside b0.cv1: K loop {st[12] - st[3]:.0f}, epilogue {st[13] - st[12]:.0f}, barrier (waiting for the slowest wave) {st[4] - st[13]:.0f} ticks")
    print(f"   (ticks per us of wall clock: {(st[11] - st[0]) / max((st[14] - st[15]) * 0.01, 1e-9):.0f})")
e.close()
