#!/usr/bin/env python3
"""Phase timing of the post-process inside real frames (UNINA_POST_STAMPS=1: wall_clock64 stamps, 100 MHz): medians over
serial unina_infer calls. Output -> profiles/rNN/*_post_phases.txt"""
import os, sys
os.environ["UNINA_POST_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import unina_yolo_dla_amd as u
from unina_yolo_dla_amd.engine import Engine
e = Engine.from_state_dict(u.synth.make_state_dict(7))
frames = [torch.from_numpy(u.rng.frame(1234 + i, 640, 640)).cuda() for i in range(4)]
e.autotune(frames[0], iters=5)
order = [7, 0, 3, 1, 2, 4, 8, 9, 5, 6]
names = ["K1 start", "decode done(wg0)", "K2 start", "list length read(wg0)", "rows fetched(wg0)", "tiles done", "loaded", "lists", "scan", "output"]
for conf in (0.5, 0.3, 0.05):
    rows = []
    for it in range(40):
        d = e.infer(frames[it % 4], conf, 0.45, 0.1)
        st = e.debug_stamps()
        rows.append([(st[k] - st[7]) / 100.0 for k in order])
    med = np.median(np.array(rows[8:]), axis=0)
    print(f"conf {conf}: kept {len(d)}; us since launch 1 started: " + "  ".join(f"{n} {v:.1f}" for n, v in zip(names, med)))
    print("      rows per scan wave:", [st[10 + k] & 0xFFFFF for k in range(4)], " wave scan end (us after lists):", [((st[10 + k] >> 20) - (st[9] & 0xFFFFFFFF)) / 100.0 for k in range(4)])
e.close()
