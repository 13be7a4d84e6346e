#!/usr/bin/env python3
"""In-kernel phase timing of the dual conv launches (P3 | P4 head layers) of the default fp16 engine: s_memtime stamps of
the mid workgroup of each conv inside ONE launch of the pair (unina_debug_dual_stamps), medians over repeated launches
with the frame's other ops replayed in front (cold weights, as in a real frame). Output -> profiles/rNN/*_dual_phases.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import unina_yolo_dla_amd as u
from unina_yolo_dla_amd.engine import Engine

from unina_yolo_dla_amd import export
STRICT = "--precision" in sys.argv and sys.argv[sys.argv.index("--precision") + 1] == "strict"
e = Engine.from_state_dict(u.synth.make_state_dict(7), precision=export.STRICT if STRICT else export.FP16)
x = torch.from_numpy(u.rng.frame(1234, 640, 640)).cuda()
e.forward(x)
ops = e.profile_ops(20)
print("phases of the mid workgroup of each conv of a dual launch; ticks = shader clock (s_memtime), GHz from the 100 MHz reference")
print(f"{'op':>3} {'kernel / conv':58s} {'us(op)':>7} | {'issue':>6} {'patch':>6} {'K loop':>7} {'epi':>6} {'total':>7} {'GHz':>5} {'us':>6}")
for i, o in enumerate(ops):
    if not o["kernel"].startswith("conv_dual_head3x3"):
        continue
    rows = []
    for rep in range(9):
        e.forward(x)                       # replay the frame: the pair meets cold weights / a just-written input
        st = e.dual_stamps(i)
        r = []
        for h in (0, 8):
            s = st[h:h + 8]
            ghz = (s[4] - s[0]) / max(1, s[6] - s[5]) * 0.1
            r.append([s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[4] - s[0], ghz, (s[6] - s[5]) * 0.01])
        rows.append(r)
    med = np.median(np.array(rows), axis=0)
    for h, tag in enumerate(("A (P3, 16x16 px x 64 ch, K 1152)", "B (P4, 8x16 px x 64 ch, K 2304)")):
        m = med[h]
        print(f"{i:3d} {o['name'][:26] + ' ' + tag:58s} {o['ms'] * 1e3:7.2f} | {m[0]:6.0f} {m[1]:6.0f} {m[2]:7.0f} {m[3]:6.0f} {m[4]:7.0f} {m[5]:5.2f} {m[6]:6.2f}")
e.close()
