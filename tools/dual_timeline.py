#!/usr/bin/env python3
"""Dispatch ramp and tail of the dual conv launches (P3 | P4 head pairs): every workgroup's start / end on the 100 MHz wall
clock from ONE stamped launch (unina_debug_dual_timeline), after a replay of the frame (cold weights).
Output -> profiles/rNN/*_dual_timeline.txt"""
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
ops = e.op_infos()
for i, o in enumerate(ops):
    if not o["kernel"].startswith("conv_dual_head3x3"):
        continue
    runs = []
    for rep in range(7):
        e.forward(x)
        runs.append(e.dual_timeline(i).astype(np.float64) * 0.01)      # us
    t = runs[len(runs) // 2]
    before, after = t[-1]
    t = t[:-1]
    t0 = t[:, 0].min()
    print(f"   marker kernel before -> first workgroup start {t0 - before:.2f} us; last workgroup end -> marker kernel after {after - t[:, 1].max():.2f} us")
    st, en = t[:, 0] - t0, t[:, 1] - t0
    print(f"op {i} {o['kernel'][:60]}: {len(t)} workgroups; first start -> last end {en.max():.2f} us")
    q = lambda a: " ".join(f"{v:6.2f}" for v in np.percentile(a, [0, 10, 50, 90, 100]))
    print(f"   start  p0/10/50/90/100: {q(st)}")
    print(f"   end    p0/10/50/90/100: {q(en)}")
    print(f"   life   p0/10/50/90/100: {q(en - st)}")
    for lo in range(0, len(t), max(1, len(t) // 8)):
        hi = min(len(t), lo + max(1, len(t) // 8))
        print(f"   wg {lo:3d}-{hi - 1:3d}: start {st[lo:hi].mean():6.2f}  end {en[lo:hi].mean():6.2f}  life {np.mean(en[lo:hi] - st[lo:hi]):6.2f}")
e.close()
