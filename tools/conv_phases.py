#!/usr/bin/env python3
"""In-kernel phase timing of selected conv ops under selected configs (debug stamps of one mid-grid workgroup)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import unina_yolo_dla_amd as u
from unina_yolo_dla_amd.engine import Engine
e = Engine.from_state_dict(u.synth.make_state_dict(7))
x = torch.from_numpy(u.rng.frame(1234, 640, 640)).cuda()
e.forward(x)
cfgs = e.conv_configs()
ops = e.op_infos()
want_ops = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [9, 43, 46, 49, 8]
for i in want_ops:
    print(f"op {i}: {ops[i]['name'][:50]}  M{ops[i]['m']} N{ops[i]['n']} K{ops[i]['k']}")
    for c, name in enumerate(cfgs):
        if not e.set_op_config(i, c):
            continue
        ms = e.profile_ops(20)[i]["ms"]
        rows = []
        for _ in range(7):
            st = e.conv_stamps(i)
            rows.append([st[k + 1] - st[k] for k in range(4)] + [(st[4] - st[0]) / max(1, st[6] - st[5]) * 0.1, st[0] - st[7]])
        med = np.median(np.array(rows), axis=0)
        print(f"   {name:38s} {ms * 1e3:6.2f} us | ticks: issue {med[0]:6.0f}  first-data {med[1]:6.0f}  loop {med[2]:7.0f}  epilogue {med[3]:6.0f}  (total {med[:4].sum():7.0f})  clock {med[4]:.2f} GHz  setup {med[5]:5.0f}")
    e.set_op_config(i, -1)
e.close()
