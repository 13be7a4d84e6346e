import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import unina_yolo_dla_amd as u
from unina_yolo_dla_amd.engine import Engine
e = Engine.from_state_dict(u.synth.make_state_dict(7))
x = torch.from_numpy(u.rng.frame(1234, 640, 640)).cuda()
e.forward(x)
ops = e.op_infos()
i = [k for k, o in enumerate(ops) if o["kernel"].startswith("conv_dual_head3x3")][0]
def phases(st, h):
    s = st[h:h + 8]
    return [s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3]]
cold, warm = [], []
for rep in range(9):
    e.forward(x)
    cold.append(phases(e.dual_stamps(i), 0))
    warm.append(phases(e.dual_stamps(i), 0))      # immediately again: same code, same data
print("cold (after a frame):", np.median(np.array(cold), axis=0))
print("warm (back to back):  ", np.median(np.array(warm), axis=0))
e.close()
