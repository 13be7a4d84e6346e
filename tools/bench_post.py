#!/usr/bin/env python3
"""Times the fused post-process kernel alone (HIP events) for different candidate counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import unina_yolo_dla_amd as u
from unina_yolo_dla_amd.engine import Engine
e = Engine.from_state_dict(u.synth.make_state_dict(7))
x = torch.from_numpy(u.rng.frame(1234, 640, 640)).cuda()
e.forward(x)
for conf in (0.9999999, 0.9, 0.7, 0.5, 0.3, 0.1, 0.0):
    d = e.postprocess(conf, 0.45, 0.1)
    st = torch.cuda.Event(enable_timing=True); en = torch.cuda.Event(enable_timing=True)
    base = e._det_buf.data_ptr()
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        e.L.unina_postprocess_async(e.h, conf, 0.45, 0.1, base + 32, base, s)
    st.record()
    for _ in range(50):
        e.L.unina_postprocess_async(e.h, conf, 0.45, 0.1, base + 32, base, s)
    en.record(); torch.cuda.synchronize()
    line = f"conf {conf:<10} kept {len(d):5d}  {st.elapsed_time(en) / 50 * 1e3:8.2f} us"
    if os.environ.get("UNINA_POST_STAMPS"):
        import ctypes
        out = np.zeros(1024, dtype=e._det_buf.cpu().numpy().dtype)
        # run through unina_infer's device result (stamps live behind the records there)
        dets = e.infer(None, conf, 0.45, 0.1)
        stamps = e.debug_stamps()
        line += "  phases(us): " + " ".join(f"{(b - a) / 100.0:6.1f}" for a, b in zip(stamps[:-1], stamps[1:]))
    print(line)
e.close()
