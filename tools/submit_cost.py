import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unina_yolo_dla_amd as u
from unina_yolo_dla_amd.engine import Engine
g = u.graph.Graph()
e = Engine.from_state_dict(u.synth.make_state_dict(7, g), g)
xs = [torch.from_numpy(u.rng.frame(1234 + i, 640, 640)).cuda() for i in range(4)]
for i in range(30): e.infer(xs[i % 4], 0.5, 0.45, 0.1)
out = torch.zeros(8 + 8 * 1024, dtype=torch.int32, device='cuda')
s = torch.cuda.current_stream()
ts = []
for i in range(200):
    torch.cuda.synchronize()
    a = time.perf_counter()
    e.infer_async(xs[i % 4], 0.5, 0.45, 0.1, out=out, stream=s)
    b = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter()
    ts.append(((b - a) * 1e6, (c - a) * 1e6))
ts = np.array(ts[20:])
print("infer_async host call us: p50 %.1f p90 %.1f ; to completion (torch sync) p50 %.1f" % (np.percentile(ts[:,0],50), np.percentile(ts[:,0],90), np.percentile(ts[:,1],50)))
# same input pointer every frame
ts = []
for i in range(200):
    torch.cuda.synchronize()
    a = time.perf_counter()
    e.infer_async(xs[0], 0.5, 0.45, 0.1, out=out, stream=s)
    b = time.perf_counter()
    torch.cuda.synchronize()
    ts.append((b - a) * 1e6)
print("same pointer: host call us p50 %.1f" % np.percentile(ts[20:], 50))
e.close()
