#!/usr/bin/env python3
"""In one process: latency (serial) and throughput (IN_FLIGHT engines) for several graph-parallelism settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import unina_yolo_dla_amd as u
from unina_yolo_dla_amd import export
from unina_yolo_dla_amd.engine import Engine
sd = u.synth.make_state_dict(7)
path = "/tmp/e.une"; export.export_engine(sd, path)
frames = [torch.from_numpy(u.rng.frame(1234 + i, 640, 640)).cuda() for i in range(8)]
def measure(ns, inflight, rounds=3):
    os.environ["UNINA_STREAMS"] = str(ns)
    engs = [Engine(path) for _ in range(inflight)]
    strs = [torch.cuda.Stream() for _ in range(inflight)]
    for e in engs: e.autotune(frames[0], iters=5, cache="/tmp/tune.json")
    res = torch.zeros((16, 8200), dtype=torch.int32, device="cuda")
    def run(n):
        for i in range(n):
            k = i % inflight
            with torch.cuda.stream(strs[k]):
                engs[k].infer_async(frames[i % 8], 0.5, 0.45, 0.1, out=res[i % 16], stream=strs[k])
    run(200); torch.cuda.synchronize()
    fps = []
    for _ in range(rounds):
        t0 = time.perf_counter(); run(1000); torch.cuda.synchronize(); fps.append(1000 / (time.perf_counter() - t0))
    lat = []
    for i in range(220):
        torch.cuda.synchronize(); a = time.perf_counter(); engs[0].infer(frames[i % 8]); lat.append((time.perf_counter() - a) * 1e3)
    # device-only time of one forward graph + post-process (events), no host sync inside
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    engs[0].infer_async(frames[0]); torch.cuda.synchronize()
    st.record()
    for i in range(50): engs[0].infer_async(frames[i % 8])
    en.record(); torch.cuda.synchronize()
    for e in engs: e.close()
    return fps, float(np.percentile(lat[20:], 50)), float(np.percentile(lat[20:], 99)), st.elapsed_time(en) / 50
for inflight in (1, 2, 3):
    for ns in (1, 2, 3, 4):
        fps, p50, p99, dev = measure(ns, inflight)
        print(f"inflight {inflight} streams {ns}: fps {' '.join(f'{f:7.0f}' for f in fps)}  lat p50 {p50:.3f} p99 {p99:.3f} ms  back-to-back one engine {dev * 1e3:.1f} us/frame", flush=True)
