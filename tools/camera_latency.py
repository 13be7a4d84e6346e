#!/usr/bin/env python3
"""Camera frame -> detections latency (perception_node.cpp:601-656): preprocess_bgra_resize + unina_infer against
unina_infer_bgra (pre-process inside the stem kernel). Usage on the GPU box: python tools/camera_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import unina_yolo_dla_amd as u
from unina_yolo_dla_amd.engine import Engine

e = Engine.from_state_dict(u.synth.make_state_dict(7))
L = e.L
rng = np.random.default_rng(3)
cams = [torch.from_numpy(rng.integers(0, 256, (720, 1280 * 4), dtype=np.uint8)).cuda() for _ in range(4)]
images = torch.empty((1, 3, 640, 640), dtype=torch.float32, device="cuda")
norm = L.create_norm_params_imagenet()
s = torch.cuda.current_stream().cuda_stream
e.autotune(images.normal_(), iters=5)


def two_step(cam):
    L.preprocess_bgra_resize(cam.data_ptr(), images.data_ptr(), 1280, 720, 1280 * 4, 640, 640, norm, s)
    return e.infer(images, 0.5, 0.45, 0.1)


def fused(cam):
    return e.infer_bgra(cam, 1280, 720, 1280 * 4, norm, 0.5, 0.45, 0.1)


for name, fn in (("preprocess_bgra_resize + unina_infer", two_step), ("unina_infer_bgra", fused)) * 2:
    lat = []
    for i in range(320):
        torch.cuda.synchronize()
        a = time.perf_counter()
        fn(cams[i % 4])
        if i >= 20:
            lat.append((time.perf_counter() - a) * 1e3)
    print(f"{name:40s} p50 {np.percentile(lat, 50):.4f} ms  p99 {np.percentile(lat, 99):.4f} ms")
e.close()
