#!/usr/bin/env python3
"""Timeline of ONE serial frame from a rocprofv3 kernel trace: per kernel of the frame graph, median start offset from
the frame's first kernel, duration, and the idle gap to the previous kernel's end (dispatch + cache maintenance).
  run:     rocprofv3 --kernel-trace -d gpurun_out/tl -o tl --output-format csv -- python3 tools/frame_timeline.py run
  summary: python tools/frame_timeline.py summary gpurun_out/tl > profiles/rNN/vXX_frame_timeline.txt"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run():
    import torch
    import unina_yolo_dla_amd as u
    from unina_yolo_dla_amd.engine import Engine
    e = Engine.from_state_dict(u.synth.make_state_dict(7))
    xs = [torch.from_numpy(u.rng.frame(1234 + i, 640, 640)).cuda() for i in range(4)]
    for i in range(400):
        e.infer(xs[i % 4], 0.5, 0.45, 0.1)
    e.close()


def summary(d):
    f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)][0]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    names = [r[2] for r in rows]
    first = next(n for n in names if "stem" in n)
    frames, cur = [], None
    for r in rows:
        if r[2] == first:
            if cur: frames.append(cur)
            cur = []
        if cur is not None: cur.append(r)
    frames = [fr for fr in frames if "post_nms" in fr[-1][2]]
    n = max(set(len(fr) for fr in frames), key=[len(fr) for fr in frames].count)
    frames = [fr for fr in frames if len(fr) == n][50:]
    st = np.array([[k[0] - fr[0][0] for k in fr] for fr in frames]) * 1e-3
    du = np.array([[k[1] - k[0] for k in fr] for fr in frames]) * 1e-3
    gap = np.array([[0] + [fr[i][0] - fr[i - 1][1] for i in range(1, n)] for fr in frames]) * 1e-3
    print(f"{len(frames)} serial frames of {n} kernels; medians in us")
    print(f"{'#':>2} {'start':>8} {'dur':>7} {'gap':>6}  kernel")
    for i in range(n):
        print(f"{i:2d} {np.median(st[:, i]):8.2f} {np.median(du[:, i]):7.2f} {np.median(gap[:, i]):6.2f}  {frames[0][i][2][:110]}")
    span = np.array([fr[-1][1] - fr[0][0] for fr in frames]) * 1e-3
    print(f"first kernel start -> last kernel end: median {np.median(span):.2f} us; sum of durations {np.median(du.sum(1)):.2f}; sum of gaps {np.median(gap.sum(1)):.2f}")
    per = np.array([frames[i + 1][0][0] - frames[i][0][0] for i in range(len(frames) - 1)]) * 1e-3
    print(f"frame period (host loop incl. hand-off): median {np.median(per):.2f} us")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else summary(sys.argv[2])
