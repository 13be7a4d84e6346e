"""fp16 error budget of the headline engine, on the CPU (no GPU needed).

Question: the fp16 engine misses the north-star tolerance (every detection IoU >= 0.999 and
|dscore| < 1e-3 against the fp32 forward) on a tail of detections -- WHICH fp16 roundings produce that tail, and what
is the cheapest set of layers to keep in a wider format that removes it?

Method: tests/emulate.py executes the exporter's op table with torch-CPU math and rounds exactly where the HIP
kernels round (fp16 folded weights, fp16 store of every activation buffer; the HIP engine sits at the same distance
from fp32 as this emulation, tests/test_gpu_parity.py::test_fp16_engine_matches_fp16_emulator). Its two switches
`precise_w` / `precise_a` turn the weight rounding / the output-store rounding of chosen ops off. With ALL roundings
off the table reproduces the fp32 oracle (that run is the reference here). Rounding errors of different sites are
independent, so head-error variances add: a run with ONLY group G's roundings on measures G's share.

    python tools/fp16_error_budget.py budget  [--size 640] [--seeds 3]      # per-group variance shares
    python tools/fp16_error_budget.py modes   [--size 640] [--seeds 10]     # detection tails of candidate mixed modes

Output of both goes to stdout as a table; profiles/r02/fp16_error_budget.txt is a committed run.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import unina_yolo_dla_amd as u                      # noqa: E402
from unina_yolo_dla_amd import export              # noqa: E402
from emulate import run_op_table                    # noqa: E402
from detcmp import iou_matrix                       # noqa: E402
from oracle import oracle                           # noqa: E402

NAMES = u.graph.OUTPUT_NAMES

GROUPS = [
    ("stem+stage1", ("backbone.stem", "backbone.stage1")),
    ("stage2", ("backbone.stage2",)),
    ("stage3", ("backbone.stage3",)),
    ("sppf+lat_p3", ("backbone.sppf", "neck.lateral_p3")),
    ("fpn_c3k2_1+lat_p2", ("neck.fpn_c3k2_1", "neck.lateral_p2")),
    ("fpn_c3k2_2", ("neck.fpn_c3k2_2",)),
    ("down1+pan_c3k2_1", ("neck.down1", "neck.pan_c3k2_1")),
    ("down2+pan_c3k2_2", ("neck.down2", "neck.pan_c3k2_2")),
    ("head_p2", ("head_p2",)),
    ("head_p3", ("head_p3",)),
    ("head_p4", ("head_p4",)),
]


def group_of(op):
    for gi, (_n, prefixes) in enumerate(GROUPS):
        if any(op.name.startswith(p) for p in prefixes):
            return gi
    raise KeyError(op.name)


def rms(a):
    return float(np.sqrt((np.asarray(a, np.float64) ** 2).mean()))


def det_stats(heads, ref_heads, conf=0.5, iou=0.45, q=0.1):
    got, _ = oracle.postprocess([heads[n] for n in NAMES], conf, iou, q)
    want, _ = oracle.postprocess([ref_heads[n] for n in NAMES], conf, iou, q)
    m = iou_matrix(got, want)
    m = np.where(got["class_id"][:, None] == want["class_id"][None, :], m, 0.0)
    j = m.argmax(1)
    best = m.max(1)
    ok = best > 0.9
    ds = np.abs(got["confidence"] - want["confidence"][j])[ok]
    return dict(n=len(want), matched=int(ok.sum()), min_iou=float(best[ok].min()), max_ds=float(ds.max()),
                p99_ds=float(np.percentile(ds, 99)), frac_bad_iou=float((best[ok] < 0.999).mean()),
                frac_bad_ds=float((ds >= 1e-3).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["budget", "modes"])
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--seeds", type=int, default=3)
    a = ap.parse_args()
    g = u.graph.Graph(in_h=a.size, in_w=a.size)
    sd = u.synth.make_state_dict(7, u.graph.Graph())
    b16 = export.EngineBuilder(sd, g)
    b32 = export.EngineBuilder(sd, g, export.FP32)
    nops = len(b16.ops)
    allops = set(range(nops))
    gid = [group_of(op) for op in b16.ops]
    members = [set(i for i in range(nops) if gid[i] == k) for k in range(len(GROUPS))]
    seeds = [1234 + i for i in range(a.seeds)]

    def run(x, pw, pa):
        return run_op_table(b16, x, fp16=True, precise_w=pw, precise_a=pa, builder32=b32)[0]

    if a.what == "budget":
        # variance share per (group, weights|activations), per head, averaged over seeds
        var = np.zeros((len(GROUPS), 2, 6))
        tot = np.zeros(6)
        for seed in seeds:
            x = u.rng.frame(seed, a.size, a.size)
            t0 = time.time()
            ref = run(x, allops, allops)
            full = run(x, (), ())
            tot += [rms(full[n] - ref[n]) ** 2 for n in NAMES]
            for k in range(len(GROUPS)):
                hw = run(x, allops - members[k], allops)          # only group k's WEIGHT rounding on
                ha = run(x, allops, allops - members[k])          # only group k's ACTIVATION-store rounding on
                var[k, 0] += [rms(hw[n] - ref[n]) ** 2 for n in NAMES]
                var[k, 1] += [rms(ha[n] - ref[n]) ** 2 for n in NAMES]
            print(f"# seed {seed}: {time.time() - t0:.0f} s", file=sys.stderr)
        var /= len(seeds)
        tot /= len(seeds)
        print(f"fp16 error budget, {a.size}x{a.size}, seeds {seeds}: share of each head's error VARIANCE (percent) that the "
              f"fp16 roundings of a layer group produce\n(w = folded weights stored in fp16, a = the group's output buffers "
              f"stored in fp16); last rows: rms error with every rounding on, and the sum of shares (100 = independent)")
        print(f"{'group':24s}" + "".join(f"{n:>16s}" for n in NAMES))
        for k, (name, _p) in enumerate(GROUPS):
            for j, tag in enumerate(("w", "a")):
                print(f"{name + ' ' + tag:24s}" + "".join(f"{100 * var[k, j, h] / tot[h]:16.1f}" for h in range(6)))
        print(f"{'rms(all on)':24s}" + "".join(f"{np.sqrt(tot[h]):16.2e}" for h in range(6)))
        print(f"{'sum of shares':24s}" + "".join(f"{100 * var[:, :, h].sum() / tot[h]:16.1f}" for h in range(6)))
        return

    # candidate mixed modes: sets of groups kept precise (weights and activations)
    G = {n: k for k, (n, _p) in enumerate(GROUPS)}
    def keep(*names):
        s = set()
        for n in names:
            s |= members[G[n]]
        return s
    heads3 = ("head_p2", "head_p3", "head_p4")
    neck = ("fpn_c3k2_1+lat_p2", "fpn_c3k2_2", "down1+pan_c3k2_1", "down2+pan_c3k2_2", "sppf+lat_p3")
    modes = [
        ("fp16 everywhere (headline)", set(), set()),
        ("heads precise", keep(*heads3), keep(*heads3)),
        ("heads + neck precise", keep(*heads3, *neck), keep(*heads3, *neck)),
        ("all weights precise", allops, set()),
        ("all activations precise", set(), allops),
        ("everything precise but stem+stage1", allops - keep("stem+stage1"), allops - keep("stem+stage1")),
    ]
    print(f"candidate mixed modes, {a.size}x{a.size}, seeds {seeds}, conf 0.5 / iou 0.45 / q 0.1: worst detection over all seeds")
    print(f"{'mode':38s}{'dets':>7s}{'min IoU':>10s}{'max|ds|':>10s}{'p99|ds|':>10s}{'IoU<.999':>10s}{'|ds|>=1e-3':>11s}"
          f"{'rms p3_cls':>11s}")
    modes.append(("SPLIT: fp16 hi+lo pairs, 3 MFMA terms", None, None))
    for name, pw, pa in modes:
        agg = dict(n=0, min_iou=1.0, max_ds=0.0, p99=[], bi=[], bd=[], r=[])
        for seed in seeds:
            x = u.rng.frame(seed, a.size, a.size)
            ref = run(x, allops, allops)
            h = run_op_table(b32, x, fp16=False, split=True)[0] if pw is None else run(x, pw, pa)
            s = det_stats(h, ref)
            agg["n"] += s["n"]
            agg["min_iou"] = min(agg["min_iou"], s["min_iou"])
            agg["max_ds"] = max(agg["max_ds"], s["max_ds"])
            agg["p99"].append(s["p99_ds"]); agg["bi"].append(s["frac_bad_iou"]); agg["bd"].append(s["frac_bad_ds"])
            agg["r"].append(rms(h["p3_cls"] - ref["p3_cls"]))
        print(f"{name:38s}{agg['n']:7d}{agg['min_iou']:10.5f}{agg['max_ds']:10.2e}{np.mean(agg['p99']):10.2e}"
              f"{np.mean(agg['bi']):10.3f}{np.mean(agg['bd']):11.3f}{np.mean(agg['r']):11.2e}", flush=True)


if __name__ == "__main__":
    main()
