"""Seeded synthetic weights for graph (A).

The reference ships no checkpoint, and its default initialisation is degenerate
for parity work (every logit ~0 => sigmoid ~0.5 => all 33 600 cells pass the 0.5
threshold and overflow MAX_DETECTIONS). This generator produces a state_dict
with the reference's key names (model.py state_dict: 378 keys incl.
``num_batches_tracked``, which is omitted here -- it does not enter eval-mode BN)
whose activations stay bounded and whose heads give a few hundred candidates:

* conv weights ~ N(0, 2/fan_in);
* BN running_var ~ U(0.8,1.2), gamma = U(0.7,0.9)*sqrt(var+eps),
  beta, running_mean ~ N(0, 0.05^2);
* head output convs: weight ~ N(0, 1/fan_in) * head_scale, bias = cls_bias / reg_bias.

``head_scale`` makes the cls logits' std 2.0 and the reg outputs' std 0.5 on the
probe frame (seed 1234); it depends on the whole network, so it is measured once
with a probe forward and tabulated in ``synth_calib.json`` (written by
tests/golden/make_golden.py, which probes with the reference model itself).
"""
from __future__ import annotations

import json
import os
from typing import Callable, Dict, Optional

import numpy as np

from . import rng
from .graph import Graph, OUTPUT_NAMES

BN_EPS = 1e-5                      # nn.BatchNorm2d default (model.py:47)
CLS_STD, REG_STD = 2.0, 0.5
CLS_BIAS, REG_BIAS = -3.5, 2.0
_CALIB_PATH = os.path.join(os.path.dirname(__file__), "synth_calib.json")


def calib_key(seed: int, g: Graph) -> str:
    return f"seed{seed}_nc{g.num_classes}_bc{g.base_channels}_lite{int(g.lite_p2)}" + ("" if g.variant == "A" else f"_{g.variant}")


def load_calib() -> Dict[str, Dict[str, float]]:
    if os.path.exists(_CALIB_PATH):
        with open(_CALIB_PATH) as f:
            return json.load(f)
    return {}


def make_state_dict(seed: int = 7, graph: Optional[Graph] = None,
                    head_scales: Optional[Dict[str, float]] = None,
                    cls_bias: float = CLS_BIAS, reg_bias: float = REG_BIAS) -> Dict[str, np.ndarray]:
    """Returns {key: fp32 ndarray}. ``head_scales`` maps output name (p2_cls...) to the
    multiplier of that head's final 1x1 weights; defaults to the tabulated calibration,
    or 1.0 (uncalibrated) if none is tabulated for this (seed, graph)."""
    g = graph or Graph()
    if head_scales is None:
        head_scales = load_calib().get(calib_key(seed, g), {})
    sd: Dict[str, np.ndarray] = {}
    out_name_of = {g.nodes[i].name: OUTPUT_NAMES[j] for j, i in enumerate(g.outputs)}
    for n in g.convs():
        fan_in = n.cin * n.k * n.k
        if n.kind == "conv":
            w = rng.normal(seed, f"{n.name}.conv.weight", n.c * fan_in) * np.sqrt(2.0 / fan_in)
            sd[f"{n.name}.conv.weight"] = w.astype(np.float32).reshape(n.c, n.cin, n.k, n.k)
            var = rng.uniform(seed, f"{n.name}.bn.running_var", n.c, 0.8, 1.2).astype(np.float32)
            fold = rng.uniform(seed, f"{n.name}.bn.fold", n.c, 0.7, 0.9)
            gamma = (fold * np.sqrt(var.astype(np.float64) + BN_EPS)).astype(np.float32)
            sd[f"{n.name}.bn.weight"] = gamma
            sd[f"{n.name}.bn.bias"] = (rng.normal(seed, f"{n.name}.bn.bias", n.c) * 0.05).astype(np.float32)
            sd[f"{n.name}.bn.running_mean"] = (rng.normal(seed, f"{n.name}.bn.running_mean", n.c) * 0.05).astype(np.float32)
            sd[f"{n.name}.bn.running_var"] = var
        else:
            oname = out_name_of[n.name]
            scale = float(head_scales.get(oname, 1.0))
            w = rng.normal(seed, f"{n.name}.weight", n.c * fan_in) * np.sqrt(1.0 / fan_in) * scale
            sd[f"{n.name}.weight"] = w.astype(np.float32).reshape(n.c, n.cin, 1, 1)
            b = cls_bias if oname.endswith("cls") else reg_bias
            sd[f"{n.name}.bias"] = np.full((n.c,), b, dtype=np.float32)
    return sd


def calibrate_head_scales(seed: int, graph: Graph,
                          probe: Callable[[Dict[str, np.ndarray], np.ndarray], Dict[str, np.ndarray]],
                          probe_seed: int = 1234) -> Dict[str, float]:
    """Measure the per-head multipliers with ``probe(state_dict, frame) -> {p2_cls: ndarray, ...}``
    (any forward implementation: the reference model in make_golden.py, or the engine's raw-head mode)."""
    sd = make_state_dict(seed, graph, head_scales={}, cls_bias=0.0, reg_bias=0.0)
    x = rng.frame(probe_seed, graph.in_h, graph.in_w)
    heads = probe(sd, x)
    scales = {}
    for name in OUTPUT_NAMES:
        std = float(np.asarray(heads[name], dtype=np.float64).std())
        target = CLS_STD if name.endswith("cls") else REG_STD
        scales[name] = target / std
    return scales
