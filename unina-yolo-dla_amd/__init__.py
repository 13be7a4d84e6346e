"""unina-yolo-dla_amd: MI355X-native engine for the UNINA-YOLO-DLA detector hot path.

Host-side mirror of the reference interfaces for that path only:
graph (model.py) -> export (export_trt.py role) -> engine (TensorRTEngine +
gpu_postprocess.h roles behind the C ABI in include/unina_mi355.h).
"""
from . import gather, graph, metrics, rng, statedict, synth  # noqa: F401

__all__ = ["gather", "graph", "metrics", "rng", "statedict", "synth"]
