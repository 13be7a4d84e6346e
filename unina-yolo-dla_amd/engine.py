"""Host-side mirror of the reference's engine interface, over the C ABI (include/unina_mi355.h).

``Engine`` plays the role of the C++ ``TensorRTEngine`` wrapper (perception_node.cpp:223-351: load / bind
tensor addresses / enqueueV3 / getInputDimensions) plus the fused per-frame path (perception_node.cpp:612-656).
All compute happens inside libunina_mi355.so (hand-written HIP); torch is used only to own device memory and
streams. There is NO fallback: if the shared library is missing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import tempfile
from typing import Dict, List, Optional

import numpy as np

from . import export as _export
from .graph import Graph, OUTPUT_NAMES

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UNINA_LIB") or os.path.join(_PKG, "libunina_mi355.so")   # (UNINA_LIB: another build of the library, for same-box A/B runs)
MAX_DETECTIONS = 1024

DET_DTYPE = np.dtype([("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("confidence", "<f4"),
                      ("class_id", "<i4"), ("valid", "<i4"), ("_pad", "<i4")])
assert DET_DTYPE.itemsize == 32

ERRORS = {1: "IO", 2: "FORMAT", 3: "HIP", 4: "ARG", 5: "STATE", 6: "UNSUPPORTED"}


class OpInfo(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("kernel", C.c_char * 64), ("kind", C.c_int), ("m", C.c_int), ("n", C.c_int),
                ("k", C.c_int), ("flops", C.c_double), ("bytes", C.c_double), ("grid", C.c_int), ("block", C.c_int)]


class NormParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("mean_r", "mean_g", "mean_b", "std_r", "std_g", "std_b")]


class EngineError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None

# every symbol include/unina_mi355.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "unina_load_engine", "unina_unload_engine", "unina_engine_input_dims", "unina_set_tensor_address",
    "unina_tensor_address", "unina_enqueue", "unina_infer", "unina_infer_bgra", "unina_infer_async", "unina_postprocess_async",
    "unina_last_error", "unina_op_count", "unina_get_op_info", "unina_profile_ops", "unina_profile_post", "unina_debug_read_buffer",
    "unina_version", "unina_conv_config_count", "unina_conv_config_name", "unina_set_op_config", "unina_autotune", "unina_debug_post_stamps", "unina_debug_conv_stamps", "unina_debug_dual_stamps", "unina_debug_dual_timeline", "unina_debug_block_stamps", "unina_serial_latency",
    "unina_set_fusion", "unina_fusion_groups", "unina_debug_fusable_groups",
    "unina_comm_unique_id", "unina_comm_init", "unina_comm_all_gather", "unina_comm_rank", "unina_comm_world", "unina_comm_destroy",
    "unina_comm_last_error",
    "create_norm_params_imagenet", "create_norm_params", "preprocess_bgra_resize", "preprocess_bgra", "preprocess_nv12",
    "allocate_preprocess_buffer", "free_preprocess_buffer", "create_preprocess_stream", "destroy_preprocess_stream",
    "init_postprocess_resources", "cleanup_postprocess_resources", "reset_detection_counter", "get_detection_count",
    "decode_yolo_head", "run_gpu_nms", "copy_valid_detections_to_host",
]


def load_library() -> C.CDLL:
    """Loads libunina_mi355.so; raises if it has not been built (``python unina-yolo-dla_amd/build.py``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                          f"(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.unina_load_engine.argtypes = [C.c_char_p, ci, C.POINTER(vp)]
    L.unina_unload_engine.argtypes = [vp]
    L.unina_unload_engine.restype = None
    L.unina_engine_input_dims.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    L.unina_set_tensor_address.argtypes = [vp, C.c_char_p, vp]
    L.unina_tensor_address.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.unina_enqueue.argtypes = [vp, vp]
    L.unina_infer.argtypes = [vp, vp, cf, cf, cf, vp, C.POINTER(ci), vp]
    L.unina_infer_async.argtypes = [vp, vp, cf, cf, cf, vp, vp, vp]
    L.unina_infer_bgra.argtypes = [vp, vp, ci, ci, ci, C.POINTER(NormParams), cf, cf, cf, vp, C.POINTER(ci), vp]
    L.unina_postprocess_async.argtypes = [vp, cf, cf, cf, vp, vp, vp]
    L.unina_last_error.argtypes = [vp]
    L.unina_last_error.restype = C.c_char_p
    L.unina_op_count.argtypes = [vp]
    L.unina_get_op_info.argtypes = [vp, ci, C.POINTER(OpInfo)]
    L.unina_profile_ops.argtypes = [vp, ci, C.POINTER(cf), vp]
    L.unina_profile_post.argtypes = [vp, ci, cf, cf, cf, C.POINTER(cf), vp]
    L.unina_debug_read_buffer.argtypes = [vp, C.c_char_p, vp, C.c_size_t, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    L.unina_version.restype = C.c_char_p
    L.unina_conv_config_name.restype = C.c_char_p
    L.unina_conv_config_name.argtypes = [ci]
    L.unina_set_op_config.argtypes = [vp, ci, ci]
    L.unina_autotune.argtypes = [vp, ci, vp]
    L.unina_set_fusion.argtypes = [vp, ci]
    L.unina_fusion_groups.argtypes = [vp]
    L.unina_debug_fusable_groups.argtypes = [C.c_char_p]
    L.unina_debug_post_stamps.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.unina_debug_conv_stamps.argtypes = [vp, ci, C.POINTER(C.c_longlong), vp]
    L.unina_debug_dual_stamps.argtypes = [vp, ci, C.POINTER(C.c_longlong), vp]
    L.unina_debug_dual_timeline.argtypes = [vp, ci, C.POINTER(C.c_longlong), ci, vp]
    L.unina_debug_block_stamps.argtypes = [vp, ci, C.POINTER(C.c_longlong), vp]
    L.unina_serial_latency.argtypes = [vp, C.POINTER(vp), ci, ci, cf, cf, cf, C.POINTER(C.c_double), vp]
    # multi-GPU: RCCL gather of detection slots behind the C ABI (csrc/comm.hip)
    L.unina_comm_unique_id.argtypes = [vp]
    L.unina_comm_init.argtypes = [C.POINTER(vp), vp, ci, ci, ci]
    L.unina_comm_all_gather.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.unina_comm_rank.argtypes = [vp]
    L.unina_comm_world.argtypes = [vp]
    L.unina_comm_destroy.argtypes = [vp]
    L.unina_comm_destroy.restype = None
    L.unina_comm_last_error.restype = C.c_char_p
    # cuda_preprocess.h drop-in symbols
    L.create_norm_params_imagenet.restype = NormParams
    L.create_norm_params.restype = NormParams
    L.create_norm_params.argtypes = [cf] * 6
    L.preprocess_bgra_resize.argtypes = [vp, vp, ci, ci, ci, ci, ci, NormParams, vp]
    L.preprocess_bgra.argtypes = [vp, vp, ci, ci, ci, NormParams, vp]
    L.preprocess_nv12.argtypes = [vp, vp, vp, ci, ci, ci, ci, NormParams, vp]
    L.allocate_preprocess_buffer.restype = vp
    L.allocate_preprocess_buffer.argtypes = [ci, ci]
    L.free_preprocess_buffer.argtypes = [vp]
    L.create_preprocess_stream.restype = vp
    L.destroy_preprocess_stream.argtypes = [vp]
    # gpu_postprocess.h drop-in symbols
    L.reset_detection_counter.argtypes = [vp]
    L.get_detection_count.argtypes = [C.POINTER(ci), vp]
    L.decode_yolo_head.argtypes = [vp, vp, vp, ci, ci, ci, ci, cf, cf, vp]
    L.run_gpu_nms.argtypes = [vp, ci, cf, vp]
    L.copy_valid_detections_to_host.argtypes = [vp, vp, ci, C.POINTER(ci), vp]
    _lib = L
    return L


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise EngineError("no MI355X visible to this process (torch.cuda.is_available() is False)")
    return torch


def _stream_ptr(stream) -> int:
    if stream is None:
        stream = _torch().cuda.current_stream()
    return stream if isinstance(stream, int) else stream.cuda_stream


class Engine:
    """One engine handle on one GPU. Use two handles to keep two frames in flight."""

    def __init__(self, path: str, device: int = 0):
        self.L = load_library()
        torch = _torch()
        self.device = device
        self.h = C.c_void_p()
        rc = self.L.unina_load_engine(path.encode(), device, C.byref(self.h))
        if rc:
            raise EngineError(f"unina_load_engine({path}) failed [{ERRORS.get(rc, rc)}]: "
                              f"{self.L.unina_last_error(None).decode()}")
        w, h, nc = C.c_int(), C.c_int(), C.c_int()
        self._check(self.L.unina_engine_input_dims(self.h, C.byref(w), C.byref(h), C.byref(nc)))
        self.width, self.height, self.num_classes = w.value, h.value, nc.value
        dev = torch.device("cuda", device)
        # the caller (this wrapper) owns the I/O buffers, as the node does (perception_node.cpp:696-707).
        # self.outputs holds the six raw head tensors as enqueue() / forward() write them. infer() / infer_async() do NOT
        # update them (the heads' output convs run inside the decode launch, include/unina_mi355.h at unina_infer): after
        # an infer call they still hold the previous forward()'s values.
        self.outputs: Dict[str, "torch.Tensor"] = {}
        for name, s in zip(OUTPUT_NAMES, (4, 4, 8, 8, 16, 16)):
            c = nc.value if name.endswith("cls") else 4
            t = torch.zeros((1, c, h.value // s, w.value // s), dtype=torch.float32, device=dev)
            self.outputs[name] = t
            self._check(self.L.unina_set_tensor_address(self.h, name.encode(), t.data_ptr()))
        self._images = None
        self._det_buf = torch.zeros((MAX_DETECTIONS * 8 + 8,), dtype=torch.int32, device=dev)

    # -- construction helpers -------------------------------------------------------------------------
    @classmethod
    def from_state_dict(cls, sd: Dict[str, np.ndarray], graph: Optional[Graph] = None, device: int = 0,
                        path: Optional[str] = None, precision: int = _export.FP16,
                        amax: Optional[Dict[str, float]] = None,
                        weight_amax: Optional[Dict[str, float]] = None) -> "Engine":
        """export_trt.py's role + load: folds/fuses `sd` into an engine file (temporary unless `path`) and loads it.
        INT8 needs `amax` (calibrate_amax below)."""
        if path is None:
            fd, tmp = tempfile.mkstemp(suffix=".une")
            os.close(fd)
            try:
                _export.export_engine(sd, tmp, graph, precision, amax, weight_amax)
                return cls(tmp, device)
            finally:
                os.unlink(tmp)
        _export.export_engine(sd, path, graph, precision, amax, weight_amax)
        return cls(path, device)

    def _check(self, rc: int):
        if rc:
            raise EngineError(f"[{ERRORS.get(rc, rc)}] {self.L.unina_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            _torch().cuda.synchronize(self.device)
            self.L.unina_unload_engine(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- TensorRTEngine-shaped path --------------------------------------------------------------------
    def bind_images(self, images) -> None:
        """images: torch cuda fp32 [1,3,H,W] contiguous (setInputTensorAddress("images", ...))."""
        assert images.is_cuda and images.dtype == _torch().float32 and images.is_contiguous()
        assert tuple(images.shape) == (1, 3, self.height, self.width), images.shape
        self._images = images
        self._check(self.L.unina_set_tensor_address(self.h, b"images", images.data_ptr()))

    def enqueue(self, stream=None) -> None:
        """enqueueV3: raw heads into self.outputs (asynchronous)."""
        self._check(self.L.unina_enqueue(self.h, _stream_ptr(stream)))

    def forward(self, images) -> Dict[str, np.ndarray]:
        """Raw-head forward, synchronous; returns {name: [C,H,W] fp32 ndarray}."""
        self.bind_images(images)
        self.enqueue()
        _torch().cuda.synchronize(self.device)
        return {k: v[0].cpu().numpy() for k, v in self.outputs.items()}

    # -- fused path ----------------------------------------------------------------------------------------
    def infer(self, images, conf_thr: float = 0.5, iou_thr: float = 0.45, conformal_q: float = 0.1, stream=None):
        """Forward + decode + NMS; returns a structured ndarray of kept detections (DET_DTYPE).
        The per-call Python work is kept minimal (the call is ~0.23 ms end to end): a tensor object is validated once,
        its address goes straight into unina_infer, and the host-side record buffer / count are reused."""
        ptr = None
        if images is not None:
            if images is not getattr(self, "_images", None):
                assert images.is_cuda and images.dtype == _torch().float32 and images.is_contiguous()
                assert tuple(images.shape) == (1, 3, self.height, self.width), images.shape
                self._images = images
            ptr = images.data_ptr()
        if getattr(self, "_host_out", None) is None:
            self._host_out = np.zeros(MAX_DETECTIONS, dtype=DET_DTYPE)
            self._host_out_ptr = self._host_out.ctypes.data
            self._host_n = C.c_int()
            self._host_n_ref = C.byref(self._host_n)
        self._check(self.L.unina_infer(self.h, ptr, conf_thr, iou_thr, conformal_q, self._host_out_ptr, self._host_n_ref,
                                       _stream_ptr(stream)))
        return self._host_out[:self._host_n.value].copy()

    def serial_latency(self, frames, n_calls: int, conf_thr: float = 0.5, iou_thr: float = 0.45, conformal_q: float = 0.1, stream=None):
        """`n_calls` serial unina_infer calls over the ring `frames` (CUDA tensors), timed INSIDE the C ABI: latencies in ms."""
        ptrs = (C.c_void_p * len(frames))(*[f.data_ptr() for f in frames])
        lat = (C.c_double * n_calls)()
        self._check(self.L.unina_serial_latency(self.h, ptrs, len(frames), n_calls, conf_thr, iou_thr, conformal_q, lat, _stream_ptr(stream)))
        return np.array(lat[:], dtype=np.float64) * 1e-3

    def infer_bgra(self, frame, width: int, height: int, pitch: int, norm: Optional[NormParams] = None,
                   conf_thr: float = 0.5, iou_thr: float = 0.45, conformal_q: float = 0.1, stream=None):
        """Camera frame (uint8 CUDA tensor, pitched BGRA) -> detections: the pre-process runs inside the stem kernel
        (perception_node.cpp:601-656 as one launch sequence, no fp32 tensor in between)."""
        if norm is None:
            norm = self.L.create_norm_params_imagenet()
        out = np.zeros(MAX_DETECTIONS, dtype=DET_DTYPE)
        n = C.c_int()
        self._check(self.L.unina_infer_bgra(self.h, frame.data_ptr(), width, height, pitch, C.byref(norm), conf_thr, iou_thr,
                                            conformal_q, out.ctypes.data, C.byref(n), _stream_ptr(stream)))
        return out[:n.value].copy()

    def infer_async(self, images, conf_thr: float = 0.5, iou_thr: float = 0.45, conformal_q: float = 0.1,
                    out=None, stream=None):
        """Asynchronous: results stay on the GPU in `out` (int32 tensor of 8 + 8*MAX_DETECTIONS words:
        word 0 = count, records from word 8). Returns `out`."""
        if images is not None:
            self.bind_images(images)
        out = self._det_buf if out is None else out
        base = out.data_ptr()
        self._check(self.L.unina_infer_async(self.h, None, conf_thr, iou_thr, conformal_q, base + 32, base,
                                             _stream_ptr(stream)))
        return out

    def postprocess(self, conf_thr: float = 0.5, iou_thr: float = 0.45, conformal_q: float = 0.1, stream=None):
        """Decode + NMS on whatever the six output tensors currently hold (synchronous)."""
        base = self._det_buf.data_ptr()
        self._check(self.L.unina_postprocess_async(self.h, conf_thr, iou_thr, conformal_q, base + 32, base,
                                                   _stream_ptr(stream)))
        return self.unpack(self._det_buf)

    @staticmethod
    def unpack(buf) -> np.ndarray:
        """int32 result tensor (see infer_async) -> structured ndarray of the kept detections."""
        host = buf.cpu().numpy()
        n = int(host[0])
        return host[8:8 + 8 * n].view(DET_DTYPE).copy()

    # -- introspection ---------------------------------------------------------------------------------------
    def op_infos(self) -> List[dict]:
        out = []
        for i in range(self.L.unina_op_count(self.h)):
            info = OpInfo()
            self._check(self.L.unina_get_op_info(self.h, i, C.byref(info)))
            out.append(dict(name=info.name.decode(), kernel=info.kernel.decode(), kind=info.kind, m=info.m, n=info.n,
                            k=info.k, flops=info.flops, bytes=info.bytes, grid=info.grid, block=info.block))
        return out

    def profile_ops(self, iters: int = 20, stream=None) -> List[dict]:
        n = self.L.unina_op_count(self.h)
        ms = (C.c_float * n)()
        self._check(self.L.unina_profile_ops(self.h, iters, ms, _stream_ptr(stream)))
        infos = self.op_infos()
        for i, d in enumerate(infos):
            d["ms"] = float(ms[i])
        return infos

    def profile_post(self, iters: int = 20, conf_thr: float = 0.5, iou_thr: float = 0.45, conformal_q: float = 0.1, stream=None):
        """(ms of the decode launch, ms of the pair-tile / scan / output launch) inside the frame sequence."""
        ms = (C.c_float * 2)()
        self._check(self.L.unina_profile_post(self.h, iters, conf_thr, iou_thr, conformal_q, ms, _stream_ptr(stream)))
        return float(ms[0]), float(ms[1])

    def autotune(self, images=None, iters: int = 10, stream=None, cache: Optional[str] = None) -> None:
        """Pick the fastest tile configuration per conv op by timing on this GPU (results are unchanged).
        `cache`: JSON tactic cache (the role of TensorRT's timing cache): reused when it matches this engine."""
        import json
        names = self.conv_configs()
        key = f"{self.width}x{self.height}:{self.op_infos()[1]['kernel'][:13]}:" + "|".join(f"{o['m']},{o['n']},{o['k']}" for o in self.op_infos())
        if cache and os.path.exists(cache):
            try:
                with open(cache) as f:
                    blob = json.load(f)
                if blob.get("key") == key and blob.get("configs") == names:
                    for i, c in enumerate(blob["choice"]):
                        if c >= 0:
                            self.set_op_config(i, c)
                    return
            except (ValueError, KeyError):
                pass
        if images is None and self._images is None:
            images = _torch().zeros((1, 3, self.height, self.width), dtype=_torch().float32,
                                    device=_torch().device("cuda", self.device))
        if images is not None:
            self.bind_images(images)
        self._check(self.L.unina_autotune(self.h, iters, _stream_ptr(stream)))
        _torch().cuda.synchronize(self.device)
        if cache:
            norm = [o["kernel"].replace("<f32,", "<f16,") for o in self.op_infos()]
            choice = [names.index(k) if k in names else -1 for k in norm]
            with open(cache, "w") as f:
                json.dump({"key": key, "configs": names, "choice": choice}, f)

    def debug_stamps(self):
        """Phase time stamps (100 MHz ticks) of the last unina_infer's post-process (needs UNINA_POST_STAMPS=1)."""
        buf = (C.c_longlong * 16)()
        self._check(self.L.unina_debug_post_stamps(self.h, buf))
        return [int(v) for v in buf]

    def conv_stamps(self, op_index: int, stream=None) -> List[int]:
        """In-kernel phase stamps (shader-clock ticks) of one launch of conv op `op_index` (debug)."""
        buf = (C.c_longlong * 8)()
        self._check(self.L.unina_debug_conv_stamps(self.h, op_index, buf, _stream_ptr(stream)))
        return [int(v) for v in buf]

    def dual_stamps(self, op_index: int, stream=None) -> List[int]:
        """In-kernel phase stamps of the dual conv launch led by op `op_index`: 8 values per conv (debug)."""
        buf = (C.c_longlong * 16)()
        self._check(self.L.unina_debug_dual_stamps(self.h, op_index, buf, _stream_ptr(stream)))
        return [int(v) for v in buf]

    def block_stamps(self, op_index: int, stream=None) -> List[int]:
        """Per-step shader-clock stamps of the fused C3k2 block led by op `op_index` (debug twin; the 40x40-level blocks)."""
        buf = (C.c_longlong * 16)()
        self._check(self.L.unina_debug_block_stamps(self.h, op_index, buf, _stream_ptr(stream)))
        return [int(v) for v in buf]

    def dual_timeline(self, op_index: int, stream=None):
        """Start / end (100 MHz wall clock) of every workgroup of the dual conv launch led by op `op_index`: array [grid + 1, 2];
        the last row = a marker kernel enqueued right before the launch, one right after it."""
        cap = 2 * 4096 + 2
        buf = (C.c_longlong * cap)()
        n = self.L.unina_debug_dual_timeline(self.h, op_index, buf, cap, _stream_ptr(stream))
        if n <= 0:
            raise RuntimeError(f"unina_debug_dual_timeline: error {-n}")
        return np.array(buf[:2 * n + 2], dtype=np.int64).reshape(n + 1, 2)

    def conv_configs(self) -> List[str]:
        return [self.L.unina_conv_config_name(i).decode() for i in range(self.L.unina_conv_config_count())]

    def set_op_config(self, op_index: int, cfg: int) -> bool:
        """Force a tile configuration for one conv op (-1 = heuristic). Returns False if it does not fit."""
        rc = self.L.unina_set_op_config(self.h, op_index, cfg)
        if rc == 6:
            return False
        self._check(rc)
        return True

    def set_fusion(self, enable: bool) -> int:
        """C3k2 block fusion on/off (bit-identical results; off = every internal buffer is written, for per-layer
        checks and calibration). Returns the number of blocks now running as one launch."""
        self._check(self.L.unina_set_fusion(self.h, int(bool(enable))))
        return self.L.unina_fusion_groups(self.h)

    def read_buffer(self, name: str) -> np.ndarray:
        """Internal activation buffer -> [C,H,W] fp32 (parity tests)."""
        c, h, w = C.c_int(-1), C.c_int(), C.c_int()
        self.L.unina_debug_read_buffer(self.h, name.encode(), None, 0, C.byref(c), C.byref(h), C.byref(w))  # dims only
        if c.value < 0:
            raise EngineError(f"unknown buffer {name!r}")
        out = np.empty(c.value * h.value * w.value, dtype=np.float32)
        self._check(self.L.unina_debug_read_buffer(self.h, name.encode(), out.ctypes.data, out.size, C.byref(c),
                                                   C.byref(h), C.byref(w)))
        return out.reshape(c.value, h.value, w.value)


def calibrate_amax(sd: Dict[str, np.ndarray], graph: Optional[Graph], frames, device: int = 0,
                   percentile: Optional[float] = None, method: Optional[str] = None, specs=None):
    """INT8 calibration on the GPU (the role of qat.py:171-220 `calibrate_model`, 30 batches in train.py:809): runs the
    fp16 engine over `frames` (iterable of [1,3,H,W] fp32 ndarrays) and records, per activation buffer, the
    (percentile of the) absolute maximum, or -- `method` = "entropy" | "mse" | "percentile" -- the range a |x| histogram over
    all frames selects (export.HistogramCalibrator: the reference's QuantDescriptor(calib_method="histogram"), qat.py:91-126).
    Feed the result to from_state_dict(..., precision=INT8, amax=...)."""
    torch = _torch()
    b = _export.EngineBuilder(sd, graph)
    names = [n for (n, _h, _w, _c, dtype, _f, _s) in b.buffers if dtype == _export.BUF_F16]
    eng = Engine.from_state_dict(sd, graph, device)
    eng.set_fusion(False)          # every internal buffer must be written: the calibrator reads them all
    try:
        def per_frame():
            for x in frames:
                eng.forward(torch.from_numpy(np.ascontiguousarray(x)).cuda(device))
                yield {n: eng.read_buffer(n) for n in names}
        if specs is not None:                      # several range selections from one pass (tools/int8_drift.py)
            return _export.calibrate_all(per_frame(), specs)
        return _export.calibrate(per_frame(), percentile, method)
    finally:
        eng.close()
