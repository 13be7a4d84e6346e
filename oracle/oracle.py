"""ctypes doorway to the CPU oracle (test infrastructure -- see oracle/unina_oracle.h).

Importable ONLY from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libunina_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libref_postprocess.so")


def build(force: bool = False) -> None:
    srcs = [os.path.join(_HERE, f) for f in ("unina_oracle.c", "postprocess_oracle.c", "preprocess_oracle.c",
                                             "unina_oracle.h", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True, capture_output=True)


class Det(C.Structure):
    _fields_ = [("x1", C.c_float), ("y1", C.c_float), ("x2", C.c_float), ("y2", C.c_float),
                ("confidence", C.c_float), ("class_id", C.c_int), ("valid", C.c_int), ("_pad", C.c_int)]


class Semantics(C.Structure):
    _fields_ = [("ge_threshold", C.c_int), ("iou_eps", C.c_float), ("strict_conf", C.c_int), ("max_det", C.c_int)]


DET_DTYPE = np.dtype([("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("confidence", "<f4"),
                      ("class_id", "<i4"), ("valid", "<i4"), ("_pad", "<i4")])
REF_DET_DTYPE = np.dtype([("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("confidence", "<f4"),
                          ("class_id", "<i4")])

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.uo_sd_load.restype = C.c_void_p
        L.uo_sd_load.argtypes = [C.c_char_p]
        L.uo_sd_free.argtypes = [C.c_void_p]
        L.uo_forward.restype = C.c_void_p
        L.uo_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.uo_forward_qat.restype = C.c_void_p
        L.uo_forward_qat.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.uo_run_get.restype = C.POINTER(C.c_float)
        L.uo_run_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.uo_run_count.argtypes = [C.c_void_p]
        L.uo_run_name.restype = C.c_char_p
        L.uo_run_name.argtypes = [C.c_void_p, C.c_int]
        L.uo_run_free.argtypes = [C.c_void_p]
        L.uo_last_error.restype = C.c_char_p
        L.uo_semantics_engine.restype = Semantics
        L.uo_semantics_cpu_header.restype = Semantics
        L.uo_sigmoid.restype = C.c_float
        L.uo_sigmoid.argtypes = [C.c_float]
        L.uo_iou.restype = C.c_float
        L.uo_iou.argtypes = [C.POINTER(Det), C.POINTER(Det), C.c_float]
        L.uo_postprocess.restype = C.c_int
        L.uo_postprocess.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(Semantics), C.c_void_p,
                                     C.POINTER(C.c_int)]
        _lib = L
    return _lib


class StateDict:
    """Loads a UNSD file (or a {name: ndarray} dict, via a temp file) into the oracle."""

    def __init__(self, src):
        L = lib()
        self._tmp = None
        if isinstance(src, dict):
            from unina_yolo_dla_amd import statedict
            fd, self._tmp = tempfile.mkstemp(suffix=".unsd")
            os.close(fd)
            statedict.save(self._tmp, src)
            src = self._tmp
        self.h = L.uo_sd_load(src.encode())
        if self._tmp:
            os.unlink(self._tmp)
        if not self.h:
            raise RuntimeError(L.uo_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            lib().uo_sd_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: module globals may already be gone
            pass


def usable_cpus() -> int:
    """CPUs this process may actually run on: the affinity mask, cut down to the cgroup CPU quota. os.cpu_count() is
    the HOST's count: on a GPU box whose container has a 16-CPU share it says 128+, and an OpenMP team that large only
    time-slices (round 1's "64 threads" figure was 64 threads on such a share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                quota, period = int(f.read()), int(g.read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def forward(sd: StateDict, x: np.ndarray, num_classes: int = 4, base_channels: int = 32, lite_p2: bool = False,
            keep_all: bool = False, nthreads: int = 0, variant: str = "A") -> Dict[str, np.ndarray]:
    """x: [1,3,H,W] or [3,H,W] fp32. Returns {name: [C,H,W] fp32}; the six heads are always present.
    variant "A" = model.py's graph, "B" = qat.py's (state_dict with qat.py key names)."""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(3, x.shape[-2], x.shape[-1])
    if nthreads <= 0:
        nthreads = min(usable_cpus(), 32)
    if variant == "B":
        run = L.uo_forward_qat(sd.h, x.ctypes.data, x.shape[1], x.shape[2], num_classes, base_channels, int(keep_all), nthreads)
    else:
        run = L.uo_forward(sd.h, x.ctypes.data, x.shape[1], x.shape[2], num_classes, base_channels, int(lite_p2),
                           int(keep_all), nthreads)
    if not run:
        raise RuntimeError(L.uo_last_error().decode())
    try:
        n = L.uo_run_count(run)
        if n < 0:
            raise RuntimeError(L.uo_last_error().decode())
        out = {}
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        for i in range(n):
            name = L.uo_run_name(run, i)
            p = L.uo_run_get(run, name, C.byref(c), C.byref(h), C.byref(w))
            out[name.decode()] = np.ctypeslib.as_array(p, shape=(c.value, h.value, w.value)).copy()
        return out
    finally:
        L.uo_run_free(run)


def semantics(kind: str = "engine") -> Semantics:
    L = lib()
    return L.uo_semantics_engine() if kind == "engine" else L.uo_semantics_cpu_header()


def _head_args(heads, strides):
    arrs = [np.ascontiguousarray(h, dtype=np.float32) for h in heads]
    ptrs = (C.c_void_p * 6)(*[a.ctypes.data for a in arrs])
    gw = (C.c_int * 3)(*[arrs[2 * i].shape[-1] for i in range(3)])
    gh = (C.c_int * 3)(*[arrs[2 * i].shape[-2] for i in range(3)])
    st = (C.c_int * 3)(*strides)
    total = sum(arrs[2 * i].shape[-1] * arrs[2 * i].shape[-2] for i in range(3))
    return arrs, ptrs, gw, gh, st, total


def postprocess(heads, conf_thr=0.5, iou_thr=0.45, conformal_q=0.1, sem: Optional[Semantics] = None,
                strides=(4, 8, 16)):
    """heads: [p2_cls,p2_reg,p3_cls,p3_reg,p4_cls,p4_reg], each [C,H,W]. Returns (dets[DET_DTYPE], n_candidates)."""
    L = lib()
    sem = sem or semantics("engine")
    arrs, ptrs, gw, gh, st, total = _head_args(heads, strides)
    nc = arrs[0].shape[0]
    out = np.zeros(max(total, 1), dtype=DET_DTYPE)
    ncand = C.c_int()
    k = L.uo_postprocess(ptrs, gw, gh, st, nc, conf_thr, iou_thr, conformal_q, C.byref(sem), out.ctypes.data,
                         C.byref(ncand))
    return out[:k].copy(), ncand.value


def have_ref() -> bool:
    return os.path.exists(_REF_PATH)


def ref_postprocess(heads, conf_thr=0.5, iou_thr=0.45, conformal_q=0.1, strides=(4, 8, 16)):
    """Runs the REFERENCE postprocess.hpp (compiled into oracle/_ref; dev container only)."""
    R = C.CDLL(_REF_PATH)
    R.ref_postprocess.restype = C.c_int
    R.ref_postprocess.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    arrs, ptrs, gw, gh, st, total = _head_args(heads, strides)
    out = np.zeros(max(total, 1), dtype=REF_DET_DTYPE)
    ncand = C.c_int()
    k = R.ref_postprocess(ptrs, gw, gh, st, arrs[0].shape[0], conf_thr, iou_thr, conformal_q, out.ctypes.data,
                          len(out), C.byref(ncand))
    return out[:k].copy(), ncand.value


class Norm(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("mean_r", "mean_g", "mean_b", "std_r", "std_g", "std_b")]


IMAGENET = (0.485, 0.456, 0.406, 0.229, 0.224, 0.225)


def preprocess_bgra(img: np.ndarray, norm=IMAGENET, dst_hw=None) -> np.ndarray:
    """img: [H, pitch_bytes] or [H, W, 4] uint8 BGRA. Returns [3, h, w] fp32 (cuda_preprocess.cu:99-128 / :144-204)."""
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[0], img.shape[1] if img.ndim == 3 else img.shape[1] // 4
    pitch = img.strides[0]
    n = Norm(*norm)
    if dst_hw is None:
        out = np.empty((3, h, w), dtype=np.float32)
        L.uo_preprocess_bgra(img.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), w, h, pitch, C.byref(n))
    else:
        dh, dw = dst_hw
        out = np.empty((3, dh, dw), dtype=np.float32)
        L.uo_preprocess_bgra_resize(img.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), w, h, pitch, dw, dh, C.byref(n))
    return out


def preprocess_nv12(y: np.ndarray, uv: np.ndarray, norm=IMAGENET) -> np.ndarray:
    """y: [H, W] uint8, uv: [H/2, W] uint8 interleaved U,V (cuda_preprocess.cu:212-253)."""
    L = lib()
    y = np.ascontiguousarray(y, dtype=np.uint8)
    uv = np.ascontiguousarray(uv, dtype=np.uint8)
    h, w = y.shape
    out = np.empty((3, h, w), dtype=np.float32)
    n = Norm(*norm)
    L.uo_preprocess_nv12(y.ctypes.data_as(C.c_void_p), uv.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), w, h,
                         y.strides[0], uv.strides[0], C.byref(n))
    return out
