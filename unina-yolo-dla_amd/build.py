"""Builds libunina_mi355.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the dev container; the .so travels to the GPU box
with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from typing import List

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "libunina_mi355.so")
ARCH = "gfx950"

# translation unit -> extra flags
UNITS = {
    "conv_igemm.hip": [],
    "c3k2_fused.hip": [],
    "head_fused.hip": [],
    "block_dual.hip": [],
    "conv_pair.hip": [],
    "stem_pool.hip": [],
    "postprocess.hip": ["-ffp-contract=off"],   # box arithmetic must round like the reference's scalar code
    "preprocess.hip": ["-ffp-contract=off"],    # normalisation arithmetic rounds as written (oracle/preprocess_oracle.c)
    "engine.hip": [],
    "comm.hip": [],                             # host code only: RCCL all-gather of detection slots, librccl loaded on first use
}
COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-result"]


def hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need the ROCm toolchain to build the MI355X engine)")


def _deps() -> List[str]:
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdrs.append(os.path.join(os.path.dirname(PKG), "include", "unina_mi355.h"))
    return hdrs


def _sha(paths: List[str], extra: str = "") -> str:
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def source_hash() -> str:
    """Hash of every source the library is built from (csrc/*.hip, csrc/*.h, include/unina_mi355.h). It is compiled into
    the library (unina_version() ends with it), so a test can tell that the loaded binary was built from the sources
    under test and not from an older tree (the .so travels to the GPU box; nothing else ties the two together)."""
    return _sha([os.path.join(CSRC, u) for u in sorted(UNITS)] + _deps())[:16]


def build_native(force: bool = False, verbose: bool = False) -> str:
    """Rebuilds by CONTENT: a unit is recompiled when the hash of its source, the headers and its flags differs from the
    one stored next to its object file (mtimes do not survive a checkout or a copy to another box)."""
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    deps = _deps()
    shash = source_hash()
    objs = []
    jobs = []
    for unit, extra in UNITS.items():
        src = os.path.join(CSRC, unit)
        obj = os.path.join(OBJ, unit + ".o")
        stamp = obj + ".hash"
        objs.append(obj)
        flags = [*COMMON, *extra]
        if unit == "engine.hip":
            flags.append(f'-DUNINA_SOURCE_HASH="{shash}"')
        want = _sha([src] + deps, " ".join(flags))
        have = open(stamp).read().strip() if os.path.exists(stamp) else ""
        if force or not os.path.exists(obj) or have != want:
            jobs.append((unit, [cc, *flags, "-c", src, "-o", obj], stamp, want))

    def compile_one(job):
        unit, cmd, stamp, want = job
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        if os.path.exists(stamp):
            os.remove(stamp)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            with open(stamp, "w") as f:
                f.write(want)
        return unit, r.returncode, r.stderr

    # the units are independent: compile them side by side (a cold build is ~3 min of hipcc time, ~1.5 min wall on 4 jobs)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(4, os.cpu_count() or 1))) as pool:
        for unit, rc, err in pool.map(compile_one, jobs):
            if rc:
                raise RuntimeError(f"hipcc failed on {unit}:\n{err}")
    rebuilt = bool(jobs)
    if rebuilt or not os.path.exists(LIB):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


HARNESS_SRC = os.path.join(os.path.dirname(PKG), "tools", "node_harness.cpp")
HARNESS = os.path.join(os.path.dirname(PKG), "tools", "node_harness")


def build_harness(force: bool = False) -> str:
    """tools/node_harness: the compiled C++ consumer of include/unina_mi355.h (the reference node's per-frame body,
    perception_node.cpp:581-689), linked against libunina_mi355.so exactly as the node would be."""
    if not force and os.path.exists(HARNESS) and os.path.getmtime(HARNESS) >= max(
            os.path.getmtime(HARNESS_SRC), os.path.getmtime(os.path.join(os.path.dirname(PKG), "include", "unina_mi355.h"))):
        return HARNESS
    cmd = [hipcc(), "-O2", "-std=c++17", "-Wall", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(os.path.dirname(PKG), "include"),
           "-I", "/opt/rocm/include", HARNESS_SRC, "-o", HARNESS, "-L", PKG, "-lunina_mi355", "-L", "/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath,$ORIGIN/../unina-yolo-dla_amd", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"node_harness build failed:\n{r.stderr}")
    return HARNESS


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
    print(build_harness(force="--force" in sys.argv))
