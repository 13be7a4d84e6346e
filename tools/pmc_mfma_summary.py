#!/usr/bin/env python3
"""Per-kernel SQ counter averages from rocprofv3 --pmc passes -> JSON, with MFMA utilisation.

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES: the fraction of the cycles in which a CU had work that its
matrix pipes were executing (both counters summed over the chip; MI355X_MICROARCH.md, rocprofv3 PMC slots / cycle
constants: MFMA_BUSY counts cycles, 16 per v_mfma_f32_16x16x32_f16 per SIMD, so a CU whose four SIMDs issue
back-to-back reads 4 x busy-CU cycles -> the ratio is divided by 4 SIMDs). Also reported against the kernel's wall
time: mfma_cycles_per_launch / (duration x 256 CUs x 4 SIMDs x clock)."""
import csv, glob, json, os, sys
res = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = res.setdefault(r["Kernel_Name"], {})
                c = k.setdefault(r["Counter_Name"], [0.0, 0])
                c[0] += float(r["Counter_Value"]); c[1] += 1
out = {}
for k, cs in sorted(res.items()):
    o = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
    o["launches"] = max(v[1] for v in cs.values())
    busy, cu = o.get("SQ_VALU_MFMA_BUSY_CYCLES"), o.get("SQ_BUSY_CU_CYCLES")
    if busy is not None and cu:
        o["mfma_util_of_busy_cu_cycles"] = busy / (4.0 * cu)
    gui = o.get("GRBM_GUI_ACTIVE")
    if busy is not None and gui:                       # rocprofv3's MfmaUtil: MFMA-busy cycles / (GPU-active cycles x 1024 SIMDs)
        o["mfma_util"] = busy / (gui * 1024.0)
    out[k] = o
json.dump(out, sys.stdout, indent=1)
