#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> JSON.
Units/corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads (16 B/lane, global_load and LDS-DMA alike), so reads are doubled.

    pmc_summary.py FETCH_DIR WRITE_DIR                         # the kernel table on stdout
    pmc_summary.py FETCH_DIR WRITE_DIR --key A:fp16:640 --merge profiles/r03/pmc_traffic.json
        # ... stored under {"configs": {"A:fp16:640": table}} of that file: bench.py looks a kernel's traffic up by
        # (configuration, kernel) -- a kernel name alone does not identify the launch."""
import csv, glob, json, os, sys


def load(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                v = float(r["Counter_Value"])
                a = out.setdefault(k, [0.0, 0])
                a[0] += v; a[1] += 1
    return out


def table(fetch_dir, write_dir):
    fetch, write = load(fetch_dir), load(write_dir)
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, [0.0, 0]); w, nw = write.get(k, [0.0, 0])
        if not nf and not nw: continue
        res[k] = {"launches": max(nf, nw),
                  "fetch_bytes_per_launch": 2.0 * 1024 * f / max(nf, 1),      # x2: gfx950 FETCH_SIZE correction for 16-B/lane reads
                  "write_bytes_per_launch": 1024 * w / max(nw, 1)}
        res[k]["hbm_bytes_per_launch"] = res[k]["fetch_bytes_per_launch"] + res[k]["write_bytes_per_launch"]
    return res


if __name__ == "__main__":
    res = table(sys.argv[1], sys.argv[2])
    if "--merge" in sys.argv:
        key = sys.argv[sys.argv.index("--key") + 1]
        path = sys.argv[sys.argv.index("--merge") + 1]
        blob = {"configs": {}}
        if os.path.exists(path):
            with open(path) as f:
                blob = json.load(f)
        blob.setdefault("configs", {})[key] = res
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(blob, f, indent=1)
        print(f"{key}: {len(res)} kernels -> {path}")
    else:
        json.dump(res, sys.stdout, indent=1)
