#!/bin/bash
# Stand-alone micro-benchmarks behind the design decisions of DESIGN.md 4.2 (results: profiles/r02/v20_probe_*.txt).
#   ifetch_probe : is straight-line (fully unrolled) MFMA code slower than a loop?        -> no (16.8 cold / 15.9 warm cycles per MFMA)
#   mix_probe    : issue cost of ds_read_b128 / address VALU ops between MFMAs, one wave per SIMD
#   ingest_probe : L2 -> CU bytes per clock against waves per workgroup, workgroups in flight, L2 hits / misses
#   l2warm_probe : does an L2 warm-up by the previous kernel survive the kernel boundary?  -> yes
#   ldsbank_probe: ds_read_b128 fragment reads against the pixel pitch of a padded LDS image -> +32 bytes is conflict-free, +16 is not
#   launch_probe : host <-> GPU round trip of a launch / a graph launch; the per-launch floor of a dependent chain inside one graph (1.6 us)
#   vmem_mfma_probe (round 3): what one 1-KiB request costs a wave between its own MFMAs; MFMA-only throughput at 1 / 2 / 4 waves per SIMD
# (round 3: ingest_probe also compares the lane -> slot maps of a 1-KiB weight block: lane order 107 B/clk/CU, the LDS-image order 55)
set -e
cd "$(dirname "$0")"
for p in ifetch_probe mix_probe ingest_probe l2warm_probe ldsbank_probe launch_probe vmem_mfma_probe; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -ftemplate-depth=2048 -o $p $p.hip
done
