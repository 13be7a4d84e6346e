#!/usr/bin/env python3
"""Summarise gpurun_out/sweep.log: best few configs per op."""
import re, sys
lines = open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/sweep.log').read().split('\n')
i = [k for k, l in enumerate(lines) if l.startswith('sweep')][0]
hdr = lines[i + 1].split()
for l in lines[i + 2:]:
    if not l.strip() or l.startswith('sum'):
        print(l); continue
    m = re.match(r'\s*(\d+) M\s*(\d+) N\s*(\d+) K\s*(\d+) (.{22}) (.*)', l)
    if not m: continue
    idx, M, N, K, name, rest = m.groups()
    vals = rest.split()
    best = sorted((float(v.rstrip('*')), c) for c, v in enumerate(vals) if v != '-')
    print(f"{idx:>2} M{M:>6} N{N:>4} K{K:>5} {name.strip()[:26]:26s} " + " ".join(f"{hdr[c].replace('f16,','')}={t:.1f}" for t, c in best[:4]))
