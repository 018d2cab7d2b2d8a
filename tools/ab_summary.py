#!/usr/bin/env python3
"""Summarise an ABAB log of tools/mixr_ab.sh (measurement tooling): python tools/ab_summary.py gpurun_out/ab.txt"""
import re, sys
runs = []
for l in open(sys.argv[1]):
    if l.startswith("=="):
        cur = {}
        runs.append((l.split()[1].split("/")[-1], cur))
        continue
    m = re.match(r"(\S+)\s+([\d.]+) us\s+([\d.]+) % HBM\s+(\S+)", l)
    if m:
        cur[m.group(1)] = (float(m.group(2)), m.group(4))
libs = []
for name, _ in runs:
    if name not in libs:
        libs.append(name)
best = {n: {} for n in libs}
inst = {n: {} for n in libs}
for name, d in runs:
    for k, (t, i) in d.items():
        best[name][k] = min(t, best[name].get(k, 1e30))
        inst[name][k] = i
base = libs[0]
for k in best[base]:
    row = f"{k:30s} {inst[base][k]:22s} {best[base][k]:7.1f}"
    for n in libs[1:]:
        if k in best[n]:
            row += f" | {n} {inst[n][k]:20s} {best[n][k]:7.1f} ratio {best[n][k] / best[base][k]:.3f}"
    print(row)
