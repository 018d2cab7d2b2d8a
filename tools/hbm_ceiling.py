#!/usr/bin/env python3
"""Measured HBM write / copy ceilings on this GPU (measurement tooling)."""
import torch
n = 100000 * 1840
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n, dtype=torch.float64, device="cuda")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps
ms = timeit(lambda: x.fill_(1.0)); print(f"fill  {n*8/1e9:.2f} GB: {ms*1e3:.1f} us -> {n*8/ms/1e6:.0f} GB/s (write only)")
ms = timeit(lambda: y.copy_(x)); print(f"copy  {n*8/1e9:.2f} GB: {ms*1e3:.1f} us -> {2*n*8/ms/1e6:.0f} GB/s (read+write)")
ms = timeit(lambda: x.zero_()); print(f"zero  {n*8/1e9:.2f} GB: {ms*1e3:.1f} us -> {n*8/ms/1e6:.0f} GB/s (write only)")
