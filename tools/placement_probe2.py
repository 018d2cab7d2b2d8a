#!/usr/bin/env python3
"""Kernel time as a function of the byte offset of the output inside one big allocation (measurement tooling)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
el, sd, deg, order, npts, batch = bench.build_element("p3tet")
ps = el.device_polyset()
pts = torch.as_tensor(bench.synth_points(sd, batch, npts, 2)).cuda()
shape = ps.out_shape(order, batch, npts)
n = 1
for s in shape: n *= s
big = torch.empty(n + (128 << 20) // 8, dtype=torch.float64, device="cuda")
print("kernel:", ps.kernel_name(order, batch, npts), " FIAT_AMD_CHUNK =", os.environ.get("FIAT_AMD_CHUNK"), f"base 0x{big.data_ptr():x}")
offs = [0, 4 << 10, 64 << 10, 256 << 10, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 12 << 20, 16 << 20, 24 << 20, 32 << 20, 48 << 20, 64 << 20, 96 << 20]
for off in offs:
    out = big[off // 8: off // 8 + n].view(shape)
    t = [ps.time_tabulate_batch(order, pts, None, out, 20) for _ in range(3)]
    print(f"offset {off / (1 << 20):8.3f} MiB: {statistics.median(t) * 1e3:7.1f} us")
