#!/usr/bin/env python3
"""Does the physical placement of the output buffer change the kernel time?  (measurement tooling)
Times the default kernel on several output buffers allocated in ONE process, with spacer
allocations of different sizes in between."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
el, sd, deg, order, npts, batch = bench.build_element("p3tet")
ps = el.device_polyset()
pts = torch.as_tensor(bench.synth_points(sd, batch, npts, 2)).cuda()
shape = ps.out_shape(order, batch, npts)
bufs, spacers = [], []
for i in range(6):
    bufs.append(torch.empty(shape, dtype=torch.float64, device="cuda"))
    spacers.append(torch.empty((1 << 20) * (1 + 3 * i), dtype=torch.uint8, device="cuda"))
print("kernel:", ps.kernel_name(order, batch, npts), " FIAT_AMD_CHUNK =", os.environ.get("FIAT_AMD_CHUNK"))
for rnd in range(2):
    for i, out in enumerate(bufs):
        t = [ps.time_tabulate_batch(order, pts, None, out, int(os.environ.get("REPS", "20"))) for _ in range(int(os.environ.get("ROUNDS", "5")))]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out.fill_(1.0)
        e1.record()
        torch.cuda.synchronize()
        fill_us = e0.elapsed_time(e1) * 100
        print(f"[fill_ {fill_us:6.1f} us] round {rnd} buffer {i} @0x{out.data_ptr():x} (mod 2MB = {out.data_ptr() % (1 << 21):#x}): {statistics.median(t) * 1e3:7.1f} us")
