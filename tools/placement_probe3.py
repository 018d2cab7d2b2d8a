#!/usr/bin/env python3
"""Are the slow XCDs tied to the XCD or to the memory region?  Same buffer, region assignment rotated. (measurement tooling)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
el, sd, deg, order, npts, batch = bench.build_element("p3tet")
ps = el.device_polyset()
pts = torch.as_tensor(bench.synth_points(sd, batch, npts, 2)).cuda()
shape = ps.out_shape(order, batch, npts)
bufs = [torch.empty(shape, dtype=torch.float64, device="cuda") for _ in range(3)]
for i, out in enumerate(bufs):
    for rot in (0, 0, 1, 2, 3, 8):
        os.environ["FIAT_AMD_ROT"] = str(rot)
        sys.stderr.write(f"--- buffer {i} rot {rot}\n"); sys.stderr.flush()
        ps.time_tabulate_batch(order, pts, None, out, 1)
