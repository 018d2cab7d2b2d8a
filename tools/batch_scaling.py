#!/usr/bin/env python3
"""Launch time against batch size (measurement tooling): a straight line through (batch, time) separates the fixed
cost of a launch (start-up, tail) from the per-request cost -- where a short launch like RT2's 25 000 requests loses
its roofline fraction.  With FIAT_AMD_LIB=.../libfiat_amd_dbg512.so FIAT_AMD_VERBOSE=1 and --lifetimes it runs a few
launches so that the instrumented build prints its wave lifetimes."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="rt2tet")
    ap.add_argument("--batches", default="12500,25000,50000,100000,200000")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--lifetimes", action="store_true")
    args = ap.parse_args()
    import torch
    import bench
    el, sd, deg, order, npts, _ = bench.build_element(args.workload)
    ps = el.device_polyset()
    batches = [int(b) for b in args.batches.split(",")]
    nmax = max(batches)
    pts = torch.as_tensor(bench.synth_points(sd, nmax, npts, 2)).cuda()
    out = torch.empty(ps.out_shape(order, nmax, npts), dtype=torch.float64, device="cuda")
    bytes_per_req = 8 * (npts * sd + int(np.prod(ps.out_shape(order, 1, npts))))
    if args.lifetimes:
        for b in batches:
            print(f"--- batch {b}", file=sys.stderr, flush=True)
            for _ in range(3):
                ps.tabulate_batch(order, pts[:b], out=out[:b])
                torch.cuda.synchronize()
        return
    # clock ramp
    for _ in range(300):
        ps.tabulate_batch(order, pts[:batches[0]], out=out[:batches[0]])
    torch.cuda.synchronize()
    times = {b: [] for b in batches}
    for _ in range(args.rounds):
        for b in batches:
            times[b].append(ps.time_tabulate_batch(order, pts[:b], None, out[:b], args.reps) * 1e3)
    med = {b: float(np.median(times[b])) for b in batches}
    A = np.array([[1.0, b] for b in batches])
    y = np.array([med[b] for b in batches])
    (c0, c1), *_ = np.linalg.lstsq(A, y, rcond=None)
    print(f"{args.workload}: kernel {ps.kernel_name(order, batches[0], npts)}")
    for b in batches:
        print(f"  batch {b:7d}: {med[b]:8.1f} us  {b * bytes_per_req / med[b] / 1e6:6.2f} TB/s  ({b * bytes_per_req / med[b] / 8e6:.3f} of 8 TB/s)")
    print(f"  fit: {c0:.1f} us fixed + {c1 * 1e3:.3f} us per 1000 requests -> asymptotic {bytes_per_req / c1 / 1e6:.2f} TB/s")


if __name__ == "__main__":
    main()
