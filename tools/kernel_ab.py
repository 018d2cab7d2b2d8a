#!/usr/bin/env python3
"""Phase ablation / A-B timing of the tabulation kernel in ONE process
(interleaved rounds, medians) -- measurement tooling, not product code.
FIAT_AMD_DEBUG bits: 1 skip recurrence, 2 skip contraction, 4 skip HBM stores."""
import argparse
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="p3tet")
    ap.add_argument("--batch", type=int, default=100000)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--variants", default="0,1,2,4,3,6,7")
    args = ap.parse_args()
    import torch
    import bench
    el, sd, deg, order, npts, _ = bench.build_element(args.workload)
    ps = el.device_polyset()
    pts = torch.as_tensor(bench.synth_points(sd, args.batch, npts, 2)).cuda()
    out = torch.empty(ps.out_shape(order, args.batch, npts), dtype=torch.float64, device="cuda")
    variants = [int(v) for v in args.variants.split(",")]
    times = {v: [] for v in variants}
    for _ in range(args.rounds):
        for v in variants:
            os.environ["FIAT_AMD_DEBUG"] = str(v)
            times[v].append(ps.time_tabulate_batch(order, pts, None, out, args.reps))
    os.environ["FIAT_AMD_DEBUG"] = "0"
    names = {0: "full", 1: "no recurrence", 2: "no contraction", 4: "no HBM store", 3: "store only",
             6: "recurrence only", 7: "empty loop", 5: "contraction only"}
    for v in variants:
        med = statistics.median(times[v])
        print(f"debug={v:<2d} {names.get(v, ''):<18s} median {med * 1e3:9.1f} us  min {min(times[v]) * 1e3:9.1f} us"
              f"  -> {args.batch / med / 1e3:8.1f} M req/s")


if __name__ == "__main__":
    main()
