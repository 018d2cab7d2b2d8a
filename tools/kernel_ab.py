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
    ap.add_argument("--verts", action="store_true", help="per-request cell geometry (random affine images of the UFC cell)")
    ap.add_argument("--kernels", default="", help="comma list of FIAT_AMD_KERNEL values to compare (debug=0 only)")
    args = ap.parse_args()
    import torch
    import bench
    el, sd, deg, order, npts, _ = bench.build_element(args.workload)
    ps = el.device_polyset()
    pts = torch.as_tensor(bench.synth_points(sd, args.batch, npts, 2)).cuda()
    out = torch.empty(ps.out_shape(order, args.batch, npts), dtype=torch.float64, device="cuda")
    verts = None
    if args.verts:
        rng = np.random.default_rng(7)
        ref = np.array(el.get_reference_element().get_vertices(), dtype=float)
        Aff = np.eye(sd) + 0.2 * rng.standard_normal((args.batch, sd, sd))
        verts = torch.as_tensor(np.einsum("vd,red->rve", ref, Aff) + rng.standard_normal((args.batch, 1, sd))).cuda()
    def ceiling():
        """HBM write ceiling of THIS box, same bytes as one launch: torch fill_ of the output buffer."""
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            out.fill_(1.0)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps
    ceiling()
    variants = [int(v) for v in args.variants.split(",")]
    times = {v: [] for v in variants}
    for _ in range(args.rounds):
        for v in variants:
            os.environ["FIAT_AMD_DEBUG"] = str(v)
            times[v].append(ps.time_tabulate_batch(order, pts, verts, out, args.reps))
    os.environ["FIAT_AMD_DEBUG"] = "0"
    if args.kernels:
        ks = args.kernels.split(",")
        kt = {k: [] for k in ks}
        for _ in range(args.rounds):
            for k in ks:
                ps.ctx.set_policy(*({"image": ["kernel_image"], "stream": ["kernel_stream"]}.get(k, [])))
                kt[k].append(ps.time_tabulate_batch(order, pts, verts, out, args.reps))
        ps.ctx.set_policy()
        cl = statistics.median(ceiling() for _ in range(args.rounds))
        print(f"fill_ of the output buffer on this box: {cl * 1e3:9.1f} us")
        for k in ks:
            med = statistics.median(kt[k])
            print(f"kernel={k:<8s} ({ps.kernel_name(order, args.batch, npts)} by default) median {med * 1e3:9.1f} us"
                  f"  min {min(kt[k]) * 1e3:9.1f} us  -> {args.batch / med / 1e3:8.1f} M req/s")
        return
    cl = statistics.median(ceiling() for _ in range(args.rounds))
    print(f"fill_ of the output buffer on this box: {cl * 1e3:9.1f} us")
    names = {0: "full", 1: "no recurrence", 2: "no contraction", 4: "no HBM store", 3: "store only",
             6: "recurrence only", 7: "empty loop", 5: "contraction only"}
    for v in variants:
        med = statistics.median(times[v])
        print(f"debug={v:<2d} {names.get(v, ''):<18s} median {med * 1e3:9.1f} us  min {min(times[v]) * 1e3:9.1f} us"
              f"  -> {args.batch / med / 1e3:8.1f} M req/s")


if __name__ == "__main__":
    main()
