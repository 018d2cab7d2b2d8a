"""Throughput of macro-element tabulation (fx_macro_tabulate_batch) on one GPU: HIP-event time per launch,
algorithmic bytes 8*(npts*sd + ntab*ndof*npts) per request against the 8 TB/s HBM peak, and a parity spot
check against the oracle.  usage: python tools/bench_macro.py [--reps 50] [--only name]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    # name: (family, sd, degree, variant, order, npts, batch)
    "p2isop1_tet": ("Lagrange", 3, 1, "equispaced,iso", 1, 23, 200_000),
    "p2isop1_tri": ("Lagrange", 2, 1, "equispaced,iso", 1, 12, 600_000),
    "cg2_alfeld_tri": ("Lagrange", 2, 2, "equispaced,alfeld", 1, 12, 400_000),
    "cg3_alfeld_tet": ("Lagrange", 3, 3, "equispaced,alfeld", 1, 23, 50_000),
    "cg2_iso_tet": ("Lagrange", 3, 2, "equispaced,iso", 1, 23, 50_000),
    "cg2_iso_tet_hess": ("Lagrange", 3, 2, "equispaced,iso", 2, 23, 20_000),
    "dg1_alfeld_tet": ("DiscontinuousLagrange", 3, 1, "equispaced_interior,alfeld", 1, 23, 100_000),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--only", default=None)
    ap.add_argument("--cpu-baseline", type=float, default=0.0, metavar="SECONDS",
                    help="also time the NumPy oracle (oracle/fiat_oracle.py macro_element_tabulate, one core) on a sample "
                         "of the same requests for about SECONDS per case")
    ap.add_argument("--debug", type=int, default=0, help="ablation bits of the lane-local kernel (FIAT_AMD_DEBUG), set "
                    "after the element has been constructed: 1 no binning, 2 no contraction, 4 no HBM stores")
    args = ap.parse_args()
    import fiat_amd
    from oracle import fiat_oracle as fo
    for name, (fam, sd, deg, variant, order, npts, batch) in CASES.items():
        if args.only and name != args.only:
            continue
        el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(sd), deg, variant)
        ps = el.device_polyset()
        if args.debug:
            os.environ["FIAT_AMD_DEBUG"] = str(args.debug)
        rng = np.random.default_rng(7)
        e = rng.exponential(size=(batch, npts, sd + 1))
        pts_h = (e / e.sum(-1, keepdims=True))[..., 1:].copy()
        pts = torch.as_tensor(pts_h).cuda()
        out = torch.empty(ps.out_shape(order, batch, npts), dtype=torch.float64, device="cuda")
        for _ in range(5):
            ps.tabulate_batch(order, pts, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            ps.tabulate_batch(order, pts, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        ntab = out.shape[1]
        bytes_per_req = 8 * (npts * sd + ntab * ps.ndof * npts)
        gbs = bytes_per_req * batch / (ms * 1e-3) / 1e9
        # parity sample
        S = el.get_reference_complex()
        es = el.get_nodal_basis().get_expansion_set()
        top = S.get_topology()
        V = np.array(S.get_vertices())
        cells = [V[list(top[sd][c])] for c in sorted(top[sd])]
        got = out[:16].cpu().numpy()
        worst = 0.0
        for r in range(16):
            ref = fo.macro_element_tabulate(np.array(S.get_parent().get_vertices()), cells, es.get_cell_node_map(deg), deg,
                                            el.get_coeffs(), order, pts_h[r], es.scale, es.variant)
            for t, a in enumerate(fo.jet_indices(sd, order)):
                worst = max(worst, np.max(np.abs(got[r, t] - ref[a])) / max(1.0, np.max(np.abs(ref[a]))))
        os.environ.pop("FIAT_AMD_DEBUG", None)
        cpu = None
        if args.cpu_baseline > 0:
            import time
            parent = np.array(S.get_parent().get_vertices())
            cmap, coeffs = es.get_cell_node_map(deg), el.get_coeffs()
            t0, done = time.perf_counter(), 0
            while time.perf_counter() - t0 < args.cpu_baseline:
                fo.macro_element_tabulate(parent, cells, cmap, deg, coeffs, order, pts_h[done % batch], es.scale, es.variant)
                done += 1
            dt = time.perf_counter() - t0
            cpu = {"value": done / dt, "unit": "tabulations/s", "cores": 1, "kind": "port",
                   "sample": f"{done} requests of the same workload, NumPy restatement of FIAT's macro tabulation "
                             f"(oracle/fiat_oracle.py) on 1 host thread, {dt:.1f} s"}
        print(json.dumps({"metric": f"element tabulations/sec ({name})", "value": batch / (ms * 1e-3), "unit": "tabulations/s",
                          "n_gpus": 1, "dtype": "f64", "data": "synthetic",
                          "config": {"workload": f"{fam} degree {deg} variant {variant}, order {order}, {npts} points/request, "
                                                 f"batch {batch}"},
                          "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                                       "traffic": None, "kernel": ps.kernel_name(order, batch, npts), "kernel_ms": ms,
                                       "algorithmic_bytes_per_request": bytes_per_req, "requests_per_launch": batch},
                          "cpu_baseline": cpu, "max_rel_err_vs_oracle": worst}), flush=True) if args.cpu_baseline > 0 else None
        print(json.dumps({"case": name, "ncell": len(cells), "ndof": ps.ndof, "order": order, "npts": npts, "batch": batch,
                          "ms": round(ms, 4), "tab_per_s": batch / (ms * 1e-3), "GBps": round(gbs, 1),
                          "frac_hbm": round(gbs / 8000.0, 4), "max_rel_err": worst}), flush=True)


if __name__ == "__main__":
    main()
