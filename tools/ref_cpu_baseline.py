#!/usr/bin/env python3
"""FIAT's own NumPy path timed on the host cores of the BUILD CONTAINER (the reference cannot travel to the GPU box):

    PYTHONPATH=oracle/restated_deps:/root/reference OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 \
        python -B tools/ref_cpu_baseline.py [--seconds 3] [--procs 1,8]

Imports the UNMODIFIED reference (``recursivenodes`` restated under oracle/restated_deps) and measures
element tabulations per second for the BASELINE configs, entry point CiarletElement.tabulate
(FIAT/finite_element.py:181) / TensorProductElement.tabulate (FIAT/tensor_product.py:231), in two modes:
  single   one tabulate() call per request (what a per-cell consumer would do),
  x100     the points of 100 requests concatenated into one call (the reference's best case),
with 1 process and with one worker process per core (requests are independent).  Uniformly random interior points,
numpy.random.default_rng(seed = 2 + worker).  Prints a Markdown table (pasted into BASELINE.md section 3)."""
import argparse
import multiprocessing as mp
import os
import platform
import time

import numpy as np

CONFIGS = [
    # name, factory (as a string: evaluated in the worker), sd, order, npts
    ("C1 Lagrange P1 triangle, order 1, 3 pts", "FIAT.Lagrange(FIAT.ufc_simplex(2), 1)", 2, 1, 3),
    ("C2 Lagrange P3 tetrahedron, order 1, 23 pts", "FIAT.Lagrange(FIAT.ufc_simplex(3), 3)", 3, 1, 23),
    ("C3 Nedelec N2 tetrahedron, order 1, 23 pts", "FIAT.Nedelec(FIAT.ufc_simplex(3), 2)", 3, 1, 23),
    ("C3 Raviart-Thomas RT2 tetrahedron, order 1, 23 pts", "FIAT.RaviartThomas(FIAT.ufc_simplex(3), 2)", 3, 1, 23),
    ("C4 DG P6 tetrahedron, order 2, 23 pts", "FIAT.DiscontinuousLagrange(FIAT.ufc_simplex(3), 6)", 3, 2, 23),
    ("C5 P4 x P4 x P4 hexahedron, order 1, 125 pts",
     "FIAT.TensorProductElement(FIAT.TensorProductElement(FIAT.Lagrange(FIAT.ufc_simplex(1), 4), "
     "FIAT.Lagrange(FIAT.ufc_simplex(1), 4)), FIAT.Lagrange(FIAT.ufc_simplex(1), 4))", -3, 1, 125),
]


def points(rng, sd, n):
    if sd < 0:                                  # hexahedron: uniform in the unit cube
        return rng.uniform(0.0, 1.0, size=(n, -sd))
    e = rng.exponential(size=(n, sd + 1))
    return (e / e.sum(axis=1, keepdims=True))[:, 1:].copy()


def work(args):
    factory, sd, order, npts, group, seconds, worker = args
    import FIAT  # noqa: F401  (the unmodified reference)
    el = eval(factory)
    rng = np.random.default_rng(2 + worker)
    pts = points(rng, sd, npts * group)
    el.tabulate(order, pts)                     # warm
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        el.tabulate(order, pts)
        done += group
    return done, time.perf_counter() - t0


def rate(cfg, group, procs, seconds):
    _, factory, sd, order, npts = cfg
    jobs = [(factory, sd, order, npts, group, seconds, w) for w in range(procs)]
    if procs == 1:
        res = [work(jobs[0])]
    else:
        with mp.get_context("spawn").Pool(procs) as pool:
            res = pool.map(work, jobs)
    return sum(d for d, _ in res) / max(t for _, t in res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--procs", default=f"1,{os.cpu_count()}")
    args = ap.parse_args()
    procs = [int(p) for p in args.procs.split(",")]
    cpu = platform.processor() or "x86_64"
    try:
        cpu = [line.split(":", 1)[1].strip() for line in open("/proc/cpuinfo") if line.startswith("model name")][0]
    except Exception:
        pass
    print(f"<!-- tools/ref_cpu_baseline.py: {cpu}, {os.cpu_count()} logical CPUs, OMP/OPENBLAS threads = "
          f"{os.environ.get('OMP_NUM_THREADS', 'unset')}, NumPy {np.__version__}, {args.seconds:g} s per cell -->")
    head = "| config | " + " | ".join(f"{mode}, {p} proc" for mode in ("single", "x100") for p in procs) + " |"
    print(head)
    print("|---|" + "---|" * (2 * len(procs)))
    for cfg in CONFIGS:
        cells = []
        for group in (1, 100):
            g = group if cfg[4] < 100 else min(group, 20)       # 125-point requests: 20 per call at most
            for p in procs:
                cells.append(f"{rate(cfg, g, p, args.seconds):.3g}")
        print(f"| {cfg[0]} | " + " | ".join(cells) + " |", flush=True)


if __name__ == "__main__":
    main()
