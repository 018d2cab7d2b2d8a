#!/usr/bin/env python3
"""Benchmark of the batched tabulate() hot path on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
Lagrange P3 tetrahedron, tabulate order 1 (values + gradient), 23 points per
request (the size of the degree-6 rule), batch of 100 000 independent requests
per GPU, synthetic uniformly random points (seed 2), fp64.

One "step" = one pass of the hot path over the whole batch, inputs and outputs
resident in HBM.  With N > 1 ranks (one process per GPU, torch.distributed over
RCCL) every rank tabulates its own 100 000 requests (weak scaling, no data-path
collective: requests are independent); ``--allgather`` additionally times the
RCCL all-gather that replicates the tables on every GPU and reports it in a
separate object.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# fp64 MFMA: 32 FLOP/clk/SIMD measured (v_mfma_f64_16x16x4 = 64 cycles, tools/ubench2.hip; the fp64 FMA rate)
# x 1024 SIMDs x 2.4 GHz.  The guide's peak table has no fp64 row.
F64_PEAK_TFLOPS = 32 * 1024 * 2.4e9 / 1e12

WORKLOADS = {
    # name: (family, sd, degree, order, npts, default batch)
    "p3tet": ("Lagrange", 3, 3, 1, 23, 100_000),
    "n2tet": ("Nedelec", 3, 2, 1, 23, 25_000),
    "rt2tet": ("RaviartThomas", 3, 2, 1, 23, 25_000),
    "dg6tet": ("DiscontinuousLagrange", 3, 6, 2, 23, 20_000),
    "dg6tet122": ("DiscontinuousLagrange", 3, 6, 2, 122, 8_000),  # C4 stress variant: 122 points (823 kB per request)
    # low-order shapes served by the generic kernel (not BASELINE configs; for tools/kernel_ab.py)
    "p1tet": ("Lagrange", 3, 1, 1, 4, 2_000_000),
    "p2tet": ("Lagrange", 3, 2, 1, 11, 300_000),
    "p4tet": ("Lagrange", 3, 4, 1, 23, 45_000),
    "p2tri": ("Lagrange", 2, 2, 1, 6, 1_250_000),
}


def synth_points(sd, nreq, npts, seed):
    """Uniform points in the UFC simplex: e ~ Exp(1)^(sd+1), x = (e / sum e)[1:]  (SURVEY.md 8d)."""
    rng = np.random.default_rng(seed)
    e = rng.exponential(size=(nreq, npts, sd + 1))
    return (e / e.sum(axis=-1, keepdims=True))[..., 1:].copy()


def host_cpu_share():
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota
    (a GPU box gives each job a share of the host, oversubscribing OpenMP threads thrashes)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except Exception:
            continue
    return max(1, min(n, int(os.environ.get("FIAT_AMD_BENCH_THREADS", "16"))))


def build_element(name):
    """Nodal coefficients through the device Vandermonde path; returns the
    device polynomial set and what the oracle needs for the CPU baseline."""
    import fiat_amd
    fam, sd, deg, order, npts, batch = WORKLOADS[name]
    el = getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(sd), deg)
    return el, sd, deg, order, npts, batch


def cpu_baseline(name, el, sd, deg, order, npts, seconds=12.0):
    """C restatement of FIAT's algorithm (oracle/fiat_oracle.c, OpenMP over requests) on the
    host cores of this box, bounded to ~`seconds` of wall time on a sample of the same workload."""
    from oracle import c_oracle, fiat_oracle as fo
    coeffs = el.get_coeffs()
    verts = fo.UFC_SIMPLEX[sd]
    variant, scale = el._expansion_variant, el._expansion_scale
    cores = host_cpu_share()
    chunk = 4096
    pts = synth_points(sd, chunk, npts, 99)
    c_oracle.tabulate_batch(verts, deg, coeffs, order, pts[:64], scale=scale, variant=variant, nthreads=cores)  # warm
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < seconds:
        c_oracle.tabulate_batch(verts, deg, coeffs, order, pts, scale=scale, variant=variant, nthreads=cores)
        done += chunk
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "tabulations/s", "cores": cores, "kind": "port",
            "sample": f"{done} requests of the same workload in batches of {chunk}, C/OpenMP restatement "
                      f"(oracle/fiat_oracle.c) on {cores} host threads, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="p3tet", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="requests per GPU (default: the workload's)")
    ap.add_argument("--allgather", action="store_true", help="also time the RCCL all-gather of the tables")
    ap.add_argument("--shared-points", action="store_true",
                    help="variant (SURVEY.md 8d): ONE 23-point rule on the reference cell pushed forward to per-request "
                         "physical cells (fx_tabulate_batch_shared) instead of per-request random points")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", type=int, default=-1,
                    help="requests verified against the CPU oracle after timing (-1: the whole batch, 0: none)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the tabulate path has no CPU fallback")
    # one process per GPU over RCCL; FIAT_AMD_BENCH_BACKEND=gloo lets several ranks share a GPU to
    # rehearse the multi-rank path on a one-GPU box (timings are then meaningless)
    backend = os.environ.get("FIAT_AMD_BENCH_BACKEND", "nccl")
    ngpu = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ngpu:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ngpu} GPUs visible")
    device_index = local_rank % max(1, ngpu)
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()

    el, sd, deg, order, npts, batch = build_element(args.workload)
    if args.batch:
        batch = args.batch
    ps = el.device_polyset()
    ntab = ps.out_shape(order, 1, 1)[1]
    rows = ps.ndof * ps.vdim
    bytes_per_req = 8 * (npts * sd + ntab * rows * npts)     # algorithmic bytes

    pts_h = synth_points(sd, batch, npts, seed=2 + rank)
    pts = torch.as_tensor(pts_h).cuda()
    out = torch.empty(ps.out_shape(order, batch, npts), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream()
    verts_h = None
    if args.shared_points:
        # cells: UFC vertices + U(-0.2, 0.2) per coordinate, the rule: 23 fixed points
        from oracle import fiat_oracle as fo
        rng = np.random.default_rng(1000 + rank)
        verts_h = fo.UFC_SIMPLEX[sd][None] + rng.uniform(-0.2, 0.2, size=(batch, sd + 1, sd))
        ref_pts_h = synth_points(sd, 1, npts, seed=6)[0]
        bary = np.concatenate([1.0 - ref_pts_h.sum(axis=1, keepdims=True), ref_pts_h], axis=1)
        pts_h = np.einsum("pv,rvd->rpd", bary, verts_h)          # the same points, per request (for the check)
        verts = torch.as_tensor(verts_h).cuda()
        ref_pts = torch.as_tensor(ref_pts_h).cuda()
        mapping = el.mapping()[0]
        bytes_per_req = 8 * ((sd + 1) * sd + ntab * rows * npts)

        def step():
            ps.tabulate_batch_shared(order, ref_pts, verts, mapping=mapping, out=out)
    else:
        def step():
            ps.tabulate_batch(order, pts, out=out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: average launch duration from HIP events on the launch stream
    if args.shared_points:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nrep = max(5, args.steps)
        e0.record(stream)
        for _ in range(nrep):
            step()
        e1.record(stream)
        torch.cuda.synchronize()
        kernel_ms = e0.elapsed_time(e1) / nrep      # reference tabulation (1 request) + the streaming kernel
    else:
        kernel_ms = ps.time_tabulate_batch(order, pts, None, out, max(5, args.steps), stream=stream)
    achieved = bytes_per_req * batch / (kernel_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": "fxk::shared_points_kernel" if args.shared_points else ps.kernel_name(order, batch, npts),
                "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_request": bytes_per_req, "requests_per_launch": batch}
    # context: what this box writes with a plain fill of the same buffer (the attainable write rate
    # varies between boxes and over time by +-10 %)
    if rank == 0:
        scratch = torch.empty_like(out)
        for _ in range(3):
            scratch.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            scratch.fill_(1.0)
        e1.record()
        torch.cuda.synchronize()
        roofline["fill_same_bytes_gbs"] = scratch.numel() * 8 / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9
        del scratch
    # algorithmic flops: contraction + recurrence; the binding roofline of the large
    # shapes (DG P6 with Hessians: 21 flop/B) is the fp64 pipe, not HBM
    nexp = math.comb(deg + sd, sd)
    flops_per_req = 2 * rows * nexp * npts * ntab + nexp * npts * (5 + 21 * (order >= 1) + 60 * (order >= 2))
    if flops_per_req / bytes_per_req > F64_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS:
        tf = flops_per_req * batch / (kernel_ms * 1e-3) / 1e12
        roofline = dict(roofline, bound="mfma", achieved=tf, peak=F64_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / F64_PEAK_TFLOPS,
                        algorithmic_flops_per_request=flops_per_req,
                        hbm={"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS})
    prof = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
    if not os.path.exists(prof):
        prof = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(prof):
        try:
            with open(prof) as f:
                tr = json.load(f)
            if tr.get("workload") == args.workload and tr.get("batch") == batch and not args.shared_points:
                roofline["traffic"] = tr["hbm_bytes_per_launch"]
        except Exception:
            pass

    # parity of the timed output against the CPU oracle, whole batch (never inside the timed region)
    max_err = None
    if args.check and rank == 0:
        from oracle import c_oracle, fiat_oracle as fo
        ncheck = batch if args.check < 0 else min(args.check, batch)
        ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[sd], deg, el.get_coeffs(), order, pts_h[:ncheck],
                                      verts=None if verts_h is None else verts_h[:ncheck],
                                      scale=el._expansion_scale, variant=el._expansion_variant)
        if args.shared_points and el.mapping()[0] != "affine":     # Piola: phi = M Phi, evaluated in NumPy
            E = np.swapaxes(verts_h[:ncheck, 1:] - verts_h[:ncheck, :1], 1, 2)          # J for the UFC reference cell
            M = np.swapaxes(np.linalg.inv(E), 1, 2) if el.mapping()[0].startswith("cov") else E / np.linalg.det(E)[:, None, None]
            r5 = ref.reshape(ncheck, ref.shape[1], -1, sd, npts)
            ref = np.einsum("rce,rtdep->rtdcp", M, r5).reshape(ref.shape)
        got = out[:ncheck].cpu().numpy().reshape(ref.shape)
        num = np.abs(got - ref).max(axis=(2, 3))
        den = np.maximum(1.0, np.abs(ref).max(axis=(2, 3)))
        max_err = float((num / den).max())

    allgather = None
    if args.allgather and world > 1:
        gathered = torch.empty((world,) + tuple(out.shape), dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(gathered, out)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        reps = max(2, args.steps // 4)
        for _ in range(reps):
            ps.tabulate_batch(order, pts, out=out)
            dist.all_gather_into_tensor(gathered, out)
        torch.cuda.synchronize()
        barrier()
        dt = (time.perf_counter() - t0) / reps
        allgather = {"ms_per_step_with_allgather": dt * 1e3,
                     "value_with_allgather": batch * world / dt,
                     "gathered_bytes_per_gpu": out.numel() * 8 * world}

    if rank == 0:
        line = {
            "metric": "element tabulations/sec (basis+grad, fp64) for batched P3 tet"
                      if args.workload == "p3tet" else f"element tabulations/sec ({args.workload})",
            "value": batch * world * args.steps / elapsed,
            "unit": "tabulations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{WORKLOADS[args.workload][0]} degree {deg} "
                                   f"{'tetrahedron' if sd == 3 else 'triangle'}, order {order}, "
                                   f"{npts} points/request, batch {batch} per GPU",
                       "requests_per_gpu": batch, "points_per_request": npts, "order": order,
                       "points": "one 23-point rule on the reference cell, per-request physical cells" if args.shared_points
                                 else "random per request",
                       "sharding": "independent requests, contiguous blocks per rank, no data-path collective"},
            "roofline": roofline,
            "max_rel_err_vs_oracle": max_err,
        }
        if allgather:
            line["allgather"] = allgather
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.workload, el, sd, deg, order, npts)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
