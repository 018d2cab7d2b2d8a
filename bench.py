#!/usr/bin/env python3
"""Benchmark of the batched tabulate() hot path on MI355X.

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on): Lagrange P3
tetrahedron, tabulate order 1 (values + gradient), 23 points per request (the size of the degree-6 rule),
100 000 independent requests per GPU, synthetic uniformly random points (seed 2), fp64.  The other BASELINE
configs are parity-test cases; ``--workload c3|n2tet|rt2tet|dg6tet|hex ...`` times them with the same line format.

One "step" = one pass of the hot path over the whole batch, inputs and outputs resident in HBM.

Multi-GPU (one process per GPU): either launched by ``python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N`` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or bare ``python bench.py --gpus N``,
which starts the N ranks itself BEFORE anything touches the GPU.  Every rank tabulates its own block of requests
(weak scaling, no data-path collective: requests are independent); ``value`` is that compute-only aggregate.  With
N > 1 the line also carries ``allgather``: the same step followed by / overlapped with the RCCL exchange that
replicates all tables on every GPU (fiat_amd/distributed.py -> fx_allgather_tables), timed separately.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import signal
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# fp64 MFMA: 32 FLOP/clk/SIMD measured (v_mfma_f64_16x16x4 = 64 cycles, tools/ubench2.hip; the fp64 FMA rate)
# x 1024 SIMDs x 2.4 GHz.  The guide's peak table has no fp64 row.
F64_PEAK_TFLOPS = 32 * 1024 * 2.4e9 / 1e12
PEER_GRACE_S = 15.0     # ranks != 0 keep waiting this much longer than rank 0's all-gather watchdog (run(): rank 0 prints)
TERM_GRACE_S = 5.0      # spawn_ranks: after the first failed rank the others get this long to leave on their own
CLOCK_RAMP_MS = 60.0    # untimed launches before the W warm-up steps: the shader clock needs tens of ms after idle

WORKLOADS = {
    # name: (family, sd, degree, order, npts, default batch per GPU)
    "p3tet": ("Lagrange", 3, 3, 1, 23, 100_000),                 # BASELINE configs[1] -- the headline
    "n2tet": ("Nedelec", 3, 2, 1, 23, 25_000),                   # configs[2], first half
    "rt2tet": ("RaviartThomas", 3, 2, 1, 23, 25_000),            # configs[2], second half
    "c3": ("Nedelec+RaviartThomas", 3, 2, 1, 23, 50_000),        # configs[2]: 25 000 N2 + 25 000 RT2 per step
    "dg6tet": ("DiscontinuousLagrange", 3, 6, 2, 23, 125_000),   # configs[3]: 1 M over 8 GPUs = 125 000 per GPU
    "dg6tet122": ("DiscontinuousLagrange", 3, 6, 2, 122, 8_000),  # C4 stress variant: 122 points (823 kB per request)
    "hex": ("P4 x P4 x P4", 3, 4, 1, 125, 200_000),              # configs[4]: batch 200 k (100 GB of tables: fits one GPU), 5^3 tensor grid
    "hex25k": ("P4 x P4 x P4", 3, 4, 1, 125, 25_000),            # the round-1/2 reading of configs[4] (200 k over 8 GPUs), kept for comparison
    # low-order shapes (not BASELINE configs; for tools/kernel_ab.py)
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
    """(element, sd, degree, order, npts, default batch) of a single-family workload (tools/ use this)."""
    import fiat_amd
    fam, sd, deg, order, npts, batch = WORKLOADS[name]
    return getattr(fiat_amd, fam)(fiat_amd.ufc_simplex(sd), deg), sd, deg, order, npts, batch


# ---------------------------------------------------------------------------------------------------
# workloads: each owns its device buffers and knows its algorithmic bytes, its kernel(s) and its oracle
class SimplexWorkload:
    """One or more simplex elements tabulated at per-request random points (or, ``shared``, at one rule pushed
    forward to per-request cells).  parts = [(element, requests)]; a step launches every part once."""

    def __init__(self, name, batch, rank, shared=False):
        import fiat_amd
        import torch
        fam, sd, deg, order, npts, default = WORKLOADS[name]
        self.name, self.sd, self.deg, self.order, self.npts = name, sd, deg, order, npts
        self.batch = batch or default
        self.shared = shared
        fams = fam.split("+")
        share = self.batch // len(fams)
        self.parts = []
        for i, f in enumerate(fams):
            el = getattr(fiat_amd, f)(fiat_amd.ufc_simplex(sd), deg)
            ps = el.device_polyset()
            n = share if i + 1 < len(fams) else self.batch - share * (len(fams) - 1)
            pts_h = synth_points(sd, n, npts, seed=2 + rank + 1000 * i)
            part = {"family": f, "el": el, "ps": ps, "n": n, "pts_h": pts_h, "verts_h": None,
                    "out": torch.empty(ps.out_shape(order, n, npts), dtype=torch.float64, device="cuda")}
            rows = ps.ndof * ps.vdim
            ntab = ps.out_shape(order, 1, 1)[1]
            part["rows"], part["ntab"] = rows, ntab
            if shared:
                rng = np.random.default_rng(1000 + rank + i)
                ufc = np.array(fiat_amd.ufc_simplex(sd).get_vertices(), dtype=float)
                part["verts_h"] = ufc[None] + rng.uniform(-0.2, 0.2, size=(n, sd + 1, sd))
                ref_h = synth_points(sd, 1, npts, seed=6)[0]
                bary = np.concatenate([1.0 - ref_h.sum(axis=1, keepdims=True), ref_h], axis=1)
                part["pts_h"] = np.einsum("pv,rvd->rpd", bary, part["verts_h"])   # the same points, for the check
                part["verts"] = torch.as_tensor(part["verts_h"]).cuda()
                part["ref_pts"] = torch.as_tensor(ref_h).cuda()
                part["bytes_per_req"] = 8 * ((sd + 1) * sd + ntab * rows * npts)
            else:
                part["pts"] = torch.as_tensor(pts_h).cuda()
                part["bytes_per_req"] = 8 * (npts * sd + ntab * rows * npts)      # algorithmic bytes
            nexp = math.comb(deg + sd, sd)
            part["flops_per_req"] = 2 * rows * nexp * npts * ntab + nexp * npts * (5 + 21 * (order >= 1) + 60 * (order >= 2))
            self.parts.append(part)

    def describe(self):
        cell = "tetrahedron" if self.sd == 3 else "triangle"
        fam = " + ".join(p["family"] for p in self.parts)
        return (f"{fam} degree {self.deg} {cell}, order {self.order}, {self.npts} points/request, "
                f"batch {self.batch} per GPU")

    def step(self):
        for p in self.parts:
            if self.shared:
                p["ps"].tabulate_batch_shared(self.order, p["ref_pts"], p["verts"], mapping=p["el"].mapping()[0], out=p["out"])
            else:
                p["ps"].tabulate_batch(self.order, p["pts"], out=p["out"])

    def produce_rows(self, lo, hi, rows):
        """Tabulate requests [lo, hi) of the (single-part) batch into ``rows`` (a view of a gather buffer)."""
        p = self.parts[0]
        p["ps"].tabulate_batch(self.order, p["pts"][lo:hi], out=rows)

    def kernel_times(self, reps, stream):
        """[(kernel name, ms per launch, algorithmic bytes per launch, flops per launch)] from HIP events on the
        launch stream."""
        import torch
        res = []
        for p in self.parts:
            if self.shared:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(reps):
                    p["ps"].tabulate_batch_shared(self.order, p["ref_pts"], p["verts"], mapping=p["el"].mapping()[0], out=p["out"])
                e1.record(stream)
                torch.cuda.synchronize()
                ms, kern = e0.elapsed_time(e1) / reps, "fxk::shared_points_kernel"   # reference tabulation + streaming kernel
            else:
                ms = p["ps"].time_tabulate_batch(self.order, p["pts"], None, p["out"], reps, stream=stream)
                kern = p["ps"].kernel_name(self.order, p["n"], self.npts)
            res.append((kern, ms, p["bytes_per_req"] * p["n"], p["flops_per_req"] * p["n"]))
        return res

    def max_rel_err(self, ncheck):
        from oracle import c_oracle, fiat_oracle as fo
        worst = 0.0
        for p in self.parts:
            # default: the whole batch while its tables stay below ~2 GB on the host, else the leading requests
            n = min(p["n"], max(1024, int(2e9 // p["bytes_per_req"]))) if ncheck < 0 else min(ncheck, p["n"])
            self.checked = getattr(self, "checked", 0) + n
            el = p["el"]
            ref = c_oracle.tabulate_batch(fo.UFC_SIMPLEX[self.sd], self.deg, el.get_coeffs(), self.order, p["pts_h"][:n],
                                          verts=None if p["verts_h"] is None else p["verts_h"][:n],
                                          scale=el._expansion_scale, variant=el._expansion_variant)
            if self.shared and el.mapping()[0] != "affine":     # Piola: phi = M Phi, evaluated in NumPy
                E = np.swapaxes(p["verts_h"][:n, 1:] - p["verts_h"][:n, :1], 1, 2)       # J for the UFC reference cell
                M = np.swapaxes(np.linalg.inv(E), 1, 2) if el.mapping()[0].startswith("cov") else E / np.linalg.det(E)[:, None, None]
                r5 = ref.reshape(n, ref.shape[1], -1, self.sd, self.npts)
                ref = np.einsum("rce,rtdep->rtdcp", M, r5).reshape(ref.shape)
            got = p["out"][:n].cpu().numpy().reshape(ref.shape)
            num = np.abs(got - ref).max(axis=(2, 3))
            den = np.maximum(1.0, np.abs(ref).max(axis=(2, 3)))
            worst = max(worst, float((num / den).max()))
        return worst

    def cpu_baseline(self, seconds=12.0):
        """C restatement of FIAT's algorithm (oracle/fiat_oracle.c, OpenMP over requests) on the host cores of this
        box, bounded to ~``seconds`` of wall time on a sample of the same workload."""
        from oracle import c_oracle, fiat_oracle as fo
        cores = host_cpu_share()
        chunk = 4096
        done, t0 = 0, time.perf_counter()
        jobs = []
        for p in self.parts:
            el = p["el"]
            pts = synth_points(self.sd, chunk, self.npts, 99)
            args = (fo.UFC_SIMPLEX[self.sd], self.deg, el.get_coeffs(), self.order)
            kw = dict(scale=el._expansion_scale, variant=el._expansion_variant, nthreads=cores)
            c_oracle.tabulate_batch(*args, pts[:64], **kw)      # warm
            jobs.append((args, pts, kw))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for args, pts, kw in jobs:                          # the mix of the workload, in equal shares
                c_oracle.tabulate_batch(*args, pts, **kw)
                done += chunk
        dt = time.perf_counter() - t0
        return {"value": done / dt, "unit": "tabulations/s", "cores": cores, "kind": "port",
                "sample": f"{done} requests of the same workload in batches of {chunk}, C/OpenMP restatement "
                          f"(oracle/fiat_oracle.c) on {cores} host threads, {dt:.1f} s"}


class HexWorkload:
    """configs[4]: P4 x P4 x P4 hexahedron, order 1, per-request 5^3 tensor grid (sum-factorised input:
    per-request 1-D coordinates), fx_tensor_tabulate_grid_batch."""

    def __init__(self, name, batch, rank, shared=False):
        import fiat_amd
        import torch
        self.name, self.sd, self.deg, self.order, self.npts = name, 3, 4, 1, 125
        self.batch = batch or WORKLOADS[name][5]
        P4 = fiat_amd.Lagrange(fiat_amd.ufc_simplex(1), 4)
        self.P4 = P4
        self.el = fiat_amd.TensorProductElement(fiat_amd.TensorProductElement(P4, P4), P4)
        rng = np.random.default_rng(5 + rank)
        self.grid_h = np.sort(rng.uniform(0, 1, size=(self.batch, 3, 5)), axis=2)
        self.grid = torch.as_tensor(self.grid_h).cuda()
        self.out = torch.empty((self.batch, 4, 125, 125), dtype=torch.float64, device="cuda")
        self.bytes_per_req = 8 * (15 + 4 * 125 * 125)           # SURVEY.md 8(d), factored 1-D inputs
        self.flops_per_req = 2 * 4 * 125 * 125
        self.parts = [None]

    def describe(self):
        return f"P4 x P4 x P4 hexahedron (nested TensorProductElement), order 1, 5^3 tensor grid/request, batch {self.batch} per GPU"

    def step(self):
        self.el.tabulate_batch(1, self.grid, out=self.out, grid=True)

    def produce_rows(self, lo, hi, rows):
        self.el.tabulate_batch(1, self.grid[lo:hi], out=rows.view(hi - lo, 4, 125, 125), grid=True)

    def kernel_times(self, reps, stream):
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            self.step()
        e1.record(stream)
        torch.cuda.synchronize()
        return [("fxk::tensor_tabulate_kernel<GRID>", e0.elapsed_time(e1) / reps, self.bytes_per_req * self.batch,
                 self.flops_per_req * self.batch)]

    def _oracle(self, r):
        from oracle import fiat_oracle as fo
        g = self.grid_h[r]
        pts = np.array([[x, y, z] for x in g[0] for y in g[1] for z in g[2]])
        nodes = np.array(self.P4.get_nodal_basis().get_expansion_set().x)
        tab = fo.hex_lagrange_tabulate(nodes, 1, pts)
        return np.stack([tab[a] for a in fo.jet_indices(3, 1)])

    def max_rel_err(self, ncheck):
        n = 64 if ncheck < 0 else min(ncheck, self.batch)     # NumPy oracle: a sample (full sizes: tests/test_gpu_fullsize.py)
        idx = np.unique(np.linspace(0, self.batch - 1, n).astype(int))
        self.checked = len(idx)
        worst = 0.0
        for r in idx:
            ref = self._oracle(r)
            got = self.out[r].cpu().numpy()
            worst = max(worst, float((np.abs(got - ref).max(axis=(1, 2)) / np.maximum(1.0, np.abs(ref).max(axis=(1, 2)))).max()))
        return worst

    def cpu_baseline(self, seconds=12.0):
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            self._oracle(done % self.batch)
            done += 1
        dt = time.perf_counter() - t0
        return {"value": done / dt, "unit": "tabulations/s", "cores": 1, "kind": "port",
                "sample": f"{done} requests of the same workload, NumPy restatement (oracle/fiat_oracle.py "
                          f"hex_lagrange_tabulate) on 1 host thread, {dt:.1f} s"}


# ---------------------------------------------------------------------------------------------------
def spawn_ranks(nranks, cmd=None, need_gpus=True):
    """``bench.py --gpus N`` without a launcher: start the N ranks (``cmd``, default this script with its own
    arguments) from a parent that never touches the GPU (torch.cuda.device_count() does not initialise it on this
    image)."""
    backend = os.environ.get("FIAT_AMD_BENCH_BACKEND", "nccl")
    if need_gpus and backend == "nccl":
        import torch
        ngpu = torch.cuda.device_count()
        if ngpu < nranks:
            print(f"bench.py: --gpus {nranks} but only {ngpu} GPU(s) visible (RCCL needs one GPU per rank; "
                  f"FIAT_AMD_BENCH_BACKEND=gloo rehearses the multi-rank path on fewer)", file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(cmd or [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stderr=subprocess.PIPE, text=True, bufsize=1))

    def relay(r, pipe):     # every line of a child's stderr carries its rank
        for text in pipe:
            sys.stderr.write(f"[rank {r}] {text}")
            sys.stderr.flush()

    relays = [threading.Thread(target=relay, args=(r, p.stderr), daemon=True) for r, p in enumerate(procs)]
    for t in relays:
        t.start()
    # wait for all; once one rank has failed the others (they would wait in a collective for ever) get TERM_GRACE_S to
    # notice and leave on their own -- rank 0 with its line -- then SIGTERM (run() turns that into the line + status 3),
    # then SIGKILL
    rc, failed, t_failed, stage = 0, None, 0.0, 0
    live = set(range(nranks))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = max(rc, abs(code))
                if failed is None:
                    failed, t_failed = r, time.monotonic()
                    print(f"bench.py: rank {r} exited with status {code}; the other ranks have {TERM_GRACE_S:.0f} s to leave",
                          file=sys.stderr)
        if failed is not None and live:
            waited = time.monotonic() - t_failed
            if stage == 0 and waited > TERM_GRACE_S:
                stage = 1
                print(f"bench.py: terminating the other ranks {sorted(live)}", file=sys.stderr)
                for q in live:
                    procs[q].terminate()
            elif stage == 1 and waited > TERM_GRACE_S + 10.0:
                stage = 2
                for q in live:
                    procs[q].kill()
        if live:
            time.sleep(0.05)
    for t in relays:
        t.join(timeout=2.0)
    return rc


def run(args):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report n_gpus for a job of another size")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the tabulate path has no CPU fallback")
    # one process per GPU over RCCL; FIAT_AMD_BENCH_BACKEND=gloo lets several ranks share a GPU to rehearse the
    # multi-rank path on a one-GPU box (timings are then meaningless)
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
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    def barrier():
        if world > 1:
            dist.barrier()

    if args.policy:
        from fiat_amd import runtime
        runtime.Context.get().set_policy(*args.policy.split(","))
    cls = HexWorkload if args.workload.startswith("hex") else SimplexWorkload
    wl = cls(args.workload, args.batch, rank, shared=args.shared_points)
    batch = wl.batch
    stream = torch.cuda.current_stream()

    # the driver's protocol WITHOUT the ramp, for the record: W warm-up + K timed launches from an idle GPU
    torch.cuda.synchronize()
    time.sleep(0.5)
    for _ in range(args.warmup):
        wl.step()
    torch.cuda.synchronize()
    t_cold = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    torch.cuda.synchronize()
    ms_per_step_cold = (time.perf_counter() - t_cold) / args.steps * 1e3

    # clock ramp (disclosed in the line, not part of W): the first launches after idle run ~25 % slower
    t_ramp = time.perf_counter()
    ramp_steps = 0
    while (time.perf_counter() - t_ramp) * 1e3 < CLOCK_RAMP_MS:
        wl.step()
        ramp_steps += 1
        if ramp_steps % 8 == 0:
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        wl.step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel(s): average launch duration from HIP events on the launch stream
    kt = wl.kernel_times(max(5, args.steps), stream)
    kernel_ms = sum(k[1] for k in kt)
    abytes = sum(k[2] for k in kt)
    aflops = sum(k[3] for k in kt)
    achieved = abytes / (kernel_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": kt[0][0] if len(kt) == 1 else [k[0] for k in kt],
                "kernel_ms": kernel_ms if len(kt) == 1 else [k[1] for k in kt],
                "algorithmic_bytes_per_launch": abytes if len(kt) == 1 else [k[2] for k in kt],
                "requests_per_launch": batch if len(kt) == 1 else [p["n"] for p in wl.parts]}
    # the driver's own protocol without the clock ramp (W + K launches from an idle GPU), same algorithmic bytes
    roofline["frac_cold"] = abytes / (ms_per_step_cold * 1e-3) / 1e9 / HBM_PEAK_GBS
    if len(kt) > 1:
        roofline["per_kernel_frac"] = [k[2] / (k[1] * 1e-3) / 1e9 / HBM_PEAK_GBS for k in kt]
    # context: what this box writes with a plain fill of the same bytes (the attainable write rate varies
    # between boxes and over time by +-10 %)
    if rank == 0:
        scratch = torch.empty(int(abytes // 8), dtype=torch.float64, device="cuda")
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
    # the binding roofline of the large shapes (DG P6 with Hessians: 21 flop/B) is the fp64 pipe, not HBM
    if aflops / abytes > F64_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS:
        tf = aflops / (kernel_ms * 1e-3) / 1e12
        roofline = dict(roofline, bound="mfma", achieved=tf, peak=F64_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / F64_PEAK_TFLOPS,
                        frac_cold=aflops / (ms_per_step_cold * 1e-3) / 1e12 / F64_PEAK_TFLOPS,
                        algorithmic_flops_per_launch=aflops,
                        hbm={"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS})
    # north_star asks for both: achieved HBM GB/s AND fp64 (MFMA) utilisation against the gfx950 peak
    tf_alg = aflops / (kernel_ms * 1e-3) / 1e12
    roofline["mfma"] = {"algorithmic_tflops": tf_alg, "peak_tflops": F64_PEAK_TFLOPS, "frac": tf_alg / F64_PEAK_TFLOPS,
                        "flop_per_byte": aflops / abytes, "busy_frac_pmc": None}
    prof = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
    if os.path.exists(prof) and not args.shared_points:
        try:
            with open(prof) as f:
                tr = json.load(f)
            if tr.get("workload") == args.workload and tr.get("batch") == batch:
                roofline["traffic"] = tr["hbm_bytes_per_launch"]
                if tr.get("mfma_busy_frac") is not None:
                    # SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles) of the profile the traffic comes from
                    roofline["mfma"]["busy_frac_pmc"] = tr["mfma_busy_frac"]
                roofline["traffic_source"] = (f"profiles/traffic_{args.workload}.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                              "of this command on an earlier box, NOT measured in this run")
        except Exception:
            pass

    # parity of the timed output against the CPU oracle (never inside the timed region)
    max_err = None
    if args.check and rank == 0:
        max_err = wl.max_rel_err(args.check)

    line = None
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
            "ms_per_step_cold": ms_per_step_cold,      # same W + K launches from an idle GPU, before the clock ramp (this rank)
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl.describe(), "requests_per_gpu": batch, "points_per_request": wl.npts, "order": wl.order,
                       "points": "one 23-point rule on the reference cell, per-request physical cells" if args.shared_points
                                 else "random per request",
                       "sharding": "independent requests, contiguous blocks per rank, no data-path collective",
                       "world_size": world, "launch": "self-spawned ranks" if os.environ.get("FIAT_AMD_BENCH_SPAWNED") else
                                     ("torch.distributed.run" if world > 1 else "single process"),
                       "clock_ramp_ms_before_warmup": CLOCK_RAMP_MS, "clock_ramp_steps": ramp_steps,
                       **({"policy": args.policy} if args.policy else {})},
            "roofline": roofline,
            "max_rel_err_vs_oracle": max_err,
            "requests_checked_vs_oracle": getattr(wl, "checked", None),
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = wl.cpu_baseline()

    if world > 1 and not args.no_allgather and len(wl.parts) == 1 and not args.shared_points:
        failed = guarded_leg(line, rank, args.allgather_timeout, barrier,
                             lambda: allgather_leg(wl, world, rank, args, barrier))
    else:
        failed = False
        if rank == 0:
            print(json.dumps(line), flush=True)
    if failed:
        print("bench.py: the all-gather leg failed (see the line's \"allgather\" field)", file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(3)     # peers may be stuck in a collective: no orderly teardown of the process group
    if world > 1:
        dist.destroy_process_group()


def guarded_leg(line, rank, timeout, barrier, leg):
    """Run ``leg()`` (the with-gather measurement) after the compute-only ``line`` is complete, under a watchdog: a stuck
    exchange must not cost the compute-only measurement, and it must not look like success either -- rank 0 still prints
    the line, with ``allgather: {"error": ...}`` inside, and the process leaves with status 3 (never re-exec'ed: it has
    touched the GPU).  Returns True when the leg failed in an orderly way (the line is printed either way).  The line
    survives every ordering of the ranks' deaths:
      * rank 0's watchdog is armed BEFORE the barrier that opens the leg, the peers' watchdogs after it and with
        PEER_GRACE_S more on the clock, so rank 0 always gives up first and prints;
      * a launcher (spawn_ranks above, torch.distributed.run) that sees another rank die SIGTERMs this one while its
        main thread sits inside a collective, where no Python signal handler can run: the C-level handler writes the
        signal number to a pipe (signal.set_wakeup_fd) and a watcher thread prints the line and leaves with status 3."""
    lock = threading.Lock()
    printed = [False]

    def emit(extra):
        with lock:
            if line is not None:
                line["allgather"] = extra
            if rank == 0 and not printed[0]:
                printed[0] = True
                print(json.dumps(line), flush=True)

    rfd, wfd = os.pipe()
    os.set_blocking(wfd, False)
    signal.signal(signal.SIGTERM, lambda *_: None)      # installs the C handler that feeds the wake-up pipe
    signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)

    def on_signal():
        os.read(rfd, 1)
        emit({"error": "terminated by the launcher: a peer rank failed before the all-gather leg finished"})
        print(f"bench.py: rank {rank} terminated during the all-gather leg", file=sys.stderr, flush=True)
        os._exit(3)
    threading.Thread(target=on_signal, daemon=True).start()

    def bail():
        emit({"error": f"no result within {timeout} s"})
        print(f"bench.py: the all-gather leg did not finish within {timeout} s", file=sys.stderr, flush=True)
        os._exit(3)

    def arm(seconds):
        dog = threading.Timer(seconds, bail)
        dog.daemon = True
        dog.start()
        return dog
    failed = False
    dog = arm(timeout) if rank == 0 else None
    try:
        barrier()       # the peers wait here for rank 0's extra work (fill probe, oracle check)
        if rank != 0:
            dog = arm(timeout + PEER_GRACE_S)
        res = leg()
    except Exception as exc:     # report, keep the compute-only result, fail the process
        res = {"error": f"{type(exc).__name__}: {exc}"}
        failed = True
    if dog is not None:
        dog.cancel()
    if isinstance(res, dict) and res.get("verify") is not None and not res["verify"].get("ok", False):
        failed = True
    emit(res)
    return failed


def allgather_leg(wl, world, rank, args, barrier):
    """Compute + replicate every table on every GPU: (a) tabulate, then one exchange; (b) chunked, the exchange of
    chunk c on a second stream under the tabulation of chunk c+1.  xGMI budget: 7T/8 bytes into every
    GPU over 7 x ~153 GB/s."""
    import torch
    from fiat_amd import distributed as D
    gather = D.TableGather(impl=args.allgather_impl, algo=args.allgather_algo)
    per = wl.batch
    out = wl.parts[0]["out"] if wl.parts[0] else wl.out
    tail = tuple(out.shape[1:])
    reps = max(2, min(10, args.steps // 4))
    res = {"impl": gather.impl, "algo": args.allgather_algo if gather.impl == "rccl" else "torch.distributed",
           "gathered_bytes_per_gpu": world * out.numel() * 8, "reps": reps}
    # test hooks (tests/test_bench_launch.py): the named rank never joins the exchange / dies in it
    if os.environ.get("FIAT_AMD_BENCH_FORCE_GATHER_TIMEOUT", "") == str(rank):
        time.sleep(args.allgather_timeout + 60.0)
    if os.environ.get("FIAT_AMD_BENCH_FORCE_GATHER_CRASH", "") == str(rank):
        os._exit(7)
    # replicated or staged?  Decided COLLECTIVELY: rank 0 keeps an extra buffer in the allocator's cache, so the ranks'
    # own free-memory readings differ and could choose different (mismatched) collectives.
    torch.cuda.empty_cache()
    free_bytes, _ = torch.cuda.mem_get_info()
    fb = torch.tensor([float(free_bytes)], dtype=torch.float64, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
    torch.distributed.all_reduce(fb, op=torch.distributed.ReduceOp.MIN)
    free_bytes = int(fb.item())
    res["free_bytes_min_over_ranks"] = free_bytes
    if args.allgather_verify:
        res["verify"] = verify_gather(wl, gather, world, rank, per, out, tail)
        if not res["verify"]["ok"]:
            gather.close()
            return res
    if args.allgather_mode == "ring" or (args.allgather_mode == "auto" and world * out.numel() * 8 > 0.6 * free_bytes):
        # too large to replicate (C4: 8 x 19 GB, at 122 points 8 x 103 GB): every table still visits every GPU, chunk by chunk,
        # through a ring of staging buffers the consumer drains (here: a reduction that touches every gathered byte)
        chunk = max(1, -(-per // max(args.allgather_chunks, 32)))
        res["mode"] = f"staging ring of 2 x {world} x {chunk} requests"
        sink = torch.zeros((), dtype=torch.float64, device="cuda")

        def ring():
            wl.produce_rows(0, per, out)
            for _, _, staged in gather.iter_gathered_chunks(out, chunk, ring=2):
                sink.add_(staged.sum())

        ring()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            ring()
        torch.cuda.synchronize()
        barrier()
        dt = (time.perf_counter() - t0) / reps
        res["ms_per_step_pipelined"] = dt * 1e3
        res["value_with_allgather"] = per * world / dt
        res["ingress_gbs_per_gpu"] = (world - 1) * out.numel() * 8 / dt / 1e9
        res["all_finite"] = bool(torch.isfinite(sink).item())
        gather.close()
        return res
    res["mode"] = "replicated"
    full = torch.empty((world * per,) + tail, dtype=torch.float64, device="cuda")
    mine = full.view(world, per, *tail)[rank]

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        barrier()
        return (time.perf_counter() - t0) / reps

    def sequential():
        wl.produce_rows(0, per, mine)
        gather.all_gather(mine, world * per, out=full)

    dt = timed(sequential)
    res["ms_per_step_sequential"] = dt * 1e3
    res["value_sequential"] = per * world / dt
    chunk = max(1, -(-per // args.allgather_chunks))

    def pipelined():
        gather.tabulate_allgather(wl.produce_rows, per, per, chunk, full)

    dt = timed(pipelined)
    res["chunks"] = -(-per // chunk)
    res["ms_per_step_pipelined"] = dt * 1e3
    res["value_with_allgather"] = per * world / dt
    res["ingress_gbs_per_gpu"] = (world - 1) * per * int(np.prod(tail)) * 8 / dt / 1e9
    # every rank's block must have arrived: compare one foreign request with a local re-tabulation of the same points
    # (ranks draw different points, so only the sizes and finiteness are checked here; parity is tests/)
    res["all_finite"] = bool(torch.isfinite(full.view(world, per, -1)[:, :: max(1, per // 16)]).all().item())
    gather.close()
    return res


def verify_gather(wl, gather, world, rank, per, out, tail):
    """Data check of the exchange on the first N > 1 run: every rank tabulates the SAME requests (rank 0's inputs,
    broadcast), so after the gather every foreign block must equal the rank's own block bit for bit -- which pins the
    stride / offset arithmetic of FX_GATHER_DIRECT, one-shot and chunked, rather than just finiteness."""
    import torch
    import torch.distributed as dist
    nv = min(per, 4096)
    dev = out.device
    inputs = wl.grid if wl.parts[0] is None else wl.parts[0]["pts"]      # hexahedron: per-request 1-D grids
    src = inputs[:nv].clone()
    if dist.get_backend() == "nccl":
        dist.broadcast(src, src=0)
    else:
        h = src.cpu()
        dist.broadcast(h, src=0)
        src.copy_(h)
    saved = inputs[:nv].clone()
    inputs[:nv].copy_(src)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    try:
        full = torch.full((world * nv,) + tail, float("nan"), dtype=torch.float64, device=dev)
        blocks = full.view(world, nv, *tail)
        mine = blocks[rank]
        wl.produce_rows(0, nv, mine)
        gather.all_gather(mine, world * nv, out=full)
        sync()
        one_shot = [bool(torch.equal(blocks[p], mine)) for p in range(world)]
        bad = [gather_mismatch("one_shot", p, blocks[p], mine, rank) for p in range(world) if not one_shot[p]]
        full.fill_(float("nan"))
        chunk = max(1, nv // 5 + 1)                                 # ragged last chunk
        gather.tabulate_allgather(wl.produce_rows, nv, nv, chunk, full)
        sync()
        chunked = [bool(torch.equal(blocks[p], mine)) for p in range(world)]
        bad += [gather_mismatch("chunked", p, blocks[p], mine, rank, chunk) for p in range(world) if not chunked[p]]
    finally:
        inputs[:nv].copy_(saved)
    ok = torch.tensor([1.0 if all(one_shot) and all(chunked) else 0.0], dtype=torch.float64,
                      device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    res = {"ok": bool(ok.item() == 1.0), "requests": nv, "one_shot_blocks_equal": one_shot, "chunked_blocks_equal": chunked,
           "rank": rank}
    if not res["ok"]:
        # where, on EVERY rank (the line is rank 0's): a first-run stride / offset bug of the direct exchange must be
        # diagnosable from the one line -- which peer's block, from which offset, how much of it, and what is there instead
        everyone = [None] * world
        dist.all_gather_object(everyone, bad[:2 * world])
        res["mismatches"] = [m for per_rank in everyone for m in (per_rank or [])][:4 * world]
    return res


def gather_mismatch(leg, peer, got, want, rank, chunk=None):
    """One differing block of the gathered tables, as seen by ``rank``: first differing double (flat offset inside the block,
    and the request / table row it falls in), how many doubles differ and how many of those were never written (still NaN),
    and whether the block equals the expected one shifted by a whole number of requests (the signature of a wrong stride)."""
    import torch
    g, w = got.reshape(got.shape[0], -1), want.reshape(want.shape[0], -1)
    diff = ~((g == w) | (torch.isnan(g) & torch.isnan(w)))
    idx = torch.nonzero(diff.reshape(-1))
    first = int(idx[0].item()) if idx.numel() else -1
    per_req = g.shape[1]
    info = {"leg": leg, "seen_by_rank": rank, "peer_block": peer, "first_offset": first, "first_request": first // per_req,
            "offset_in_request": first % per_req, "differing": int(idx.numel()), "never_written": int((diff & torch.isnan(g)).sum().item()),
            "doubles_per_block": int(g.numel())}
    if chunk:
        info["chunk_requests"] = chunk
    for shift in (1, -1, chunk or 0, -(chunk or 0)):
        if shift and abs(shift) < g.shape[0]:
            a, b = (g[shift:], w[:-shift]) if shift > 0 else (g[:shift], w[-shift:])
            if bool(torch.equal(a, b)):
                info["equals_expected_shifted_by_requests"] = shift
                break
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="p3tet", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="requests per GPU (default: the workload's)")
    ap.add_argument("--no-allgather", action="store_true", help="N > 1: skip the with-gather leg")
    ap.add_argument("--allgather", action="store_true", help="(kept for compatibility: the leg runs by default when N > 1)")
    ap.add_argument("--allgather-impl", default="auto", choices=["auto", "rccl", "torch"],
                    help="rccl = fx_allgather_tables through the C ABI, torch = torch.distributed collectives")
    ap.add_argument("--allgather-algo", default="direct", choices=["direct", "ring"])
    ap.add_argument("--allgather-chunks", type=int, default=8)
    ap.add_argument("--allgather-mode", default="auto", choices=["auto", "replicated", "ring"],
                    help="replicated: every GPU ends with all tables; ring: tables too large to replicate pass through staging buffers")
    ap.add_argument("--allgather-timeout", type=float, default=120.0)
    ap.add_argument("--allgather-verify", dest="allgather_verify", action="store_true", default=True,
                    help="N > 1 (default): before timing the exchange, every rank tabulates the same requests and compares every "
                         "gathered block with its own (bit for bit); a mismatch fails the run (exit 3)")
    ap.add_argument("--no-allgather-verify", dest="allgather_verify", action="store_false")
    ap.add_argument("--shared-points", action="store_true",
                    help="variant (SURVEY.md 8d): ONE 23-point rule on the reference cell pushed forward to per-request "
                         "physical cells (fx_tabulate_batch_shared) instead of per-request random points")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--policy", default="", help="measurement only: kernel-selection policy names (fx_ctx_set_policy), comma separated; "
                                                 "disclosed in config.policy -- the default line is the one without it")
    ap.add_argument("--check", type=int, default=-1,
                    help="requests verified against the CPU oracle after timing (-1: the whole batch, 0: none)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.environ["FIAT_AMD_BENCH_SPAWNED"] = "1"
        sys.exit(spawn_ranks(args.gpus))
    run(args)


if __name__ == "__main__":
    main()
