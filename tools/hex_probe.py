#!/usr/bin/env python3
"""Why is the hexahedron launch (BASELINE config 5) bimodal by ~20 %?  One process, one binary (measurement tooling):
  A. six FRESHLY allocated output tensors (free, empty_cache, allocate again), 20 timed launches each;
  B. six repeats on the SAME tensor;
  C. the same batch written into a 2 MiB-aligned window of one large arena, at six different offsets;
with rocm-smi clocks / power / temperature before and after every measurement and the output pointer's alignment.
If the mode follows the allocation it is physical placement; if it follows time or temperature, the clocks say so."""
import argparse
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def smi():
    out = {}
    try:
        txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True,
                             timeout=20).stdout
        data = json.loads(txt)
        card = data[sorted(data)[0]]
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl or "mclk" in kl or "fclk" in kl or "power" in kl or ("temperature" in kl and ("junction" in kl or "memory" in kl or "edge" in kl or "hotspot" in kl)):
                out[k] = v
    except Exception as exc:   # not fatal: the timings are the point
        out["error"] = f"{type(exc).__name__}: {exc}"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=25000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=6)
    ap.add_argument("--series", default="ABC")
    args = ap.parse_args()
    import numpy as np
    import torch
    import fiat_amd
    P4 = fiat_amd.Lagrange(fiat_amd.ufc_simplex(1), 4)
    el = fiat_amd.TensorProductElement(fiat_amd.TensorProductElement(P4, P4), P4)
    rng = np.random.default_rng(5)
    grid = torch.as_tensor(np.sort(rng.uniform(0, 1, size=(args.batch, 3, 5)), axis=2)).cuda()
    shape = (args.batch, 4, 125, 125)
    nbytes = int(np.prod(shape)) * 8

    def timed(out):
        for _ in range(3):
            el.tabulate_batch(1, grid, out=out, grid=True)
        torch.cuda.synchronize()
        before = smi()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        per = []
        for _ in range(args.steps):
            e0.record()
            el.tabulate_batch(1, grid, out=out, grid=True)
            e1.record()
            torch.cuda.synchronize()
            per.append(e0.elapsed_time(e1))
        # and back to back, as bench.py times it
        e0.record()
        for _ in range(args.steps):
            el.tabulate_batch(1, grid, out=out, grid=True)
        e1.record()
        torch.cuda.synchronize()
        b2b = e0.elapsed_time(e1) / args.steps
        after = smi()
        fill0, fill1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fill0.record()
        for _ in range(5):
            out.fill_(1.0)
        fill1.record()
        torch.cuda.synchronize()
        return {"ms_single_median": float(np.median(per)), "ms_single_min": float(min(per)), "ms_single_max": float(max(per)),
                "ms_back_to_back": b2b, "tb_s_back_to_back": nbytes / b2b / 1e9, "fill_tb_s": nbytes / (fill0.elapsed_time(fill1) / 5) / 1e9,
                "ptr": hex(out.data_ptr()), "ptr_mod_2MiB": out.data_ptr() % (2 << 20), "ptr_mod_1GiB": out.data_ptr() % (1 << 30),
                "smi_before": before, "smi_after": after}

    # clock ramp
    warm = torch.empty(shape, dtype=torch.float64, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        el.tabulate_batch(1, grid, out=warm, grid=True)
    torch.cuda.synchronize()
    del warm
    print(json.dumps({"batch": args.batch, "bytes": nbytes}), flush=True)
    for i in range(args.repeats if "A" in args.series else 0):
        torch.cuda.empty_cache()
        out = torch.empty(shape, dtype=torch.float64, device="cuda")
        print(json.dumps({"series": "A fresh allocation", "i": i, **timed(out)}), flush=True)
        # a spacer of a different size each time moves the next allocation
        spacer = torch.empty((1 + i) * 37_000_000, dtype=torch.float64, device="cuda")
        del out
        torch.cuda.empty_cache()
        del spacer
    torch.cuda.empty_cache()
    if not set("BCD") & set(args.series):
        return
    out = torch.empty(shape, dtype=torch.float64, device="cuda")
    for i in range(args.repeats if "B" in args.series else 0):
        print(json.dumps({"series": "B same tensor", "i": i, **timed(out)}), flush=True)
    del out
    torch.cuda.empty_cache()
    if "D" in args.series:
        # D. fresh allocations of PADDED sizes: does the size class of the allocation decide the mode?
        for pad_name, padded in (("exact", nbytes), ("to 1 GiB multiple", -(-nbytes // (1 << 30)) * (1 << 30)),
                                 ("+1.6 GB", nbytes + 1_600_000_000), ("x2", 2 * nbytes)):
            for i in range(max(3, args.repeats // 2)):
                torch.cuda.empty_cache()
                buf = torch.empty(padded // 8, dtype=torch.float64, device="cuda")
                out = buf[:nbytes // 8].view(shape)
                print(json.dumps({"series": f"D fresh allocation, {pad_name} ({padded} B)", "i": i, **timed(out)}), flush=True)
                spacer = torch.empty((1 + i) * 41_000_000, dtype=torch.float64, device="cuda")
                del out, buf
                torch.cuda.empty_cache()
                del spacer
        torch.cuda.empty_cache()
    if "C" not in args.series:
        return
    arena = torch.empty(nbytes // 8 + 6 * (256 << 20) // 8 + (2 << 20) // 8, dtype=torch.float64, device="cuda")
    base = (-arena.data_ptr()) % (2 << 20) // 8
    for i in range(args.repeats if "C" in args.series else 0):
        off = base + i * (256 << 20) // 8 + (i % 2) * 16        # odd windows: 128 B past a 2 MiB boundary
        out = arena[off:off + nbytes // 8].view(shape)
        print(json.dumps({"series": "C window of one arena", "i": i, **timed(out)}), flush=True)


if __name__ == "__main__":
    main()
