#!/usr/bin/env python3
"""Per-kernel summary of a tools/profile_shapes.sh directory (measurement tooling): calls and average duration from the kernel
trace; HBM bytes per launch from WRITE_SIZE (exact, KiB) and FETCH_SIZE (KiB, x2: gfx950 reports half of wide coalesced reads,
MI355X_MICROARCH.md); MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8).  Large launches only."""
import collections, csv, glob, os, sys
src = sys.argv[1]
newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime, reverse=True)
dur = {}
for f in newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[:1]:
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for d in ("pmc_write", "pmc_fetch", "pmc_mfma"):
    for f in newest(os.path.join(src, d, "*", "*_counter_collection.csv"))[:1]:
        for r in csv.DictReader(open(f)):
            if int(r["Grid_Size"]) < 60000:
                continue
            cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[r["Kernel_Name"]] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("Scratch_Size"), r.get("LDS_Block_Size"))
for name, c in sorted(cnt.items()):
    if "tabulate" not in name and "piola" not in name and "mix" not in name:
        continue
    med = lambda k: sorted(c[k])[len(c[k]) // 2] if c.get(k) else None
    w, f, mb, ga = med("WRITE_SIZE"), med("FETCH_SIZE"), med("SQ_VALU_MFMA_BUSY_CYCLES"), med("GRBM_GUI_ACTIVE")
    line = name[:110]
    if name in dur:
        line += f" | calls {dur[name][0]} avg {dur[name][1]:.1f} us min {dur[name][2]:.1f} us"
    if w is not None:
        line += f" | write {w * 1024 / 1e6:.1f} MB"
    if f is not None:
        line += f" read {2 * f * 1024 / 1e6:.1f} MB"
    if mb is not None and ga:
        line += f" | mfma busy {mb / (1024 * ga / 8):.3f}"
    line += f" | vgpr {meta[name][0]}+{meta[name][1]} scratch {meta[name][2]} lds {meta[name][3]}"
    print(line)
