#!/usr/bin/env python3
"""Approximate VGPR liveness over a kernel's ISA listing (measurement tooling).

usage: vgpr_live.py kernel.s [loop_label]
Treats the text as straight-line code (branches ignored; with a loop label the range
label..last branch to it is iterated twice), prints the live-VGPR count every 25
instructions and the instructions around the maximum, so that the source of register
pressure in a fully unrolled kernel can be located."""
import re
import sys

STORES = ("ds_write", "global_store", "scratch_store", "buffer_store", "flat_store", "ds_bpermute_nothing")
RMW = ("v_fmac", "v_writelane", "v_mac", "v_dot", "v_pk_fmac")


def regs(op):
    op = op.strip()
    m = re.match(r"^v\[(\d+):(\d+)\]$", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", op)
    if m:
        return {int(m.group(1))}
    m = re.match(r"^a\[(\d+):(\d+)\]$", op)
    if m:
        return set(1000 + r for r in range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^a(\d+)$", op)
    if m:
        return {1000 + int(m.group(1))}
    return set()


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    parts = line.split(None, 1)
    mn = parts[0]
    ops = [o for o in (parts[1].split(",") if len(parts) > 1 else [])]
    ops = [re.sub(r"\s+(offset|offset0|offset1|off|glc|slc|sc0|sc1|nt|cbsz|abid|blgp|op_sel|op_sel_hi|neg_lo|neg_hi|row_shr|quad_perm|row_mask|bank_mask|bound_ctrl).*$", "", o.strip()) for o in ops]
    defs, uses = set(), set()
    if mn.startswith(STORES):
        for o in ops:
            uses |= regs(o)
    else:
        if ops:
            defs |= regs(ops[0])
            if mn.startswith(RMW):
                uses |= regs(ops[0])
            if mn.startswith("v_swap") and len(ops) > 1:
                defs |= regs(ops[1])
        for o in ops[1:]:
            uses |= regs(o)
    return mn, defs, uses, line


def main():
    lines = open(sys.argv[1]).read().splitlines()
    label = sys.argv[2] if len(sys.argv) > 2 else None
    ins = []
    lab_at = None
    last_branch = None
    for ln in lines:
        s = ln.split(";")[0].strip()
        if label and s == label + ":":
            lab_at = len(ins)
        p = parse(ln)
        if p:
            if label and label in s and s.startswith("s_cbranch"):
                last_branch = len(ins)
            ins.append(p)
    live_after = [None] * len(ins)

    def backward(lo, hi, live):
        for i in range(hi, lo - 1, -1):
            mn, d, u, _ = ins[i]
            live_after[i] = set(live)
            live = (live - d) | u
        return live

    if lab_at is not None and last_branch is not None:
        tail = backward(last_branch + 1, len(ins) - 1, set())
        top = backward(lab_at, last_branch, tail)
        top = backward(lab_at, last_branch, tail | top)
        backward(0, lab_at - 1, top)
    else:
        backward(0, len(ins) - 1, set())
    counts = [len(x) for x in live_after]
    mx = max(range(len(ins)), key=lambda i: counts[i])
    for i in range(0, len(ins), 25):
        print(f"{i:5d} live={counts[i]:3d}  {ins[i][3][:90]}")
    print(f"\nmax live = {counts[mx]} after instruction {mx}: {ins[mx][3]}")
    for i in range(max(0, mx - 6), min(len(ins), mx + 6)):
        print(f"   {i:5d} live={counts[i]:3d}  {ins[i][3][:100]}")


if __name__ == "__main__":
    main()
