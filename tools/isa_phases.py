#!/usr/bin/env python3
"""Static per-basic-block instruction counts of one kernel in a `hipcc -S` listing: VALU / SALU /
LDS / VMEM per block, with the branch targets, so that loop bodies can be told apart and weighed
by their trip counts (profiles/r04_secondary/k7_rows_phases.md).

    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Iinclude -Iecoflap_amd/csrc \\
        -S --cuda-device-only -o /tmp/wanda.s ecoflap_amd/csrc/wanda.hip
    python3 tools/isa_phases.py /tmp/wanda.s _Z22wanda_rows_wave_kernelILi2ELi4EEv9RowsGroup
"""
import re
import sys


def blocks(path, kernel):
    out, cur, on = [], None, False
    for line in open(path):
        if line.startswith(kernel + ":"):
            on = True
            cur = {"label": "entry", "ins": []}
            out.append(cur)
            continue
        if not on:
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB[0-9_]+):", t)
        if m:
            cur = {"label": m.group(1), "ins": []}
            out.append(cur)
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur["ins"].append(t.split(";")[0].strip())
    return out


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    show = set(sys.argv[3:])
    tot = {}
    for b in blocks(path, kernel):
        c = {}
        targets = []
        ops = {}
        for ins in b["ins"]:
            op = ins.split()[0]
            c[kind(op)] = c.get(kind(op), 0) + 1
            ops[op] = ops.get(op, 0) + 1
            if op.startswith(("s_cbranch", "s_branch")):
                targets.append(ins.split()[-1])
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
        print(f"{b['label']:14s} valu {c.get('valu', 0):4d}  salu {c.get('salu', 0):4d}  lds {c.get('lds', 0):3d}  "
              f"vmem {c.get('vmem', 0):3d}  -> {' '.join(targets)}")
        if b["label"] in show:
            for op, n in sorted(ops.items(), key=lambda t: -t[1]):
                print(f"      {n:4d} {op}")
    print("static total", tot)


if __name__ == "__main__":
    main()
