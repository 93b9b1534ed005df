#!/usr/bin/env python3
"""Per-position breakdown of one sampler reverse step from a rocprofv3 --kernel-trace CSV.

A step = the dispatches from the step's first kernel (`conv_first_kernel`, or `step_prepare_kernel` on shapes the first-layer
kernel does not take) up to (not including) the next.  For every position in the
step the script prints the kernel, its grid (in workgroups) and the median duration and median gap to the previous
kernel over all complete steps found, then totals per kernel family.   python tools/step_breakdown.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("ddk::", "")


def main(path, out=sys.stdout):
    rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    first = "conv_first_kernel" if any("conv_first_kernel" in r["Kernel_Name"] for r in rows) else "step_prepare_kernel"
    starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    steps = []
    for a, b in zip(starts, starts[1:]):
        steps.append(rows[a:b])
    if not steps:
        print("no sampler steps in trace", file=out)
        return
    # keep the modal step length (drops the step that borders other work)
    n = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if len(s) == n]
    sig0 = [short(r["Kernel_Name"]) for r in steps[0]]
    steps = [s for s in steps if [short(r["Kernel_Name"]) for r in s] == sig0]
    print(f"{len(steps)} complete steps of {n} launches", file=out)
    import statistics
    durs = [[] for _ in range(n)]
    gaps = [[] for _ in range(n)]
    for s in steps:
        for i, r in enumerate(s):
            durs[i].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            if i:
                gaps[i].append(int(r["Start_Timestamp"]) - int(s[i - 1]["End_Timestamp"]))
    # medians: one step that borders other work (warm-up / timed boundary, a graph launch) must not smear a gap over all of them
    dur = [statistics.median(d) for d in durs]
    gap = [statistics.median(g) if g else 0.0 for g in gaps]
    k = 1
    fam = collections.OrderedDict()
    tot_d = tot_g = 0.0
    for i, r in enumerate(steps[0]):
        wg = [int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"])) for a in "XYZ"]
        d, g = dur[i] / k / 1e3, gap[i] / k / 1e3
        tot_d += d
        tot_g += g
        print(f"{i:3d} {sig0[i]:44s} wg {wg[0]:5d}x{wg[1]:3d}x{wg[2]:3d} lds {int(r['LDS_Block_Size']):6d} {d:8.2f} us  gap {g:6.2f}", file=out)
        e = fam.setdefault(sig0[i], [0, 0.0])
        e[0] += 1
        e[1] += d
    print(f"\nkernel time {tot_d:.1f} us + gaps {tot_g:.1f} us = {tot_d + tot_g:.1f} us per step", file=out)
    for name, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"  {name:44s} x{c:3d} {d:8.1f} us  {100 * d / (tot_d + tot_g):5.1f} %", file=out)


if __name__ == "__main__":
    main(sys.argv[1])
