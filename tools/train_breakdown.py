#!/usr/bin/env python3
"""Breakdown of one optimiser step from a rocprofv3 --kernel-trace CSV of `tools/train_profile.py <cfg> graph`.

A step = the dispatches after one adam_kernel up to and including the next (2 micro-batches of forward + backward, the deferred
reduces, the weight refresh, clip + Adam).  Prints the totals per kernel (launches, time, share) of the median-length step among the
last ones, the kernel time, the idle time between kernels, and with -v the launch sequence.   python tools/train_breakdown.py <csv> [-v]"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"at::native::vectorized_elementwise_kernel<\d+, at::native::(\w+)<.*", r"at::\1", name)
    name = re.sub(r"at::native::(\w+)<.*", r"at::\1", name)
    return name.replace("ddk::", "")


def main(path, verbose):
    rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    steps = [rows[a + 1:b + 1] for a, b in zip(ends, ends[1:])]
    if len(steps) < 2:
        print("fewer than two complete optimiser steps in the trace")
        return
    steps = steps[-4:]
    n = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
    step = [s for s in steps if len(s) == n][-1]
    t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
    fam = collections.OrderedDict()
    busy = 0
    for r in step:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        busy += d
        e = fam.setdefault(short(r["Kernel_Name"]), [0, 0])
        e[0] += 1
        e[1] += d
    # kernels of different branches of the captured step (side streams) run side by side: the union of the busy intervals, not their sum
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
    covered, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > cur_e:
            covered += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    covered += cur_e - cur_s
    print(f"one optimiser step: {n} launches, {busy / 1e3:.0f} us of kernel time ({(busy - covered) / 1e3:.0f} us of it beside another "
          f"kernel) in a span of {(t1 - t0) / 1e3:.0f} us (idle {(t1 - t0 - covered) / 1e3:.0f} us); steps seen {[len(s) for s in steps]}")
    for name, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"  {name:60s} x{c:4d} {d / 1e3:9.1f} us  {100 * d / busy:5.1f} %   avg {d / c / 1e3:7.2f}")
    if verbose:
        prev = None
        for i, r in enumerate(step):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            wg = [int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"])) for a in "XYZ"]
            gap = 0 if prev is None else (s - prev) / 1e3
            print(f"{i:4d} {short(r['Kernel_Name']):60s} wg {wg[0]:6d}x{wg[1]:3d}x{wg[2]:3d} {(e - s) / 1e3:8.2f} us gap {gap:6.2f}")
            prev = e


if __name__ == "__main__":
    main(sys.argv[1], "-v" in sys.argv)
