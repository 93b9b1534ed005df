#!/usr/bin/env python3
"""Summarise rocprofv3 counter passes of ONE kernel into the JSON bench.py reads (profiles/r03_*_pmc.json).

    python tools/pmc_summary.py <kernel-name substring> <kernel source file> <out.json> <pass dir> [<pass dir> ...]

Every <pass dir> is the -d directory of one `rocprofv3 --kernel-trace --pmc <counters> --output-format csv` run (counters that do
not fit one pass go to separate runs: MI355X_MICROARCH.md, rocprofv3 PMC slots); the per-dispatch values of the matching kernel are
averaged over its dispatches (the first 2 are warm-up and dropped).  Derived figures: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024
SIMDs over SQ_BUSY_CYCLES / 32 shader engines; traffic = 2 x FETCH_SIZE KB (gfx950 tallies 128-B requests as 64 B: the guide's
HBM section) + WRITE_SIZE KB; l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS).  The JSON records the sha256 of the kernel's source file so
that bench.py can tell when the figures were taken on another version of the kernel."""
import csv
import glob
import hashlib
import json
import os
import sys


def main():
    name, src, out = sys.argv[1], sys.argv[2], sys.argv[3]
    vals, durs, meta = {}, [], {}
    for d in sys.argv[4:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = {}
            for r in csv.DictReader(open(f)):
                if name not in r["Kernel_Name"]:
                    continue
                per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                meta = {"grid": r.get("Grid_Size"), "workgroup": r.get("Workgroup_Size"), "vgpr": r.get("VGPR_Count"),
                        "lds": r.get("LDS_Block_Size"), "kernel_name": r["Kernel_Name"][:120]}
            for k, v in per.items():
                v = v[2:] if len(v) > 4 else v
                vals[k] = sum(v) / len(v)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
            if rows and not glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                dd = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
                durs = dd[2:] if len(dd) > 4 else dd
    res = {"kernel": meta.get("kernel_name", name), "launch": meta, "counters": vals}
    if durs:
        res["duration_us_unprofiled_trace"] = sum(durs) / len(durs)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_BUSY_CYCLES" in vals:
        res["mfma_busy_cycles_per_simd"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
        res["sq_busy_cycles_per_se"] = vals["SQ_BUSY_CYCLES"] / 32
        res["mfma_busy"] = res["mfma_busy_cycles_per_simd"] / res["sq_busy_cycles_per_se"]
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in vals:
        res["executed_mfma_gflop"] = vals["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / 1e9
    if "FETCH_SIZE" in vals:
        res["fetch_bytes_per_launch"] = vals["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in vals:
        res["write_bytes_per_launch"] = vals["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        res["traffic_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
    if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals and vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"] > 0:
        res["l2_hit"] = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
    if "TCP_TCC_READ_REQ_sum" in vals:
        res["l1_to_l2_read_requests_per_launch"] = vals["TCP_TCC_READ_REQ_sum"]
    # a kernel that lives in included files names them all, '+'-separated: the hash covers every one (in the order given)
    h = hashlib.sha256()
    for part in src.split("+"):
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), part), "rb") as f:
            h.update(f.read())
    res["kernel_source"] = src
    res["kernel_source_sha16"] = h.hexdigest()[:16]
    res["note"] = ("per-launch averages over the kernel's dispatches of `python3 tools/pmc_one.py <case>` (first two dropped); SQ_BUSY_CYCLES is "
                   "summed over 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs; FETCH_SIZE doubled per MI355X_MICROARCH.md")
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ("counters", "note")}, indent=1))


if __name__ == "__main__":
    main()
