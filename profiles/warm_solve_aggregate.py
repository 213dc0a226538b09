#!/usr/bin/env python3
"""Warm-solve part of a tools/trace.py run: per-kernel time in launch order (kernel trace) and SQ counters (pmc pass).
usage: warm_solve_aggregate.py KT_DIR PMC_DIR OUT.json   (the warm part = everything after the 2nd emission_kernel)"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from aggregate_pmc import short


def rows(d, pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


kt = rows(sys.argv[1], "*kernel_trace.csv")
kt.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [short(r["Kernel_Name"]) for r in kt]
em = [i for i, k in enumerate(names) if k.startswith("emission_kernel")]
start = em[1] if len(em) > 1 else 0  # (an empty kernel-trace dir gives an empty table)
seq, per = [], {}
for r, k in list(zip(kt, names))[start:]:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "ORIENT" in k or "strip" in k:
        k = k + ("<1>" if "ILi1E" in r["Kernel_Name"] or "<1>" in r["Kernel_Name"] else "<0>")
    seq.append([k, round(us, 1)])
    e = per.setdefault(k, [0, 0.0])
    e[0] += 1
    e[1] += us
out = {"warm_solve_kernels_us": {k: {"launches": v[0], "total_us": round(v[1], 1)} for k, v in sorted(per.items(), key=lambda x: -x[1][1])},
       "launch_order_us": seq}
pm = rows(sys.argv[2], "*counter_collection.csv")
if pm:
    did = sorted({int(r["Dispatch_Id"]) for r in pm})
    # the same program: dispatch order is the same as in the kernel-trace pass
    emd = sorted({int(r["Dispatch_Id"]) for r in pm if short(r["Kernel_Name"]).startswith("emission_kernel")})
    d0 = emd[1] if len(emd) > 1 else did[0]
    agg = {}
    for r in pm:
        if int(r["Dispatch_Id"]) < d0:
            continue
        k = short(r["Kernel_Name"])
        e = agg.setdefault(k, {})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, v in agg.items():
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc > 0:
            for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU"):
                if c in v:
                    v[c + "/WAVE_CYCLES"] = round(v[c] / wc, 4)
        if v.get("SQ_BUSY_CYCLES", 0) > 0:
            v["resident_waves_per_busy_cycle"] = round(wc / v["SQ_BUSY_CYCLES"], 3)
    out["warm_solve_sq"] = agg
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["warm_solve_kernels_us"], indent=1))
