#!/usr/bin/env python3
"""Per-kernel sums of SQ counters from one rocprofv3 --pmc pass (csv).  usage: aggregate_sq.py DIR OUT.json"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from aggregate_pmc import short

out = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            e = out.setdefault(k, {})
            e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
keep = {k: v for k, v in out.items() if any(s in k for s in ("strip_kernel", "chain_kernel", "comp_table", "cc_union", "propose",
                                                             "alpha_mask", "energy_kernel", "posterior", "emission", "strip_scan", "strip_multi", "cc_", "comp_"))}
for k, v in keep.items():
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    if wc > 0:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if c in v:
                v[c + "/WAVE_CYCLES"] = round(v[c] / wc, 4)
    if v.get("SQ_LDS_IDX_ACTIVE", 0) > 0 and "SQ_LDS_BANK_CONFLICT" in v:
        v["LDS_BANK_CONFLICT/LDS_IDX_ACTIVE"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 4)
json.dump(keep, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: {c: x for c, x in v.items() if "/" in c} for k, v in keep.items()}, indent=1))
