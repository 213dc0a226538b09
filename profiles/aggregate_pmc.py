#!/usr/bin/env python3
"""Aggregate rocprofv3 counter-collection CSVs (one --pmc pass per directory) into per-kernel sums.

usage: aggregate_pmc.py FETCH_DIR WRITE_DIR OUT_BY_KERNEL.json OUT_TRAFFIC.json
FETCH_DIR / WRITE_DIR hold the outputs of
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  --output-format csv -d FETCH_DIR -- python3 bench.py --steps 2 --warmup 5 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d WRITE_DIR -- python3 bench.py --steps 2 --warmup 5 --no-cpu-baseline
(separate passes: the TCC counters do not fit one pass, MI355X_MICROARCH.md).  Counter values are KiB.
"""
import csv, glob, json, os, sys

CLASSES = {"strip": ["strip_cols_kernel", "strip_kernel"], "fusion": ["fusion_cols_kernel"], "chain": ["chain_kernel"], "icm": ["icm_kernel"], "posterior_stats": ["posterior_kernel"],
           "energy": ["energy_kernel", "energy_grid_kernel", "energy_delta_grid_kernel", "energy_diff_grid_kernel"],
           "propose": ["propose_kernel", "propose_grid_kernel", "strip_scan_kernel", "unary_planes_kernel"],
           "coarse": ["coarsen_kernel", "coarse_apply_kernel"],
           "emission": ["emission_kernel"],
           "component": ["cc_init_kernel", "cc_union_kernel", "cc_flatten_kernel", "comp_table_kernel", "comp_decide_kernel",
                         "comp_block_kernel", "comp_apply_kernel"]}


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = name.split("(")[0]
    name = name.split("<")[0]
    return name.split("::")[-1].split()[-1].strip() if name.strip() else name


def collect(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = short(row["Kernel_Name"])
                e = out.setdefault(k, {"sum": 0.0, "launches": 0})
                e["sum"] += float(row["Counter_Value"])
                e["launches"] += 1
    return out


def main():
    fdir, wdir, out_k, out_t = sys.argv[1:5]
    fetch, write = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    json.dump({"fetch": fetch, "write": write}, open(out_k, "w"), indent=1)
    traffic = {}
    for cls, kernels in CLASSES.items():
        fs = sum(fetch.get(k, {}).get("sum", 0.0) for k in kernels)
        ws = sum(write.get(k, {}).get("sum", 0.0) for k in kernels)
        # a class "launch" is one launch of its main kernel(s): both strip kernels for the strip class
        n = (sum(fetch.get(k, {}).get("launches", 0) for k in kernels) if cls == "strip"
             else fetch.get(kernels[0], {}).get("launches", 0))
        if not n:
            continue
        traffic[cls] = {"fetch_size_kb_per_launch": fs / n, "write_size_kb_per_launch": ws / n, "launches_profiled": n,
                        "hbm_bytes_per_launch": int(round((fs + ws) / n * 1024))}
    note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 2 --warmup 5 "
            "--no-cpu-baseline` (whole-genome workload, 7 EM iterations of which 6 are steady-state warm starts; averages over all launches of all blocks). Raw counter values x 1024; "
            "on gfx950 FETCH_SIZE under-reports wide (16 B/lane) coalesced streams by 2x and is uncalibrated for narrower "
            "accesses (MI355X_MICROARCH.md, HBM), so the true read traffic lies between the raw value and twice it.")
    json.dump({"cfg3": traffic, "note": note}, open(out_t, "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
