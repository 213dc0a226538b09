#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs: the TCC counters of both do not fit one pass,
MI355X_MICROARCH.md) -> profiles/pmc_by_kernel.json: HBM bytes per kernel NAME, tagged with the build and the command,
which bench.py reports as roofline.traffic only while the tag matches the build it runs.

usage: aggregate_pmc_by_kernel.py FETCH_DIR WRITE_DIR OUT.json WORKLOAD "COMMAND" [VALU_DIR]
(VALU_DIR, round 6: a third pass with --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES -> per kernel `valu`: the share of the
vector pipes' issue capacity it used, which bench.py reports as roofline.valu)

Units and corrections (MI355X_MICROARCH.md, HBM): the counters are KiB.  On gfx950 FETCH_SIZE tallies a 128-byte
request at 64 bytes: wide coalesced streams read exactly half their bytes, narrower access widths are uncalibrated.
Both figures are kept: `fetch_bytes_raw` and `fetch_bytes_x2`; `hbm_bytes` = fetch x 2 + write, the guide's correction
(an upper bound where the kernel's reads are narrow gathers)."""
import csv, glob, hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("phmrf::", "")
    base = name.split("(")[0]
    tmpl = ""
    if "<" in base:
        base, rest = base.split("<", 1)
        tmpl = "<" + rest.split(">")[0] + ">"
    base = base.split()[-1] if base.strip() else base
    return base + tmpl


def collect(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                e = out.setdefault(short(row["Kernel_Name"]), [0.0, 0])
                e[0] += float(row["Counter_Value"])
                e[1] += 1
    return out


def source_hash():
    h = hashlib.sha1()
    d = os.path.join(ROOT, "phylo_hmrf_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def collect_valu(d):
    """third pass (round 6): --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES, with the kernel trace of the same pass.
    -> per kernel name: wave64 vector instructions, the clocks the GPU was active for its dispatches (GRBM_GUI_ACTIVE is
    summed over the 8 XCDs: / 8), the dispatches' own durations (the pass runs one block at a time), waves"""
    out = {}
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                dur[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    seen = set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                e = out.setdefault(k, {"launches": 0, "SQ_INSTS_VALU": 0.0, "GRBM_GUI_ACTIVE": 0.0, "SQ_WAVES": 0.0, "duration_ns": 0.0})
                if row["Counter_Name"] in e:
                    e[row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Dispatch_Id"] not in seen:
                    seen.add(row["Dispatch_Id"])
                    e["launches"] += 1
                    e["duration_ns"] += dur.get(row["Dispatch_Id"], 0)
    return out


SIMDS = 256 * 4      # MI355X: 256 CUs x 4 SIMDs; a wave64 vector instruction keeps a SIMD's pipe busy for 4 clocks


def main():
    fdir, wdir, out, workload, command = sys.argv[1:6]
    vdir = sys.argv[6] if len(sys.argv) > 6 else None
    fetch, write = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    valu = collect_valu(vdir) if vdir else {}
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, fl = fetch.get(k, [0.0, 0])
        w, wl = write.get(k, [0.0, 0])
        n = max(fl, wl)
        kernels[k] = {"launches": n, "fetch_bytes_raw": f * 1024.0, "fetch_bytes_x2": 2.0 * f * 1024.0,
                      "write_bytes": w * 1024.0, "hbm_bytes": 2.0 * f * 1024.0 + w * 1024.0,
                      "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0 / max(n, 1),
                      "fetch_raw_per_launch": f * 1024.0 / max(n, 1), "write_per_launch": w * 1024.0 / max(n, 1)}
    for k, v in valu.items():
        clocks = v["GRBM_GUI_ACTIVE"] / 8.0
        e = kernels.setdefault(k, {"launches": v["launches"]})
        e["valu"] = {"launches": v["launches"], "sq_insts_valu": v["SQ_INSTS_VALU"], "active_clocks": clocks,
                     "sq_waves": v["SQ_WAVES"], "duration_ns": v["duration_ns"],
                     "clock_ghz_implied": (round(clocks / v["duration_ns"], 3) if v["duration_ns"] > 0 else None),
                     # share of the vector pipes' issue capacity the kernel used while it ran: instructions x 4 clocks / (active clocks x SIMDs)
                     "valu_busy": (round(v["SQ_INSTS_VALU"] * 4.0 / (clocks * SIMDS), 4) if clocks > 0 else None)}
    try:
        rev = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        rev = os.environ.get("PHMRF_GIT_REV", "unknown")
    json.dump({"source_hash": source_hash(), "git_rev": rev, "workload": workload, "command": command, "kernels": kernels,
               "note": __doc__.split("Units and corrections")[1].strip()}, open(out, "w"), indent=1)
    top = sorted(kernels.items(), key=lambda kv: -kv[1].get("hbm_bytes", 0.0))[:8]
    for k, v in top:
        print("%-34s launches %6d  fetch raw %8.2f MB/launch  write %8.2f MB/launch  valu busy %s" % (
            k[:34], v["launches"], v.get("fetch_raw_per_launch", 0.0) / 1e6, v.get("write_per_launch", 0.0) / 1e6,
            (v.get("valu") or {}).get("valu_busy")))


if __name__ == "__main__":
    main()
