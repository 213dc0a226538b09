#!/bin/bash
# GPU box: texture-addresser (vector memory pipeline) counters per kernel over the warm solve of the chr1-sized block:
# bash tools/job_ta.sh [lib]   (rocprofv3 --pmc, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export PHMRF_TRACE_PERT=0.05
[ -n "$1" ] && [ "$1" != product ] && export PHMRF_LIB=variants/libphmrf_$1.so
rm -rf gpurun_out/ta_pm
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD TA_FLAT_READ_WAVEFRONTS_sum SQ_INSTS_VALU --output-format csv -d gpurun_out/ta_pm -- python3 tools/trace.py 20 4980 1000 > /dev/null 2> gpurun_out/ta_pm.err
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/ta_pm/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
# the warm part: dispatches after the LAST emission_kernel
ids = sorted({int(r["Dispatch_Id"]) for r in rows})
em = [int(r["Dispatch_Id"]) for r in rows if "emission_kernel" in r["Kernel_Name"]]
start = max(em)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in rows:
    d = int(r["Dispatch_Id"])
    if d < start: continue
    m = re.search(r"(\w+_kernel(?:<[^(]*>)?)", r["Kernel_Name"])
    k = (m.group(1) if m else r["Kernel_Name"])[:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (d, k) not in seen: seen.add((d, k)); cnt[k] += 1
print("%-36s %5s %12s %10s %12s %12s %12s" % ("kernel", "n", "GUI_ACTIVE/8", "TA_BUSY%", "VMEM_RD", "TA_RD_WAVES", "VALU"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    ga = v.get("GRBM_GUI_ACTIVE", 0) / 8
    if ga < 1000: continue
    print("%-36s %5d %12.0f %10.1f %12.0f %12.0f %12.0f" % (k, cnt[k], ga, 100.0 * v.get("TA_BUSY_avr", 0) / max(ga, 1), v.get("SQ_INSTS_VMEM_RD", 0), v.get("TA_FLAT_READ_WAVEFRONTS_sum", 0), v.get("SQ_INSTS_VALU", 0)))
PY
rm -rf gpurun_out/ta_pm
