#!/bin/bash
# GPU box: the timeline of ONE warm solve of the chr1 block (rocprofv3 --kernel-trace on tools/trace.py): where the time
# between the kernels goes
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
export PHMRF_TRACE_PERT=0.05
rm -rf $O/gaps_kt
rocprofv3 --kernel-trace --output-format csv -d $O/gaps_kt -- python3 tools/trace.py 20 ${GAPS_N:-4980} 1000 > $O/gaps_trace.out 2> $O/gaps_trace.err
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/gaps_kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the warm solve = everything after the LAST emission kernel but one ... find the last emission launch and take what follows
em = [i for i, r in enumerate(rows) if "emission_kernel" in r["Kernel_Name"]]
start = em[-1]
seg = rows[start:]
# cut at the posterior kernel
end = [i for i, r in enumerate(seg) if "posterior_kernel" in r["Kernel_Name"]][0]
seg = seg[:end + 1]
t0 = int(seg[0]["Start_Timestamp"]); t1 = int(seg[-1]["End_Timestamp"])
dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print("warm E-step: %d launches, span %.3f ms, sum of durations %.3f ms, idle %.3f ms" % (len(seg), (t1 - t0) / 1e6, dur / 1e6, (t1 - t0 - dur) / 1e6))
gaps = []
for a, b in zip(seg, seg[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps.append((g, a["Kernel_Name"][:50], b["Kernel_Name"][:50]))
import collections
big = sorted(gaps, reverse=True)[:15]
for g, a, b in big:
    print("gap %.1f us  after %-50s before %s" % (g / 1e3, a, b))
small = [g for g, _, _ in gaps if g < 20000]
print("gaps < 20 us: %d, sum %.3f ms, mean %.1f us;  gaps >= 20 us: %d, sum %.3f ms" % (len(small), sum(small) / 1e6, sum(small) / max(len(small), 1) / 1e3, len(gaps) - len(small), (sum(g for g, _, _ in gaps) - sum(small)) / 1e6))
by = collections.Counter()
for r in seg:
    by[r["Kernel_Name"].split("(")[0][-40:]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in by.most_common(12):
    print("  %-42s %.3f ms" % (k, v / 1e6))
import re
print("launch by launch (us):", [((re.search(r"(\w+_kernel|fillBuffer|copyBuffer)", r["Kernel_Name"]) or [None, "?"])[1][:14], round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)) for r in seg])
PY
rm -rf $O/gaps_kt
