#!/bin/bash
# development: per-kernel durations of a few warm E-steps of one chr1-sized block (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/ks
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ks -o ks --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/estep_phases.py 20 4980 4 0.005 > $GRAFT_REPO_ROOT/gpurun_out/ks.out 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/ks -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    print("%-60s calls %6s  avg %9.1f us  total %9.1f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
P
