#!/bin/bash
# GPU box: the driver's command on this tree and on a checkout of round 4's last commit (variants/r4tree: `git archive b51af9b`
# + make), alternating, on ONE box: bash tools/job_ab_r4.sh [runs]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
RUNS=${1:-3}
for i in $(seq 1 $RUNS); do
  (cd variants/r4tree && python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit) > gpurun_out/ab_r4_$i.json 2> gpurun_out/ab_r4_$i.err
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit > gpurun_out/ab_r5_$i.json 2> gpurun_out/ab_r5_$i.err
done
python3 - <<PY
import json
for tag in ("r4", "r5"):
    v = []
    for i in range(1, $RUNS + 1):
        d = json.loads(open("gpurun_out/ab_%s_%d.json" % (tag, i)).read().strip().splitlines()[-1])
        v.append((d["ms_per_step"], d["estep_ms"], d["ms_per_step_median"]))
    print(tag, "ms/step", [round(x[0], 1) for x in v], "E-step", [round(x[1], 1) for x in v], "median step", [round(x[2], 1) for x in v])
PY
