#!/bin/bash
# GPU box (needs tools/patches/r6_front_repeat.patch or r6_front_coarse.patch applied and built: a measured negative):
# extra moves for the labels of a FRONT (api.hip solve_round_decide, PHMRF_FRONT_MIN of the development
# library) against the schedule without: solver rounds per EM iteration (one block at a time, traced), then the driver's
# command's timed region, alternating.  usage: job_r6_front.sh [min_cells=32] [reps=3]
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export PHMRF_LIB=$GRAFT_REPO_ROOT/phylo_hmrf_amd/libphmrf_dev.so
M=${1:-32}; REPS=${2:-3}
for m in 0 $M; do
  PHMRF_FRONT_MIN=$m PHMRF_SOLVE_TRACE=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit --no-kernel-timing --block-threads 1 > $O/r6_front_trace_$m.json 2> $O/r6_front_trace_$m.err || { tail -5 $O/r6_front_trace_$m.err; exit 1; }
  python3 - <<PY
import json, re
d = json.loads(open('$O/r6_front_trace_$m.json').read().strip().splitlines()[-1])
n0 = [i for i, l in enumerate(open('$O/r6_front_trace_$m.err')) if ' round 0 ' in l]
lines = [l for l in open('$O/r6_front_trace_$m.err') if re.match(r'\[phmrf solve\] round', l)]
per_it, cur, seen = [], 0, 0
for l in lines:
    if ' round 0 ' in l:
        seen += 1
        if seen > 1 and (seen - 1) % 26 == 0:
            per_it.append(cur); cur = 0
    cur += 1
per_it.append(cur)
print('front_min $m (one block at a time): E %.1f ms, rounds per iteration (timed 20): %s  total %d | cost1 %s' % (
    d['estep_ms'], per_it[6:26], sum(per_it[6:26]), [round(c, 4) for c in d['cost1'][-4:]]))
PY
done
for rep in $(seq 1 $REPS); do
  for m in 0 $M; do
    PHMRF_FRONT_MIN=$m python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit > $O/r6_front_${m}_$rep.json 2> $O/r6_front.err || { tail -5 $O/r6_front.err; exit 1; }
    python3 -c "
import json;d=json.loads(open('$O/r6_front_${m}_$rep.json').read().strip().splitlines()[-1])
print('front_min $m rep $rep: ms/step %.2f E %.2f M %.2f max E %.1f | cost1 %s'%(d['ms_per_step'],d['estep_ms'],d['mstep_ms'],max(d['estep_ms_by_step']),[round(c,4) for c in d['cost1'][-3:]]))"
  done
done
