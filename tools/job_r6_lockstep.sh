#!/bin/bash
# GPU box: the E-step driven by ONE host thread in lockstep rounds (bench.py --block-threads 0: phmrf_mrf_solve_group) against
# 14 threads and against one block at a time; then one emulated rank both ways
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
for rep in 1 2; do
  for t in 14 0 1; do
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit --block-threads $t > $O/r6_lock_${t}_$rep.json 2> $O/r6_lock.err || { tail -5 $O/r6_lock.err; exit 1; }
    python3 -c "
import json;d=json.loads(open('$O/r6_lock_${t}_$rep.json').read().strip().splitlines()[-1])
print('threads $t rep $rep: ms/step %.2f E %.2f M %.2f | cost1 %s'%(d['ms_per_step'],d['estep_ms'],d['mstep_ms'],[round(c,4) for c in d['cost1'][-3:]]))"
  done
done
for t in 14 0; do
  for r in 0 3; do
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit --block-threads $t --emulate-world 8 --emulate-rank $r > $O/r6_lock_emu_${t}_$r.json 2>> $O/r6_lock.err || exit 1
    python3 -c "
import json;d=json.loads(open('$O/r6_lock_emu_${t}_$r.json').read().strip().splitlines()[-1])
print('emulated rank $r threads $t: E %.2f M %.2f step %.2f'%(d['estep_ms'],d['mstep_ms'],d['ms_per_step']))"
  done
done
