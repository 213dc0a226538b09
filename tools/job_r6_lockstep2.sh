#!/bin/bash
# GPU box: the group solve taken round by round as the rounds END (one host thread, bench.py --block-threads 0) against 14 threads
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
python3 -m pytest tests/test_gpu_estep.py tests/test_gpu_fit.py -m gpu -x -q -k "group or lockstep" > $O/r6_lock2_tests.log 2>&1 || { tail -20 $O/r6_lock2_tests.log; exit 1; }
tail -1 $O/r6_lock2_tests.log
for rep in 1 2 3; do
  for t in 14 0; do
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit --no-kernel-timing --block-threads $t > $O/r6_lock2_${t}_$rep.json 2> $O/r6_lock2.err || { tail -5 $O/r6_lock2.err; exit 1; }
    python3 -c "
import json;d=json.loads(open('$O/r6_lock2_${t}_$rep.json').read().strip().splitlines()[-1])
print('threads $t rep $rep: ms/step %.2f E %.2f M %.2f'%(d['ms_per_step'],d['estep_ms'],d['mstep_ms']))"
  done
done
for t in 14 0; do
  for r in 0 3; do
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit --no-kernel-timing --block-threads $t --emulate-world 8 --emulate-rank $r > $O/r6_lock2_emu_${t}_$r.json 2>> $O/r6_lock2.err || exit 1
    python3 -c "
import json;d=json.loads(open('$O/r6_lock2_emu_${t}_$r.json').read().strip().splitlines()[-1])
print('emulated rank $r threads $t: E %.2f M %.2f step %.2f'%(d['estep_ms'],d['mstep_ms'],d['ms_per_step']))"
  done
done
