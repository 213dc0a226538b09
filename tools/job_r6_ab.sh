#!/bin/bash
# GPU box: bench.py --ab-env -- the same E-steps (same labellings, same parameters) under two values of a knob of the
# development library.  usage: job_r6_ab.sh NAME=A,B [ab_steps=30] [tag] [extra bench args...]
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export PHMRF_LIB=$GRAFT_REPO_ROOT/phylo_hmrf_amd/libphmrf_dev.so
AB=$1; N=${2:-30}; TAG=${3:-ab}; shift 3
python3 bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit --no-old-tolerance --no-kernel-timing --ab-env "$AB" --ab-steps $N "$@" > $O/r6_$TAG.json 2> $O/r6_$TAG.err || { tail -5 $O/r6_$TAG.err; exit 1; }
python3 -c "
import json;d=json.loads(open('$O/r6_$TAG.json').read().strip().splitlines()[-1]);ab=d['ab']
print(json.dumps({k:ab[k] for k in ab if k not in ('cost1',)}, indent=0))"
