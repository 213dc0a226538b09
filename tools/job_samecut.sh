#!/bin/bash
# GPU box: the fusion pass on the expansions' own cut (variants/libphmrf_samecut.so: -DPHMRF_FUSION_SAME_CUT in api.hip)
# against the product library, warm solves from the SAME labelling
cd "$GRAFT_REPO_ROOT"
python3 tools/warm_from.py 20 4980 save gpurun_out/lab0.npy 2>/dev/null
for pert in 0.05 0.02 0.15; do
for v in product samecut; do
  if [ $v = product ]; then unset PHMRF_LIB; else export PHMRF_LIB=$PWD/variants/libphmrf_$v.so; fi
  echo "== $v pert $pert"
  python3 tools/warm_from.py 20 4980 load gpurun_out/lab0.npy $pert 2>/dev/null | grep "^warm"
done; done
