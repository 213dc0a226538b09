#!/bin/bash
# development helper (GPU box): run the quick warm-solve profile with each prebuilt variants/libphmrf_NAME.so in turn
cd "$GRAFT_REPO_ROOT"
cp phylo_hmrf_amd/libphmrf.so /tmp/libphmrf_keep.so
for v in "$@"; do
  cp variants/libphmrf_$v.so phylo_hmrf_amd/libphmrf.so && bash profiles/warm_solve_quick.sh v_$v
done
cp /tmp/libphmrf_keep.so phylo_hmrf_amd/libphmrf.so
