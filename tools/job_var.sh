cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in "$@"; do
  PHMRF_LIB=variants/libphmrf_$v.so bash profiles/warm_solve_quick.sh v_$v 2>&1 | head -2
done
