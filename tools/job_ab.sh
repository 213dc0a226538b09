cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_estep.py -x -q -m gpu -k "strip_multi or moves_never" 2>&1 | tail -3
bash profiles/warm_solve_quick.sh r3_cols
