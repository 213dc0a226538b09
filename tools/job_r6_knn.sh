#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_estep.py tests/test_gpu_fit.py -q -m gpu -k "no_grid or chain_graph or general or not_a_grid" -s > gpurun_out/r6_knn_tests.log 2>&1; rc=$?
grep -E "^knn|passed|failed|Error|assert" gpurun_out/r6_knn_tests.log | cut -c1-300 | head -40
exit 0
