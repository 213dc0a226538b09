#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_estep.py -q -m gpu -k "graph_expansion or no_grid or chain_graph" -s > gpurun_out/r6_mf_tests.log 2>&1
grep -E "^knn|passed|failed|Error|assert |AssertionError" gpurun_out/r6_mf_tests.log | cut -c1-300 | head -40
PHMRF_SOLVE_TRACE=1 timeout -k 10 300 python3 tools/knn_trace.py 2 2>&1 | cut -c1-300 | tail -14
