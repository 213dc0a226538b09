#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r6_gputests.log 2>&1; rc=$?
tail -25 gpurun_out/r6_gputests.log | cut -c1-400
exit $rc
