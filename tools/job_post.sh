cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_estep.py tests/test_gpu_example.py -x -q -m gpu -k "posterior or reference_labels" 2>&1 | tail -2
bash profiles/warm_solve_quick.sh r3_post > /dev/null 2>&1
python3 -c "
import json
d=json.load(open('gpurun_out/r3_post_warm_solve.json'))
print({k:v for k,v in d['warm_solve_kernels_us'].items() if 'posterior' in k})"
