#!/bin/bash
# GPU box: what a lockstep round's exchange costs.  The chr1 block of cfg3 as two row tiles
#   (a) both on one rank (--tile-parts 2): the exchange is host packing only
#   (b) on two ranks, both on GPU 0, over gloo (RCCL refuses two ranks on one device): + one small all-reduce per round
# -> gpurun_out/${T}_tiles_local.json, gpurun_out/${T}_tiles_2ranks.json (bench lines with the tile_rounds object)
T=${1:-r5}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python3 bench.py --workload cfg3-chr1 --tile-parts 2 --steps 20 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit > gpurun_out/${T}_tiles_local.json 2> gpurun_out/${T}_tiles_local.err
PHMRF_ONE_GPU=1 PHMRF_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=4 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29771 bench.py --gpus 2 --workload cfg3-chr1 --steps 20 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit > gpurun_out/${T}_tiles_2ranks.json 2> gpurun_out/${T}_tiles_2ranks.err
python3 bench.py --workload cfg3-chr1 --steps 20 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit > gpurun_out/${T}_tiles_whole.json 2> gpurun_out/${T}_tiles_whole.err
TILE_TAG=$T python3 - <<'PY'
import json
import os
T = os.environ.get("TILE_TAG", "r5")
for f in (T + "_tiles_whole", T + "_tiles_local", T + "_tiles_2ranks"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json" % f).read().strip().splitlines() if l.startswith("{")][-1])
        print(f, "ms/step %.2f (E %.2f + M %.2f)" % (d["ms_per_step"], d["estep_ms"], d["mstep_ms"]), d.get("tile_rounds"))
    except Exception as e:
        print(f, "FAILED", e)
PY
