"""GPU: the SHARDED path on HIP kernels -- two ranks, each owning some of the syntenic blocks, the statistics all-reduced,
the M-step on rank 0 and broadcast (the reference: one process per block + the parent's reduction,
base.py:357-394).  One GPU is all a test box has, so both ranks use device 0 and the collective runs over gloo (RCCL
refuses two ranks on one device); the ranks are fresh child processes that set up torch.distributed before they touch
the GPU.  Compared with the single-rank run of the same problem."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# a cfg5-shaped workload in small: three syntenic blocks of one "chromosome split at the centromere" (two diagonal blocks
# and the off-diagonal block between them, utility.py:381-393), K = 8, S = 4
FIT_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=world)
from phylo_hmrf_amd import synthetic
from phylo_hmrf_amd.graph_host import grid_edges
from phylo_hmrf_amd.hmrf import phyloHMRF
from phylo_hmrf_amd.tree import PhyloTree
S, K = 4, 8
rng = np.random.default_rng(3)
tree = PhyloTree(synthetic.tree_for(S))
params = synthetic.sample_ou_params(rng, tree, K)
means, covars = tree.mean_cov(params)
covars = covars + 1e-3 * np.eye(S)
Lc = np.linalg.cholesky(covars)
Xs, len_vec, edges, start = [], [], [], 0
for rid, (H, W, diag) in enumerate([(90, 90, True), (70, 70, True), (90, 70, False)]):
    img = synthetic.label_image(rng, H, W, K)
    lab = img[np.triu_indices(H)] if diag else img.reshape(-1)
    X = np.maximum(means[lab] + np.einsum("nij,nj->ni", Lc[lab], rng.standard_normal((lab.shape[0], S))), 0.0)
    n = X.shape[0]
    Xs.append(X)
    edges.append(grid_edges(X, H, W, diag, 8))
    len_vec.append([n, start, start + n, H, W, 0, 0, rid, 1 if diag else 0, 1])
    start += n
X = np.concatenate(Xs)
m = phyloHMRF(n_components=K, run_id=0, n_samples=X.shape[0], n_features=S, observation=X, edge_list=synthetic.tree_for(S),
              len_vec=len_vec, type_id=1, branch_list=[1.0] * 7, edge_list_1=edges, cons_param=1.0, beta=1.0, beta1=0.5,
              initial_mode=0, initial_weight=0.3, initial_weight1=0.1, initial_magnitude=1.0, learning_rate=0.001,
              estimate_type=3, max_iter=100, n_iter=5000, tol=1e-7, random_state=11, quiet=True, mstep_workers=1,
              solver_opts=dict(energy_tol_ppb=0))
assert m.world == world
owned = sorted(int(r) for r in m.my_regions)
res = m.fit_accumulate_test(X, len_vec, 1e-3, "t", 5)        # t_labels are kept from iteration 3 on (base.py:422-426)
out = dict(rank=m.rank, owned=owned, cost_vec=res[5].tolist(), labels=res[6].astype(int).tolist(),
           means=m.means_.tolist())
m.close()
if m.rank == 0:
    json.dump(out, open(%(out)r, "w"))
else:
    json.dump(dict(owned=owned), open(%(out)r + ".r1", "w"))
if world > 1:
    dist.destroy_process_group()
'''


def _run_fit(tmp_path, world, port):
    out = str(tmp_path / ("fit_w%d.json" % world))
    script = tmp_path / ("fit_worker_w%d.py" % world)
    script.write_text(FIT_WORKER % {"root": ROOT, "out": out})
    procs = []
    for r in range(world):
        # (PHMRF_DETERMINISTIC: the solver's reductions do not depend on the order of the atomics, so what is left between
        #  the two runs is the order of the f64 sums of the statistics -- per rank first, then across ranks)
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", PHMRF_DETERMINISTIC="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      cwd=ROOT))
    for p in procs:
        o, _ = p.communicate(timeout=900)
        assert p.returncode == 0, o.decode()[-3000:]
    d = json.load(open(out))
    if world > 1:
        d["owned_r1"] = json.load(open(out + ".r1"))["owned"]
    return d


def test_two_ranks_on_one_gpu_fit_matches_the_single_rank_fit(tmp_path):
    """phyloHMRF(world=2): rank 0 owns the largest block, rank 1 the other two (dist.lpt_assign); five EM iterations with
    the solver at its exact fixed point.  Against the single-rank fit: the same costs in the first iterations up to the
    order of the sums, the same fit within the tolerance a chaotic EM trajectory allows afterwards."""
    one = _run_fit(tmp_path, 1, 29741)
    two = _run_fit(tmp_path, 2, 29743)
    from phylo_hmrf_amd.dist import lpt_assign
    owner = lpt_assign([90 * 91 // 2, 70 * 71 // 2, 90 * 70], 2)       # the off-diagonal block alone, the two diagonal ones together
    assert one["owned"] == [0, 1, 2]
    assert two["owned"] == [r for r in range(3) if owner[r] == 0] == [2]
    assert two["owned_r1"] == [r for r in range(3) if owner[r] == 1] == [0, 1]
    c1, c2 = np.array(one["cost_vec"]), np.array(two["cost_vec"])
    assert c1.shape == c2.shape == (5, 4)
    # The first iteration -- initialisation, E-step of every block on its rank, all-reduce of the statistics -- agrees to
    # the order of the f64 sums (blocks summed per rank, then across ranks).  From the first M-step on an EM fit is a
    # chaotic map: each state's SLSQP run starts 60 % random (phylo_hmrf.py:1378-1380) and can end in another local
    # optimum when its statistics differ in the last digits (measured: the second iteration's costs then differ by up to
    # 0.6 %); those iterations are held to a loose tolerance only.
    np.testing.assert_allclose(c2[:1], c1[:1], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(c2, c1, rtol=6e-2, atol=5e-3)
    np.testing.assert_allclose(np.array(two["means"]), np.array(one["means"]), rtol=1e-1, atol=1e-1)
    l1, l2 = np.array(one["labels"]), np.array(two["labels"])
    assert l1.shape == l2.shape and (l1 != l2).mean() < 3e-2, float((l1 != l2).mean())


def test_bench_two_ranks_on_one_gpu_shards_the_blocks(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), both ranks on device 0
    over gloo: the workload's blocks are DEALT (strong scaling), every rank times its own blocks, the statistics are
    all-reduced, rank 0 prints the one JSON line.  Against --gpus 1 on the same workload and seed: the same EM
    trajectory (cost1 per iteration) up to the order of the sums."""
    def run(world, port):
        env = dict(os.environ, PHMRF_ONE_GPU="1", PHMRF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   OMP_NUM_THREADS="2", PHMRF_DETERMINISTIC="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--workload",
               "small", "--steps", "2", "--warmup", "2", "--no-cpu-baseline"]
        if world == 1:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "small", "--steps", "2",
                   "--warmup", "2", "--no-cpu-baseline"]
        out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
        return json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    d1 = run(1, 29751)
    d2 = run(2, 29753)
    n0, n1 = 300 * 301 // 2, 200 * 260
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong"
    assert d2["config"]["blocks_per_rank"] == [1, 1] and sorted(d2["config"]["nodes_per_rank"]) == sorted([n0, n1])
    assert d1["config"]["nodes_per_rank"] == [n0 + n1]
    assert d2["value"] > 0 and d2["ms_per_step"] > 0
    np.testing.assert_allclose(d2["cost1"][:1], d1["cost1"][:1], rtol=1e-6, atol=1e-9)      # (see the fit test: chaotic afterwards)
    np.testing.assert_allclose(d2["cost1"], d1["cost1"], rtol=6e-2, atol=5e-3)
