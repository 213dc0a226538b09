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
              solver_opts=dict(energy_tol_ppb=0), tile_parts=json.loads(os.environ.get("TEST_TILE_PARTS", "{}")))
assert m.world == world
owned = sorted(int(r) for r in m.my_regions)
tiles_held = sorted([int(r), int(t)] for r, g in m.tile_groups.items() for t in g.local)
if os.environ.get("TEST_FIXED_SCHEDULE") == "1":
    # the M-step replaced by a fixed parameter schedule (as tests/test_em_driver.py does for the reference's trace): what
    # is left to differ between the runs is the sharding itself
    srng = np.random.default_rng(77)
    sched = [np.clip(params * (1.0 + 0.08 * srng.standard_normal(params.shape)), 1e-3, 50.0) for _ in range(8)]
    lab0 = np.concatenate([np.argmax(-((Xb[:, None, :] - means[None]) ** 2).sum(-1), axis=1) for Xb in Xs])
    it_box = [0]

    def set_params(p):
        m.params_vec1 = p.copy()
        m._ou_param_varied_constraint(p)
        m._covars_ = m._covars_ + 1e-3 * np.eye(S)

    def fake_init(X_, lengths=None):
        m.startprob_ = np.full(K, 1.0 / K)
        m.transmat_ = np.full((K, K), 1.0 / K)
        m.init_ou_params = sched[0].copy()
        set_params(sched[0])
        m._upload_labels(lab0)

    def fake_mstep(stats):
        it_box[0] += 1
        set_params(sched[it_box[0]])

    m._init = fake_init
    m._do_mstep = fake_mstep
res = m.fit_accumulate_test(X, len_vec, 1e-3, "t", 5)        # t_labels are kept from iteration 3 on (base.py:422-426)
pred = {}
if os.environ.get("TEST_PREDICT_TILED") == "1":
    # predict() of the split regions, called by the ranks that hold tiles of them ONLY (a collective of the tile group, not
    # of the world: a rank without a tile of the region must not be needed)
    import hashlib
    for r in sorted(m.tile_groups):
        st, lp = m.predict(X, r)
        pred[str(r)] = [hashlib.sha1(np.ascontiguousarray(st, dtype=np.int32).tobytes()).hexdigest(), float(lp.sum()),
                        int(st.shape[0]), list(lp.shape)]
    for r in m.split_regions:
        try:
            m._compute_posteriors_graph(X, None, None, r)
            raise SystemExit("no error for a split region")
        except NotImplementedError:
            pass
        try:
            m._predict_posteriors(X, len_vec, r)
            raise SystemExit("no error for a split region")
        except NotImplementedError:
            pass
out = dict(rank=m.rank, owned=owned, tiles=tiles_held, cost_vec=res[5].tolist(), labels=res[6].astype(int).tolist(),
           means=m.means_.tolist(), predict=pred)
m.close()
if m.rank == 0:
    json.dump(out, open(%(out)r, "w"))
else:
    json.dump(dict(owned=owned, tiles=tiles_held, predict=pred), open(%(out)r + ".r1", "w"))
if world > 1:
    dist.destroy_process_group()
'''


def _run_fit(tmp_path, world, port, extra_env=None):
    out = str(tmp_path / ("fit_w%d.json" % world))
    script = tmp_path / ("fit_worker_w%d.py" % world)
    script.write_text(FIT_WORKER % {"root": ROOT, "out": out})
    procs = []
    for r in range(world):
        # (PHMRF_DETERMINISTIC: the solver's reductions do not depend on the order of the atomics, so what is left between
        #  the two runs is the order of the f64 sums of the statistics -- per rank first, then across ranks)
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", PHMRF_DETERMINISTIC="1")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      cwd=ROOT))
    for p in procs:
        o, _ = p.communicate(timeout=900)
        assert p.returncode == 0, o.decode()[-3000:]
    d = json.load(open(out))
    if world > 1:
        r1 = json.load(open(out + ".r1"))
        d["owned_r1"], d["tiles_r1"], d["predict_r1"] = r1["owned"], r1["tiles"], r1.get("predict", {})
    return d


def test_sharding_alone_changes_nothing(tmp_path):
    """The sharded reduction by itself: the M-step replaced by a fixed parameter schedule, the solver deterministic
    (PHMRF_DETERMINISTIC=1).  All five iterations of the two-rank run then equal the one-rank run: the costs to the order of
    the f64 sums (per rank first, then across ranks: 1e-9), the labels exactly."""
    env = {"TEST_FIXED_SCHEDULE": "1"}
    one = _run_fit(tmp_path, 1, 29761, env)
    two = _run_fit(tmp_path, 2, 29763, env)
    c1, c2 = np.array(one["cost_vec"]), np.array(two["cost_vec"])
    assert c1.shape == c2.shape == (5, 4)
    np.testing.assert_allclose(c2, c1, rtol=1e-9, atol=1e-12)
    assert np.array_equal(np.array(one["labels"]), np.array(two["labels"]))


def test_row_tiles_on_two_ranks_equal_the_same_tiles_on_one(tmp_path):
    """Blocks 0 and 2 cut into 2 and 3 row tiles (tiles.py).  On one rank all five tiles are local; on two ranks they are
    dealt with the whole block (tiles of one block on DIFFERENT ranks: lockstep rounds over gloo, halo rows exchanged).  Same
    algorithm, same numbers: costs to 1e-9, labels exactly, over five EM iterations with a fixed parameter schedule."""
    env = {"TEST_FIXED_SCHEDULE": "1", "TEST_TILE_PARTS": json.dumps({"0": 2, "2": 3}), "TEST_PREDICT_TILED": "1"}
    one = _run_fit(tmp_path, 1, 29765, env)
    two = _run_fit(tmp_path, 2, 29767, env)
    assert one["tiles"] == [[0, 0], [0, 1], [2, 0], [2, 1], [2, 2]] and one["owned"] == [1]
    held = sorted(two["tiles"] + two["tiles_r1"])
    assert held == one["tiles"]
    # at least one block has tiles on both ranks
    r0_blocks, r1_blocks = set(t[0] for t in two["tiles"]), set(t[0] for t in two["tiles_r1"])
    assert r0_blocks & r1_blocks, (two["tiles"], two["tiles_r1"])
    c1, c2 = np.array(one["cost_vec"]), np.array(two["cost_vec"])
    np.testing.assert_allclose(c2, c1, rtol=1e-9, atol=1e-12)
    assert np.array_equal(np.array(one["labels"]), np.array(two["labels"]))
    # predict() of a split region is a collective of its holders alone, and every holder gets the whole region: what the
    # one-rank run returns (labels exactly, log-likelihoods to the order of the sum)
    assert sorted(one["predict"]) == ["0", "2"]
    for ranks_pred in (two["predict"], two["predict_r1"]):
        for r, (sha, lpsum, nlab, shape) in ranks_pred.items():
            assert sha == one["predict"][r][0] and nlab == one["predict"][r][2] and shape == one["predict"][r][3]
            assert abs(lpsum - one["predict"][r][1]) <= 1e-9 * abs(one["predict"][r][1])
    assert set(two["predict"]) | set(two["predict_r1"]) == {"0", "2"}


def test_two_ranks_on_one_gpu_fit_matches_the_single_rank_fit(tmp_path):
    """phyloHMRF(world=2) with the REAL M-step (its states dealt to the two ranks): rank 0 owns the largest block, rank 1 the
    other two (dist.lpt_assign); five EM iterations with the solver at its exact fixed point.  Against the single-rank fit:
    the same costs in the first iteration up to the order of the sums; afterwards EM is a chaotic map (see below) and only
    loose bounds hold -- what the sharding itself changes is tested exactly in test_sharding_alone_changes_nothing."""
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
    # (the final means are not compared state by state: a state whose restart lands in another local optimum ends elsewhere
    #  -- seen: 2 of 32 entries off by 0.8 with every cost within the bounds above)
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
               "small", "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--split-above", "0.4"]
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
    # --split-above 0.4: both blocks hold more than 0.4 of a rank's share (97,150 / 2) and are cut into three row tiles each;
    # the six tiles are dealt longest first, which puts tiles of the SAME block on both ranks (lockstep rounds over gloo)
    assert d2["config"]["units_per_rank"] == [3, 3] and sum(d2["config"]["nodes_per_rank"]) == n0 + n1
    assert max(d2["config"]["nodes_per_rank"]) < 0.52 * (n0 + n1)
    assert d1["config"]["nodes_per_rank"] == [n0 + n1]
    assert d2["value"] > 0 and d2["ms_per_step"] > 0
    # (round 6) every rank's own clocks travel on the line: which rank is slow, and in what
    pr = d2["per_rank"]
    assert len(pr["estep_ms"]) == 2 and all(x > 0 for x in pr["estep_ms"]) and pr["nodes"] == d2["config"]["nodes_per_rank"]
    assert pr["units"] == [3, 3] and all(x > 0 for x in pr["exchange_us"]) and d2["fit_surface"] is None and d1["fit_surface"]
    # (one block solved as two tiles: a different local optimum of the same energy in the very first iteration already;
    #  chaotic afterwards, see the fit tests)
    np.testing.assert_allclose(d2["cost1"], d1["cost1"], rtol=6e-2, atol=5e-3)
    assert d2["fit"]["iterations"] >= 6 and d1["fit"]["iterations"] >= 6


def test_bench_four_ranks_on_one_gpu_two_split_blocks_on_disjoint_rank_pairs(tmp_path):
    """bench.py --gpus 4 under torch.distributed.run, all four ranks on device 0 over gloo, the small workload: each of its
    two blocks holds more than a rank's share and is cut into row tiles (2 and 3), dealt longest first so that block 0 sits
    on ranks 0 and 1 and block 1 on ranks 2 and 3 -- the shape of the 8-GPU deal of the whole-genome workload: every rank
    creates both tile groups and is a bystander of one of them, the tiles' lockstep rounds run inside each pair, the
    statistics and the dealt M-step (5 states on 4 ranks) go through the world."""
    env = dict(os.environ, PHMRF_ONE_GPU="1", PHMRF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", "29757", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "small", "--steps", "2",
           "--warmup", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    d = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    n0, n1 = 300 * 301 // 2, 200 * 260
    assert d["n_gpus"] == 4 and d["scaling"] == "strong"
    assert sorted(d["config"]["units_per_rank"]) == [1, 1, 1, 2] and sum(d["config"]["nodes_per_rank"]) == n0 + n1
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["fit"]["iterations"] >= 6
    assert all(np.isfinite(c) for c in d["cost1"])
    assert len(d["per_rank"]["estep_ms"]) == 4 and sum(d["per_rank"]["nodes"]) == n0 + n1


def test_cli_under_torchrun_two_ranks_one_block_in_two_tiles(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 phylo_hmrf.py --synthetic 64 ...`: the reference's command line on
    two ranks (both on GPU 0, over gloo).  The one synthetic block (2,080 nodes) is more than a rank's share, so it is cut
    into two row tiles, one per rank, built from the host edge list; rank 0 writes the reference's .mat."""
    import scipy.io
    out = str(tmp_path)
    env = dict(os.environ, PHMRF_ONE_GPU="1", PHMRF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29771", os.path.join(ROOT, "phylo_hmrf.py"), "-n", "5", "-r", "3", "--miter", "6", "--output", out,
           "--synthetic", "64", "--seed", "7", "-g", "3", "--quiet", "1"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    d = scipy.io.loadmat(os.path.join(out, "estimate_ou_3_1.00_5.mat"))
    n = 64 * 65 // 2
    assert d["state_vec"].size == n and d["params_vec1"].shape == (5, 23)
    cv = d["cost_vec"]
    assert cv.shape[1] == 4 and 1 <= cv.shape[0] <= 6 and np.all(np.isfinite(cv))
    lab = d["state_vec"].ravel()
    assert lab.min() >= 0 and lab.max() < 5 and len(np.unique(lab)) >= 2
