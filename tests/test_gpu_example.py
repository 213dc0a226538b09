"""GPU: BASELINE config 1 (example_input) -- the E-step on REAL Hi-C, against the reference's own run.

tests/golden/example_chr22_em.npz holds the reference's fit_accumulate_test (K=20, --miter 5; gco swap through pygco's
quantisation) on the first 300 bins of the example's chr22 synteny block, loaded by the reference's own loader: per EM
iteration the E-step inputs (means_, _covars_, the warm-start labels_local) and what the reference made of them
(labels, E_float, cost_vec row, sufficient statistics).  Deviations from config 1 (data the reference does not ship)
are listed in tests/golden/make_golden_example.py.

north_star: "final MRF energy <= the reference's" -- asserted STRICTLY here (both labellings are scored by the same
float64 energy function, so there is no evaluation slack to allow for)."""
import os

import numpy as np
import pytest

from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ex():
    g = np.load(os.path.join(G, "example_chr22_em.npz"))
    d = {k: g[k] for k in g.files}
    d["w"], d["eid"] = R.edge_weights_from_distance(d["edges"], float(d["beta1"]))
    return d


def _block(ex):
    from phylo_hmrf_amd import Block
    n = ex["X"].shape[0]
    b = Block(n, 4, int(ex["K"]))
    b.set_observations(ex["X"])
    b.set_graph(ex["eid"], ex["w"])
    lv = ex["len_vec"][0]
    b.set_grid(int(lv[3]), int(lv[4]), bool(lv[8]), 8)
    return b


@pytest.mark.parametrize("tol_ppb", [0, 1000, 10000])
@pytest.mark.parametrize("it", [0, 1, 2, 3, 4])
def test_energy_below_the_reference_on_real_hic(ex, it, tol_ppb):
    beta = float(ex["beta"])
    b = _block(ex)
    b.emission(ex["it_means"][it], ex["it_covars"][it])
    b.set_labels(ex["it_init"][it])
    res = b.solve(beta, energy_tol_ppb=tol_ppb)
    lab = b.get_labels()
    b.close()
    lp = R.log_multivariate_normal_density_full(ex["X"], ex["it_means"][it], ex["it_covars"][it])
    e_init = R.mrf_energy(np.int64(ex["it_init"][it]), lp, ex["eid"], ex["w"], beta)[0]
    e_ref_lab = R.mrf_energy(np.int64(ex["it_labels"][it]), lp, ex["eid"], ex["w"], beta)[0]
    np.testing.assert_allclose([e_init, e_ref_lab], [ex["it_efloat_init"][it][0], ex["it_efloat"][it][0]], rtol=1e-9)
    e_mine = R.mrf_energy(lab, lp, ex["eid"], ex["w"], beta)[0]
    print("iteration %d tol %d ppb: E init %.2f  reference (gco swap via pygco) %.2f  GPU %.2f  rounds %d"
          % (it, tol_ppb, e_init, e_ref_lab, e_mine, res["rounds"]))
    assert e_mine <= e_ref_lab                       # strictly: <= the reference's labelling
    assert e_mine <= e_init                          # and never above the warm start (the reference's is, on this data:
    #                                                  pygco's quantisation zeroes most edge weights when |logprob| is large)
    np.testing.assert_allclose(res["energy"], e_mine, rtol=2e-5)     # the device's f32 logprob vs the f64 oracle


@pytest.fixture(scope="module")
def exfull():
    """BASELINE config 1 at its real block size: the FULL chr22 synteny block of example_input (683 bins, 233,586 nodes),
    the reference's own K=20 --miter 5 run (tests/golden/make_golden_chr22_full.py).  The edge list is rebuilt from X
    with the builder that tests/golden/grid_edges.npz pins bit-exact on the reference's (the generator handed the
    reference exactly these edges)."""
    g = np.load(os.path.join(G, "example_chr22_full.npz"))
    d = {k: g[k] for k in g.files}
    d["X"] = np.float64(d["X"])
    lv = d["len_vec"][0]
    edges = R.grid_edges(d["X"], int(lv[3]), int(lv[4]), True, 8)
    assert edges.shape[0] == int(d["n_edges"])
    d["w"], d["eid"] = R.edge_weights_from_distance(edges, float(d["beta1"]))
    d["edges3"] = edges
    return d


@pytest.mark.parametrize("it", [0, 1, 2, 3, 4])
def test_energy_below_the_reference_on_the_full_chr22_block(exfull, it):
    """north_star's "final MRF energy <= the reference's" on real data at config 1's own block size, every EM iteration
    of the reference's run, from the reference's own warm start, at the exact fixed point and at the fit's stopping
    tolerance: strictly at or below the labelling gco's swap returned through pygco (scored by the same float64 function)."""
    ex = exfull
    beta = float(ex["beta"])
    lp = R.log_multivariate_normal_density_full(ex["X"], ex["it_means"][it], ex["it_covars"][it])
    e_init = R.mrf_energy(np.int64(ex["it_init"][it]), lp, ex["eid"], ex["w"], beta)[0]
    e_ref_lab = R.mrf_energy(np.int64(ex["it_labels"][it]), lp, ex["eid"], ex["w"], beta)[0]
    np.testing.assert_allclose([e_init, e_ref_lab], [ex["it_efloat_init"][it][0], ex["it_efloat"][it][0]], rtol=1e-9)
    b = _block(ex)
    b.emission(ex["it_means"][it], ex["it_covars"][it])
    for tol_ppb in (0, 1000, 10000):
        b.set_labels(ex["it_init"][it])
        res = b.solve(beta, energy_tol_ppb=tol_ppb)
        e_mine = R.mrf_energy(b.get_labels(), lp, ex["eid"], ex["w"], beta)[0]
        print("full chr22 block, iteration %d tol %d ppb: E init %.2f  reference (gco swap via pygco) %.2f  GPU %.2f  rounds %d"
              % (it, tol_ppb, e_init, e_ref_lab, e_mine, res["rounds"]))
        assert res["converged"]
        assert e_mine <= e_ref_lab                   # strictly: <= the reference's labelling
        assert e_mine <= e_init                      # and never above the warm start (the reference's is, at iteration 1)
    b.close()


@pytest.mark.parametrize("FIT_TOL_PPB,tile_gap", [(1000, 5e-4), (10000, 1e-3)])     # (the fit's tolerance: 1e-5 since round 6, 1e-6 before)
@pytest.mark.parametrize("parts", [1, 2, 3])
def test_warm_start_policy_and_row_tiles_stay_below_the_reference_at_every_iteration(exfull, parts, FIT_TOL_PPB, tile_gap):
    """The fit's own E-step sequence on the full chr22 block: five EM iterations of the reference's parameters, each
    labelling started from the reference's labels_local of that iteration OR from this build's previous result, whichever
    has the lower energy (Block.warm_start, the fit's default), at the fit's stopping tolerance -- as one block (parts = 1)
    and cut into 2 and 3 row tiles solved in lockstep rounds (tiles.py).  Every iteration's labelling is strictly at or
    below the reference's (gco swap through pygco, same float64 energy function); the tiled solves end at most 5e-4 (1e-3 at 1e-5) above
    the unsplit one (two local minima of one energy on a 233,586-node block in a run whose energy falls from 536,000 to 92,000
    in four iterations: measured +4e-5, +2.8e-4, -4.2e-4, -2.1e-3 -- the reference's own results are 0.2 % to 166 % above
    either; at 2,001,000 nodes tiled and unsplit agree to 1e-5, tests/test_gpu_tiles.py)."""
    from phylo_hmrf_amd import Block, tiles
    from phylo_hmrf_amd.base import SLOT_LOCAL
    ex = exfull
    beta, K = float(ex["beta"]), int(ex["K"])
    lv = ex["len_vec"][0]
    H = int(lv[3])
    n = ex["X"].shape[0]
    whole = _block(ex)
    grp = None
    if parts > 1:
        rows = tiles.split_rows(H, H, True, parts)

        def load(tl):
            tl.b.set_observations(ex["X"][tl.global_slice()])

        grp = tiles.make_group(0, (H, H, True), rows, [0] * parts, 0, 4, K, Block, load, None, 8, float(ex["beta1"]),
                               edges=ex["edges3"])
        cond = tiles.Conductor([grp])
    for it in range(5):
        lp = R.log_multivariate_normal_density_full(ex["X"], ex["it_means"][it], ex["it_covars"][it])
        e_ref_lab = R.mrf_energy(np.int64(ex["it_labels"][it]), lp, ex["eid"], ex["w"], beta)[0]
        init = np.int64(ex["it_init"][it])
        # the unsplit block under the same policy
        whole.emission(ex["it_means"][it], ex["it_covars"][it])
        if it == 0:
            whole.set_labels(init)
        cur = whole.get_labels()
        whole.set_labels(init)
        whole.save_labels(SLOT_LOCAL)
        whole.set_labels(cur)
        ec, es, took = whole.warm_start(beta, SLOT_LOCAL)
        e_start = min(ec, es)
        np.testing.assert_allclose(es, R.mrf_energy(init, lp, ex["eid"], ex["w"], beta)[0], rtol=1e-6)
        whole.solve_fast(beta, energy_tol_ppb=FIT_TOL_PPB)
        e_whole = R.mrf_energy(whole.get_labels(), lp, ex["eid"], ex["w"], beta)[0]
        assert e_whole <= e_ref_lab and e_whole <= e_start * (1 + 1e-9), (it, e_whole, e_ref_lab, e_start)
        if grp is None:
            continue
        for tl in grp.local.values():
            cur = tl.b.get_labels() if it else init[tl.global_slice()]
            tl.b.set_labels(init[tl.global_slice()])
            tl.b.save_labels(SLOT_LOCAL)
            tl.b.set_labels(cur)
        cond.solve(beta, dict(energy_tol_ppb=FIT_TOL_PPB), prepare=lambda tl: tl.b.emission(ex["it_means"][it], ex["it_covars"][it]),
                   warm_slot=SLOT_LOCAL)
        lab = np.zeros(n, dtype=np.int64)
        for tl in grp.local.values():
            lab[tl.owned_global_slice()] = tl.b.get_labels()[tl.owned_local_slice()]
        e_tiled = R.mrf_energy(lab, lp, ex["eid"], ex["w"], beta)[0]
        print("full chr22, iteration %d, %d tiles: reference %.2f  unsplit %.2f  tiled %.2f (%+.1e)"
              % (it, parts, e_ref_lab, e_whole, e_tiled, (e_tiled - e_whole) / abs(e_whole)))
        assert e_tiled <= e_ref_lab, (it, e_tiled, e_ref_lab)          # strictly at or below the reference
        # (measured over the five iterations: +4e-5 ... -2.1e-3 with 2 and 3 tiles; worst +2.8e-4)
        assert e_tiled <= e_whole + tile_gap * abs(e_whole), (it, e_tiled, e_whole)
    whole.close()
    if grp is not None:
        for tl in grp.local.values():
            tl.b.close()


def test_live_gco_fine_quantisation_on_real_hic(ex):
    """The same inputs with gco at its finest safe quantisation (not what the reference runs): also strictly below."""
    from oracle import gco_ref
    import json
    it, beta, K = 2, float(ex["beta"]), int(ex["K"])
    lp = R.log_multivariate_normal_density_full(ex["X"], ex["it_means"][it], ex["it_covars"][it])
    init = np.int64(ex["it_init"][it])
    # recorded by tests/golden/make_golden_live_gco.py (gco compiled from /root/reference); reproduced live where the
    # binary is present -- the test never skips
    e_fine = json.load(open(os.path.join(G, "live_gco_energies.json")))["real_hic_chr22_300_it2_fine"]
    if gco_ref.available():
        fine = gco_ref.cut_general_graph(ex["eid"], ex["w"], -lp, R.potts_matrix(K, beta), n_iter=5000, algorithm="swap",
                                         init_labels=init, quant="fine")
        np.testing.assert_allclose(R.mrf_energy(fine, lp, ex["eid"], ex["w"], beta)[0], e_fine, rtol=1e-12)
    b = _block(ex)
    b.set_logprob(lp)
    b.set_labels(init)
    b.solve(beta, energy_tol_ppb=0)
    e_mine = R.mrf_energy(b.get_labels(), lp, ex["eid"], ex["w"], beta)[0]
    b.close()
    print("real Hi-C, iteration 2: GPU %.3f  gco swap (fine quantisation) %.3f  gap %.2e" % (e_mine, e_fine, (e_mine - e_fine) / abs(e_fine)))
    assert e_mine <= e_fine          # strictly: every measured gap has been negative (-4.8e-4 here), so no allowance


@pytest.mark.parametrize("it", [0, 3])
def test_drop_in_cut_general_graph_as_the_reference_calls_it(ex, it):
    """phylo_hmrf.py:496-498 verbatim: edges, weights, unary, pairwise, n_iter=5000, algorithm='swap', init_labels --
    no geometry argument; the shim recovers the grid from the edge list."""
    from phylo_hmrf_amd.pygco_compat import cut_general_graph
    beta, K = float(ex["beta"]), int(ex["K"])
    lp = R.log_multivariate_normal_density_full(ex["X"], ex["it_means"][it], ex["it_covars"][it])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                  # no "general-graph moves only" warning
        lab = cut_general_graph(ex["eid"], ex["w"], -lp, R.potts_matrix(K, beta), n_iter=5000, algorithm="swap",
                                init_labels=np.float64(ex["it_init"][it]))   # the reference hands float labels (base.py:381)
    assert lab.dtype == np.int32 and lab.shape == (lp.shape[0],)
    e_mine = R.mrf_energy(lab, lp, ex["eid"], ex["w"], beta)[0]
    assert e_mine <= ex["it_efloat"][it][0]


@pytest.mark.parametrize("it", [0, 2, 4])
def test_posterior_costs_and_statistics_of_the_reference_labels(ex, it):
    """b3 on real data: the reference's labels in, its cost_vec row and sufficient statistics out."""
    beta, n = float(ex["beta"]), ex["X"].shape[0]
    b = _block(ex)
    b.emission(ex["it_means"][it], ex["it_covars"][it])
    b.set_labels(ex["it_labels"][it])
    stats, costs, _ = b.posterior_stats(beta, 3)
    b.close()
    row = ex["cost_vec"][it]                         # [iteration, pairwise_cost_normalize, unary_cost, cost1] (base.py:410)
    np.testing.assert_allclose(costs[1:4] / n, row[1:4], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(stats["post"], ex["it_stats_post"][it], rtol=2e-4, atol=1e-3)
    np.testing.assert_allclose(stats["obs"], ex["it_stats_obs"][it], rtol=2e-4, atol=5e-3)
    np.testing.assert_allclose(stats["obs*obs.T"], ex["it_stats_oo"][it], rtol=2e-4, atol=2e-2)


FIT_SCRIPT = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["PHMRF_ROOT"])
from phylo_hmrf_amd.hmrf import phyloHMRF
from phylo_hmrf_amd import mstep
g = np.load(os.path.join(os.environ["PHMRF_ROOT"], "tests", "golden", "example_chr22_em.npz"))
X, K = g["X"], int(g["K"])
tree = [[0, 1], [1, 2], [1, 3], [3, 4], [4, 5], [4, 6], [3, 7]]
runs = []
for seed in (22, 23, 24):
    m = phyloHMRF(n_components=K, run_id=0, n_samples=X.shape[0], n_features=4, observation=X, edge_list=tree,
                  len_vec=g["len_vec"].tolist(), type_id=1, branch_list=[0, 32, 20, 6, 6, 6, 12], edge_list_1=[g["edges"]],
                  cons_param=1.0, beta=1.0, beta1=0.5, initial_mode=0, initial_weight=0.3, initial_weight1=0.1,
                  initial_magnitude=1.0, learning_rate=0.001, estimate_type=3, max_iter=100, n_iter=5000, tol=1e-7,
                  random_state=seed, quiet=True)                    # mstep_workers=None: the default (host threads)
    # the M-step and the initialisation's per-cluster fits run natively on host threads: no worker processes are forked
    assert mstep.native_available() and mstep._POOL is None
    res = m.fit_accumulate_test(X, g["len_vec"].tolist(), 0.001, "t", int(g["m_iter"]))
    runs.append(dict(general=m.general_graph_regions, cost_vec=res[5].tolist(), tmax=int(res[6].max()), tn=int(res[6].shape[0])))
    m.close()
    assert mstep._POOL is None            # (three fits in one process, the GPU runtime up from the first on)
mstep.close_pool()
assert mstep._POOL is None
print("RESULT " + json.dumps(runs))
"""

# how far above the reference's best cost1 (1.418 on this block) a fit may end: 3.5 % of it.  (Measured: the fits end
# near 0.78, well below the reference's own run -- the labellings have lower energy and the M-step uses exact gradients.)
COST1_MARGIN = 0.05


def test_fit_on_real_hic_reaches_the_reference_cost(ex):
    """The whole drop-in: phyloHMRF.fit_accumulate_test on the same real block, same K and --miter, with the DEFAULT
    M-step settings (all states in one native call on host threads; three fits in ONE fresh process, so the second and
    third start with the GPU runtime up -- nothing forks), from three seeds.  EM trajectories are not comparable step by step (other initial clustering, other labellings); judged best
    to best: EVERY seed's best cost1 is at most the reference's best cost1 plus a stated margin."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PHMRF_ROOT=root)
    out = subprocess.run([sys.executable, "-c", FIT_SCRIPT], capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    runs = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    ref_best = float(ex["cost_vec"][:, 3].min())
    assert len(runs) == 3
    for r in runs:
        cost_vec = np.array(r["cost_vec"])
        assert r["general"] == [] and cost_vec.shape == (5, 4) and np.all(np.isfinite(cost_vec))
        assert r["tn"] == ex["X"].shape[0] and r["tmax"] < int(ex["K"])
        print("cost1 per iteration: GPU fit", np.round(cost_vec[:, 3], 4), " reference", np.round(ex["cost_vec"][:, 3], 4))
        assert cost_vec[:, 3].min() <= ref_best + COST1_MARGIN
