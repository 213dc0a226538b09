"""GPU end-to-end: the reference's command line / fit surface on a seeded synthetic block."""
import os

import numpy as np
import pytest
import scipy.io

pytestmark = pytest.mark.gpu


def test_cli_run_synthetic_writes_reference_outputs(tmp_path):
    import phylo_hmrf as cli
    out = str(tmp_path)
    opts = cli.parse_args(["-n", "5", "-r", "3", "--miter", "6", "--output", out, "--synthetic", "64", "--seed", "7",
                           "-g", "3", "--quiet", "1"])
    mat = cli.run(opts.num_states, opts.chromvec, opts.root_path, opts.multiple, opts.species_name, opts.sort_states,
                  opts.run_id, opts.cons_param, opts.method_mode, opts.initial_mode, opts.initial_weight,
                  opts.initial_weight1, opts.initial_magnitude, opts.position1, opts.position2, opts.filter_sigma,
                  opts.beta, opts.beta1, opts.num_neighbor, opts.filter_mode, opts.threshold, opts.estimate_type,
                  opts.simu_version, opts.annotation, opts.reload, opts.dtype, opts.miter, opts.resolution, opts.quantile,
                  opts.ref_species, opts.output, synthetic=opts.synthetic, seed=opts.seed, quiet=opts.quiet)
    assert os.path.basename(mat) == "estimate_ou_3_1.00_5.mat"
    d = scipy.io.loadmat(mat)
    n = 64 * 65 // 2
    for key in ("state_vec", "len_vec", "params_vec1", "params_vec2", "iter_id1", "iter_id2", "cost_vec"):
        assert key in d
    assert d["state_vec"].size == n and d["params_vec1"].shape == (5, 23)
    cv = d["cost_vec"]
    assert cv.shape[1] == 4 and 1 <= cv.shape[0] <= 6 and np.all(np.isfinite(cv))
    assert list(cv[:, 0]) == list(range(cv.shape[0]))
    # cache files in the reference's names / formats (phylo_hmrf.py:1676-1704), and --reload 1 reads them back
    for f in ("data.50Kb.observed.3.npy", "edgelist.50Kb.observed.3.npy", "lenvec.50Kb.observed.3.txt"):
        assert os.path.exists(os.path.join(out, f))
    samples, len_vec, edge_list_vec = cli.load_cache(out, 50000, 3)
    assert samples.shape == (n, 4) and len_vec.shape == (1, 10) and edge_list_vec[0].shape[1] == 3


def test_fit_surface_and_predict_shapes():
    import phylo_hmrf as cli
    from phylo_hmrf_amd.hmrf import phyloHMRF
    from oracle import ref_numpy as R
    X, len_vec, edge_list_vec, tree = cli.synthetic_cache(50, 4, 4, 8, 11)
    n = X.shape[0]
    m = phyloHMRF(n_components=4, run_id=0, n_samples=n, n_features=4, observation=X, edge_list=tree, len_vec=len_vec,
                  type_id=1, branch_list=[1.0] * 7, edge_list_1=edge_list_vec, cons_param=1.0, beta=1.0, beta1=0.5,
                  initial_mode=0, initial_weight=0.3, initial_weight1=0.1, initial_magnitude=1.0, learning_rate=0.001,
                  estimate_type=3, max_iter=100, n_iter=5000, tol=1e-7, random_state=5, quiet=True, mstep_workers=1)
    res = m.fit_accumulate_test(X, len_vec, 1e-3, "t", 5)
    params_vec, params_vec1, params_vecList, it1, it2, cost_vec, t_labels = res
    assert params_vec.shape == (4, 23) and params_vecList.shape[1:] == (4, 23) and t_labels.shape == (n,)
    assert cost_vec.shape[1] == 4
    # b1: _compute_log_likelihood == float64 oracle density within the emission tolerance
    lp = m._compute_log_likelihood(X)
    ref = R.log_multivariate_normal_density_full(X, m.means_, m._covars_)
    assert np.all(np.abs(lp - ref) <= 2e-6 * np.abs(ref) + 2e-4)
    # predict(): labels + logprob of a region; _compute_posteriors_graph(): the reference's 5-tuple
    state, logprob = m.predict(X, 0)
    assert state.shape == (n,) and logprob.shape == (n, 4)
    post, pc, pcn, uc, c1 = m._compute_posteriors_graph(X, state, logprob, 0)
    w, eid = R.edge_weights_from_distance(edge_list_vec[0], 0.5)
    post_ref, pc_r, pcn_r, uc_r, c1_r = R.compute_posteriors_graph(state, logprob, eid, w, R.potts_matrix(4, 1.0), 3)
    finite = np.isfinite(post_ref).all(axis=1)
    assert np.max(np.abs(post[finite] - post_ref[finite])) < 2e-5
    np.testing.assert_allclose([pc, uc], [pc_r, uc_r], rtol=1e-5)
    # drop-in for the pygco call
    from phylo_hmrf_amd.pygco_compat import cut_general_graph
    lab = cut_general_graph(eid, w, -logprob, m.edge_potential, n_iter=5000, algorithm="swap", init_labels=state)
    assert lab.shape == (n,) and lab.dtype == np.int32
    e_new = R.mrf_energy(lab, logprob, eid, w, 1.0)[0]
    e_old = R.mrf_energy(state, logprob, eid, w, 1.0)[0]
    assert e_new <= e_old + 1e-6 * abs(e_old)
    with pytest.raises(ValueError):
        cut_general_graph(eid, w, -logprob, np.arange(16.0).reshape(4, 4))
    m.close()


def test_blocks_in_flight_give_the_sequential_result():
    """Three independent blocks: the E-steps driven concurrently (one HIP stream per block, host thread pool) must
    produce the statistics, costs and labels of the one-block-at-a-time run."""
    import phylo_hmrf as cli
    from phylo_hmrf_amd.hmrf import phyloHMRF
    Xs, lens, edges, tree, off = [], [], [], None, 0
    for N, seed in ((40, 1), (57, 2), (33, 3)):
        X, lv, ev, tree = cli.synthetic_cache(N, 4, 4, 8, seed)
        n = X.shape[0]
        row = list(lv[0])
        row[1], row[2] = off, off + n
        Xs.append(X)
        lens.append(row)
        edges.append(ev[0])
        off += n
    X = np.concatenate(Xs)

    def fit(threads):
        m = phyloHMRF(n_components=4, run_id=0, n_samples=X.shape[0], n_features=4, observation=X, edge_list=tree,
                      len_vec=lens, type_id=1, branch_list=[1.0] * 7, edge_list_1=edges, cons_param=1.0, beta=1.0,
                      beta1=0.5, initial_mode=0, initial_weight=0.3, initial_weight1=0.1, initial_magnitude=1.0,
                      learning_rate=0.001, estimate_type=3, max_iter=100, n_iter=5000, tol=1e-7, random_state=5,
                      quiet=True, mstep_workers=1, block_threads=threads, init_method="sklearn")
        # (the host initialiser: what is compared is the E-step with 1 and with 3 blocks in flight, from bit-identical
        #  starting parameters; the default initialiser's device-side moments carry f64 atomics of their own)
        res = m.fit_accumulate_test(X, lens, 1e-3, "t", 4)
        m.close()
        return res

    a, b = fit(1), fit(3)
    # (not bit-for-bit by contract: the component move table is summed with float atomics in either mode)
    np.testing.assert_allclose(a[5], b[5], rtol=1e-6)     # cost_vec: every iteration's costs
    np.testing.assert_allclose(a[0], b[0], rtol=1e-4, atol=1e-6)
    assert np.mean(a[6] == b[6]) > 0.999                  # labels


@pytest.mark.parametrize("tag", ["diag", "offdiag", "chain"])
def test_pygco_drop_in_without_geometry_beats_the_reference(tag):
    """Row b2: the shim called exactly like phylo_hmrf.py:496-498 (no geometry argument) on the graphs whose labelling
    was recorded through the reference's own predict(): the grid is inferred from the edge list and the float energy is
    <= the reference's (strictly: same float64 scoring of both labellings)."""
    import warnings
    from oracle import ref_numpy as R
    from phylo_hmrf_amd.pygco_compat import cut_general_graph
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gco_%s.npz" % tag))
    K, beta = int(g["K"]), float(g["beta"])
    w, eid = R.edge_weights_from_distance(g["edges"], 0.5)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        lab = cut_general_graph(eid, w, -g["logprob"], R.potts_matrix(K, beta), n_iter=5000, algorithm="swap",
                                init_labels=g["init"])
    e_mine = R.mrf_energy(lab, g["logprob"], eid, w, beta)[0]
    assert e_mine <= float(g["efloat_swap_pygco"][0])
    assert e_mine <= float(g["efloat_swap_fine"][0]) + 1e-6 * abs(float(g["efloat_swap_fine"][0]))


def test_pygco_drop_in_on_a_graph_that_is_no_grid():
    """A general graph gets the general-graph moves and says so; strict_grid=True refuses."""
    from oracle import ref_numpy as R
    from phylo_hmrf_amd.pygco_compat import cut_general_graph
    rng = np.random.default_rng(3)
    n, K = 200, 5
    eid = np.array([[i, j] for i in range(n) for j in rng.choice(np.arange(i + 1, n), size=min(3, n - 1 - i), replace=False)])
    w = rng.random(eid.shape[0])
    unary = rng.random((n, K)) * 3
    init = rng.integers(0, K, n)
    V = R.potts_matrix(K, 1.0)
    with pytest.warns(RuntimeWarning, match="general-graph moves only"):
        lab = cut_general_graph(eid, w, unary, V, n_iter=5000, algorithm="swap", init_labels=init)
    assert R.mrf_energy(lab, -unary, eid, w, 1.0)[0] <= R.mrf_energy(init, -unary, eid, w, 1.0)[0]
    with pytest.raises(ValueError, match="not the stencil"):
        cut_general_graph(eid, w, unary, V, n_iter=5000, algorithm="swap", init_labels=init, strict_grid=True)


def test_fit_warns_when_a_region_is_not_a_grid_block():
    import phylo_hmrf as cli
    from phylo_hmrf_amd.hmrf import phyloHMRF
    X, len_vec, edge_list_vec, tree = cli.synthetic_cache(20, 4, 3, 8, 2)
    bad = [np.vstack([edge_list_vec[0], [[0.0, float(X.shape[0] - 1), 0.3]]])]      # one edge across the whole block
    with pytest.warns(RuntimeWarning, match="general-graph moves only"):
        m = phyloHMRF(n_components=3, run_id=0, n_samples=X.shape[0], n_features=4, observation=X, edge_list=tree,
                      len_vec=len_vec, type_id=1, branch_list=[1.0] * 7, edge_list_1=bad, cons_param=1.0, beta=1.0,
                      beta1=0.5, initial_mode=0, initial_weight=0.3, initial_weight1=0.1, initial_magnitude=1.0,
                      estimate_type=3, random_state=1, quiet=True, mstep_workers=1)
    assert m.general_graph_regions == [0]
    m.close()
    with pytest.raises(ValueError, match="at most 8 species"):
        phyloHMRF(n_components=3, run_id=0, n_samples=4, n_features=9, observation=np.zeros((4, 9)), edge_list=tree,
                  len_vec=[[4, 0, 4, 2, 2, 0, 0, 0, 0, 1]], type_id=1, branch_list=[1.0] * 7, edge_list_1=[np.zeros((0, 3))],
                  cons_param=1.0, beta=1.0, beta1=0.5, initial_mode=0, initial_weight=0.3, initial_weight1=0.1,
                  initial_magnitude=1.0)


@pytest.mark.parametrize("filter_mode", ["0", "1"])
def test_cli_runs_from_raw_hic_text(tmp_path, filter_mode):
    """Row f4: `python phylo_hmrf.py -n 6 --chromvec 22 -p <dir>` on example_input-style raw files (a 120-bin window of
    the example's chr22 rows, tests/golden/example_loader.npz) -> cache files + the .mat.  (The diffusion filter on this
    path is the build's own Perona-Malik restatement, medpy being absent; tests/test_preprocess.py checks it against the
    published update rule.)"""
    import phylo_hmrf as cli
    from tests.test_preprocess import SPECIES, _write_dir
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example_loader.npz"))
    d, flist = _write_dir(tmp_path, g, "22", int(g["first_bin"]), 120, 0, g["a_synteny"])
    with open(os.path.join(d, "edge.1.txt"), "w") as f:
        f.write("0\t1\n1\t2\n1\t3\n3\t4\n4\t5\n4\t6\n3\t7\n")
    with open(os.path.join(d, "branch_length.1.txt"), "w") as f:
        f.write("0\t32\t20\t6\t6\t6\t12\n")
    with open(os.path.join(d, "species_name.1.txt"), "w") as f:
        f.write("\n".join(SPECIES) + "\n")
    with open(os.path.join(d, "path_list.txt"), "w") as f:
        f.write("\n".join("hic_" + s for s in SPECIES) + "\n")
    out = os.path.join(d, "out")
    cwd = os.getcwd()
    os.chdir(d)                                   # chrom_quantile_test.txt goes to the working directory (:1659-1661)
    try:
        o = cli.parse_args(["-n", "6", "-r", "1", "--miter", "4", "--chromvec", "22", "-p", d, "--output", out, "-g", "3",
                            "--seed", "4", "--quiet", "1", "--filter_mode", filter_mode])
        mat = cli.run(o.num_states, o.chromvec, o.root_path, o.multiple, o.species_name, o.sort_states, o.run_id,
                      o.cons_param, o.method_mode, o.initial_mode, o.initial_weight, o.initial_weight1,
                      o.initial_magnitude, o.position1, o.position2, o.filter_sigma, o.beta, o.beta1, o.num_neighbor,
                      o.filter_mode, o.threshold, o.estimate_type, o.simu_version, o.annotation, o.reload, o.dtype,
                      o.miter, o.resolution, o.quantile, o.ref_species, o.output, seed=o.seed, quiet=o.quiet)
    finally:
        os.chdir(cwd)
    samples, len_vec, edge_list_vec = cli.load_cache(out, 50000, 1)
    if filter_mode == "0":
        assert np.array_equal(samples, g["a_diffusion_samples"]) and np.array_equal(len_vec, g["a_diffusion_lenvec"])
    else:
        # --filter_mode 1: the bilateral filter (utility.py:1575-1582; restated, scikit-image being absent): same nodes, other
        # features than the unfiltered loader run of the reference
        assert np.array_equal(len_vec, g["a_none_lenvec"]) and samples.shape == g["a_none_samples"].shape
        assert np.all(np.isfinite(samples)) and not np.allclose(samples, g["a_none_samples"])
    assert os.path.exists(os.path.join(d, "chrom_quantile_test.txt"))
    dm = scipy.io.loadmat(mat)
    assert dm["state_vec"].size == samples.shape[0] and np.all(np.isfinite(dm["cost_vec"]))


def test_checkpoint_and_resume_continue_the_fit_exactly(tmp_path, monkeypatch):
    """SURVEY.md section 5 (checkpoint/resume; the reference keeps its state in RAM only, base.py:412): six EM iterations
    straight against three iterations + a checkpoint + a resumed fit to six, solver deterministic (PHMRF_DETERMINISTIC=1) and
    the reference's own M-step with its random restarts (the generator's state travels in the checkpoint).  The resumed fit
    continues with the same bookkeeping (base.py:402-435): every row of cost_vec equal to 1e-12, labels exactly, the
    parameter history and the returned iteration ids equal."""
    import phylo_hmrf as cli
    from phylo_hmrf_amd.hmrf import phyloHMRF
    monkeypatch.setenv("PHMRF_DETERMINISTIC", "1")
    Xs, lens, edges, tree, off = [], [], [], None, 0
    for N, seed in ((48, 1), (61, 2)):
        X, lv, ev, tree = cli.synthetic_cache(N, 4, 5, 8, seed)
        n = X.shape[0]
        row = list(lv[0])
        row[1], row[2], row[7] = off, off + n, len(lens)
        Xs.append(X)
        lens.append(row)
        edges.append(ev[0])
        off += n
    X = np.concatenate(Xs)
    ck = str(tmp_path / "fit.ckpt.npz")

    def model(**kw):
        return phyloHMRF(n_components=5, run_id=0, n_samples=X.shape[0], n_features=4, observation=X, edge_list=tree,
                         len_vec=lens, type_id=1, branch_list=[1.0] * 7, edge_list_1=edges, cons_param=1.0, beta=1.0,
                         beta1=0.5, initial_mode=0, initial_weight=0.3, initial_weight1=0.1, initial_magnitude=1.0,
                         learning_rate=0.001, estimate_type=3, max_iter=100, n_iter=5000, tol=1e-7, random_state=9,
                         quiet=True, mstep_workers=1, block_threads=1, init_method="sklearn", **kw)

    m = model()
    straight = m.fit_accumulate_test(X, lens, 1e-12, "t", 6)
    m.close()
    m = model(checkpoint_path=ck, checkpoint_every=1)
    first = m.fit_accumulate_test(X, lens, 1e-12, "t", 3)
    m.close()
    assert os.path.exists(ck) and not os.path.exists(ck + ".tmp.npz")
    z = np.load(ck)
    assert int(z["it_next"]) == 3 and z["cost_vec"].shape == (3, 4) and z["labels_local"].shape == (X.shape[0],)
    np.testing.assert_allclose(first[5], straight[5][:3], rtol=1e-12, atol=0)
    m = model(resume_from=ck)
    resumed = m.fit_accumulate_test(X, lens, 1e-12, "t", 6)
    m.close()
    assert resumed[5].shape == straight[5].shape == (6, 4)
    np.testing.assert_allclose(resumed[5], straight[5], rtol=1e-12, atol=0)         # cost_vec, all six rows
    assert np.array_equal(resumed[6], straight[6])                                   # t_labels
    assert resumed[3] == straight[3] and resumed[4] == straight[4]                   # iter_id1, iter_id2
    np.testing.assert_allclose(resumed[2], straight[2], rtol=1e-12, atol=0)         # params_vecList
    np.testing.assert_allclose(resumed[0], straight[0], rtol=1e-12, atol=0)
    # a checkpoint of another model is refused
    m = phyloHMRF(n_components=4, run_id=0, n_samples=X.shape[0], n_features=4, observation=X, edge_list=tree, len_vec=lens,
                  type_id=1, branch_list=[1.0] * 7, edge_list_1=edges, cons_param=1.0, beta=1.0, beta1=0.5, initial_mode=0,
                  initial_weight=0.3, initial_weight1=0.1, initial_magnitude=1.0, learning_rate=0.001, estimate_type=3,
                  max_iter=100, n_iter=5000, tol=1e-7, random_state=9, quiet=True, mstep_workers=1, resume_from=ck)
    with pytest.raises(ValueError):
        m.fit_accumulate_test(X, lens, 1e-12, "t", 6)
    m.close()


def test_fit_surface_at_the_chr1_block_size():
    """Row b4 at the headline size of ONE block: the product's own surface -- a phyloHMRF object built from host float64
    samples, fit_accumulate_test driving the EM loop (phylo_hmrf_amd/base.py <-> reference base.py:301-455) -- on a
    4,980-bin diagonal block (12,402,690 nodes: the chr1 block of the whole-genome 50 kb workload), K = 20, S = 4, graph
    built on the device.  Five iterations: the return tuple has the reference's shapes, every cost is finite, t_labels (set
    from iteration 3 on, base.py:422-426) comes back as float64 labels in [0, K) through the one-byte-per-label gather, and
    the loop's own clocks say a warm EM iteration of this block stays under a quarter of a second (measured: ~15 ms)."""
    import torch
    from phylo_hmrf_amd import synthetic
    from phylo_hmrf_amd.hmrf import phyloHMRF
    from phylo_hmrf_amd.tree import PhyloTree
    K, S, N = 20, 4, 4980
    tree = PhyloTree(synthetic.tree_for(S))
    rng = np.random.default_rng(0)
    P = synthetic.sample_ou_params(rng, tree, K)
    mu, cv = tree.mean_cov(P)
    cv = cv + 1e-3 * np.eye(S)
    Xd = synthetic.device_observations(torch, torch.device("cuda", 0), 1, N, N, True, K, mu, cv)
    X = Xd.cpu().numpy().astype(np.float64)
    del Xd
    torch.cuda.empty_cache()
    n = N * (N + 1) // 2
    assert X.shape == (n, S)
    len_vec = [[n, 0, n, N, N, 0, 0, 0, 1, 1]]
    m = phyloHMRF(n_components=K, run_id=0, n_samples=n, n_features=S, observation=X, edge_list=synthetic.tree_for(S),
                  len_vec=len_vec, type_id=1, branch_list=[1.0] * tree.branch_dim, edge_list_1=[None], cons_param=1.0, beta=1.0,
                  beta1=0.5, initial_mode=0, initial_weight=0.3, initial_weight1=0.1, initial_magnitude=1.0, estimate_type=3,
                  random_state=3, quiet=True, device_graph=True, init_method="device")
    try:
        params_vec, params_vec1, params_vecList, it1, it2, cost_vec, t_labels = m.fit_accumulate_test(X, len_vec, 0.0, "t", 5)
        assert params_vec.shape == (K, tree.n_params) and params_vecList.shape == (5, K, tree.n_params)
        assert cost_vec.shape == (5, 4) and np.all(np.isfinite(cost_vec))
        assert t_labels.shape == (n,) and t_labels.dtype == np.float64
        assert t_labels.min() >= 0 and t_labels.max() < K and len(np.unique(t_labels)) > K // 2
        it_ms = np.asarray(m.timing_["iteration"]) * 1e3
        assert len(it_ms) == 5 and np.all(it_ms[2:] < 250.0), it_ms
    finally:
        m.close()


def test_fit_with_the_blocks_in_lockstep_from_one_thread_equals_the_threaded_fit(monkeypatch):
    """block_threads=0 (round 6): the E-step of every whole block driven from the calling thread in lockstep rounds
    (phmrf_mrf_solve_group) -- same start, same M-step draws, deterministic solver: the fit's costs equal the threaded fit's to
    1e-9 and its labels exactly, iteration by iteration (the group call interleaves the blocks' own state machines)."""
    monkeypatch.setenv("PHMRF_DETERMINISTIC", "1")
    import phylo_hmrf as cli
    from phylo_hmrf_amd.hmrf import phyloHMRF
    Xa, lva, ea, tree = cli.synthetic_cache(70, 4, 4, 8, 21)
    Xb, lvb, eb, _ = cli.synthetic_cache(50, 4, 4, 8, 22)
    X = np.vstack([Xa, Xb])
    na, nb = Xa.shape[0], Xb.shape[0]
    len_vec = [lva[0], [nb, na, na + nb, 50, 50, 0, 0, 1, 1, 2]]
    edges = [ea[0], eb[0]]

    def fit(threads, **kw):
        m = phyloHMRF(n_components=4, run_id=0, n_samples=na + nb, n_features=4, observation=X, edge_list=tree, len_vec=len_vec,
                      type_id=1, branch_list=[1.0] * 7, edge_list_1=edges, cons_param=1.0, beta=1.0, beta1=0.5, initial_mode=0,
                      initial_weight=0.3, initial_weight1=0.1, initial_magnitude=1.0, estimate_type=3, random_state=5, quiet=True,
                      mstep_workers=1, block_threads=threads, init_method="sklearn", **kw)
        try:
            return m.fit_accumulate_test(X, len_vec, 0.0, "t", 4)
        finally:
            m.close()

    a, b = fit(2), fit(0)
    np.testing.assert_allclose(a[5], b[5], rtol=1e-9)
    assert np.array_equal(a[6], b[6])
    # with the first block cut into two row tiles: the tiles' lockstep rounds run on the calling thread, the whole block's
    # group solve on ONE helper thread meanwhile (base.py) -- the fit of the same tiles with the blocks on the runner's threads
    c, d = fit(2, tile_parts={0: 2}), fit(0, tile_parts={0: 2})
    np.testing.assert_allclose(c[5], d[5], rtol=1e-9)
    assert np.array_equal(c[6], d[6])
