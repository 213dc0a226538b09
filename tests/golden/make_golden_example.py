#!/usr/bin/env python3
"""Generate the example_input-derived fixtures from the REFERENCE ITSELF (build container only; companion of
make_golden.py, whose lib2to3 import shim it reuses; nothing of the reference's source is written into the repo).

    python tests/golden/make_golden_example.py        # needs /root/reference and `make -C oracle ref`

BASELINE config 1 is "example_input chr21+chr22, 4 species".  The reference ships only part of that data
(/root/reference/.MISSING_LARGE_BLOBS): chr22 for gorGor4, panTro5, panPan2; hg38 chr21/chr22 and two chr21 files are
absent.  Deviations, stated once here and in DESIGN.md:
  * chr22 only;
  * hg38 chr22 is a deterministic SURROGATE: per locus the median of the species that have a (non-NaN) value there,
    over the union of the three present files, written with 4 decimals like the real files;
  * unless a fixture says otherwise the smoothing filter is off (filter_mode 2, sigma 0): medpy's anisotropic diffusion
    (filter_mode 0, the CLI default) is not installed, and where it is exercised below the reference runs with the
    build's own restatement (phylo_hmrf_amd.preprocess.anisotropic_diffusion) plugged in as `medpy`.

Fixtures:
  grid_edges.npz            edge_weightlist_grid3_undirected_unsym / _undirected (utility.py:1871-2053) on small
                            diagonal and off-diagonal blocks, 8- and 4-neighbour, with a zero-norm row
  example_loader.npz        raw contact rows of a 160-bin window of chr22 (4 species) + what the reference's
                            quantile_contact_vec / load_data_chromosome2 make of (a) its first 120 bins as chr22 with
                            filter off / Gaussian 0.25 / diffusion, (b) the window moved onto the chr3 centromere gap
                            (utility.py:385): two diagonal regions and one off-diagonal region
  example_chr22_em.npz      the reference's fit_accumulate_test (K=20, --miter 5, gco swap through pygco's
                            quantisation) on the first 300 bins of the chr22 synteny block: per iteration the E-step
                            inputs (means_, _covars_, labels_local) and the reference's labels, costs and E_float
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402
from oracle import gco_ref, ref_numpy  # noqa: E402

REF = MG.REF
EX = os.path.join(REF, "example_input")
SPECIES = ["gorGor4", "panTro5", "panPan2", "hg38"]
RES = 50000
SYNTENY22 = (16554072, 50780062)          # example_input/chr22.synteny.txt
FIRST_BIN = 331                            # first bin whose start is >= the synteny start (16554072 // 50000 + 1)


def read_species_tables():
    """The three chr22 files the reference ships + the hg38 surrogate -> {species: (pos1, pos2, value)}."""
    import pandas as pd
    tabs = {}
    for s in SPECIES[:3]:
        t = pd.read_csv("%s/test_data/hic_%s/chr22.%dK.txt" % (EX, s, RES // 1000), header=None, sep="\t")
        tabs[s] = (np.asarray(t[0], dtype=np.int64), np.asarray(t[1], dtype=np.int64), np.asarray(t[2], dtype=np.float64))
    keys = np.unique(np.concatenate([(p1 << 32) + p2 for p1, p2, _ in tabs.values()]))
    vals = np.full((keys.shape[0], 3), np.nan)
    for i, s in enumerate(SPECIES[:3]):
        p1, p2, v = tabs[s]
        vals[np.searchsorted(keys, (p1 << 32) + p2), i] = v
    have = ~np.all(np.isnan(vals), axis=1)
    med = np.round(np.nanmedian(vals[have], axis=1), 4)
    tabs["hg38"] = (keys[have] >> 32, keys[have] & ((1 << 32) - 1), med)
    return tabs


def write_input_dir(tabs, chrom, synteny, chrom_sizes_line=None):
    """An example_input-style directory in a temp dir -> (dir, filename_list)."""
    d = tempfile.mkdtemp(prefix="phmrf_exin_")
    for f in ("edge.1.txt", "branch_length.1.txt", "species_name.1.txt", "hg38.chrom.sizes"):
        shutil.copy(os.path.join(EX, f), d)
    flist = []
    for s in SPECIES:
        p = "%s/hic_%s" % (d, s)
        os.makedirs(p)
        flist.append(p)
        p1, p2, v = tabs[s]
        with open("%s/chr%s.%dK.txt" % (p, chrom, RES // 1000), "w") as f:
            for a, b, c in zip(p1, p2, v):
                f.write("%d\t%d\t%s\n" % (a, b, "NaN" if np.isnan(c) else "%.4f" % c))
    with open("%s/chr%s.synteny.txt" % (d, chrom), "w") as f:
        f.write("%d\t%d\t%d\n" % (synteny[0], synteny[1], synteny[1] - synteny[0]))
    return d, flist


def window(tabs, bin0, nbins, shift_bins=0):
    out = {}
    lo, hi = bin0 * RES, (bin0 + nbins) * RES
    for s, (p1, p2, v) in tabs.items():
        k = (p1 >= lo) & (p1 < hi) & (p2 >= lo) & (p2 < hi)
        out[s] = (p1[k] + shift_bins * RES, p2[k] + shift_bins * RES, v[k])
    return out


def main():
    if not gco_ref.available():
        raise SystemExit("build oracle/_ref first: make -C oracle ref")
    mod, tmp = MG.import_reference()
    import utility
    from phylo_hmrf_amd import preprocess
    sys.modules["medpy.filter.smoothing"].anisotropic_diffusion = (
        lambda img, niter=1, kappa=50, gamma=0.1, voxelspacing=None, option=1:
        preprocess.anisotropic_diffusion(img, niter=niter, kappa=kappa, gamma=gamma, option=option))
    utility.anisotropic_diffusion = sys.modules["medpy.filter.smoothing"].anisotropic_diffusion
    import scipy.ndimage
    if not hasattr(scipy.ndimage, "filters"):                       # utility.py:1588 spells it scipy.ndimage.filters
        import types
        scipy.ndimage.filters = types.SimpleNamespace(gaussian_filter=scipy.ndimage.gaussian_filter)
    rng = np.random.default_rng(20261003)

    # ---------------- grid_edges.npz: the two edge builders --------------------------------------------------
    ge = {}
    for tag, H, W, diag in (("diag", 9, 9, True), ("off", 7, 10, False)):
        n = H * (H + 1) // 2 if diag else H * W
        X = np.abs(rng.standard_normal((n, 4))) + 0.05
        X[3] = 0.0                                                   # zero-norm row: d = 0 / (0 + 1e-16)
        if diag:
            ii, jj = np.triu_indices(H)
            serial = ii * W + jj
        else:
            serial = np.arange(H * W)
        ge[tag + "_X"] = X
        for nn in (8, 4):
            if diag:
                e = MG.quiet(utility.edge_weightlist_grid3_undirected_unsym, X, serial, H, "", nn)
            else:
                e = MG.quiet(utility.edge_weightlist_grid3_undirected, X, serial, (H, W), "", nn)
            ge["%s_nn%d" % (tag, nn)] = np.asarray(e, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "grid_edges.npz"), **ge)

    # ---------------- example_loader.npz ---------------------------------------------------------------------
    tabs = read_species_tables()
    rec = {}
    raw = window(tabs, FIRST_BIN, 160)
    for s in SPECIES:
        rec["raw_%s_pos1" % s], rec["raw_%s_pos2" % s], rec["raw_%s_value" % s] = (
            raw[s][0].astype(np.int32), raw[s][1].astype(np.int32), raw[s][2])
    rec["first_bin"] = FIRST_BIN
    # (a) chr22, first 120 bins
    w120 = window(tabs, FIRST_BIN, 120)
    syn = (FIRST_BIN * RES, (FIRST_BIN + 120) * RES)
    d22, fl22 = write_input_dir(w120, "22", syn)
    ref_sizes = d22 + "/hg38.chrom.sizes"
    mv = MG.quiet(utility.quantile_contact_vec, [22], RES, ref_sizes, fl22, SPECIES)
    rec["a_quantile"] = mv
    x_max = float(np.median(mv[:, 6]))
    rec["a_synteny"] = np.array(syn)
    for tag, fm, sigma in (("none", 2, 0.0), ("gauss", 2, 0.25), ("diffusion", 0, 0.25)):
        samples, len_vec, elv = MG.quiet(utility.load_data_chromosome2, [22], x_max, 0, RES, 8, fm, sigma, 0, ref_sizes,
                                         fl22, SPECIES, d22, "golden")
        rec["a_%s_samples" % tag] = samples
        rec["a_%s_lenvec" % tag] = np.asarray(len_vec, dtype=np.int64)
        if tag == "none":
            rec["a_none_edges"] = np.asarray(elv[0], dtype=np.float64)
        else:                                                        # the edge builder is pinned by grid_edges.npz
            N = int(len_vec[0][3])
            assert np.array_equal(np.asarray(elv[0]), ref_numpy.grid_edges(samples, N, N, True, 8))
    # 4-neighbour, diagonal regions only (--dtype 1)
    samples, len_vec, elv = MG.quiet(utility.load_data_chromosome2, [22], x_max, 0, RES, 4, 2, 0.0, 1, ref_sizes, fl22,
                                     SPECIES, d22, "golden")
    assert np.array_equal(samples, rec["a_none_samples"])
    rec["a_none_edges_nn4"] = np.asarray(elv[0], dtype=np.float64)
    # (b) the same 160 bins moved onto the chr3 centromere gap: bins 1760 .. 1919 = 88.0 .. 96.0 Mb
    shift = 1760 - FIRST_BIN
    w3 = window(tabs, FIRST_BIN, 160, shift_bins=shift)
    syn3 = (1760 * RES, 1920 * RES)
    d3, fl3 = write_input_dir(w3, "3", syn3)
    mv3 = MG.quiet(utility.quantile_contact_vec, [3], RES, d3 + "/hg38.chrom.sizes", fl3, SPECIES)
    x_max3 = float(np.median(mv3[:, 6]))
    samples, len_vec, elv = MG.quiet(utility.load_data_chromosome2, [3], x_max3, 0, RES, 8, 0, 0.25, 0,
                                     d3 + "/hg38.chrom.sizes", fl3, SPECIES, d3, "golden")
    rec["b_shift_bins"] = shift
    rec["b_synteny"] = np.array(syn3)
    rec["b_quantile"] = mv3
    rec["b_samples"] = samples
    rec["b_lenvec"] = np.asarray(len_vec, dtype=np.int64)
    for i, e in enumerate(elv):
        rec["b_edges%d" % i] = np.asarray(e, dtype=np.float64)
    assert len(elv) == 3 and [int(lv[8]) for lv in len_vec] == [1, 0, 1], len_vec
    np.savez_compressed(os.path.join(HERE, "example_loader.npz"), **rec)

    # ---------------- example_chr22_em.npz: the reference's own fit on real Hi-C ----------------------------------
    NB = 300
    dfull, flfull = write_input_dir(tabs, "22", (SYNTENY22[0], SYNTENY22[0] + NB * RES))
    mvf = MG.quiet(utility.quantile_contact_vec, [22], RES, dfull + "/hg38.chrom.sizes", flfull, SPECIES)
    x_maxf = float(np.median(mvf[:, 6]))
    samples, len_vec, elv = MG.quiet(utility.load_data_chromosome2, [22], x_maxf, 0, RES, 8, 2, 0.0, 0,
                                     dfull + "/hg38.chrom.sizes", flfull, SPECIES, dfull, "golden")
    K, beta, beta1, m_iter = 20, 1.0, 0.5, 5
    np.random.seed(22)                                    # the reference draws from the global NumPy state
    m = MG.quiet(mod.phyloHMRF, n_components=K, run_id=0, n_samples=samples.shape[0], n_features=4, observation=samples,
                 edge_list=MG.TREE4, len_vec=len_vec, type_id=1, branch_list=[0, 32, 20, 6, 6, 6, 12],
                 edge_list_1=elv, cons_param=1.0, beta=beta, beta1=beta1, initial_mode=0, initial_weight=0.3,
                 initial_weight1=0.1, initial_magnitude=1.0, learning_rate=0.001, estimate_type=3, max_iter=100,
                 n_iter=5000, tol=1e-7)
    # NumPy 2 raises LinAlgError from np.linalg.cond when SLSQP (SciPy 1.15) steps the initial per-cluster OU fit
    # (phylo_hmrf.py:1427-1498, off the hot path, no try/except there) into parameters whose covariance is NaN; the
    # optimiser is told "very bad point" instead, and the reference's own retry / random-restart logic carries on.
    lik_single = m._ou_lik_varied_single

    def lik_single_guarded(params, X_):
        try:
            v = lik_single(params, X_)
        except np.linalg.LinAlgError:
            return 1e10
        return v if np.isfinite(v) else 1e10

    m._ou_lik_varied_single = lik_single_guarded
    trace = []
    state = {"local": None}
    orig_mstep, orig_init = m._do_mstep, m._init

    def init_hook(X_, lengths=None):
        orig_init(X_, lengths=lengths)
        state["local"] = np.asarray(m.labels_local).copy()

    def mstep_hook(stats):
        # called right after the E-step of an iteration and its bookkeeping: self.labels are this iteration's labels,
        # means_/_covars_ the parameters they were computed with, state['local'] the warm start they started from
        lp = m._compute_log_likelihood(samples)
        eid, w = m.edge_idList_undirected_vec[0], m.edge_weightList_undirected_vec[0]
        lab = np.int64(m.labels)
        trace.append(dict(means=m.means_.copy(), covars=m._covars_.copy(), init=np.int64(state["local"]), labels=lab,
                          efloat=np.array(ref_numpy.mrf_energy(lab, lp, eid, w, beta)),
                          efloat_init=np.array(ref_numpy.mrf_energy(np.int64(state["local"]), lp, eid, w, beta)),
                          stats_post=stats["post"].copy(), stats_obs=stats["obs"].copy(),
                          stats_oo=stats["obs*obs.T"].copy()))
        state["local"] = np.asarray(m.labels_local).copy()
        orig_mstep(stats)

    m._init, m._do_mstep = init_hook, mstep_hook
    res = MG.quiet(m.fit_accumulate_test, samples, len_vec, 0.001, "golden", m_iter)
    params_vec, params_vec1, params_vecList, it1, it2, cost_vec, t_labels = res
    assert len(trace) == m_iter
    out = dict(X=samples, len_vec=np.asarray(len_vec, dtype=np.int64), edges=np.asarray(elv[0], dtype=np.float64),
               quantile=mvf, K=K, beta=beta, beta1=beta1, m_iter=m_iter, cost_vec=cost_vec, iter_id1=it1, iter_id2=it2,
               t_labels=np.uint8(t_labels), params_vec=params_vec, params_vec1=params_vec1)
    for key in trace[0]:
        arr = np.stack([t[key] for t in trace])
        out["it_" + key] = np.uint8(arr) if key in ("init", "labels") else arr
    np.savez_compressed(os.path.join(HERE, "example_chr22_em.npz"), **out)

    os.chdir(ROOT)
    for d in (d22, d3, dfull, tmp):
        shutil.rmtree(d, ignore_errors=True)
    print("example_input fixtures written to", HERE)


if __name__ == "__main__":
    main()
