"""CPU: the host pre-processing (phylo_hmrf_amd/preprocess.py, graph_host.py, libphmrf_host.so) and the oracle's edge
builder against fixtures recorded from the reference's OWN loader and edge builders
(tests/golden/make_golden_example.py: utility.quantile_contact_vec, load_data_chromosome2,
edge_weightlist_grid3_undirected_unsym / _undirected run on example_input chr22 rows)."""
import os

import numpy as np
import pytest

from oracle import ref_numpy as R
from phylo_hmrf_amd import graph_host, preprocess

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SPECIES = ["gorGor4", "panTro5", "panPan2", "hg38"]
RES = 50000
HG38_SIZES = "chr3\t198295559\nchr21\t46709983\nchr22\t50818468\n"        # public genome facts (UCSC hg38.chrom.sizes)


# ---- f2: the edge builders (utility.py:1871-2053) -----------------------------------------------------------------
@pytest.mark.parametrize("tag,H,W,diag", [("diag", 9, 9, True), ("off", 7, 10, False)])
@pytest.mark.parametrize("nn", [8, 4])
@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_grid_edges_match_reference(tag, H, W, diag, nn, impl):
    g = np.load(os.path.join(G, "grid_edges.npz"))
    fn = R.grid_edges if impl == "oracle" else graph_host.grid_edges
    e = fn(g[tag + "_X"], H, W, diag, nn)
    ref = g["%s_nn%d" % (tag, nn)]
    assert e.shape == ref.shape
    assert np.array_equal(e[:, :2], ref[:, :2])
    assert np.array_equal(e[:, 2], ref[:, 2])             # same operations in the same order: bit-exact
    assert ref[:, 2].max() > 1e15                          # the zero-norm row: d = |x_j|^2 / (0 + 1e-16)


# ---- median fill: native scan vs the reference's loops restated in Python -------------------------------------------
def _median_fill_py(mtx, symmetric):
    """near_interpolation1 / 1a (utility.py:603-660), restated: the checker of the native version."""
    n1, n2 = mtx.shape
    for i in range(2, n1 - 1):
        for j in range(i if symmetric else 2, n2 - 1):
            if mtx[i, j] < preprocess.THRESH1:
                w = np.delete(mtx[i - 1:i + 2, j - 1:j + 2].ravel(), 4)
                m1 = np.median(w)
                if m1 > preprocess.THRESH1:
                    mtx[i, j] = m1
                    if symmetric:
                        mtx[j, i] = m1
    return mtx


@pytest.mark.parametrize("symmetric", [True, False])
def test_median_fill_native_vs_python(symmetric):
    rng = np.random.default_rng(5)
    n1, n2 = (23, 23) if symmetric else (17, 29)
    a = rng.random((n1, n2)) * (rng.random((n1, n2)) > 0.45)
    if symmetric:
        a = np.triu(a) + np.triu(a, 1).T
    b = a.copy()
    cnt = preprocess.median_fill(a, symmetric)
    _median_fill_py(b, symmetric)
    assert np.array_equal(a, b)
    assert cnt[0] >= cnt[1] > 0
    with pytest.raises(ValueError):
        preprocess.median_fill(np.zeros((3, 4)), True)
    with pytest.raises(ValueError):
        preprocess.median_fill(np.zeros((4, 4), dtype=np.float32), True)
    preprocess.median_fill(np.zeros((2, 2)), True)        # smaller than the scan window: nothing to do


# ---- the loader on example_input rows -----------------------------------------------------------------------------
def _write_dir(tmp, g, chrom, first, nbins, shift, synteny):
    d = str(tmp)
    with open(os.path.join(d, "hg38.chrom.sizes"), "w") as f:
        f.write(HG38_SIZES)
    flist = []
    lo, hi = first * RES, (first + nbins) * RES
    for s in SPECIES:
        p = os.path.join(d, "hic_" + s)
        os.makedirs(p, exist_ok=True)
        flist.append(p)
        p1, p2, v = g["raw_%s_pos1" % s].astype(np.int64), g["raw_%s_pos2" % s].astype(np.int64), g["raw_%s_value" % s]
        k = (p1 >= lo) & (p1 < hi) & (p2 >= lo) & (p2 < hi)
        with open(os.path.join(p, "chr%s.%dK.txt" % (chrom, RES // 1000)), "w") as f:
            for a, b, c in zip(p1[k] + shift * RES, p2[k] + shift * RES, v[k]):
                f.write("%d\t%d\t%s\n" % (a, b, "NaN" if np.isnan(c) else "%.4f" % c))
    with open(os.path.join(d, "chr%s.synteny.txt" % chrom), "w") as f:
        f.write("%d\t%d\t%d\n" % (synteny[0], synteny[1], synteny[1] - synteny[0]))
    return d, flist


@pytest.fixture(scope="module")
def loader_golden():
    return np.load(os.path.join(G, "example_loader.npz"))


def test_loader_chr22_window(tmp_path, loader_golden):
    """The raw loader on a window of example_input's chr22 against the reference's own loader run: bit-exact for the
    'none' and 'gauss' settings.  The 'diffusion' arrays of the fixture are SELF-GENERATED as far as the filter goes (the
    reference's loader was run with this build's Perona-Malik restatement plugged in for the absent medpy,
    tests/golden/make_golden_example.py:103-106): that case pins the pipeline around the filter; the filter itself is
    checked by test_anisotropic_diffusion_against_the_published_update_rule."""
    g = loader_golden
    first = int(g["first_bin"])
    d, flist = _write_dir(tmp_path, g, "22", first, 120, 0, g["a_synteny"])
    sizes = os.path.join(d, "hg38.chrom.sizes")
    mv = preprocess.quantile_contact_vec([22], RES, sizes, flist, SPECIES)
    assert np.allclose(mv, g["a_quantile"], rtol=1e-12, atol=0)
    x_max = float(np.median(mv[:, 6]))                                        # phylo_hmrf.py:1662-1663
    for tag, fm, sigma, exact in (("none", 2, 0.0, True), ("gauss", 2, 0.25, True), ("diffusion", 0, 0.25, True)):
        samples, len_vec, elv = preprocess.load_data_chromosome2([22], x_max, 0, RES, 8, fm, sigma, 0, sizes, flist,
                                                                 SPECIES, d, "t")
        assert np.array_equal(np.asarray(len_vec), g["a_%s_lenvec" % tag]), tag
        ref = g["a_%s_samples" % tag]
        assert samples.shape == ref.shape
        # filter off / Gaussian: the reference's own arithmetic, reproduced to the last bit.  'diffusion': the fixture
        # was recorded with THIS build's Perona-Malik restatement standing in for medpy (absent): it pins the pipeline
        # around the filter, not the filter (parity unpinned, see preprocess.py).
        assert np.array_equal(samples, ref), (tag, np.abs(samples - ref).max())
        if tag == "none":
            assert np.array_equal(elv[0], g["a_none_edges"])
    samples, len_vec, elv = preprocess.load_data_chromosome2([22], x_max, 0, RES, 4, 2, 0.0, 1, sizes, flist, SPECIES, d)
    assert np.array_equal(elv[0], g["a_none_edges_nn4"])
    # real Hi-C: the window has empty cells that the median fill closed and cells it could not close
    assert (g["a_none_samples"] == 0).any() and (g["a_none_samples"] > 0).mean() > 0.5


def test_loader_centromere_split(tmp_path, loader_golden):
    """chr3 spans a centromere gap (utility.py:385): two diagonal regions + the off-diagonal region between them."""
    g = loader_golden
    first, shift = int(g["first_bin"]), int(g["b_shift_bins"])
    d, flist = _write_dir(tmp_path, g, "3", first, 160, shift, g["b_synteny"])
    sizes = os.path.join(d, "hg38.chrom.sizes")
    mv = preprocess.quantile_contact_vec([3], RES, sizes, flist, SPECIES)
    assert np.allclose(mv, g["b_quantile"], rtol=1e-12, atol=0)
    x_max = float(np.median(mv[:, 6]))
    samples, len_vec, elv = preprocess.load_data_chromosome2([3], x_max, 0, RES, 8, 0, 0.25, 0, sizes, flist, SPECIES, d)
    assert np.array_equal(np.asarray(len_vec), g["b_lenvec"])
    assert [lv[8] for lv in len_vec] == [1, 0, 1]
    assert np.array_equal(samples, g["b_samples"])
    for i in range(3):
        assert np.array_equal(elv[i], g["b_edges%d" % i])
    # --dtype 1 keeps the diagonal regions only (utility.py:396-400)
    s1, lv1, e1 = preprocess.load_data_chromosome2([3], x_max, 0, RES, 8, 0, 0.25, 1, sizes, flist, SPECIES, d)
    assert [lv[8] for lv in lv1] == [1, 1] and len(e1) == 2
    assert lv1[1][1] == lv1[0][2] and lv1[1][2] == s1.shape[0]


def test_loader_errors(tmp_path, loader_golden):
    g = loader_golden
    d, flist = _write_dir(tmp_path, g, "22", int(g["first_bin"]), 40, 0, g["a_synteny"])
    sizes = os.path.join(d, "hg38.chrom.sizes")
    with pytest.raises(IOError):
        preprocess.quantile_contact_vec([21], RES, sizes, flist, SPECIES)          # no chr21 files
    with pytest.raises(ValueError):
        preprocess.quantile_contact_vec([7], RES, sizes, flist, SPECIES)           # chr7 not in the sizes file


def test_loader_filter_mode_1_runs_the_bilateral_filter(tmp_path, loader_golden):
    """--filter_mode 1 (utility.py:1575-1582): same nodes, lengths and edge structure as the other modes, the features
    smoothed by denoise_bilateral(sigma_color=0.5, sigma_spatial=5) -- checked against the loader's unfiltered image run
    through the oracle's restatement of the filter."""
    g = loader_golden
    d, flist = _write_dir(tmp_path, g, "22", int(g["first_bin"]), 40, 0, g["a_synteny"])
    sizes = os.path.join(d, "hg38.chrom.sizes")
    mv = preprocess.quantile_contact_vec([22], RES, sizes, flist, SPECIES)
    x_max = float(np.median(mv[:, 6]))                                        # phylo_hmrf.py:1662-1663
    s1, lv1, e1 = preprocess.load_data_chromosome2([22], x_max, 0, RES, 8, 1, 0.25, 0, sizes, flist, SPECIES, d)
    s2, lv2, e2 = preprocess.load_data_chromosome2([22], x_max, 0, RES, 8, 2, 0.0, 0, sizes, flist, SPECIES, d)   # no filter
    assert s1.shape == s2.shape and [list(a) for a in lv1] == [list(a) for a in lv2]
    assert all(np.array_equal(a[:, :2], b[:, :2]) for a, b in zip(e1, e2))
    assert np.all(np.isfinite(s1)) and not np.allclose(s1, s2)
    # the first block is diagonal: rebuild its image from the unfiltered nodes, filter it with the oracle, compare
    n, _, _, H, W = [int(v) for v in lv2[0][:5]]
    assert H == W and n == H * (H + 1) // 2
    iu = np.triu_indices(H)
    for c in range(s2.shape[1]):
        img = np.zeros((H, W))
        img[iu] = s2[:n, c]
        img = img + img.T - np.diag(np.diag(img))
        ref = R.denoise_bilateral(img, sigma_color=0.5, sigma_spatial=5)
        np.testing.assert_allclose(s1[:n, c], ref[iu], rtol=1e-12, atol=1e-12)


def _perona_malik_by_the_book(img, niter, kappa, gamma):
    """Perona & Malik 1990, eq. (7)-(8) as published, pixel by pixel: I <- I + gamma * sum over the four neighbours q of
    g(I_q - I_p) (I_q - I_p) with g(d) = exp(-(d / kappa)^2); a missing neighbour (image border) contributes no flux.
    Written independently of preprocess.anisotropic_diffusion (which works on forward-difference / flux arrays)."""
    cur = np.array(img, dtype=np.float64)
    H, W = cur.shape
    for _ in range(niter):
        nxt = cur.copy()
        for i in range(H):
            for j in range(W):
                acc = 0.0
                for di, dj in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                    a, b = i + di, j + dj
                    if 0 <= a < H and 0 <= b < W:
                        d = cur[a, b] - cur[i, j]
                        acc += np.exp(-(d / kappa) ** 2) * d
                nxt[i, j] = cur[i, j] + gamma * acc
        cur = nxt
    return cur


@pytest.mark.parametrize("shape,niter,kappa", [((3, 3), 1, 50.0), ((5, 5), 5, 50.0), ((5, 5), 10, 50.0), ((4, 7), 5, 2.0)])
def test_anisotropic_diffusion_against_the_published_update_rule(shape, niter, kappa):
    """An independent check of the Perona-Malik restatement (option 1; the reference's parameters are niter 5 or 10,
    kappa 50, gamma 0.1: utility.py:1566-1573): the published per-pixel update rule written out by hand in the test,
    on small images, one case with kappa near the data's differences so that the edge-stopping function matters.
    (The `a_diffusion_*` arrays of tests/golden/example_loader.npz were recorded from the reference's loader WITH THIS
    BUILD'S filter plugged in where medpy is missing, so they pin everything around the filter and nothing of it.)"""
    rng = np.random.default_rng(5)
    img = rng.random(shape) * 6.0
    want = _perona_malik_by_the_book(img, niter, kappa, 0.1)
    got = preprocess.anisotropic_diffusion(img, niter=niter, kappa=kappa, gamma=0.1, option=1)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)          # the filter works in float32 like medpy
    if shape == (3, 3):
        # the centre pixel of a 3 x 3 image after one step, written out
        c, nb = img[1, 1], (img[0, 1], img[2, 1], img[1, 0], img[1, 2])
        by_hand = c + 0.1 * sum(np.exp(-((q - c) / kappa) ** 2) * (q - c) for q in nb)
        assert abs(float(got[1, 1]) - by_hand) < 2e-6


def test_anisotropic_diffusion_properties():
    """Perona-Malik restatement (parity unpinned): conserves the mean (zero-flux borders), never widens the range,
    leaves a constant image alone, and smooths small differences more than large ones."""
    rng = np.random.default_rng(0)
    img = rng.random((40, 50)) * 3.0
    out = preprocess.anisotropic_diffusion(img, niter=5, kappa=50, gamma=0.1, option=1)
    assert out.dtype == np.float32 and out.shape == img.shape
    assert abs(out.mean() - img.mean()) < 1e-5
    assert out.min() >= img.min() - 1e-6 and out.max() <= img.max() + 1e-6
    assert out.std() < img.std()
    c = np.full((8, 9), 2.5)
    assert np.array_equal(preprocess.anisotropic_diffusion(c, niter=3), c.astype(np.float32))
    step = np.zeros((1, 20)); step[0, 10:] = 1000.0           # an edge far above kappa survives
    o2 = preprocess.anisotropic_diffusion(step, niter=5, kappa=50, gamma=0.1)
    assert abs(o2[0, 9]) < 1e-3 and abs(o2[0, 10] - 1000.0) < 1e-3


# ---- filter_mode 1: bilateral filter (scikit-image absent: parity unpinned; native vs the oracle's restatement) ----------
@pytest.mark.parametrize("shape,sc,ss", [((23, 31), 0.5, 5), ((40, 40), 0.5, 5), ((7, 5), 0.2, 1), ((16, 9), None, 2)])
def test_bilateral_native_matches_the_oracle(shape, sc, ss):
    rng = np.random.default_rng(5)
    img = np.abs(rng.standard_normal(shape)) * 2.0 + (np.arange(shape[1]) > shape[1] // 2) * 3.0     # an edge + noise
    got = preprocess.denoise_bilateral(img, sigma_color=sc, sigma_spatial=ss)
    ref = R.denoise_bilateral(img, sigma_color=sc, sigma_spatial=ss)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13)
    assert got.shape == img.shape and np.all(got >= 0) and got.max() <= img.max() + 1e-12


def test_bilateral_by_the_book_pixel_loop():
    """The published per-pixel loop (window, zero padding, two look-up tables), written out for single pixels,
    independently of the vectorised oracle and of the native code."""
    rng = np.random.default_rng(8)
    img = rng.random((12, 14)) * 4.0
    sc, ss, bins = 0.5, 1.5, 10000
    win = max(5, 2 * int(np.ceil(3 * ss)) + 1)
    ext = (win - 1) // 2
    mx = img.max()
    got = preprocess.denoise_bilateral(img, sigma_color=sc, sigma_spatial=ss)
    for r, c in [(0, 0), (5, 7), (11, 13), (3, 0), (0, 9)]:
        tot = wsum = 0.0
        for wr in range(-ext, ext + 1):
            for wc in range(-ext, ext + 1):
                rr, cc = r + wr, c + wc
                v = img[rr, cc] if (0 <= rr < 12 and 0 <= cc < 14) else 0.0
                b = min(int(abs(img[r, c] - v) * (bins / mx)), bins - 1)
                w = np.exp(-0.5 * (np.hypot(wr, wc) / ss) ** 2) * np.exp(-0.5 * (b * mx / bins / sc) ** 2)
                tot += v * w
                wsum += w
        assert abs(got[r, c] - tot / wsum) < 1e-12


def test_bilateral_properties_and_errors():
    flat = np.full((9, 9), 2.5)
    assert np.array_equal(preprocess.denoise_bilateral(flat, 0.5, 5), flat)             # min == max: returned unchanged
    rng = np.random.default_rng(2)
    # an interior far from the zero padding: a step edge survives, the noise on either side shrinks
    img = np.where(np.arange(80)[None, :] < 40, 1.0, 6.0) + 0.05 * rng.random((80, 80))
    out = preprocess.denoise_bilateral(img, sigma_color=0.5, sigma_spatial=2)
    core = (slice(20, 60), slice(20, 60))
    assert out[core][:, :18].std() < 0.5 * img[core][:, :18].std()
    assert out[30, 45] - out[30, 34] > 4.9                                               # the edge is not blurred away
    with pytest.raises(ValueError):
        preprocess.denoise_bilateral(np.array([[1.0, -0.5], [0.2, 0.3]]), 0.5, 1)        # skimage: ValueError
    with pytest.raises(ValueError):
        preprocess.denoise_bilateral(np.ones((4, 4, 2)), 0.5, 1)
