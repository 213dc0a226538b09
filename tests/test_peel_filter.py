"""CPU: the exact filter in front of the strip alpha-expansions (oracle/mrf_moves.peel, the model of
strip_cols_kernel's sweeps) against the exact strip DP (oracle/mrf_moves.strip_fusion): on seeded problems with
integer and with real-valued costs, every node the DP switches lies in the filter's set U and every strip the DP
changes has a seed -- the filter never loses a move."""
import numpy as np
import pytest

from oracle import mrf_moves as M
from oracle import ref_numpy as R
from oracle import synth


def _problem(seed, H, W, K, diagonal, integer):
    rng = np.random.default_rng(seed)
    n = H * (H + 1) // 2 if diagonal else H * W
    X = rng.uniform(0.5, 2, (n, 2))
    e = R.grid_edges(X, H, W, diagonal, 8)
    eid = np.int64(e[:, :2])
    img = synth.label_image(rng, H, W, K, mean_run=6)
    truth = img[np.triu_indices(H)] if diagonal else img.reshape(-1)
    if integer:
        w = rng.integers(1, 9, len(eid)) / 8.0
        un = rng.integers(0, 12, (n, K)).astype(np.float64) * 0.5
    else:
        w = np.exp(-0.5 * e[:, 2])
        un = rng.gamma(2.0, 1.5, (n, K))
    un[np.arange(n), truth] -= 2.0
    # a labelling close to a local optimum (as in a warm-started E-step): truth with 5 % noise
    lab = np.where(rng.random(n) < 0.05, rng.integers(0, K, n), truth).astype(np.int64)
    return n, eid, w, un, lab


@pytest.mark.parametrize("seed,H,W,K,diagonal,integer", [(1, 23, 70, 4, False, True), (2, 41, 41, 6, True, True),
                                                          (3, 30, 130, 5, False, False), (4, 50, 50, 8, True, False)])
@pytest.mark.parametrize("max_sweeps", [None, 1, 3])
def test_filter_never_loses_a_move(seed, H, W, K, diagonal, integer, max_sweeps):
    n, eid, w, un, lab0 = _problem(seed, H, W, K, diagonal, integer)
    g = M.Graph(n, eid, w)
    beta = 0.75
    n_moves = n_units = n_seeded = 0
    for orient, sr, sc in [(0, 0, 0), (1, 3, 17), (0, 5, 40)]:
        nodes, _ = M.strip_node_table(H, W, diagonal, orient, sr, sc)
        sid = -np.ones(n, dtype=np.int64)
        ss, pp = np.nonzero(nodes >= 0)
        sid[nodes[ss, pp]] = ss
        lab = lab0.copy()
        for alpha in range(K):
            U, seeded = M.peel(g, un, lab, beta, H, W, diagonal, orient, sr, sc, alpha, max_sweeps)
            before = lab.copy()
            M.strip_fusion(g, un, lab, np.full(n, alpha), beta, H, W, diagonal, orient, sr, sc)
            moved = lab != before
            assert not (moved & ~U).any()                              # switched nodes lie in U
            assert seeded[np.unique(sid[moved])].all()                 # changed strips have a seed
            n_moves += len(np.unique(sid[moved]))
            n_units += len(seeded)
            n_seeded += int(seeded.sum())
    assert n_moves > 0 and n_seeded < n_units      # the test saw moves, and the filter settled some pairs without a DP
