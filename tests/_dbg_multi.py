import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import mrf_moves as M
from tests.test_gpu_estep import _block, _integer_problem
H, W, K, diagonal = 23, 70, 4, False
n, eid, w, lp, init = _integer_problem(6, H, W, K, diagonal)
g = M.Graph(n, eid, w)
for mode in ("single-label launches",):
    b = _block(n, 2, K); b.set_graph(eid, w); b.set_grid(H, W, diagonal, 8); b.set_logprob(lp); b.set_labels(init)
    lab = init.astype(np.int64).copy()
    for (orient, sr, sc) in [(0, 0, 0), (1, 3, 17)]:
        if mode == "all labels":
            for a in range(K):
                M.strip_fusion(g, -lp, lab, np.full(n, a), 1.0, H, W, diagonal, orient, sr, sc)
            b.strip_multi_pass(1.0, orient, sr, sc, None)
            got = b.get_labels().astype(np.int64)
            bad = np.flatnonzero(got != lab)
            print(mode, orient, "mismatches", bad, [(divmod(int(i), W), int(got[i]), int(lab[i])) for i in bad[:10]])
            b.set_labels(lab)
        else:
            for a in range(K):
                before = lab.copy()
                M.strip_fusion(g, -lp, lab, np.full(n, a), 1.0, H, W, diagonal, orient, sr, sc)
                b.strip_multi_pass(1.0, orient, sr, sc, [a])
                got = b.get_labels().astype(np.int64)
                bad = np.flatnonzero(got != lab)
                print(mode, orient, "alpha", a, "model moved", int((lab != before).sum()), "mismatches",
                      [(divmod(int(i), W), int(got[i]), int(lab[i]), int(before[i])) for i in bad[:10]])
                b.set_labels(lab)
    b.close()
