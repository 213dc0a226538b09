"""The one-hop flow certificate of the coarse child strips (strip_kernel's look, HISTORY.md 3.1 item 6 (v)), checked by brute
force on small problems (CPU, NumPy): whenever every negative cell of a switch set S is SETTLED by the rule the kernel uses,
S costs at least 0 -- so a strip whose negative cells are all settled has nothing better than all-keep."""
import itertools

import numpy as np

DIRS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]


def _problem(rng, H, W, neg_share, pin_share):
    """switch costs D[H, W] (some negative, some pinned at 1e30) and symmetric pair weights lam[(a, b)] >= 0 on the 8-neighbour grid"""
    D = rng.gamma(2.0, 1.5, size=(H, W))
    D[rng.random((H, W)) < neg_share] *= -rng.uniform(0.2, 1.5)
    D[rng.random((H, W)) < pin_share] = 1.0e30
    lam = {}
    for i in range(H):
        for j in range(W):
            for di, dj in DIRS:
                a, b = (i, j), (i + di, j + dj)
                if 0 <= b[0] < H and 0 <= b[1] < W and a < b:
                    lam[(a, b)] = float(rng.gamma(1.5, 0.6)) if rng.random() > 0.1 else 0.0
    return D, lam


def _w(lam, a, b):
    return lam.get((a, b) if a < b else (b, a), 0.0)


def settled_cells(D, lam):
    """the kernel's rule: a cell with D_B > 0 offers each of its n_B negative neighbours min(lambda_AB, D_B / n_B) (a pinned
    cell, D >= 1e29, offers lambda_AB); a negative cell is settled if what it is offered covers -D_A (with the kernel's margins)"""
    H, W = D.shape
    neg = D < 0
    out = np.zeros((H, W), dtype=bool)
    for i in range(H):
        for j in range(W):
            if not neg[i, j]:
                continue
            recv = 0.0
            for di, dj in DIRS:
                b = (i + di, j + dj)
                if not (0 <= b[0] < H and 0 <= b[1] < W):
                    continue
                nb = sum(1 for ei, ej in DIRS if 0 <= b[0] + ei < H and 0 <= b[1] + ej < W and neg[b[0] + ei, b[1] + ej])
                cap = max(float(D[b]), 0.0)
                recv += min(_w(lam, (i, j), b), cap / nb)          # (nb >= 1: the cell itself is one of b's negative neighbours)
            out[i, j] = recv >= -D[i, j] * 1.0001 + 1e-6
    return out


def cost(D, lam, S):
    c = sum(float(D[a]) for a in S)
    for (a, b), w in lam.items():
        if (a in S) != (b in S):
            c += w
    return c


def test_sets_whose_negative_cells_are_settled_cost_at_least_zero():
    rng = np.random.default_rng(11)
    checked = with_settled = 0
    for trial in range(60):
        H, W = (3, 4) if trial % 2 else (2, 6)
        D, lam = _problem(rng, H, W, neg_share=rng.choice([0.15, 0.3, 0.5]), pin_share=rng.choice([0.0, 0.2]))
        ok = settled_cells(D, lam)
        cells = [(i, j) for i in range(H) for j in range(W)]
        with_settled += int(ok.sum())
        for mask in range(1, 1 << len(cells)):
            S = {c for k, c in enumerate(cells) if (mask >> k) & 1}
            if any(D[c] >= 1e29 for c in S):
                continue                                         # (a pinned cell is in no switch set the DP considers)
            if all(ok[c] for c in S if D[c] < 0):
                checked += 1
                assert cost(D, lam, S) >= -1e-9, (trial, sorted(S))
    assert checked > 10000 and with_settled > 50                 # (the rule did settle cells, and sets were tested)


def test_a_cell_the_rule_leaves_open_can_have_an_improving_set():
    """the other direction is not claimed -- and must not be vacuous: a deep negative cell among cheap neighbours stays open and does pay"""
    D = np.full((3, 3), 0.2)
    D[1, 1] = -5.0
    lam = {}
    for i in range(3):
        for j in range(3):
            for di, dj in DIRS:
                a, b = (i, j), (i + di, j + dj)
                if 0 <= b[0] < 3 and 0 <= b[1] < 3 and a < b:
                    lam[(a, b)] = 0.3
    assert not settled_cells(D, lam)[1, 1]
    assert cost(D, lam, {(1, 1)}) < 0
