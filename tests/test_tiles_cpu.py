"""CPU: the lockstep orchestration of row tiles (phylo_hmrf_amd/tiles.py) with a NumPy test double in place of the GPU
blocks (tests/fake_tile_block.py: a round = one ICM sweep of the oracle's move model) -- all tiles in one process, and the
tiles of one block on TWO RANKS over gloo.  The HIP kernels behind the same orchestration: tests/test_gpu_tiles.py,
tests/test_gpu_dist.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import mrf_moves as M
from oracle import ref_numpy as R
from oracle import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(H, W, diag, K, seed=3):
    blk = synth.make_block(seed, H, W, 4, K, diag)
    lp = R.log_multivariate_normal_density_full(blk["X"], blk["means"], blk["covars"])
    w, eid = R.edge_weights_from_distance(blk["edges"], 0.5)
    init = np.random.default_rng(seed + 1).integers(0, K, blk["X"].shape[0])
    return blk, lp, w, eid, init


def _make_group(blk, lp, init, H, W, diag, K, parts, owners, rank, comm):
    from phylo_hmrf_amd import tiles
    from tests.fake_tile_block import FakeTileBlock
    rows = tiles.split_rows(H, W, diag, parts)

    def load(tl):
        tl.b.set_observations(blk["X"][tl.global_slice()])

    g = tiles.make_group(0, (H, W, diag), rows, owners, rank, 4, K, FakeTileBlock, load, comm, 8, 0.5, edges=blk["edges"])
    for tl in g.local.values():
        tl.b.set_logprob(lp[tl.global_slice()])
        tl.b.set_labels(init[tl.global_slice()])
    return g


def _solve(g):
    energies = []
    g.begin(1.0, {})
    while True:
        g.launch()
        st = g.finish_round()
        energies.append(next(iter(g.local.values())).b.last_energy)          # the sum over ALL tiles, as every tile sees it
        if st != 0:
            break
    return g.end(want_result=True), energies


@pytest.mark.parametrize("H,W,diag,parts", [(40, 40, True, 2), (48, 48, True, 3), (30, 44, False, 3)])
def test_lockstep_rounds_with_all_tiles_local(H, W, diag, parts):
    K = 5
    blk, lp, w, eid, init = _problem(H, W, diag, K)
    n = lp.shape[0]
    g = _make_group(blk, lp, init, H, W, diag, K, parts, [0] * parts, 0, None)
    res, energies = _solve(g)
    labels = np.zeros(n, dtype=np.int64)
    for tl in g.local.values():
        labels[tl.owned_global_slice()] = tl.b.get_labels()[tl.owned_local_slice()]
    graph = M.Graph(n, eid, w)
    e = M.energy(graph, -lp, labels, 1.0)[0]
    # the tiles' energies over their owned rows add up to the whole block's, and it never went up from round to round
    assert abs(res["energy"] - e) <= 1e-9 * abs(e)
    assert abs(energies[-1] - e) <= 1e-9 * abs(e)
    for a, c in zip(energies, energies[1:]):
        assert c <= a + 1e-9 * abs(a)
    assert e < M.energy(graph, -lp, init, 1.0)[0]
    # halo rows hold what their owners hold
    for tl in g.local.values():
        assert np.array_equal(tl.b.get_labels(), np.int32(labels[tl.global_slice()]))
    # a fixed point of single-site moves on the WHOLE block (two quiet rounds, one of each pin parity, ended the solve)
    full = labels.copy()
    M.icm_sweep(graph, -lp, full, 1.0, *M.icm_colours(H, W, diag))
    assert np.array_equal(full, labels)
    assert res["converged"]


WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from phylo_hmrf_amd import tiles
from tests.test_tiles_cpu import _problem, _make_group, _solve
H, W, diag, K, parts = 48, 48, True, 5, 3
blk, lp, w, eid, init = _problem(H, W, diag, K)
owners = [0, 1, 0]                                   # the middle tile on the other rank: both of its cuts cross ranks
comm = tiles.GroupComm(owners, None)
g = _make_group(blk, lp, init, H, W, diag, K, parts, owners, rank, comm)
res, energies = _solve(g)
out = {str(t): tl.b.get_labels()[tl.owned_local_slice()].tolist() for t, tl in g.local.items()}
json.dump(dict(labels=out, energy=res["energy"], energies=energies, rounds=res["rounds"]), open(%(out)r + ".%%d" %% rank, "w"))
dist.destroy_process_group()
'''


def test_tiles_of_one_block_on_two_ranks_over_gloo(tmp_path):
    """tile 1 of 3 on rank 1, tiles 0 and 2 on rank 0: every cut crosses the ranks.  The labels, the per-round energies and
    the number of rounds equal those of the same three tiles in one process."""
    import json
    H, W, diag, K, parts = 48, 48, True, 5, 3
    blk, lp, w, eid, init = _problem(H, W, diag, K)
    g = _make_group(blk, lp, init, H, W, diag, K, parts, [0] * parts, 0, None)
    res1, energies1 = _solve(g)
    want = {t: tl.b.get_labels()[tl.owned_local_slice()] for t, tl in g.local.items()}
    out = str(tmp_path / "tiles")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": out})
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29781", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT))
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, o.decode()[-3000:]
    got = {}
    for r in range(2):
        d = json.load(open(out + ".%d" % r))
        for t, lab in d["labels"].items():
            got[int(t)] = np.array(lab)
        assert d["rounds"] == res1["rounds"]
        np.testing.assert_allclose(d["energies"], energies1, rtol=1e-12)
        np.testing.assert_allclose(d["energy"], res1["energy"], rtol=1e-12)
    assert sorted(got) == [0, 1, 2]
    for t in range(3):
        assert np.array_equal(got[t], want[t])


WORKER4 = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from phylo_hmrf_amd import tiles
from tests.test_tiles_cpu import _problem, _make_group, _solve
H, W, diag, K, parts = 48, 48, True, 5, 2
OWNERS = {0: [2, 3], 1: [1, 0]}                      # block 0 on ranks 2 and 3, block 1 on ranks 1 and 0 (top tile on rank 1)
out, groups = {}, []
for bi in (0, 1):                                    # every rank creates every group, in block order (new_group is collective)
    comm = tiles.GroupComm(OWNERS[bi], None)
    if rank not in OWNERS[bi]:
        continue
    blk, lp, w, eid, init = _problem(H, W, diag, K, seed=3 + bi)
    groups.append((bi, _make_group(blk, lp, init, H, W, diag, K, parts, OWNERS[bi], rank, comm)))
for bi, g in groups:
    res, energies = _solve(g)
    out[str(bi)] = dict(labels={str(t): tl.b.get_labels()[tl.owned_local_slice()].tolist() for t, tl in g.local.items()},
                        energy=res["energy"], rounds=res["rounds"])
# what follows the tiles' rounds in an EM iteration: a collective of the whole world (the statistics)
t = torch.tensor([float(sum(v["energy"] for v in out.values()))], dtype=torch.float64)
dist.all_reduce(t)
out["world_sum"] = float(t.item())
json.dump(out, open(%(out)r + ".%%d" %% rank, "w"))
dist.destroy_process_group()
'''


def test_two_split_blocks_on_disjoint_rank_pairs_of_a_world_of_four(tmp_path):
    """The 8-GPU deal in small: two split blocks, each on its own pair of ranks, every rank a bystander of the other block's
    group (it creates the group -- torch.distributed.new_group is collective -- and takes no part in its rounds), then a
    collective of the whole world.  Labels, energies and round counts equal those of the same tiles in one process."""
    import json
    H, W, diag, K, parts = 48, 48, True, 5, 2
    want = {}
    for bi in (0, 1):
        blk, lp, w, eid, init = _problem(H, W, diag, K, seed=3 + bi)
        g = _make_group(blk, lp, init, H, W, diag, K, parts, [0] * parts, 0, None)
        res1, _ = _solve(g)
        want[bi] = (res1, {t: tl.b.get_labels()[tl.owned_local_slice()] for t, tl in g.local.items()})
    out = str(tmp_path / "tiles4")
    script = tmp_path / "worker4.py"
    script.write_text(WORKER4 % {"root": ROOT, "out": out})
    procs = []
    for r in range(4):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29783", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT))
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, o.decode()[-3000:]
    got = {0: {}, 1: {}}
    sums = []
    for r in range(4):
        d = json.load(open(out + ".%d" % r))
        sums.append(d.pop("world_sum"))
        assert sorted(d) == [str(bi) for bi in (0, 1) if r in {0: [2, 3], 1: [1, 0]}[bi]]        # a rank ran its own block only
        for bi, v in d.items():
            assert v["rounds"] == want[int(bi)][0]["rounds"]
            np.testing.assert_allclose(v["energy"], want[int(bi)][0]["energy"], rtol=1e-12)
            for t, lab in v["labels"].items():
                got[int(bi)][int(t)] = np.array(lab)
    for bi in (0, 1):
        assert sorted(got[bi]) == [0, 1]
        for t in range(2):
            assert np.array_equal(got[bi][t], want[bi][1][t])
    # every member reported its block's energy: each block twice
    np.testing.assert_allclose(sums, [2 * (want[0][0]["energy"] + want[1][0]["energy"])] * 4, rtol=1e-12)


def test_rccl_selftest_runs_over_gloo_with_forced_tile_groups():
    """tools/rccl_selftest.py -- the first thing to run on a multi-GPU node -- on two gloo ranks here: the world's three
    message shapes, and the tile sub-groups of cfg3 with blocks above 0.2 of a rank's share cut (so that groups exist at
    world size 2); every rank finishes and rank 0's JSON line names the groups."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PHMRF_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "tools", "rccl_selftest.py"), "--rounds", "3", "--workloads", "cfg3",
           "--split-above", "0.2"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    rep = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rep["ranks_that_finished"] == 2 and rep["backend"] == "gloo"
    assert any("E-step statistics" in k for k in rep["messages"]) and any("tile round" in k for k in rep["messages"])
    g = rep["tile_groups"]["cfg3"]
    assert g["split_blocks"] >= 1 and all(len(x["ranks"]) >= 1 for x in g["groups"])
