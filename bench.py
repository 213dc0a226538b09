#!/usr/bin/env python3
"""Benchmark of the Phylo-HMRF EM hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A STEP is one full EM iteration over every syntenic block of the workload, exactly as fit_accumulate_test drives it
(base.py:347-442): per block  emission log-likelihoods, labels <- labels_local or the previous labelling, whichever has
the lower energy (--warm-start local: labels_local as the reference), MRF labelling, posteriors + costs + sufficient
statistics;  then the reduction of the statistics (one RCCL all-reduce when N > 1), the cost bookkeeping, and the host
M-step (SciPy's SLSQP over the OU parameters of every state; the states dealt to the ranks, one all-reduce of their rows).
Inputs (X, graph) are resident in HBM before the timed region.  Metric (BASELINE.json):
EM-iterations/sec x nodes = N_tot * steps / wall.

Multi-GPU: the syntenic blocks are independent MRFs and are DEALT to the ranks, longest first (the reference forks one
process per block, base.py:357-362); a block that holds more than --split-above x a rank's share is cut into row tiles
first (see the end of this text).  --scaling strong (default): the workload's units are shared out, total work fixed --
north_star's "whole-genome run ... at 8 GPUs".  --scaling weak: N copies of the workload (an N-genome cohort) are shared
out the same way, work per GPU fixed.

Extra objects on the JSON line:
  "roofline"          the kernel with the largest device time: `frac` = algorithmic bytes (SURVEY.md 8d accounting x the
                      units the launches actually processed, COUNTED ON THE DEVICE) / the sum of the launches' OWN durations /
                      the 8 TB/s HBM peak -- own durations from HIP events with one block in flight at a time (the timed region
                      itself under --block-threads 1, otherwise an isolated pass right after it), split into `full_sweep` (first
                      round of a solve) and `mop_up` launches.  `class_throughput` is what round 4 called frac: the class's bytes
                      over the time during which at least one of its launches was running, fourteen blocks in flight.  In the
                      timed region only this class carries events (the timers of all ten cost 4 - 5 ms per EM iteration);
                      `kernels` is filled by an instrumented pass after it.
  "roofline_limiter"  what actually bounds that kernel, from this run's device counters only (LDS bytes, pairs, cells, DP
                      steps); the explanation -- instruction issue of in-order waves at 4 waves per SIMD -- and the
                      measurements behind it are in HISTORY.md 3.3 and profiles/README.md.
  "cpu_baseline"      (rank 0, N=1) the reference's CPU path on a bounded sample -- see cpu_baseline().
  "cold_first_iteration_ms"  the first EM iteration of the warm-up (from argmax labels; child blocks allocated).
  "fit"               the whole fit under the reference's own stopping rules, run after the timed region (not part of
                      `value`): iterations executed, wall, node-iterations/s over the whole fit, per-iteration E-step ms.
  "fit_reference_start"  the same whole fit with every labelling started from labels_local, as the reference does
                      (phylo_hmrf.py:479): the drop-in's number under the reference's own rule next to `fit` (--warm-start best).

Multi-GPU, round 4: a block that holds more than --split-above x a rank's share of the nodes is cut into ROW TILES held by
different ranks (phylo_hmrf_amd/tiles.py: lockstep rounds, one small all-reduce per block and round), and the M-step's K
states are dealt to the ranks.  --emulate-world N --emulate-rank R rehearses one rank's load on one GPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "cfg3-chr1", "cfg4", "cfg5", "cfg5-chr1", "small"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the workload's blocks dealt to the ranks (total work fixed); weak = N copies of "
                         "the workload dealt to the ranks (work per GPU fixed)")
    ap.add_argument("--split-above", type=float, default=1.0,
                    help="N > 1: a block that holds more than this many times a rank's share of the nodes is cut into row "
                         "tiles that different ranks hold (phylo_hmrf_amd/tiles.py)")
    ap.add_argument("--tile-parts", type=int, default=0,
                    help="cut EVERY block into this many row tiles whatever its size (measurement of the lockstep rounds' "
                         "overhead on one GPU: the tiles of a block are then all local)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="one process, one GPU: build and time only what rank --emulate-rank of a run on this many GPUs holds")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--no-through-fit", action="store_true",
                    help="skip `fit_surface`: the same EM iterations once more THROUGH THE PRODUCT'S OWN SURFACE -- a phyloHMRF "
                         "object's fit_accumulate_test (phylo_hmrf_amd/base.py <-> reference base.py:301-455), what the CLI's "
                         "run() calls -- timed by the loop's own clocks (one GPU only; not part of `value`)")
    ap.add_argument("--no-fit", action="store_true",
                    help="skip the whole-fit measurement after the timed region (the `fit` object of the JSON line)")
    ap.add_argument("--warm-start", default="best", choices=["best", "local"],
                    help="start of an E-step's labelling: local = labels_local as the reference (phylo_hmrf.py:479); best = "
                         "labels_local or the previous E-step's labels, whichever has the lower energy under the new parameters")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--mean-run", type=int, default=25,
                    help="synthetic data: mean side (bins) of the ground-truth label image's rectangles (SURVEY.md 8d: 25)")
    ap.add_argument("--noise", type=float, default=1.0,
                    help="synthetic data: factor on the states' standard deviations (1: as sampled; another data regime for the "
                         "label solver's schedule)")
    ap.add_argument("--beta", type=float, default=1.0)
    ap.add_argument("--beta1", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ab-env", default="",
                    help="development (libphmrf_dev.so via PHMRF_LIB; or a solver option such as energy_tol_ppb): NAME=A,B -- after everything else, --ab-steps more EM "
                         "iterations whose E-step runs TWICE from the same labellings and parameters, once with the environment "
                         "variable NAME set to A and once to B (order alternating); both times and cost1 go to `ab` on the line")
    ap.add_argument("--ab-steps", type=int, default=20)
    ap.add_argument("--no-prepare", action="store_true",
                    help="A/B: do not queue the next E-step's connected components behind the M-step (phmrf_block_prepare_components)")
    ap.add_argument("--block-threads", type=int, default=14,   # 0: ONE host thread, all blocks in lockstep rounds (phmrf_mrf_solve_group)
                    help="host threads driving blocks concurrently, each block on its own HIP stream (1 = sequential)")
    ap.add_argument("--mstep-workers", type=int, default=0,
                    help="processes fitting the states in the M-step (0 = min(K, cores); 1 = in this process, no fork: "
                         "the setting for runs under rocprofv3 --pmc, whose preloaded library initialises the GPU first)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not record HIP events around the kernel launches (no per-kernel breakdown / roofline)")
    ap.add_argument("--cpu-sample", type=int, default=652,
                    help="side N of the diagonal block used for the CPU baseline (652: the size of config 1's chr21 block)")
    ap.add_argument("--no-expansion", action="store_true")
    ap.add_argument("--energy-tol-ppb", type=int, default=10000,
                    help="stop the label solver when a round lowers the energy by less than this many ppb (0 = exact "
                         "fixed point).  10000 = 1e-5 |E|: the level at which two solves of the same inputs differ (1.4e-5 per "
                         "E-step, profiles/r6_late_round_economies.txt); rounds 3-5 benched 1000, which the line still "
                         "carries as `at_tol_1000ppb`; energy parity with gco at 0, 1000 and 10000: tests/test_gpu_estep.py")
    ap.add_argument("--no-old-tolerance", action="store_true",
                    help="skip the second window at the stopping tolerance of rounds 3-5 (`at_tol_1000ppb`)")
    return ap.parse_args()


def kernel_bytes(cls, n, K, S, D=8):
    """ALGORITHMIC HBM bytes of ONE launch of a kernel class on a block of n nodes (DESIGN.md section 4;
    f32 data, int32 indices, u8 labels, explicit adjacency -- the accounting of SURVEY.md 8d).  The strip classes
    ("strip": the multi-label expansions, "fusion": strip_kernel) and the proposals are priced from the work the
    launches actually did, counted on the device (strip_bytes / fusion_bytes / propose_bytes below)."""
    nbr = D * (4 + 4 + 1)                     # neighbour id + weight + neighbour label
    if cls == "emission":
        return n * (4 * S + 4 * K)
    if cls == "icm":                          # one colour class = n/4 nodes
        return n / 4.0 * (4 * K + nbr + 1 + 1)
    if cls == "chain":                        # one colour of one family: 4n nodes over 10 launches per round
        return 0.4 * n * (4 * K + nbr + 1 + 1)
    if cls == "posterior_stats":
        return n * (4 * S + 4 * K + nbr + 1)
    if cls == "energy":
        return n * (4 + nbr + 1)
    if cls == "component":                    # one pass over the unary rows and the adjacency (move table), four over
        return n * (4 * K + nbr + 4 * (4 * D + 1) + 12)   # neighbour ids + labels (union-find, block test), roots / flags
    return None


def strip_bytes(work, D=8):
    """SURVEY.md 8d's "one MRF sweep = 4K + 9 deg + 1 bytes per node", with the K the launches actually ran: the
    expansion kernel sweeps a strip's cells ONCE for all listed labels -- explicit adjacency + label + write, 73 B per
    cell swept -- and reads one unary entry per cell and listed label (4 B).  Cells and label-cells counted on the device."""
    return (9 * D + 1) * work["swept_cells"] + 4.0 * work["label_cells"]


def fusion_bytes(work, D=8):
    """a fusion / single-proposal strip unit re-decides its cells between two labels: two unary entries, label,
    proposal, explicit adjacency = 83 B per cell of the unit (cells counted on the device)"""
    return (8 + 9 * D + 2 + 1) * work["cells"]


def propose_bytes(work, K, grid=True, D=8):
    """proposals are recomputed only in node tiles that carry a new change stamp (nodes counted on the device): on a
    grid block K unary entries from the planes, the forward-edge records of the node and of its four backward
    neighbours' shares, nine labels, one write = 98 B per node at K = 20; explicit adjacency otherwise"""
    per_node = (4 * K + 16 + 1 + 1) if grid else (4 * K + 9 * D + 1 + 1)
    return per_node * work["proposal_nodes"]


def source_hash():
    """identifies the build a committed profile was taken with: SHA-1 over the device sources"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "phylo_hmrf_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (a.gpus, world))
    # --emulate-world N --emulate-rank R (one process, one GPU): build and time ONLY what rank R of an N-rank run would hold
    # -- its whole blocks, its row tiles (their neighbours absent: the halo rows stay as they are) and its share of the
    # M-step's states.  A rehearsal of one rank's load for the scaling projection in DESIGN.md, not the metric.
    emulating = a.emulate_world > 1
    if emulating and world > 1:
        raise SystemExit("--emulate-world is a one-process rehearsal: run it without torch.distributed.run")
    eworld = a.emulate_world if emulating else world
    erank = a.emulate_rank if emulating else rank
    if not 0 <= erank < eworld:
        raise SystemExit("--emulate-rank must be in [0, --emulate-world)")

    from phylo_hmrf_amd import mstep, tiles, workloads
    blocks_def, S, K, nn, desc = workloads.workload(a.workload)
    my_states = [c for c in range(K) if c % eworld == erank]      # the M-step's states are dealt to the ranks
    workers = a.mstep_workers if a.mstep_workers > 0 else max(1, min(len(my_states), os.cpu_count() or 1))
    if not mstep.native_available():
        mstep._pool(workers)                  # (Python fall-back of the M-step only: fork BEFORE anything initialises the GPU)

    import torch
    import torch.distributed as dist
    from phylo_hmrf_amd import Block, _lib
    from phylo_hmrf_amd.base import SLOT_LOCAL
    from phylo_hmrf_amd.block import unpack_stats
    from phylo_hmrf_amd.tree import PhyloTree
    _lib.require_gpu()
    # PHMRF_ONE_GPU=1 (tests): every rank on device 0 -- the sharded path on one card, with PHMRF_DIST_BACKEND=gloo (RCCL
    # refuses two ranks on one device); the production launch is one rank per GPU over RCCL
    if os.environ.get("PHMRF_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    _lib.check(_lib.load().phmrf_set_device(local_rank))
    dev = torch.device("cuda", local_rank)
    # PHMRF_FORCE_DIST=1 exercises the RCCL path (all-reduce, barrier) with a single rank
    use_dist = world > 1 or os.environ.get("PHMRF_FORCE_DIST") == "1"
    backend = os.environ.get("PHMRF_DIST_BACKEND", "nccl")
    coll_dev = dev if backend == "nccl" else torch.device("cpu")     # where the collectives' tensors live
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- synthetic multi-species Hi-C, generated on the device ----------------------------
    from phylo_hmrf_amd import synthetic
    tree = PhyloTree(synthetic.tree_for(S))
    rng = np.random.default_rng(a.seed)
    params_true = synthetic.sample_ou_params(rng, tree, K)
    means_true, cov_true = tree.mean_cov(params_true)
    cov_true = cov_true + 1e-3 * np.eye(S)                     # EM-time covariances carry 2e-3 (phylo_hmrf.py:1522-1524)
    t_setup = time.time()
    # units of work -> ranks: whole blocks, and ROW TILES of the blocks that hold more than --split-above x a rank's share
    # (phylo_hmrf_amd/tiles.py), dealt longest first; this rank builds and keeps only its own
    all_blocks = list(blocks_def) * (world if a.scaling == "weak" else 1)
    units = tiles.plan(all_blocks, eworld, a.split_above,
                       force_parts={bi: a.tile_parts for bi in range(len(all_blocks))} if a.tile_parts > 1 else None)
    owner = tiles.assign(units, eworld)
    sizes = [workloads.block_nodes(*bd) for bd in all_blocks]
    n_global = int(sum(sizes))
    nodes_per_rank = [int(sum(u["nodes"] for u, o in zip(units, owner) if o == r)) for r in range(eworld)]
    need_gb = max(nodes_per_rank) * workloads.BYTES_PER_NODE * (K / 20.0) / 1e9
    if need_gb > 260:
        raise SystemExit("workload %s needs about %.0f GB on the fullest of %d GPU(s) (288 GB each): run it on more GPUs "
                         "(cfg5 needs >= 4) or use --workload cfg5-chr1" % (a.workload, need_gb, eworld))
    blocks, groups = [], []                                    # whole blocks / tile groups of split blocks on this rank
    tile_comm_dev = coll_dev if backend == "nccl" else None
    for bi, (H, W, diag) in enumerate(all_blocks):
        mine = [(i, u) for i, u in enumerate(units) if u["block"] == bi]
        owners = [owner[i] for i, _ in mine]
        split = len(mine) > 1
        comm = tiles.GroupComm(owners, tile_comm_dev) if (split and world > 1) else None   # (collective: every rank, block order)
        if erank not in owners:
            continue
        Xd = synthetic.device_observations(torch, dev, a.seed * 1000 + bi, H, W, diag, K, means_true, cov_true,
                                           mean_run=a.mean_run, noise=a.noise)
        torch.cuda.synchronize()
        if not split:
            b = Block(workloads.block_nodes(H, W, diag), S, K)
            b.set_observations_dev(Xd.data_ptr())
            b.sync()
            b.build_grid_graph(H, W, diag, nn, a.beta1)        # stencil graph + w = exp(-beta1 d) built on the device
            blocks.append(b)
        else:
            def load(tl, Xd=Xd):
                tl.b.set_observations_dev(Xd.data_ptr() + tl.node0 * S * 4)     # the tile's stored rows of the block
                tl.b.sync()
            groups.append(tiles.make_group(bi, (H, W, diag), [(u["r0"], u["r1"]) for _, u in mine], owners, erank, S, K,
                                           Block, load, comm, nn, a.beta1))
        del Xd
    torch.cuda.empty_cache()
    conductor = tiles.Conductor(groups)
    local_tiles = [g.local[t] for g in conductor.groups for t in sorted(g.local)]
    unit_blocks = blocks + [tl.b for tl in local_tiles]         # every Block this rank drives (stats row = position here)
    n_local = int(sum(b.n for b in blocks) + sum(tl.own_hi - tl.own_lo for tl in local_tiles))
    n_norm = n_local if emulating else n_global                # the nodes the statistics of an E-step refer to
    n_stats = K * (1 + S + S * S)
    # EM starts from perturbed parameters; first labels = argmax_k logprob + one ICM sweep -> labels_local
    params_cur = np.clip(params_true * (1.0 + 0.15 * rng.standard_normal(params_true.shape)), 1e-3, 50.0)
    init_ou = params_cur.copy()
    rng_state_at_start = rng.bit_generator.state               # (fit_surface replays the same M-step draws)
    means, covars = tree.mean_cov(params_cur)
    covars = covars + 1e-3 * np.eye(S)
    SLOT_INIT = 3
    for b in unit_blocks:
        b.emission(means, covars)
        b.solve_fast(a.beta, max_rounds=1, use_chains=False, use_components=False, use_strips=False, use_expansion=False,
                     init_mode=1)
    for g in conductor.groups:
        g.sync_halos()                                         # (the ICM sweep saw a tile's halo rows without their other side)
    for b in unit_blocks:
        b.save_labels(SLOT_LOCAL)
        b.save_labels(SLOT_INIT)
        b.sync()
    setup_s = time.time() - t_setup
    stats_dev = torch.zeros((max(len(unit_blocks), 1), n_stats + 4), dtype=torch.float64, device=dev)
    state = dict(min_cost=1e30, params=params_cur.copy(), means=means, covars=covars, warm_start=a.warm_start)
    solver = dict(max_rounds=64, use_chains=True, use_components=True, use_strips=True, use_expansion=not a.no_expansion,
                  energy_tol_ppb=a.energy_tol_ppb)
    t_e, t_m = [], []
    untimed = [0.0]                                            # emulation only: the other ranks' M-step states, fitted here

    # The blocks are independent (the reference forks one process per block, base.py:357-362): a few host threads
    # drive them concurrently, each block on its own HIP stream, largest first, so the latency-bound launches of the
    # small blocks fill the GPU next to the large ones.  ctypes drops the GIL for the duration of a library call.
    # The row tiles of split blocks run their lockstep rounds on THIS thread meanwhile (tiles.Conductor).
    from phylo_hmrf_amd.concurrent import BlockRunner
    runner = BlockRunner(max(a.block_threads, 1), local_rank)      # (--block-threads 0: the lockstep group solve, estep_lockstep)
    order = sorted(range(len(blocks)), key=lambda i: -blocks[i].n)

    block_trace = [] if os.environ.get("PHMRF_BLOCK_TRACE") else None      # development: (step start, block, n, t0, t1)

    def estep_block(i):
        b = blocks[i]
        tb0 = time.time()
        b.emission(state["means"], state["covars"])
        if state["warm_start"] == "best":
            b.warm_start(a.beta, SLOT_LOCAL, report=False)     # labels_local or the previous E-step's labels: the lower energy
        else:
            b.restore_labels(SLOT_LOCAL)                       # init_labels = labels_local (phylo_hmrf.py:479)
        b.solve_fast(a.beta, **solver)
        b.posterior_stats_dev(a.beta, 3, stats_dev[i].data_ptr())
        b.sync()
        if block_trace is not None:
            block_trace.append((i, b.n, tb0, time.time()))

    def tile_prepare(tl):
        if state["warm_start"] != "best":
            tl.b.restore_labels(SLOT_LOCAL)
        tl.b.emission(state["means"], state["covars"])

    def tile_finish(tl):
        tl.b.posterior_stats_dev(a.beta, 3, stats_dev[len(blocks) + local_tiles.index(tl)].data_ptr())

    def estep_lockstep():
        """--block-threads 0: ONE host thread drives every whole block -- emission and warm start queued on the blocks' streams,
        then all solves round by round as the rounds end (phmrf_mrf_solve_group), then the statistics: the same library work per
        block as estep_block, the same overlap on the GPU, no thread pool"""
        for i in order:
            b = blocks[i]
            b.emission(state["means"], state["covars"])
            if state["warm_start"] == "best":
                b.warm_start(a.beta, SLOT_LOCAL, report=False)
            else:
                b.restore_labels(SLOT_LOCAL)
        Block.solve_group([blocks[i] for i in order], a.beta, **solver)
        for i in order:
            blocks[i].posterior_stats_dev(a.beta, 3, stats_dev[i].data_ptr())
        for i in order:
            blocks[i].sync()

    def estep_all(sequential=False):
        if block_trace is not None:
            del block_trace[:]
        t0 = time.time()
        if sequential:
            for i in order:
                estep_block(i)
            pending = None
        elif a.block_threads == 0:
            # one thread for all whole blocks: this one, unless the rank holds row tiles too (their rounds run here meanwhile)
            if conductor.groups and blocks:
                pending = runner.start_one(estep_lockstep)
            else:
                estep_lockstep()
                pending = None
        else:
            pending = runner.start(estep_block, order)
        if conductor.groups:
            conductor.solve(a.beta, solver, prepare=tile_prepare, finish=tile_finish,
                            warm_slot=SLOT_LOCAL if state["warm_start"] == "best" else None)
            for tl in local_tiles:
                tl.b.sync()
        if pending is not None:
            pending.results()
        if block_trace is not None and rank == 0:
            sys.stderr.write("[block trace] E-step %.2f ms: " % ((time.time() - t0) * 1e3) + " ".join(
                "%d:%.1fM[%.1f-%.1f]" % (i, n / 1e6, (s0 - t0) * 1e3, (s1 - t0) * 1e3) for i, n, s0, s1 in sorted(block_trace, key=lambda r: r[2])) + "\n")
        tot = stats_dev.sum(dim=0).to(coll_dev)
        if use_dist:
            dist.all_reduce(tot)                               # RCCL: K(1+S+S^2)+4 doubles
        return tot.cpu().numpy()

    def mstep_all(stats, gen):
        """the K states dealt to the ranks (state c -> rank c mod N), one all-reduce of the rows (phyloHMRF._do_mstep)"""
        base = int(gen.integers(0, 2 ** 62))
        p, mu, cv, _ = mstep.do_mstep(tree, stats, state["params"], init_ou, n_norm, 1.0, 0, 0.3, 0.1, 1.0, gen,
                                      workers=workers, states=my_states if eworld > 1 else None, base=base)
        packed = np.concatenate([p.ravel(), mu.ravel(), cv.ravel()])
        if use_dist and world > 1:
            t = torch.from_numpy(packed).to(coll_dev)
            dist.all_reduce(t)
            packed = t.cpu().numpy()
        elif emulating:      # the states of the absent ranks, so that the EM trajectory is a real one (not part of the timings)
            tu = time.time()
            others = [c for c in range(K) if c not in my_states]
            p2, mu2, cv2, _ = mstep.do_mstep(tree, stats, state["params"], init_ou, n_norm, 1.0, 0, 0.3, 0.1, 1.0, gen,
                                             workers=min(len(others), os.cpu_count() or 1), states=others, base=base)
            packed = packed + np.concatenate([p2.ravel(), mu2.ravel(), cv2.ravel()])
            untimed[0] += time.time() - tu
        P = tree.n_params
        state["params"] = packed[:K * P].reshape(K, P)
        state["means"] = packed[K * P:K * P + K * S].reshape(K, S)
        state["covars"] = packed[K * P + K * S:].reshape(K, S, S)

    cost1_log = []                                             # cost1 of every EM iteration (warm-up included)

    def em_step():
        t0 = time.time()
        tot = estep_all()
        stats = unpack_stats(tot[:n_stats], K, S)
        cost1 = tot[n_stats + 3] / n_norm
        cost1_log.append(float(cost1))
        if cost1 < state["min_cost"]:                          # base.py:416-420
            state["min_cost"] = cost1
            for b in unit_blocks:
                b.save_labels(SLOT_LOCAL)
        t1 = time.time()
        u0 = untimed[0]
        # the labels-only part of the next E-step's component passes, queued behind the host's M-step (the GPU is idle then)
        ahead = (runner.start(lambda i: blocks[i].prepare_components(), order)
                 if (blocks and not a.no_prepare and solver["use_components"]) else None)
        mstep_all(stats, rng)
        if ahead is not None:
            ahead.results()
        t2 = time.time()
        t_e.append(t1 - t0)
        t_m.append(t2 - t1 - (untimed[0] - u0))

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    cold_first_ms = None
    dominant = None                                            # the kernel class whose launches the timed region times
    for w_it in range(a.warmup):
        # (which class leads is asked of a WARM iteration: with fewer than two warm-up steps the only candidate is the cold
        #  first iteration, whose coarse rounds say nothing about the timed region -- every class is timed then, as before)
        last = w_it == a.warmup - 1 and a.warmup >= 2
        if last and not a.no_kernel_timing:                    # the last warm-up iteration with every class timed: which one leads?
            for b in unit_blocks:
                b.enable_timing(True)
                b.reset_timing()
        tc = time.time()
        em_step()
        if w_it == 0:
            cold_first_ms = (time.time() - tc) * 1e3           # the cold first EM iteration (= --steps 1 --warmup 0)
        if last and not a.no_kernel_timing and unit_blocks:
            tot = {}
            for b in unit_blocks:
                for name, (ms, _) in b.timing().items():
                    if name in KERNELS_OF_CLASS:
                        tot[name] = tot.get(name, 0.0) + ms
            dominant = max(tot.items(), key=lambda kv: kv[1])[0]
    # HIP events around EVERY launch group cost 4 - 5 ms of a 75 ms EM iteration (tools/job_ab_timing.sh): in the timed region
    # only the dominant class is timed (its launch durations are what `roofline` needs); the per-class breakdown (`kernels`)
    # comes from an instrumented pass after the timed region, which is not part of `value`
    for b in unit_blocks:
        b.enable_timing(not a.no_kernel_timing, classes=[dominant] if dominant else None)
        b.reset_timing()
    del t_e[:], t_m[:]
    untimed[0] = 0.0
    barrier()
    from phylo_hmrf_amd.block import time_base_reset
    time_base_reset()                                          # t = 0 of the kernel-interval time line
    for g in conductor.groups:
        g.reset_clock()
    t0 = time.time()
    for _ in range(a.steps):
        em_step()
    barrier()
    elapsed = time.time() - t0 - untimed[0]
    # the lockstep rounds of this rank's split blocks in the timed region: host time per round and group of the exchange
    # (boundary rows out of the pinned buffers + pack, the all-reduce over the block's holders, unpack + halo rows in)
    tile_rounds = None
    if conductor.groups:
        rs = sum(g.clock["rounds"] for g in conductor.groups)
        tile_rounds = {"groups": len(conductor.groups), "rounds": int(rs), "rounds_per_solve": round(rs / float(a.steps * len(conductor.groups)), 2),
                       "tile_round_exchange_us": round(1e6 * sum(g.clock["exchange"] for g in conductor.groups) / max(rs, 1), 1),
                       "of_which_allreduce_us": round(1e6 * sum(g.clock["allreduce"] for g in conductor.groups) / max(rs, 1), 1),
                       "collect_wait_us": round(1e6 * sum(g.clock["collect"] for g in conductor.groups) / max(rs, 1), 1),
                       "decide_us": round(1e6 * sum(g.clock["decide"] for g in conductor.groups) / max(rs, 1), 1),
                       "transport": ("local (every tile of the block on this rank)" if all(g.comm is None or g.comm.group is None
                                                                                          for g in conductor.groups)
                                     else "torch.distributed %s over %d ranks" % (backend, world)),
                       "note": "host wall time per lockstep round and split block on rank 0: exchange = boundary rows + pack, the "
                               "all-reduce over the ranks that hold the block's tiles, unpack + halo rows queued; collect_wait = "
                               "waiting for the round's kernels (the tiles' E-step work itself)"}
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    t_e_timed, t_m_timed = list(t_e), list(t_m)
    # every rank's own clocks on the line (round 6: a first run on a multi-GPU node should say WHICH rank is slow and in what):
    # mean E-step / M-step per step, the slowest E-step, nodes and units held, and the tile rounds' exchange
    per_rank = None
    mine = np.zeros((world, 7))
    mine[rank] = [np.mean(t_e_timed) * 1e3 if t_e_timed else 0.0, np.mean(t_m_timed) * 1e3 if t_m_timed else 0.0,
                  np.max(t_e_timed) * 1e3 if t_e_timed else 0.0, n_local, len(unit_blocks),
                  (tile_rounds or {}).get("tile_round_exchange_us", 0.0), (tile_rounds or {}).get("rounds_per_solve", 0.0)]
    if use_dist:
        tpr = torch.from_numpy(mine).to(coll_dev)
        dist.all_reduce(tpr)
        mine = tpr.cpu().numpy()
    per_rank = {"estep_ms": [round(float(x), 3) for x in mine[:, 0]], "mstep_ms": [round(float(x), 3) for x in mine[:, 1]],
                "estep_ms_max": [round(float(x), 3) for x in mine[:, 2]], "nodes": [int(x) for x in mine[:, 3]],
                "units": [int(x) for x in mine[:, 4]], "exchange_us": [round(float(x), 1) for x in mine[:, 5]],
                "tile_rounds_per_solve": [round(float(x), 2) for x in mine[:, 6]],
                "note": "one entry per rank, this rank's own host clocks over the timed steps: E-step (its blocks and tiles, to the "
                        "all-reduce of the statistics), M-step (its share of the K states + the all-reduce of the rows), "
                        "exchange_us = host time per lockstep round of its split blocks"}

    # ---- per-kernel-class device time (HIP events recorded on the blocks' streams during the timed region) ----
    def union_ms(iv):
        """total length of the union of [start, end] intervals (ms)"""
        if iv.shape[0] == 0:
            return 0.0
        iv = iv[np.argsort(iv[:, 0])]
        tot, cs, ce = 0.0, iv[0, 0], iv[0, 1]
        for s0, e0 in iv[1:]:
            if s0 > ce:
                tot += ce - cs
                cs, ce = s0, e0
            elif e0 > ce:
                ce = e0
        return float(tot + ce - cs)

    def collect(first=False):
        """per kernel class [stream ms, launches, algorithmic bytes] and the device-counted work, over this rank's blocks
        (first: only the launches of the FIRST round of every solve -- a warm start's full sweeps)"""
        agg_, work_ = {}, dict(units=0, cells=0, staged_cells=0, dp_steps=0, launches=0, swept_cells=0, label_cells=0,
                               proposal_nodes=0, mask_label_cells=0, mask_strip_cells=0)
        for b in unit_blocks:
            for name, (ms, ln) in (b.timing_first() if first else b.timing()).items():
                d = agg_.setdefault(name, [0.0, 0, 0.0])
                d[0] += ms
                d[1] += ln
                kb = kernel_bytes(name, b.n, K, S)
                if kb is not None:
                    d[2] += kb * ln
            for k, v in (b.work_first() if first else b.work()).items():
                work_[k] += v
        if "strip" in agg_:
            agg_["strip"][2] = strip_bytes(work_)
        if "fusion" in agg_:
            agg_["fusion"][2] = fusion_bytes(work_)
        if "propose" in agg_:
            agg_["propose"][2] = propose_bytes(work_, K)
        return agg_, work_

    agg_dom, work = collect()                                  # the timed region: the dominant class's events, all classes' work
    agg_dom_first, _ = collect(first=True)
    busy_dom = {name: union_ms(np.concatenate([b.intervals(name) for b in unit_blocks] or [np.zeros((0, 2))]))
                for name in agg_dom} if not a.no_kernel_timing and unit_blocks else {}
    # ---- the instrumented pass: a few more EM iterations with every class timed (after the timed region, not part of `value`)
    agg, busy, inst_steps, work_inst = agg_dom, busy_dom, 0, work
    if not a.no_kernel_timing and unit_blocks and dominant:
        inst_steps = min(a.steps, 5)
        for b in unit_blocks:
            b.enable_timing(True)
            b.reset_timing()
        barrier()
        time_base_reset()
        for _ in range(inst_steps):
            em_step()
        barrier()
        agg, work_inst = collect()
        busy = {name: union_ms(np.concatenate([b.intervals(name) for b in unit_blocks] or [np.zeros((0, 2))])) for name in agg}

    # ---- the same E-step once more, ONE BLOCK AT A TIME (after the timed region; not part of `value`): with a single
    #      stream in flight a launch's event time is the kernel's own duration, which the concurrent streams of the timed
    #      region cannot give (there a launch shares the GPU with up to thirteen others)
    isolated, agg_i, agg_i_first = {}, {}, {}
    if not a.no_kernel_timing and unit_blocks:
        for b in unit_blocks:
            b.reset_timing()
        estep_all(sequential=True)
        torch.cuda.synchronize()
        agg_i, _ = collect()
        agg_i_first, _ = collect(first=True)
        isolated = {k: {"launches": int(v[1]), "avg_launch_us": round(v[0] * 1e3 / max(v[1], 1), 2),
                        "GBps": (round(v[2] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 and v[2] > 0 else None)}
                    for k, v in agg_i.items()}

    # ---- the same warm-up + timed EM iterations REPLAYED from the bench's start (same parameters, first labelling and M-step
    #      draws) at the stopping tolerance of rounds 3 - 5 (1e-6 |E|; not part of `value`): the number that continues
    #      BENCH_r03 .. r05 beside the one at this round's default tolerance.  (Later iterations of the same EM are calmer: a
    #      window that merely FOLLOWED the timed region would flatter whichever setting it ran.)
    at_old_tol = None
    if not a.no_old_tolerance and a.energy_tol_ppb != 1000 and unit_blocks:
        for b in unit_blocks:                        # (the same event timers as in the timed region)
            b.enable_timing(not a.no_kernel_timing, classes=[dominant] if dominant else None)
            b.reset_timing()
            b.restore_labels(SLOT_INIT)
            b.save_labels(SLOT_LOCAL)
        keep = solver["energy_tol_ppb"]
        solver["energy_tol_ppb"] = 1000
        state.update(min_cost=1e30, params=params_cur.copy(), means=means, covars=covars, warm_start=a.warm_start)
        rng.bit_generator.state = rng_state_at_start
        for _ in range(a.warmup):
            em_step()
        n_e, n_m = len(t_e), len(t_m)
        u_old = untimed[0]
        barrier()
        t_old = time.time()
        for _ in range(a.steps):
            em_step()
        barrier()
        el_old = time.time() - t_old - (untimed[0] - u_old)
        solver["energy_tol_ppb"] = keep
        if use_dist:
            t = torch.tensor([el_old], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_old = float(t.item())
        at_old_tol = {"energy_tol_ppb": 1000, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el_old / a.steps * 1e3,
                      "value": n_norm * a.steps / el_old, "estep_ms": float(np.mean(t_e[n_e:]) * 1e3),
                      "mstep_ms": float(np.mean(t_m[n_m:]) * 1e3),
                      "estep_ms_by_step": [round(x * 1e3, 1) for x in t_e[n_e:]],
                      "note": "the warm-up + timed EM iterations replayed from the bench's start (same parameters, first labelling and "
                              "M-step draws; the EM is chaotic: another draw of the same iterations) with the label solver's stopping "
                              "tolerance at 1e-6 |E| as benched in rounds 3 - 5; `value` is measured at config.mrf_solver.energy_tol_ppb"}
        del t_e[n_e - a.warmup:], t_m[n_m - a.warmup:]

    # ---- the WHOLE FIT (not part of `value`): the same workload from the same start under the reference's own stopping
    #      rules -- threshold 0.001 and at most 60 iterations (phylo_hmrf.py:1555, :1561), the relative-change tests after
    #      iteration 5 and the 50-iterations-past-the-minimum test (base.py:428-435), no M-step after the last E-step --
    #      cold first iteration included.  Same code path as the timed region; kernel timers off.
    def run_fit(warm_start):
        """the whole fit from the bench's start under the reference's stopping rules, with the given start of an E-step's
        labelling ("local": labels_local as the reference, phylo_hmrf.py:479; "best": that or the previous labelling)"""
        for b in unit_blocks:
            b.enable_timing(False)
            b.restore_labels(SLOT_INIT)
            b.save_labels(SLOT_LOCAL)
        state.update(min_cost=1e30, params=params_cur.copy(), means=means, covars=covars, warm_start=warm_start)
        gen = np.random.default_rng(a.seed + 4242)
        threshold, m_iter, max_iter1 = 0.001, 60, 50
        pre = [0.001, 0.001, 0.001]
        min_cost, min_cost1 = [0, 1000.0], [0, 1000.0]
        e_ms, m_ms, c1 = [], [], []
        untimed[0] = 0.0
        barrier()
        tf0 = time.time()
        stop = "m_iter"
        for it in range(m_iter):
            ta = time.time()
            tot = estep_all()
            stats = unpack_stats(tot[:n_stats], K, S)
            _, pairwise_cost, unary_cost, cost1 = (tot[n_stats:n_stats + 4] / n_norm).tolist()
            d1, d2, d3 = (abs((pairwise_cost - pre[0]) / pre[0]), abs((unary_cost - pre[1]) / pre[1]),
                          abs((cost1 - pre[2]) / pre[2]))
            pre = [pairwise_cost, unary_cost, cost1]
            c1.append(round(float(cost1), 6))
            if cost1 < min_cost[1]:                                # base.py:416-420
                min_cost = [it, cost1]
                for b in unit_blocks:
                    b.save_labels(SLOT_LOCAL)
            if cost1 < min_cost1[1] and it >= 3:                   # base.py:422-426
                min_cost1 = [it, cost1]
            e_ms.append((time.time() - ta) * 1e3)
            if ((d1 < threshold and d2 < threshold) or d3 < threshold) and it > 5:      # base.py:428-429
                stop = "relative change below the threshold"
                break
            if it - min_cost1[0] > max_iter1:                      # base.py:434-435
                stop = "50 iterations past the minimum"
                break
            tb = time.time()
            u0 = untimed[0]
            mstep_all(stats, gen)
            m_ms.append((time.time() - tb - (untimed[0] - u0)) * 1e3)
        barrier()
        fit_wall = time.time() - tf0 - untimed[0]
        if use_dist:
            t = torch.tensor([fit_wall], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fit_wall = float(t.item())
        iters = len(e_ms)
        state["warm_start"] = a.warm_start
        return {"warm_start": warm_start, "iterations": iters, "stopped_by": stop, "wall_s": round(fit_wall, 4),
                "value": n_norm * iters / fit_wall, "unit": "node-iterations/s over the whole fit (cold first iteration included)",
                "estep_ms": [round(x, 2) for x in e_ms], "mstep_ms_mean": round(float(np.mean(m_ms)), 3) if m_ms else None,
                # the iterations past the fifth (what `value`'s timed window holds under --warmup 5), E-step + mean M-step:
                # the steady cost of an EM iteration under THIS start rule, beside `ms_per_step`
                "steady_ms_per_iteration": (round(float(np.mean(e_ms[5:])) + (float(np.mean(m_ms)) if m_ms else 0.0), 3)
                                            if iters > 5 else None),
                "cost1": c1, "rules": "threshold 0.001, m_iter 60 (phylo_hmrf.py:1555,1561); base.py:416-435"}

    # `fit`: with the start of an E-step's labelling this run was asked for (--warm-start; default "best");
    # `fit_reference_start`: the same whole fit under the REFERENCE's own rule -- every labelling starts from labels_local
    # (phylo_hmrf.py:479, base.py:416-420) -- so that the line carries the drop-in's number under the reference's policy next
    # to the build's (--warm-start local makes the two the same run: it is then not repeated)
    fit = fit_ref = None
    if not a.no_fit:
        fit = run_fit(a.warm_start)
        fit_ref = fit if a.warm_start == "local" else run_fit("local")

    # ---- `fit_surface` (not part of `value`): the same warm-up + timed EM iterations, from the same start and with the same
    #      M-step draws, THROUGH THE PRODUCT: a phyloHMRF object (host observations in, device graph, blocks dealt to its own
    #      runner) whose fit_accumulate_test runs the loop -- E-steps by _estep_region, the statistics through the Reducer, the
    #      bookkeeping of base.py:402-435, _do_mstep, the components prepared behind it.  What the reference's run()
    #      (phylo_hmrf.py:1738) would call; `value` above is measured around the same library calls without that object.
    def through_fit():
        from phylo_hmrf_amd.base import _BaseGraph
        from phylo_hmrf_amd.hmrf import phyloHMRF
        t_build = time.time()
        parts, len_vec, s0 = [], [], 0
        for bi, (H, W, diag) in enumerate(all_blocks):
            Xd = synthetic.device_observations(torch, dev, a.seed * 1000 + bi, H, W, diag, K, means_true, cov_true,
                                               mean_run=a.mean_run, noise=a.noise)
            parts.append(Xd.cpu().numpy().astype(np.float64))          # the reference hands float64 samples over (utility.py:332)
            del Xd
            nb = parts[-1].shape[0]
            len_vec.append([nb, s0, s0 + nb, H, W, 0, 0, bi, 1 if diag else 0, bi + 1])      # utility.py:455-456, :528
            s0 += nb
        Xh = np.concatenate(parts, axis=0)
        del parts
        torch.cuda.empty_cache()

        class FromTheBenchStart(phyloHMRF):
            """phyloHMRF with the bench's start instead of the k-means initialisation (phylo_hmrf.py:205-264): the perturbed
            true parameters, first labels = argmax_k logprob + one ICM sweep -> labels_local; the generator of the M-step's
            restarts in the state the bench's was in.  Nothing of the timed loop is touched."""

            def _init(self, X, lengths=None):
                _BaseGraph._init(self, X, lengths=lengths)
                self.params_vec1 = params_cur.copy()
                self.init_ou_params = init_ou.copy()
                self.means_, self._covars_ = means.copy(), covars.copy()
                self.rng = np.random.default_rng(0)
                self.rng.bit_generator.state = rng_state_at_start
                for b, _, _, _ in self._local_units():
                    b.emission(self.means_, self._covars_)
                    b.solve_fast(self.beta, max_rounds=1, use_chains=False, use_components=False, use_strips=False,
                                 use_expansion=False, init_mode=1)
                    b.save_labels(SLOT_LOCAL)
                    b.sync()

        B = tree.branch_dim
        m = FromTheBenchStart(n_components=K, run_id=0, n_samples=Xh.shape[0], n_features=S, observation=Xh,
                              edge_list=synthetic.tree_for(S), len_vec=len_vec, type_id=1, branch_list=[1.0] * B,
                              edge_list_1=[None] * len(all_blocks), cons_param=1.0, beta=a.beta, beta1=a.beta1, initial_mode=0,
                              initial_weight=0.3, initial_weight1=0.1, initial_magnitude=1.0, estimate_type=3, num_neighbor=nn,
                              random_state=a.seed, quiet=True, mstep_workers=workers, device_graph=True,
                              block_threads=a.block_threads, warm_start=a.warm_start, solver_opts=dict(solver))
        build_s = time.time() - t_build
        try:
            torch.cuda.synchronize()
            t_fit = time.time()
            res = m.fit_accumulate_test(Xh, len_vec, 0.0, "bench", a.warmup + a.steps)     # threshold 0: no early stop
            torch.cuda.synchronize()
            fit_s = time.time() - t_fit
            it_ms = np.asarray(m.timing_["iteration"]) * 1e3
            e_ms, m_ms = np.asarray(m.timing_["estep"]) * 1e3, np.asarray(m.timing_["mstep"]) * 1e3
            w = a.warmup if len(it_ms) > a.warmup else 0
            ms_step = float(np.mean(it_ms[w:]))
            return {"ms_per_step": round(ms_step, 3), "value": n_global / (ms_step * 1e-3), "unit": "node-iterations/s",
                    "estep_ms": round(float(np.mean(e_ms[w:])), 3), "mstep_ms": round(float(np.mean(m_ms[w:])), 3),
                    "iterations_timed": int(len(it_ms) - w), "warmup": int(w),
                    "ms_per_step_by_iteration": [round(float(x), 1) for x in it_ms],
                    "cost1": [round(float(c), 6) for c in np.asarray(res[5])[-min(len(res[5]), 8):, 3]],
                    "build_s": round(build_s, 2), "fit_wall_s": round(fit_s, 3),
                    "how": "phyloHMRF(observation = the workload's host float64 samples, device_graph=True, block_threads=%d, "
                           "warm_start=%r).fit_accumulate_test(X, len_vec, 0.0, 'bench', warmup + steps) from the bench's start "
                           "(same parameters, first labelling and M-step draws as the timed region); ms_per_step = mean of the "
                           "loop's own per-iteration clock (model.timing_['iteration']: E-step, reduction, bookkeeping, M-step) over "
                           "the iterations after the warm-up" % (a.block_threads, a.warm_start)}
        finally:
            m.close()

    # ---- development: the same E-steps under two settings of a knob of the development library (not part of `value`) --------
    ab = None
    if a.ab_env and unit_blocks and not conductor.groups:
        name, vals = a.ab_env.split("=", 1)
        va, vb = vals.split(",")
        SLOT_AB = 1
        for b in unit_blocks:
            b.enable_timing(False)
        ms = {va: [], vb: []}
        c1 = {va: [], vb: []}
        solver_before = dict(solver)
        threads_before = a.block_threads
        for it in range(a.ab_steps):
            for b in unit_blocks:
                b.save_labels(SLOT_AB)                 # the labelling the previous E-step left
            tot = None
            for v in ((va, vb) if it % 2 == 0 else (vb, va)):
                if name in solver:                     # a solver option (energy_tol_ppb, max_rounds, ...) instead of a knob
                    solver[name] = type(solver[name])(int(v))
                elif name == "block_threads":          # 0: all whole blocks from this thread (group solve); else the runner's threads
                    a.block_threads = int(v)
                else:
                    os.environ[name] = v
                for b in unit_blocks:
                    b.restore_labels(SLOT_AB)
                    if not a.no_prepare and solver["use_components"]:
                        b.prepare_components()
                barrier()
                ta = time.time()
                tot = estep_all()
                barrier()
                ms[v].append((time.time() - ta) * 1e3)
                c1[v].append(float(tot[n_stats + 3] / n_norm))
            # the EM goes on from the second run's result (the order alternates)
            stats = unpack_stats(tot[:n_stats], K, S)
            cost1 = tot[n_stats + 3] / n_norm
            if cost1 < state["min_cost"]:
                state["min_cost"] = cost1
                for b in unit_blocks:
                    b.save_labels(SLOT_LOCAL)
            mstep_all(stats, rng)
        solver.update(solver_before)
        a.block_threads = threads_before
        ab = {"env": name, "values": [va, vb], "steps": a.ab_steps,
              "estep_ms_mean": {v: round(float(np.mean(ms[v])), 3) for v in ms},
              "estep_ms_median": {v: round(float(np.median(ms[v])), 3) for v in ms},
              "estep_ms": {v: [round(x, 1) for x in ms[v]] for v in ms},
              "cost1_mean": {v: round(float(np.mean(c1[v])), 7) for v in c1},
              "cost1_diff_mean (B - A)": float(np.mean(np.array(c1[vb]) - np.array(c1[va]))),
              "cost1": {v: [round(x, 6) for x in c1[v]] for v in c1}}

    fit_surface = None
    if not a.no_through_fit and world == 1 and not emulating and a.scaling == "strong":
        fit_surface = through_fit()
        fit_surface["ratio_to_ms_per_step"] = round(fit_surface["ms_per_step"] / (elapsed / a.steps * 1e3), 4)

    roofline = roofline_limiter = None
    # the dominant class among those with a byte model (the coarse expansions' gathers have none)
    with_model = {k: v for k, v in agg.items() if v[2] > 0}
    dom_name = max(with_model.items(), key=lambda kv: kv[1][0])[0] if with_model else None
    dom_name = dominant if (dominant in agg_dom and agg_dom[dominant][2] > 0) else dom_name
    def own_split(all_, first_, name):
        """launches of a class whose event times are the kernel's OWN durations (one stream in flight), split into the full
        sweeps (first round of a solve) and the mop-up launches (later rounds: strips whose inputs changed)"""
        ms, ln, by = all_[name]
        fms, fln, fby = first_.get(name, [0.0, 0, 0.0])

        def part(ms_, ln_, by_):
            return {"launches": int(ln_), "bytes": int(by_), "own_ms": round(ms_, 4),
                    "avg_launch_us": round(ms_ * 1e3 / max(ln_, 1), 2),
                    "GBps": (round(by_ / (ms_ * 1e-3) / 1e9, 1) if ms_ > 0 and by_ > 0 else None)}
        return part(ms, ln, by), part(fms, fln, fby), part(ms - fms, ln - fln, max(by - fby, 0.0))

    if dom_name and busy_dom.get(dom_name, 0) > 0 and agg_dom[dom_name][2] > 0:
        dom_ms, dom_launches, dom_bytes = agg_dom[dom_name]           # (measured in the timed region)
        busy_d = busy_dom[dom_name]
        cls_ach = dom_bytes / (busy_d * 1e-3) / 1e9
        kernel_names = KERNELS_OF_CLASS.get(dom_name, [])
        traffic, traffic_note = pmc_traffic(a.workload, kernel_names)
        valu = pmc_valu(a.workload, kernel_names)
        # The kernel's fraction: algorithmic bytes / the launches' OWN durations / peak.  With one block in flight at a time
        # (--block-threads 1: the profiled serial runs) the timed region's own event times are that; otherwise the isolated
        # pass right after the timed region supplies them (same EM trajectory, one block at a time).
        serial = a.block_threads == 1 and not conductor.groups
        if serial:
            own_all, own_full, own_mop = own_split(agg_dom, agg_dom_first, dom_name)
            own_from = "the timed region (--block-threads 1: one stream in flight, an event pair times the kernel alone)"
        elif dom_name in agg_i:
            own_all, own_full, own_mop = own_split(agg_i, agg_i_first, dom_name)
            own_from = "the isolated pass: the same E-step one block at a time right after the timed region (not part of `value`)"
        else:
            own_all = own_full = own_mop = None
            own_from = None
        ach = own_all["GBps"] if own_all and own_all["GBps"] else None
        hbm_frac = (round(ach / HBM_PEAK_GBS, 5) if ach else None)
        # (round 6) the bound that is actually the higher of the two for this kernel: its share of the vector pipes' issue
        # capacity (a committed rocprofv3 --pmc pass of this build, see pmc_valu) against its share of the HBM roofline.
        # achieved / peak / unit / frac / traffic stay the HBM figures the contract asks for, whichever bound is named.
        bound = "valu" if (valu.get("busy") is not None and hbm_frac is not None and valu["busy"] > hbm_frac) else "hbm"
        roofline = {"bound": bound, "kernel": dom_name, "kernel_names": kernel_names,
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": hbm_frac,
                    "valu": valu,
                    "bound_note": "bound = whichever is higher for the dominant kernel: valu.busy (SQ_INSTS_VALU x 4 clocks / (active "
                                  "clocks x 1024 SIMDs): the share of the vector pipes' issue capacity) or frac (algorithmic HBM bytes / own "
                                  "duration / 8 TB/s).  achieved, peak, unit, frac and traffic are the HBM figures either way",
                    "traffic": traffic, "traffic_source": traffic_note,
                    "own_durations_from": own_from,
                    "launches": (own_all or {}).get("launches"),
                    "algorithmic_bytes_per_launch": (int(own_all["bytes"] / max(own_all["launches"], 1)) if own_all else None),
                    "avg_launch_us": (own_all or {}).get("avg_launch_us"),
                    "full_sweep": own_full, "mop_up": own_mop,
                    "class_throughput": {"achieved": round(cls_ach, 2), "frac": round(cls_ach / HBM_PEAK_GBS, 5),
                                         "launches": int(dom_launches),
                                         "algorithmic_bytes_per_launch": int(dom_bytes / max(dom_launches, 1)),
                                         "avg_launch_us": round(dom_ms * 1e3 / max(dom_launches, 1), 2),
                                         "busy_ms_per_step": round(busy_d / a.steps, 3),
                                         "note": "timed region, blocks in flight on their own streams: bytes of all launches of the "
                                                 "class / time during which >= 1 of them was running -- a throughput of the CLASS with "
                                                 "overlapping launches (round 4 reported this as `frac`), not the kernel's fraction"},
                    "note": "frac = algorithmic bytes (SURVEY.md 8d accounting x the cells the launches processed, counted on the "
                            "device) / the sum of the launches' OWN durations / peak, over every launch of the dominant kernel in "
                            "the pass named by own_durations_from -- the number to hold against AverageNs of the kernel in the "
                            "rocprofv3 --kernel-trace --stats summary of a --block-threads 1 run (profiles/, tests/"
                            "test_roofline_profiles.py).  full_sweep = the launches of the first round of a solve (both "
                            "orientations over the whole block), mop_up = the later rounds' (strips whose inputs changed: bound by "
                            "one wave's label loop, few bytes).  traffic = PMC FETCH_SIZE x 2 + WRITE_SIZE per launch of the named "
                            "kernels from the committed rocprofv3 passes, only if taken with this build (traffic_source)"}
        # (round 6) the whole E-step beside the dominant kernel: what the committed passes of THIS build say about every
        # kernel of the warm path together -- vector issue capacity (profiles/r6_regime.json: 14 blocks in flight) and HBM
        # bytes (profiles/pmc_by_kernel.json) over this run's measured E-step time
        roofline["whole_estep"] = whole_estep_figures(a.workload, float(np.mean(t_e_timed)) if len(t_e_timed) else None)
        if dom_name in ("strip", "fusion"):
            # what the strip kernels move through LDS (device counters): a DP step reads 64 lanes x 8 B and its cell's
            # table (128 B) was written once; staging writes 17 B per staged cell and reads 9 x 5 B back per strip cell
            # (bytes and time of the instrumented pass, where both strip classes are timed)
            lds_bytes = (work_inst["dp_steps"] * (512.0 + 128.0) + work_inst["staged_cells"] * 17.0
                         + (work_inst["cells"] + work_inst["swept_cells"]) * 45.0)
            t_lds = max((busy.get("strip", 0.0) + busy.get("fusion", 0.0)) * 1e-3, 1e-9)
            lds_ach = lds_bytes / t_lds / 1e12
            lds_peak = 256 * 256 * 2.4e9 / 1e12      # 256 B/clk/CU x 256 CUs x 2.4 GHz (MI355X_MICROARCH.md, LDS)
            roofline_limiter = {"bound": "instruction issue of in-order waves (vector + scalar), 4 waves per SIMD",
                                "kernel": dom_name,
                                "lds": {"achieved": round(lds_ach, 2), "peak": round(lds_peak, 1), "unit": "TB/s",
                                        "frac": round(lds_ach / lds_peak, 4)},
                                "units": int(work["units"]), "single_proposal_cells": int(work["cells"]),
                                "swept_cells": int(work["swept_cells"]), "label_cells": int(work["label_cells"]),
                                "dp_steps": int(work["dp_steps"]), "proposal_nodes": int(work["proposal_nodes"]),
                                "note": "device counters of this run only.  The expansion kernel is bound neither by HBM nor "
                                        "by the LDS pipe: an exact filter settles most (strip, label) pairs in ~500 vector "
                                        "and ~300 scalar instructions, and a SIMD issues at most ~0.4 of either per cycle "
                                        "(tools/ubench/issue.hip); the measurements behind this are in HISTORY.md 3.3, DESIGN.md 3.2 and "
                                        "profiles/README.md, not repeated in this line"}
    kernels = {k: {"ms": round(v[0], 3), "launches": int(v[1]), "busy_ms": round(busy.get(k, 0.0), 3),
                   "GBps": (round(v[2] / (busy[k] * 1e-3) / 1e9, 1) if busy.get(k, 0) > 0 and v[2] > 0 else None)}
               for k, v in agg.items()}
    if dominant:
        kernels_from = ("an instrumented pass of %d EM iterations right after the timed region with every kernel class timed (not "
                        "part of `value`); in the timed region only the dominant class of the last warm-up iteration (%s) carries "
                        "HIP events -- the timers of all classes cost 4 - 5 ms per EM iteration" % (inst_steps, dominant))
    else:
        kernels_from = "the timed region (every class timed: fewer than two warm-up steps to choose the dominant class from)"

    if rank == 0:
        value = n_norm * a.steps / elapsed
        n_split = len(set(u["block"] for u in units if u["ntiles"] > 1))
        out = {
            "metric": "EM-iterations/sec x nodes (bin-pairs)", "value": value, "unit": "node-iterations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s%s; E-step start: %s" % (
                           a.workload, desc, ("" if (a.mean_run == 25 and a.noise == 1.0) else
                                              " [data regime: mean run %d bins, noise x %.2f]" % (a.mean_run, a.noise)),
                           START_POLICY[a.warm_start]),
                       "sharding": ("%d blocks (%d nodes) as %d units (%d blocks cut into row tiles: those above %.2f x a rank's "
                                    "share) dealt to %d rank(s) by longest-processing-time-first; %s"
                                    % (len(all_blocks), n_global, len(units), n_split, a.split_above, eworld,
                                       "total work fixed" if a.scaling == "strong" else "%d copies of the workload" % world)),
                       "units_per_rank": [int(sum(1 for o in owner if o == r)) for r in range(eworld)],
                       "nodes_per_rank": nodes_per_rank,
                       "S": S, "K": K, "num_neighbor": nn, "beta": a.beta, "beta1": a.beta1,
                       "step": "full EM iteration: GPU E-step of every block + stats reduction + host M-step (SLSQP; the K states "
                               "dealt to the ranks, %d of them on %d host threads here)" % (len(my_states), workers),
                       "mrf_solver": dict(solver, warm_start=a.warm_start)},
            "estep_ms": float(np.mean(t_e_timed) * 1e3), "mstep_ms": float(np.mean(t_m_timed) * 1e3),
            # the E-step of every timed step (ms, this rank): warm-started label solves make the steps unequal -- an
            # iteration that renews labels_local or restarts from it moves more labels than the one after it
            "estep_ms_by_step": [round(x * 1e3, 1) for x in t_e_timed],
            # (not the metric: `value` is units / the whole timed region; the median step tells a window that caught a far-off
            #  iteration from one that did not)
            "ms_per_step_median": float(np.median(np.asarray(t_e_timed) + np.asarray(t_m_timed)) * 1e3),
            "cold_first_iteration_ms": cold_first_ms,
            "fit": fit,
            "fit_reference_start": fit_ref,
            "fit_surface": fit_surface,
            "tile_rounds": tile_rounds,
            "per_rank": per_rank,
            **({"ab": ab} if ab else {}),
            "at_tol_1000ppb": at_old_tol,
            "cost1": [round(c, 6) for c in cost1_log[-min(len(cost1_log), 8):]],     # the last iterations' cost1 (base.py:410)
            "build": {"source_hash": source_hash()},
            "value_estep_only": n_norm * a.steps / float(np.sum(t_e_timed)),
            "setup_s": setup_s, "block_threads": a.block_threads, "components_prepared_behind_mstep": bool(blocks and not a.no_prepare),
            "kernels": kernels, "kernels_from": kernels_from,
            "kernels_steps": inst_steps or a.steps, "roofline": roofline,
            "roofline_limiter": roofline_limiter,
        }
        if emulating:
            out["emulated"] = {"world": eworld, "rank": erank, "nodes": n_local, "whole_blocks": len(blocks),
                               "tiles": [[g.block_id, t] for g in conductor.groups for t in sorted(g.local)],
                               "mstep_states": len(my_states),
                               "note": "a ONE-GPU REHEARSAL of what this rank of the N-rank run holds: its blocks, its row tiles "
                                       "(neighbour tiles absent: no halo traffic) and its share of the M-step's states; EM on "
                                       "this rank's nodes only.  `value` is this rank's nodes x steps / wall, not the metric"}
        if world == 1 and not a.no_cpu_baseline and not emulating:
            out["cpu_baseline"] = cpu_baseline(a, S, K, nn, len(all_blocks), n_global, max(sizes))
    runner.close()
    for b in unit_blocks:
        b.close()
    mstep.close_pool()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:     # the JSON line must be the LAST line on stdout: RCCL (NCCL_DEBUG=VERSION) leaves a banner in
        try:          # libc's stdout buffer that would otherwise be flushed after it, at exit
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


START_POLICY = {"best": "warm_start=best (labels_local or the previous labelling, whichever has the lower energy; the "
                        "reference's rule is warm_start=local: see fit_reference_start)",
                "local": "warm_start=local (labels_local, as the reference: phylo_hmrf.py:479)"}
KERNELS_OF_CLASS = {"strip": ["strip_cols_kernel"], "fusion": ["fusion_cols_kernel", "strip_kernel"],
                    "propose": ["propose_grid_kernel", "propose_kernel"], "emission": ["emission_kernel"],
                    "energy": ["energy_grid_kernel", "energy_delta_grid_kernel", "energy_kernel"],
                    "posterior_stats": ["posterior_kernel"]}


def pmc_traffic(workload, kernel_names):
    """HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, corrected as
    MI355X_MICROARCH.md prescribes) of the named kernels, from profiles/pmc_by_kernel.json -- but only if that file was
    recorded with THIS build (its source_hash equals the hash of phylo_hmrf_amd/csrc now) on this workload; otherwise
    None and the reason.  -> (bytes or None, provenance string)"""
    fn = os.path.join(ROOT, "profiles", "pmc_by_kernel.json")
    try:
        d = json.load(open(fn))
    except Exception:
        return None, "no profiles/pmc_by_kernel.json"
    if d.get("source_hash") != source_hash():
        return None, "profiles/pmc_by_kernel.json was recorded with another build (source_hash %s, now %s)" % (
            d.get("source_hash"), source_hash())
    if d.get("workload") != workload:
        return None, "profiles/pmc_by_kernel.json was recorded on workload %s" % d.get("workload")
    tot_b, tot_l = 0.0, 0
    for name, rec in d.get("kernels", {}).items():
        if any(name.startswith(k) for k in kernel_names):
            tot_b += rec["hbm_bytes"]
            tot_l += rec["launches"]
    if tot_l == 0:
        return None, "kernels %s not in profiles/pmc_by_kernel.json" % (kernel_names,)
    return int(tot_b / tot_l), "profiles/pmc_by_kernel.json: %s (git %s)" % (d.get("command"), d.get("git_rev"))


WARM_PATH_KERNELS = ("strip_cols", "fusion_cols", "propose_grid", "emission", "posterior", "comp_", "cc_", "energy", "choose",
                     "unary")


def whole_estep_figures(workload, estep_s):
    """-> {"valu_busy", "hbm_frac": [lower, upper], ...} of the whole E-step, from the committed counter passes of this build
    (None values + the reason when they are another build's or another workload's)"""
    out = {"valu_busy": None, "hbm_frac": None}
    try:
        reg = json.load(open(os.path.join(ROOT, "profiles", "r6_regime.json")))
        if (reg.get("build") or {}).get("source_hash") == source_hash() and workload == "cfg3":
            v = reg["estep_valu_pipe_busy_as_benched"]
            out["valu_busy"] = v["value"]
            out["valu_source"] = ("profiles/r6_regime.json: SQ_INSTS_VALU per EM iteration (counter pass of the benched command) x 4 / "
                                  "(E-step of its kernel-trace pass, %.1f ms, x %.1f GHz x 1024 SIMDs)" % (v["estep_ms"], v["assumed_GHz"]))
        else:
            out["valu_source"] = "profiles/r6_regime.json is another build's or another workload's"
    except Exception as e:          # noqa
        out["valu_source"] = "no profiles/r6_regime.json (%s)" % type(e).__name__
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_by_kernel.json")))
        if d.get("source_hash") == source_hash() and d.get("workload") == workload and estep_s:
            its = 7.0                                     # the pass runs --steps 2 --warmup 5 (its cold first iteration included)
            raw = wr = 0.0
            for name, rec in d.get("kernels", {}).items():
                if any(name.startswith(k) for k in WARM_PATH_KERNELS):
                    raw += rec.get("fetch_bytes_raw", 0.0)
                    wr += rec.get("write_bytes", 0.0)
            lo, hi = (raw + wr) / its, (2.0 * raw + wr) / its
            out["hbm_bytes_per_iteration"] = [int(lo), int(hi)]
            out["hbm_frac"] = [round(lo / estep_s / (HBM_PEAK_GBS * 1e9), 4), round(hi / estep_s / (HBM_PEAK_GBS * 1e9), 4)]
            out["hbm_source"] = ("profiles/pmc_by_kernel.json: FETCH_SIZE (raw .. x 2, the guide's gfx950 correction as an upper bound) + "
                                 "WRITE_SIZE of the warm path's kernels over the pass's 7 EM iterations, per iteration / this run's "
                                 "E-step (%.1f ms) / %.0f GB/s" % (estep_s * 1e3, HBM_PEAK_GBS))
        else:
            out["hbm_source"] = "profiles/pmc_by_kernel.json is another build's or another workload's"
    except Exception as e:          # noqa
        out["hbm_source"] = "no profiles/pmc_by_kernel.json (%s)" % type(e).__name__
    return out


def pmc_valu(workload, kernel_names):
    """roofline.valu: the named kernels' share of the vector pipes' issue capacity -- SQ_INSTS_VALU x 4 clocks (a wave64
    vector instruction occupies a SIMD's pipe for 4) / (GRBM_GUI_ACTIVE / 8 XCDs x 1,024 SIMDs) over their dispatches in the
    committed --block-threads 1 pass (profiles/pmc_by_kernel.json, `valu`; profiles/run_pmc_by_kernel.sh) -- reported only
    while that file's source_hash is this build's.  -> dict (busy None + the reason otherwise)"""
    fn = os.path.join(ROOT, "profiles", "pmc_by_kernel.json")
    try:
        d = json.load(open(fn))
    except Exception:
        return {"busy": None, "source": "no profiles/pmc_by_kernel.json"}
    if d.get("source_hash") != source_hash():
        return {"busy": None, "source": "profiles/pmc_by_kernel.json was recorded with another build (source_hash %s, now %s)"
                                        % (d.get("source_hash"), source_hash())}
    if d.get("workload") != workload:
        return {"busy": None, "source": "profiles/pmc_by_kernel.json was recorded on workload %s" % d.get("workload")}
    insts = clocks = 0.0
    launches = 0
    for name, rec in d.get("kernels", {}).items():
        v = rec.get("valu")
        if v and any(name.startswith(k) for k in kernel_names):
            insts += v["sq_insts_valu"]
            clocks += v["active_clocks"]
            launches += v["launches"]
    if launches == 0 or clocks <= 0:
        return {"busy": None, "source": "no SQ_INSTS_VALU pass for %s in profiles/pmc_by_kernel.json" % (kernel_names,)}
    return {"busy": round(insts * 4.0 / (clocks * 1024.0), 4), "sq_insts_valu_per_launch": int(insts / launches),
            "active_clocks_per_launch": int(clocks / launches), "launches": int(launches), "simds": 1024,
            "source": "profiles/pmc_by_kernel.json: %s (git %s)" % (d.get("command"), d.get("git_rev"))}


def _cpu_sample_main():
    """child process of cpu_baseline's parallel leg: `python bench.py --cpu-sample-child N S K nn seed beta beta1` runs the
    faithful warm-start E-step of one N x N diagonal block (no GPU, no torch) and prints its seconds"""
    N, S, K, nn, seed = (int(x) for x in sys.argv[2:7])
    beta, beta1 = float(sys.argv[7]), float(sys.argv[8])
    from oracle import gco_ref, ref_numpy as R, synth
    blk = synth.make_block(seed, N, N, S, K, True, nn)
    X = blk["X"]
    w, eid = R.edge_weights_from_distance(blk["edges"], beta1)
    V = R.potts_matrix(K, beta)
    have_ref = gco_ref.available()

    def label(lp, init):
        if have_ref:
            return gco_ref.cut_general_graph(eid, w, -lp, V, n_iter=5000, algorithm="swap", init_labels=init)
        from oracle import estep_c
        u_i, w_i, v_i = gco_ref.quantise(w, -lp, V, "pygco")
        return estep_c.swap_int(eid, w_i, u_i, v_i, init)[0]

    rng = np.random.default_rng(seed + 99)
    lp_prev = R.log_multivariate_normal_density_full(X, blk["means"] * (1.0 + 0.03 * rng.standard_normal(blk["means"].shape)),
                                                     blk["covars"])
    labels_prev = label(lp_prev, np.argmax(lp_prev, axis=1))
    print("READY", flush=True)
    sys.stdin.readline()                                   # all children start the timed part together
    t0 = time.time()
    lp = R.log_multivariate_normal_density_full(X, blk["means"], blk["covars"])
    labels = label(lp, labels_prev)
    pp = R.pairwise_compare_loops(labels, eid, w, V, 3)
    wp = np.exp(lp - pp)
    post = wp / wp.sum(axis=1, keepdims=True)
    R.pairwise_cost_ensemble_loops(labels, eid, w, V, 3)
    R.sufficient_statistics(post, X)
    print("SECONDS %.4f %d" % (time.time() - t0, X.shape[0]), flush=True)


def cpu_parallel(a, S, K, nn, procs):
    """the faithful sample of cpu_baseline in `procs` fresh processes at once (one block each, as the reference runs one process
    per block, base.py:357-362): a MEASURED multi-core figure next to the extrapolated bound.  -> dict or None"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-sample-child", str(a.cpu_sample), str(S), str(K), str(nn), str(a.seed),
           str(a.beta), str(a.beta1)]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    ch = []
    try:
        ch = [subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, cwd=ROOT)
              for _ in range(procs)]
        for p in ch:
            if p.stdout.readline().strip() != "READY":
                raise RuntimeError("child did not start")
        t0 = time.time()
        for p in ch:
            p.stdin.write("go\n")
            p.stdin.flush()
        secs, n = [], 0
        for p in ch:
            parts = p.stdout.readline().split()
            secs.append(float(parts[1]))
            n = int(parts[2])
            p.wait(timeout=60)
        wall = time.time() - t0
        return {"processes": procs, "value": procs * n / wall, "unit": "node-iterations/s", "wall_s": round(wall, 2),
                "per_process_s": [round(x, 2) for x in secs],
                "note": "MEASURED: %d processes at once, one %d-node block each (ref-faithful variant, OMP/BLAS threads 1)" % (procs, n)}
    except Exception as err:          # a baseline leg must never take the bench line down
        for p in ch:
            try:
                p.kill()
            except Exception:
                pass
        return {"processes": procs, "value": None, "error": "%s: %s" % (type(err).__name__, err)}


def cpu_baseline(a, S, K, nn, n_blocks=1, n_total=0, n_max=0):
    """The reference's CPU E-step, timed on this box's host cores on ONE bounded block of the workload's shape: a
    652-bin diagonal block (212,878 nodes: the size of config 1's chr21 block), S species, K states.

    Like `value`, it is a WARM-START iteration: the labelling starts from the labels of a previous iteration (here: the
    reference's own result under slightly different parameters, computed untimed), as phylo_hmrf.py:479 does.
      emission   NumPy float64 restatement of sklearn-0.18's density (phylo_hmrf.py:266-268)
      labelling  the reference's gco alpha-beta swap through pygco's quantisation (oracle/_ref = the compiled reference;
                 kind "reference") or, if that library is absent, the oracle's C restatement of gco's swap (kind "port")
      posteriors / costs / statistics, two variants:
         ref-faithful    the reference's per-node Python loops (phylo_hmrf.py:398-468) -- what its users experience
         ref-vectorised  the same arithmetic in vectorised NumPy -- a fair CPU ceiling for that part
    One core = one reference block process (base.py:357-362).  The reference runs one process per block, so on C cores
    its whole-workload rate is at most min(#blocks, C, N_tot / n_largest_block) times the single-core figure
    (`all_cores_upper_bound`, extrapolated, not measured); `measured_parallel` is the same faithful sample run in
    min(#blocks, cores) fresh processes at once on this box's cores (one block each, as the reference would): measured.  The reference's M-step (K SLSQP runs, ~0.6 s each) is NOT
    included although `value` includes the build's M-step.  A stated baseline, never the target."""
    from oracle import gco_ref, ref_numpy as R, synth
    N = a.cpu_sample
    blk = synth.make_block(a.seed, N, N, S, K, True, nn)
    X = blk["X"]
    n = X.shape[0]
    w, eid = R.edge_weights_from_distance(blk["edges"], a.beta1)
    V = R.potts_matrix(K, a.beta)
    have_ref = gco_ref.available()
    kind = "reference" if have_ref else "port"

    def label(lp, init):
        if have_ref:
            return gco_ref.cut_general_graph(eid, w, -lp, V, n_iter=5000, algorithm="swap", init_labels=init)
        from oracle import estep_c
        u_i, w_i, v_i = gco_ref.quantise(w, -lp, V, "pygco")
        return estep_c.swap_int(eid, w_i, u_i, v_i, init)[0]

    # untimed: the previous iteration (parameters 3 % off) leaves its labels behind
    rng = np.random.default_rng(a.seed + 99)
    lp_prev = R.log_multivariate_normal_density_full(X, blk["means"] * (1.0 + 0.03 * rng.standard_normal(blk["means"].shape)),
                                                     blk["covars"])
    labels_prev = label(lp_prev, np.argmax(lp_prev, axis=1))
    # timed: one warm-start E-step
    t0 = time.time()
    lp = R.log_multivariate_normal_density_full(X, blk["means"], blk["covars"])
    t_em = time.time() - t0
    t0 = time.time()
    labels = label(lp, labels_prev)
    t_cut = time.time() - t0
    t0 = time.time()
    pp = R.pairwise_compare_loops(labels, eid, w, V, 3)                      # phylo_hmrf.py:398-436
    wp = np.exp(lp - pp)
    post = wp / wp.sum(axis=1, keepdims=True)
    R.pairwise_cost_ensemble_loops(labels, eid, w, V, 3)                     # phylo_hmrf.py:438-468
    R.sufficient_statistics(post, X)
    t_loops = time.time() - t0
    t0 = time.time()
    post_v, _, _, _, _ = R.compute_posteriors_graph(labels, lp, eid, w, V, 3)
    R.sufficient_statistics(post_v, X)
    t_vec = time.time() - t0
    faithful = n / (t_em + t_cut + t_loops)
    cores = os.cpu_count() or 1
    par = min(n_blocks, cores, (n_total / float(n_max)) if n_max else 1.0) if n_blocks else 1.0
    return {"value": faithful, "unit": "node-iterations/s", "cores": 1, "kind": kind,
            "variant": "ref-faithful (Python posterior loops)",
            "vectorised": {"value": n / (t_em + t_cut + t_vec), "unit": "node-iterations/s", "cores": 1,
                           "variant": "ref-vectorised (same arithmetic in NumPy)"},
            "measured_parallel": cpu_parallel(a, S, K, nn, int(max(1, min(cores, n_blocks or 1)))),
            "all_cores_upper_bound": {"faithful": faithful * par, "vectorised": n / (t_em + t_cut + t_vec) * par,
                                      "processes": "one per block as in base.py:357-362: at most min(%d blocks, %d cores, "
                                                   "N_tot / largest block = %.1f) = %.1f x one core; extrapolated, not measured"
                                                   % (n_blocks, cores, (n_total / float(n_max)) if n_max else 1.0, par)},
            "seconds": {"emission": round(t_em, 3), "labelling": round(t_cut, 3), "posterior_loops": round(t_loops, 3),
                        "posterior_vectorised": round(t_vec, 3)},
            "sample": "one warm-start E-step of one %dx%d diagonal block (%d nodes), S=%d K=%d: emission %.2fs, labelling (%s, "
                      "from the previous iteration's labels) %.2fs, posteriors/costs/statistics %.2fs as the reference's "
                      "Python loops or %.2fs vectorised; the reference's M-step (K SLSQP runs) is NOT included"
                      % (N, N, n, S, K, t_em, "reference gco swap via pygco's quantisation" if have_ref
                         else "C restatement of gco swap", t_cut, t_loops, t_vec)}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-sample-child":
        _cpu_sample_main()
    else:
        main()
