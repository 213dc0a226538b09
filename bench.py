#!/usr/bin/env python3
"""Benchmark of the Phylo-HMRF EM hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A STEP is one full EM iteration over every syntenic block of the workload, exactly as fit_accumulate_test drives it
(base.py:347-442): per block  labels <- labels_local, emission log-likelihoods, MRF labelling, posteriors + costs +
sufficient statistics;  then the reduction of the statistics (one RCCL all-reduce when N > 1), the cost
bookkeeping, and the host M-step (SciPy SLSQP over the OU parameters of every state; rank 0 + broadcast).
Inputs (X, graph) are resident in HBM before the timed region.  Metric (BASELINE.json):
EM-iterations/sec x nodes = N_tot * steps / wall.  Weak scaling: every rank owns one copy of the workload's blocks.

Extra objects on the JSON line: "roofline" for the kernel class with the largest device time (HIP events on the
blocks' streams, resolved after the timed region) and "cpu_baseline" (rank 0, N=1): the reference's CPU path --
NumPy emission + the reference's gco alpha-beta swap (oracle/_ref, through pygco's quantisation) + the reference's
per-node Python posterior/cost loops -- timed on a bounded sample on this box's host cores.
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
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "cfg4", "small"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--beta", type=float, default=1.0)
    ap.add_argument("--beta1", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--block-threads", type=int, default=12,
                    help="host threads driving blocks concurrently, each block on its own HIP stream (1 = sequential)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not record HIP events around the kernel launches (no per-kernel breakdown / roofline)")
    ap.add_argument("--cpu-sample", type=int, default=360, help="side N of the diagonal block used for the CPU baseline")
    ap.add_argument("--no-expansion", action="store_true")
    ap.add_argument("--energy-tol-ppb", type=int, default=1000,
                    help="stop the label solver when a round lowers the energy by less than this many ppb (0 = exact "
                         "fixed point); energy parity with gco at this setting: tests/test_gpu_estep.py")
    return ap.parse_args()


def kernel_bytes(cls, n, K, S, D=8):
    """ALGORITHMIC HBM bytes of ONE launch of a kernel class on a block of n nodes (DESIGN.md section 4;
    f32 data, int32 indices, u8 labels, explicit adjacency -- the accounting of SURVEY.md 8d)."""
    nbr = D * (4 + 4 + 1)                     # neighbour id + weight + neighbour label
    if cls == "emission":
        return n * (4 * S + 4 * K)
    if cls == "icm":                          # one colour class = n/4 nodes
        return n / 4.0 * (4 * K + nbr + 1 + 1)
    if cls == "chain":                        # one colour of one family: 4n nodes over 10 launches per round
        return 0.4 * n * (4 * K + nbr + 1 + 1)
    if cls == "strip":                        # 5 of 6 rows x 63 of 64 columns; two unary entries, label, proposal
        return (5.0 / 6.0) * (63.0 / 64.0) * n * (8 + nbr + 2 + 1)
    if cls == "propose":
        return n * (4 * K + nbr + 1 + 1)
    if cls == "posterior_stats":
        return n * (4 * S + 4 * K + nbr + 1)
    if cls == "energy":
        return n * (4 + nbr + 1)
    return None


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (a.gpus, world))

    from phylo_hmrf_amd import mstep, workloads
    blocks_def, S, K, nn, desc = workloads.workload(a.workload)
    workers = min(K, os.cpu_count() or 1)
    if rank == 0:
        mstep._pool(workers)                  # fork the M-step workers BEFORE anything initialises the GPU

    import torch
    import torch.distributed as dist
    from phylo_hmrf_amd import Block, _lib
    from phylo_hmrf_amd.base import SLOT_LOCAL
    from phylo_hmrf_amd.block import unpack_stats
    from phylo_hmrf_amd.tree import PhyloTree
    _lib.require_gpu()
    torch.cuda.set_device(local_rank)
    _lib.check(_lib.load().phmrf_set_device(local_rank))
    dev = torch.device("cuda", local_rank)
    # PHMRF_FORCE_DIST=1 exercises the RCCL path (all-reduce, broadcast, barrier) with a single rank
    use_dist = world > 1 or os.environ.get("PHMRF_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- synthetic multi-species Hi-C, generated on the device ----------------------------
    from phylo_hmrf_amd import synthetic
    tree = PhyloTree(synthetic.tree_for(S))
    rng = np.random.default_rng(a.seed)
    params_true = synthetic.sample_ou_params(rng, tree, K)
    means_true, cov_true = tree.mean_cov(params_true)
    cov_true = cov_true + 1e-3 * np.eye(S)                     # EM-time covariances carry 2e-3 (phylo_hmrf.py:1522-1524)
    t_setup = time.time()
    blocks, n_total = [], 0
    for bi, (H, W, diag) in enumerate(blocks_def):
        n = workloads.block_nodes(H, W, diag)
        b = Block(n, S, K)
        Xd = synthetic.device_observations(torch, dev, a.seed * 1000 + rank * 100 + bi, H, W, diag, K, means_true, cov_true)
        torch.cuda.synchronize()
        b.set_observations_dev(Xd.data_ptr())
        b.sync()
        del Xd
        b.build_grid_graph(H, W, diag, nn, a.beta1)            # stencil graph + w = exp(-beta1 d) built on the device
        blocks.append(b)
        n_total += n
    torch.cuda.empty_cache()
    n_stats = K * (1 + S + S * S)
    # EM starts from perturbed parameters; first labels = argmax_k logprob + one ICM sweep -> labels_local
    params_cur = np.clip(params_true * (1.0 + 0.15 * rng.standard_normal(params_true.shape)), 1e-3, 50.0)
    init_ou = params_cur.copy()
    means, covars = tree.mean_cov(params_cur)
    covars = covars + 1e-3 * np.eye(S)
    for b in blocks:
        b.emission(means, covars)
        b.solve_fast(a.beta, max_rounds=1, use_chains=False, use_components=False, use_strips=False, use_expansion=False,
                     init_mode=1)
        b.save_labels(SLOT_LOCAL)
        b.sync()
    setup_s = time.time() - t_setup
    stats_dev = torch.zeros((len(blocks), n_stats + 4), dtype=torch.float64, device=dev)
    state = dict(min_cost=1e30, params=params_cur, means=means, covars=covars)
    solver = dict(max_rounds=64, use_chains=True, use_components=True, use_strips=True, use_expansion=not a.no_expansion,
                  energy_tol_ppb=a.energy_tol_ppb)
    n_global = n_total * world
    t_e, t_m = [], []

    # The blocks are independent (the reference forks one process per block, base.py:357-362): a few host threads
    # drive them concurrently, each block on its own HIP stream, largest first, so the latency-bound launches of the
    # small blocks fill the GPU next to the large ones.  ctypes drops the GIL for the duration of a library call.
    from phylo_hmrf_amd.concurrent import BlockRunner
    runner = BlockRunner(a.block_threads, local_rank)
    order = sorted(range(len(blocks)), key=lambda i: -blocks[i].n)

    def estep_block(i):
        b = blocks[i]
        b.restore_labels(SLOT_LOCAL)                           # init_labels = labels_local (phylo_hmrf.py:479)
        b.emission(state["means"], state["covars"])
        b.solve_fast(a.beta, **solver)
        b.posterior_stats_dev(a.beta, 3, stats_dev[i].data_ptr())
        b.sync()

    def em_step():
        t0 = time.time()
        runner.map(estep_block, order)
        tot = stats_dev.sum(dim=0)
        if use_dist:
            dist.all_reduce(tot)                               # RCCL: K(1+S+S^2)+4 doubles
        tot = tot.cpu().numpy()
        stats = unpack_stats(tot[:n_stats], K, S)
        cost1 = tot[n_stats + 3] / n_global
        if cost1 < state["min_cost"]:                          # base.py:416-420
            state["min_cost"] = cost1
            for b in blocks:
                b.save_labels(SLOT_LOCAL)
        t1 = time.time()
        if rank == 0:
            p, mu, cv, _ = mstep.do_mstep(tree, stats, state["params"], init_ou, n_global, 1.0, 0, 0.3, 0.1, 1.0, rng,
                                          workers=workers)
            packed = np.concatenate([p.ravel(), mu.ravel(), cv.ravel()])
        else:
            packed = np.zeros(K * (tree.n_params + S + S * S))
        if use_dist:
            t = torch.from_numpy(packed).to(dev)
            dist.broadcast(t, src=0)
            packed = t.cpu().numpy()
        P = tree.n_params
        state["params"] = packed[:K * P].reshape(K, P)
        state["means"] = packed[K * P:K * P + K * S].reshape(K, S)
        state["covars"] = packed[K * P + K * S:].reshape(K, S, S)
        t2 = time.time()
        t_e.append(t1 - t0)
        t_m.append(t2 - t1)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        em_step()
    for b in blocks:
        b.enable_timing(not a.no_kernel_timing)
        b.reset_timing()
    del t_e[:], t_m[:]
    barrier()
    t0 = time.time()
    for _ in range(a.steps):
        em_step()
    barrier()
    elapsed = time.time() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel-class device time (HIP events recorded on the blocks' streams during the timed region) ----
    def collect():
        agg = {}
        for b in blocks:
            for name, (ms, ln) in b.timing().items():
                d = agg.setdefault(name, [0.0, 0, 0.0])
                d[0] += ms
                d[1] += ln
                kb = kernel_bytes(name, b.n, K, S)
                if kb is not None:
                    d[2] += kb * ln
        return agg

    def roofline_of(agg, dom_name):
        dom_ms, dom_launches, dom_bytes = agg[dom_name]
        if not (dom_launches and dom_ms > 0 and dom_bytes > 0):
            return None
        ach = dom_bytes / (dom_ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": dom_name, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(a.workload, dom_name),
                "avg_launch_us": round(dom_ms * 1e3 / dom_launches, 2), "launches": int(dom_launches),
                "algorithmic_bytes_per_launch": int(dom_bytes / dom_launches)}

    agg = collect()
    dom_name = max(agg.items(), key=lambda kv: kv[1][0])[0]
    roofline = roofline_of(agg, dom_name)
    kernels = {k: {"ms": round(v[0], 3), "launches": int(v[1]),
                   "GBps": (round(v[2] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 and v[2] > 0 else None)} for k, v in agg.items()}
    # With several blocks in flight the event intervals above contain the kernels of OTHER streams sharing the GPU:
    # they are what the timed region saw, but not a kernel's own duration.  One extra E-step, blocks one after the
    # other, gives the uncontended figure of the same kernel (outside the timed region, not part of `value`).
    roofline_isolated = None
    if roofline is not None and runner.n_threads > 1 and len(blocks) > 1 and not a.no_kernel_timing:
        for b in blocks:
            b.reset_timing()
        for i in order:
            estep_block(i)
        roofline_isolated = roofline_of(collect(), dom_name)

    if rank == 0:
        value = n_global * a.steps / elapsed
        out = {
            "metric": "EM-iterations/sec x nodes (bin-pairs)", "value": value, "unit": "node-iterations/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": a.workload + ": " + desc, "blocks_per_gpu": len(blocks), "nodes_per_gpu": n_total,
                       "S": S, "K": K, "num_neighbor": nn, "beta": a.beta, "beta1": a.beta1,
                       "step": "full EM iteration: GPU E-step of every block + stats reduction + host M-step (SLSQP, %d workers)" % workers,
                       "mrf_solver": solver},
            "estep_ms": float(np.mean(t_e) * 1e3), "mstep_ms": float(np.mean(t_m) * 1e3),
            "value_estep_only": n_global * a.steps / float(np.sum(t_e)),
            "setup_s": setup_s, "block_threads": runner.n_threads, "kernels": kernels, "roofline": roofline,
            "roofline_isolated": roofline_isolated,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a, S, K, nn)
    runner.close()
    for b in blocks:
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


def pmc_traffic(workload, kernel_class):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE and WRITE_SIZE collected in separate runs of this same command; see profiles/README.md).
    None when no measurement for this workload / kernel is on file."""
    fn = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
    try:
        d = json.load(open(fn))
        rec = d[workload][kernel_class]
        return int(rec["hbm_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(a, S, K, nn):
    """The reference's CPU E-step on one bounded diagonal block: NumPy float64 emission (sklearn-0.18 density
    restated), gco alpha-beta swap through pygco's quantisation (the compiled reference, oracle/_ref) or -- if that
    library is not present -- the oracle's own move model, and the reference's per-node Python loops for the
    posteriors / costs (phylo_hmrf.py:398-468).  One core, like one reference block process (base.py:357-362)."""
    from oracle import gco_ref, mrf_moves, ref_numpy as R, synth
    N = a.cpu_sample
    blk = synth.make_block(a.seed, N, N, S, K, True, nn)
    X = blk["X"]
    n = X.shape[0]
    w, eid = R.edge_weights_from_distance(blk["edges"], a.beta1)
    V = R.potts_matrix(K, a.beta)
    t0 = time.time()
    lp = R.log_multivariate_normal_density_full(X, blk["means"], blk["covars"])
    t_em = time.time() - t0
    init = np.argmax(lp, axis=1)
    t0 = time.time()
    if gco_ref.available():
        kind = "reference"
        labels = gco_ref.cut_general_graph(eid, w, -lp, V, n_iter=5000, algorithm="swap", init_labels=init)
    else:                                   # the oracle's plain-C restatement of gco's swap (oracle/estep_oracle.c)
        from oracle import estep_c
        kind = "port"
        u_i, w_i, v_i = gco_ref.quantise(w, -lp, V, "pygco")
        labels, _, _ = estep_c.swap_int(eid, w_i, u_i, v_i, init)
    t_cut = time.time() - t0
    t0 = time.time()
    pp = R.pairwise_compare_loops(labels, eid, w, V, 3)                      # phylo_hmrf.py:398-436
    wp = np.exp(lp - pp)
    post = wp / wp.sum(axis=1, keepdims=True)
    R.pairwise_cost_ensemble_loops(labels, eid, w, V, 3)                     # phylo_hmrf.py:438-468
    R.sufficient_statistics(post, X)
    t_post = time.time() - t0
    total = t_em + t_cut + t_post
    return {"value": n / total, "unit": "node-iterations/s", "cores": 1, "kind": kind,
            "sample": "E-step of one %dx%d diagonal block (%d nodes), S=%d K=%d: emission %.2fs, labelling (%s) %.2fs, "
                      "posterior/cost Python loops %.2fs; the reference's M-step (K SLSQP runs) is NOT included"
                      % (N, N, n, S, K, t_em, "reference gco swap" if kind == "reference" else "C restatement of gco swap", t_cut, t_post)}


if __name__ == "__main__":
    main()
