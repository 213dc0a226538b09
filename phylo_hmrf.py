#!/usr/bin/env python3
"""Command line of the reference (`python phylo_hmrf.py -n K --chromvec 21,22 --miter M ...`, README.md:7-51;
parse_args / run, phylo_hmrf.py:1531-1760) driving the MI355X E-step.

Same flags, same defaults as the reference CODE (which differ from its README: --cons_param 1, --beta1 0.5,
--num_neighbor 8, --estimate_type 0; phylo_hmrf.py:1541,1552,1553,1556), same cache files
(data.<res>Kb.observed.<run>.npy, edgelist.<res>Kb.observed.<run>.npy, lenvec.<res>Kb.observed.<run>.txt,
phylo_hmrf.py:1676-1704) and the same output `estimate_ou_<run>_<lambda0>_<K>.mat` with the fields
state_vec, len_vec, params_vec1, params_vec2, iter_id1, iter_id2, cost_vec (:1743-1748).

Raw input (edge.1.txt, branch_length.1.txt, species_name.1.txt, path_list.txt, <ref>.chrom.sizes, chr<c>.synteny.txt and
the species' chr<c>.<res>K.txt contact files under --root_path, README.md:53-75) is pre-processed on the host by
phylo_hmrf_amd/preprocess.py, a restatement of utility.load_data_chromosome2 pinned on the reference's own loader
(tests/test_preprocess.py); --filter_mode 0 uses the build's Perona-Malik restatement of medpy's filter (medpy is not
installed: parity unpinned for that one function), --filter_mode 1 its restatement of skimage's bilateral filter
(scikit-image is not installed either: parity unpinned likewise).
--synthetic N instead generates a seeded multi-species block in the same cache format.
"""
from __future__ import print_function

import os
import sys
import time
from optparse import OptionParser

import numpy as np
import scipy.io

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args(argv=None):
    parser = OptionParser(usage="Phylo-HMRF state estimation", add_help_option=False)
    parser.add_option("-n", "--num_states", default="10", help="Set the number of states to estimate")
    parser.add_option("-f", "--chromosome", default="1", help="Chromosome name")
    parser.add_option("-l", "--length", default="one", help="Filename of length vectors")
    parser.add_option("-p", "--root_path", default=".", help="Root directory of the data files")
    parser.add_option("-m", "--multiple", default="true", help="Use multivariate data (true, default)")
    parser.add_option("-a", "--species_name", default="human", help="Species to estimate states")
    parser.add_option("-o", "--sort_states", default="false", help="Whether to sort the states")
    parser.add_option("-r", "--run_id", default="0", help="experiment id")
    parser.add_option("-c", "--cons_param", default="1", help="constraint parameter")
    parser.add_option("-t", "--method_mode", default="1", help="method_mode: 1: Phylo-HMRF")
    parser.add_option("-d", "--initial_mode", default="0", help="initial mode")
    parser.add_option("-i", "--initial_weight", default="0.3", help="initial weight 0 for initial parameters")
    parser.add_option("-k", "--initial_weight1", default="0.1", help="initial weight 1 for initial parameters")
    parser.add_option("-j", "--initial_magnitude", default="1", help="initial magnitude for initial parameters")
    parser.add_option("-s", "--simu_version", default="1", help="dataset version")
    parser.add_option("-u", "--position1", default="0", help="position1")
    parser.add_option("-v", "--position2", default="50000", help="position2")
    parser.add_option("-w", "--filter_sigma", default="0.25", help="sigma of filter")
    parser.add_option("-b", "--beta", default="1", help="beta")
    parser.add_option("--beta1", default="0.5", help="beta1")
    parser.add_option("--num_neighbor", default="8", help="number of neighbors")
    parser.add_option("--filter_mode", default="0", help="filter method")
    parser.add_option("-e", "--threshold", default="0.001", help="convergence threshold")
    parser.add_option("-g", "--estimate_type", default="0", help="the method used for estimating label")
    parser.add_option("-q", "--annotation", default="test", help="annotation of the filename")
    parser.add_option("--dtype", default="0", help="diagonal type")
    parser.add_option("--reload", default="0", help="reload existing processed data")
    parser.add_option("--quantile", default="1", help="whether to compute signal quantiles")
    parser.add_option("--miter", default="60", help="max number of iterations")
    parser.add_option("--resolution", default="50000", help="genomic bin size")
    parser.add_option("--ref_species", default="hg38", help="reference species id")
    parser.add_option("--chromvec", default="1", help="chromosomes to perform estimation")
    parser.add_option("--output", default=".", help="output directory to save files")
    # additions of this build
    parser.add_option("--synthetic", default="0", help="generate a seeded synthetic diagonal block of this side and "
                                                        "write it in the reference's cache format")
    parser.add_option("--seed", default="", help="random seed (default: non-deterministic like the reference)")
    parser.add_option("--quiet", default="0", help="1: suppress the per-iteration prints")
    parser.add_option("--init", default="minibatch", help="initial clustering: 'minibatch' = the reference's MiniBatchKMeans "
                      "(phylo_hmrf.py:234-238) for the centres, the assignment of all nodes and the per-cluster "
                      "statistics on the GPU; 'sklearn' = the reference's initialisation verbatim on the host; "
                      "'device' = Lloyd k-means on the GPU")
    parser.add_option("--warm_start", default="best", help="start of an E-step's labelling: 'local' = labels_local, the labels of "
                      "the iteration with the lowest cost so far, as the reference does (phylo_hmrf.py:479); 'best' (default) = "
                      "that or the previous E-step's labels, whichever has the lower energy under the new parameters")
    parser.add_option("--energy_tol_ppb", default="10000", help="the label solver of an E-step stops when a whole round lowers the "
                      "energy by less than this many parts per billion of it (default 10000 = 1e-5, the level at which two solves of "
                      "the same inputs differ; 1000 until round 6; 0 = to the exact fixed point, ~25x the E-step time)")
    parser.add_option("--checkpoint", default="", help="write an .npz checkpoint of the fit here after every "
                      "--checkpoint_every-th EM iteration (parameters, cost bookkeeping, labellings, random generator)")
    parser.add_option("--checkpoint_every", default="1", help="EM iterations between checkpoints")
    parser.add_option("--resume", default="", help="continue a fit from this checkpoint (same data, states and options)")
    parser.add_option("-h", "--help", action="help")
    opts, _ = parser.parse_args(argv)
    return opts


def cache_names(output_path, resolution, run_id, annot1="observed"):
    kb = int(resolution / 1000)
    return ("%s/data.%dKb.%s.%d.npy" % (output_path, kb, annot1, run_id),
            "%s/edgelist.%dKb.%s.%d.npy" % (output_path, kb, annot1, run_id),
            "%s/lenvec.%dKb.%s.%d.txt" % (output_path, kb, annot1, run_id))


def write_cache(output_path, resolution, run_id, samples, edge_list_vec, len_vec):
    """(several ranks: every one builds the same data, rank 0 writes the cache -- each file under a temporary name first and
    renamed when complete, so that no reader ever sees a truncated file)"""
    if int(os.environ.get("RANK", "0")) != 0:
        return
    f1, f2, f3 = cache_names(output_path, resolution, run_id)
    arr = np.empty(len(edge_list_vec), dtype=object)
    for i, e in enumerate(edge_list_vec):
        arr[i] = e
    with open(f1 + ".tmp", "wb") as fh:
        np.save(fh, samples)
    with open(f2 + ".tmp", "wb") as fh:
        np.save(fh, arr, allow_pickle=True)
    np.savetxt(f3 + ".tmp", np.asarray(len_vec), fmt="%d", delimiter="\t")
    for f in (f1, f2, f3):
        os.replace(f + ".tmp", f)


def init_process_group():
    """several GPUs (`python -m torch.distributed.run --nproc-per-node N phylo_hmrf.py ...`): the process group is set up FIRST
    -- before any data is loaded, so that a rank that fails early does not leave the others waiting in the rendezvous, and
    before anything touches the GPU.  -> (rank, world)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29519")
            backend = os.environ.get("PHMRF_DIST_BACKEND", "nccl")
            if backend == "nccl":
                local = 0 if os.environ.get("PHMRF_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
                torch.cuda.set_device(local)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def all_ranks_agree(value):
    """rank 0's value on every rank (a decision about files must be ONE decision)"""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return value
    import torch.distributed as dist
    box = [value]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def load_cache(output_path, resolution, run_id):
    f1, f2, f3 = cache_names(output_path, resolution, run_id)
    samples = np.load(f1)
    edge_list_vec = list(np.load(f2, allow_pickle=True))
    len_vec = np.atleast_2d(np.loadtxt(f3, dtype="int32", delimiter="\t"))
    return samples, len_vec, edge_list_vec


def synthetic_cache(N, S, K, num_neighbor, seed):
    """One N x N diagonal block in the reference's (samples, len_vec, edge_list_vec) form."""
    from phylo_hmrf_amd import synthetic
    from phylo_hmrf_amd.graph_host import grid_edges
    from phylo_hmrf_amd.tree import PhyloTree
    rng = np.random.default_rng(seed)
    tree = PhyloTree(synthetic.tree_for(S))
    params = synthetic.sample_ou_params(rng, tree, K)
    means, covars = tree.mean_cov(params)
    covars = covars + 1e-3 * np.eye(S)
    img = synthetic.label_image(rng, N, N, K)
    ii, jj = np.triu_indices(N)
    lab = img[ii, jj]
    L = np.linalg.cholesky(covars)
    X = np.maximum(means[lab] + np.einsum("nij,nj->ni", L[lab], rng.standard_normal((lab.shape[0], S))), 0.0)
    edges = grid_edges(X, N, N, True, num_neighbor)
    n = X.shape[0]
    len_vec = [[n, 0, n, N, N, 0, 0, 0, 1, 1]]        # utility.py:455-456, :528
    return X, len_vec, [edges], synthetic.tree_for(S)


def run(num_states, chromvec, root_path, multiple, species_name, sort_states, run_id1, cons_param, method_mode,
        initial_mode, initial_weight, initial_weight1, initial_magnitude, position1, position2, filter_sigma, beta,
        beta1, num_neighbor, filter_mode, conv_threshold, estimate_type, simu_version, annotation, reload_mode,
        diagonal_type, m_iter, resolution, quantile, ref_species, output_path, synthetic="0", seed="", quiet="0",
        init_method="minibatch", warm_start="best", checkpoint="", checkpoint_every="1", resume="", energy_tol_ppb="10000"):
    run_id = int(run_id1)
    n_components1 = int(num_states)
    cons_param = float(cons_param)
    initial_mode = int(initial_mode)
    initial_weight, initial_weight1, initial_magnitude = float(initial_weight), float(initial_weight1), float(initial_magnitude)
    method_mode = int(method_mode)
    version = int(simu_version)
    beta, beta1 = float(beta), float(beta1)
    num_neighbor = int(num_neighbor)
    conv_threshold = float(conv_threshold)
    estimate_type = int(estimate_type)
    annotation = str(annotation)
    reload_mode = int(reload_mode)
    m_iter = int(m_iter)
    resolution = int(resolution)
    output_path = str(output_path)
    data_path = str(root_path)
    synthetic = int(synthetic)
    seed = None if seed == "" else int(seed)
    print("estimate type %d" % estimate_type)
    os.makedirs(output_path, exist_ok=True)
    rank, world = init_process_group()

    from phylo_hmrf_amd import mstep
    from phylo_hmrf_amd.tree import load_tree_files
    if not mstep.native_available():      # (only the Python fall-back of the M-step uses worker processes: forked before the GPU is touched)
        mstep._pool(min(n_components1, os.cpu_count() or 1))

    start = time.time()
    if synthetic > 0:
        from phylo_hmrf_amd import synthetic as _syn
        S = 4
        samples, len_vec, edge_list_vec, edge_list = synthetic_cache(synthetic, S, n_components1, num_neighbor, seed or 0)
        branch_list = [1.0] * (len(edge_list))
        write_cache(output_path, resolution, run_id, samples, edge_list_vec, len_vec)
    else:
        edge_list, branch_list, species = load_tree_files(data_path)       # phylo_hmrf.py:1607-1631
        # (rank 0 looks, everybody follows: with --reload 1 no rank may find the cache half there while another writes it)
        have_cache = all_ranks_agree(all(os.path.exists(f) for f in cache_names(output_path, resolution, run_id)))
        if reload_mode == 1 and not have_cache:
            print("%s does not exist" % cache_names(output_path, resolution, run_id)[0])   # :1682-1684
            reload_mode = 0
        if reload_mode == 1:
            samples, len_vec, edge_list_vec = load_cache(output_path, resolution, run_id)
        else:
            from phylo_hmrf_amd import preprocess
            with open("%s/path_list.txt" % data_path) as f:                 # :1633-1639
                filename_list = [line.strip() for line in f if line.strip()]
            # path_list.txt holds the species' directories, relative to the working directory in the reference's example
            filename_list = [p if os.path.isabs(p) or os.path.isdir(p) else os.path.join(data_path, p) for p in filename_list]
            chrom_vec = list(range(1, 23)) if str(chromvec) == "-1" else [int(c) for c in str(chromvec).split(",")]
            ref_filename = "%s/%s.chrom.sizes" % (data_path, str(ref_species))
            quantile = int(quantile)
            qfile = "chrom_quantile_test.txt"                               # :1648-1664
            # (rank 0 looks, everybody follows -- as for the cache files: no rank may find the file half written by another;
            #  the file appears by rename, and x_max itself is rank 0's on every rank)
            if quantile == 0 and all_ranks_agree(os.path.exists(qfile)):
                x_max = float(np.median(np.atleast_2d(np.loadtxt(qfile, delimiter="\t"))[:, 6])) if rank == 0 else 0.0
            else:
                m_vec_list = preprocess.quantile_contact_vec(chrom_vec, resolution, ref_filename, filename_list, species)
                if rank == 0:
                    np.savetxt(qfile + ".tmp", m_vec_list, fmt="%.4f", delimiter="\t")
                    os.replace(qfile + ".tmp", qfile)
                x_max = float(np.median(m_vec_list[:, 6]))
            x_max = float(all_ranks_agree(x_max))
            print(x_max)
            samples, len_vec, edge_list_vec = preprocess.load_data_chromosome2(
                chrom_vec, x_max, 0, resolution, num_neighbor, int(filter_mode), float(filter_sigma), int(diagonal_type),
                ref_filename, filename_list, species, data_path, annotation)
            write_cache(output_path, resolution, run_id, samples, edge_list_vec, len_vec)     # :1697-1704
    print("use time load data: %s" % (time.time() - start))
    print(samples.shape)
    print(np.asarray(len_vec))

    if method_mode == 1:
        from phylo_hmrf_amd.hmrf import phyloHMRF
        # several GPUs: one process per GPU (init_process_group above); the blocks (and row tiles of the ones larger than a
        # GPU's share) are dealt to the ranks, every rank loads the data, rank 0 writes the result
        tree1 = phyloHMRF(n_components=n_components1, run_id=run_id, n_samples=samples.shape[0],
                          n_features=samples[0].shape[-1], observation=samples, edge_list=edge_list, len_vec=len_vec,
                          type_id=version, branch_list=branch_list, edge_list_1=edge_list_vec, cons_param=cons_param,
                          beta=beta, beta1=beta1, initial_mode=initial_mode, initial_weight=initial_weight,
                          initial_weight1=initial_weight1, initial_magnitude=initial_magnitude, learning_rate=0.001,
                          estimate_type=estimate_type, max_iter=100, n_iter=5000, tol=1e-7, num_neighbor=num_neighbor,
                          random_state=seed, quiet=bool(int(quiet)), init_method=init_method, warm_start=warm_start,
                          checkpoint_path=checkpoint or None, checkpoint_every=int(checkpoint_every), resume_from=resume or None,
                          solver_opts=dict(energy_tol_ppb=int(energy_tol_ppb)))
        print("fitting...")
        lambda_0 = cons_param
        filename = "%s/estimate_ou_%d_%.2f_%d_%s" % (output_path, run_id, lambda_0, n_components1, annotation)
        start = time.time()
        params_vec1, params_vec2, params_vecList, iter_id1, iter_id2, cost_vec, state_vec = tree1.fit_accumulate_test(
            samples, len_vec, conv_threshold, filename, m_iter)
        print("fit use time: %s" % (time.time() - start))
        mdict = {"state_vec": state_vec, "len_vec": np.asarray(len_vec), "params_vec1": params_vec1,
                 "params_vec2": params_vec2, "iter_id1": iter_id1, "iter_id2": iter_id2, "cost_vec": cost_vec}
        filename3 = "%s/estimate_ou_%d_%.2f_%d.mat" % (output_path, run_id, lambda_0, n_components1)
        if rank == 0:                                   # (every rank holds the same result: state_vec is all-reduced)
            scipy.io.savemat(filename3, mdict)
        print(params_vecList.shape)
        tree1.close()
        mstep.close_pool()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return filename3
    mstep.close_pool()
    return None


if __name__ == "__main__":
    opts = parse_args()
    run(opts.num_states, opts.chromvec, opts.root_path, opts.multiple, opts.species_name, opts.sort_states, opts.run_id,
        opts.cons_param, opts.method_mode, opts.initial_mode, opts.initial_weight, opts.initial_weight1,
        opts.initial_magnitude, opts.position1, opts.position2, opts.filter_sigma, opts.beta, opts.beta1,
        opts.num_neighbor, opts.filter_mode, opts.threshold, opts.estimate_type, opts.simu_version, opts.annotation,
        opts.reload, opts.dtype, opts.miter, opts.resolution, opts.quantile, opts.ref_species, opts.output,
        synthetic=opts.synthetic, seed=opts.seed, quiet=opts.quiet, init_method=opts.init, warm_start=opts.warm_start,
        checkpoint=opts.checkpoint, checkpoint_every=opts.checkpoint_every, resume=opts.resume,
        energy_tol_ppb=opts.energy_tol_ppb)
