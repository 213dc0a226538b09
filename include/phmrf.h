/* libphmrf -- MI355X (gfx950) E-step of Phylo-HMRF behind a plain C ABI.
 *
 * One `phmrf_block_t` is one syntenic block (an independent MRF; reference: one OS process per
 * block, base.py:357-362) resident on ONE GPU: observations X[n,S], the neighbour graph, the
 * emission log-likelihoods logprob[n,K] and the labels stay in HBM across EM iterations.  Per
 * iteration the host sends K*(S+S*S) doubles (means_, _covars_) and receives K*(1+S+S*S)+4
 * doubles (sufficient statistics + cost numerators).
 *
 * Every entry point returns a status int (PHMRF_OK == 0); nothing throws across the boundary
 * (the reference's native layer signals by GCException, gco_source/GCoptimization.cpp:1072-1076;
 * here it is a status + phmrf_last_error()).  Host arrays are plain C-order arrays owned by the
 * caller (NumPy-compatible), float64 / int32 / int64 as the reference passes them.
 *
 * Reference interfaces replaced (file:line in /root/reference):
 *   b1  phyloHMRF._compute_log_likelihood            phylo_hmrf.py:266-268  -> phmrf_emission
 *   b2  pygco.cut_general_graph(... 'swap' ...)      phylo_hmrf.py:496-498  -> phmrf_mrf_solve
 *       (gco handle API it sits on: GCoptimization.h:551-597)                  phmrf_block_set_graph
 *   b3  phyloHMRF._compute_posteriors_graph + stats  phylo_hmrf.py:334-355, :311-314
 *                                                                           -> phmrf_posterior_stats
 *   a-E judged energy                                      -> phmrf_mrf_energy
 */
#ifndef PHMRF_H_
#define PHMRF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHMRF_API __attribute__((visibility("default")))

/* ---- status codes --------------------------------------------------------------------------- */
#define PHMRF_OK 0
#define PHMRF_ERR_INVALID 1      /* bad argument (shape, range, NULL)                              */
#define PHMRF_ERR_HIP 2          /* a HIP runtime call failed (see phmrf_last_error)               */
#define PHMRF_ERR_NO_DEVICE 3    /* no gfx950 device visible                                       */
#define PHMRF_ERR_UNSUPPORTED 4  /* valid request outside what the kernels cover (K>64, deg>64...) */
#define PHMRF_ERR_STATE 5        /* call order: e.g. solve before the graph or logprob was set     */
#define PHMRF_ERR_NOT_PD 6       /* a state covariance is not positive definite (sklearn raises
                                    ValueError at the same point)                                  */

/* Threading: every block owns a HIP stream.  Calls on DIFFERENT blocks may be made concurrently from different host
 * threads (each thread selects the device with phmrf_set_device first: the current device is per thread); calls on
 * the SAME block must be serialised by the caller.  phmrf_last_error() is per thread. */
typedef struct phmrf_block* phmrf_block_t;

/* ---- library ---------------------------------------------------------------------------------- */
/* ABI version = major * 100 + minor.  110 (round 4): PHMRF_NUM_KERNEL_CLASSES is 10 and phmrf_block_get_timing takes the
 * capacity of the caller's arrays; phmrf_block_get_work writes 8 values; the resumable solve (phmrf_mrf_solve_begin ...
 * _end) and the row-tile entry points are new.  120 (round 5): phmrf_block_get_work_first, phmrf_block_get_timing_first; phmrf_solve_opts.coarse_start.  121: phmrf_block_prepare_components.  122 (round 6): phmrf_block_get_work_ex.  123: phmrf_mrf_graph_expansion.  124: phmrf_mrf_solve_group.  A binding checks phmrf_version() == PHMRF_VERSION when it loads the library
 * (phylo_hmrf_amd/_lib.py does). */
#define PHMRF_VERSION 124
PHMRF_API int phmrf_version(void);
PHMRF_API const char* phmrf_last_error(void);
PHMRF_API const char* phmrf_status_string(int status);
PHMRF_API int phmrf_device_count(int* count);
PHMRF_API int phmrf_set_device(int device);

/* ---- block lifetime --------------------------------------------------------------------------- */
/* n nodes (Hi-C bin pairs), S species (leaves), K states.  K <= 64, S <= 16, n < 2^31 - 64.
 * Limits per entry point: the emission (b1) covers S <= 16; phmrf_posterior_stats (b3) covers S <= 8 at every K <= 64
 * (its LDS tile -- 256, 128 or 64 nodes x (K + 1 + S + 1) floats plus K (1 + S + S(S+1)/2) doubles: 29 KB at K = 20, S = 4,
 * up to 158 KB of the CU's 160 at K = 64, S = 8) and returns PHMRF_ERR_UNSUPPORTED for S > 8; node degree <= 64
 * (phmrf_block_set_graph). */
PHMRF_API int phmrf_block_create(int64_t n, int S, int K, phmrf_block_t* out);
/* Environment, read when a block is created: PHMRF_DETERMINISTIC=1 makes the label solver's reductions independent of
 * the order in which atomics land (the component moves' per-component sums and the round-by-round energies are then
 * accumulated as fixed-point integers; 8 instead of 4 bytes per node and label of scratch for the component table): two
 * solves of the same inputs return identical labels.  Default: f32 / f64 atomics, labellings that can differ in a few
 * nodes from run to run (the reference is not deterministic either: random restarts, k-means).
 * SCOPE: the label solver only (phmrf_mrf_solve and the move passes).  The posterior statistics (f64 atomics across
 * workgroups) and the k-means moments of the initialisation (f32 LDS / f64 global atomics) stay order-dependent in their
 * last bits, so a seeded FIT is reproducible to rounding, not bit for bit. */
PHMRF_API int phmrf_block_destroy(phmrf_block_t b);
/* Run this block's kernels on a caller-owned hipStream_t (e.g. torch's current stream); NULL = the
 * block's own stream. */
PHMRF_API int phmrf_block_set_stream(phmrf_block_t b, void* hip_stream);
PHMRF_API int phmrf_block_sync(phmrf_block_t b);

/* X: host float64 [n,S] C-order (reference `samples[s1:s2]`, phylo_hmrf.py:303); stored as f32. */
PHMRF_API int phmrf_block_set_observations(phmrf_block_t b, const double* X);
/* Same from a device f32 [n,S] buffer (device-to-device copy on the block's stream). */
PHMRF_API int phmrf_block_set_observations_dev(phmrf_block_t b, const float* X_dev);

/* Undirected graph as the reference hands it to pygco (phylo_hmrf.py:496): edges int64 [E,2]
 * (one row per unordered pair, any order, no self loops, no duplicates), w float64 [E] >= 0
 * (= exp(-beta1*d), phylo_hmrf.py:585).  Builds the device adjacency once; the graph is constant
 * across EM iterations (phylo_hmrf.py:101). */
PHMRF_API int phmrf_block_set_graph(phmrf_block_t b, int64_t E, const int64_t* edges, const double* w);
/* Optional geometry of the block (len_vec fields 3,4,8: H, W, type):
 * diagonal=1: nodes are the first H <= W rows of the W x W upper triangle, row-major (row i holds columns i .. W-1; the
 * reference's diagonal blocks have H == W; H < W is a ROW TILE of one, see phmrf_block_set_tile); diagonal=0: full H x W
 * row-major.
 * num_neighbor 8 or 4 (utility.py:1898-1905).  Checked against the edge list (every edge must join
 * grid neighbours); enables the 1-D chain moves (rows / columns / diagonals). */
PHMRF_API int phmrf_block_set_grid(phmrf_block_t b, int H, int W, int diagonal, int num_neighbor);

/* Build the neighbour graph of a grid block ON THE DEVICE from the resident observations, replacing the host
 * edge list: the stencil and distances of utility.py:1871-2053 (d = |xi-xj|^2/(|xi||xj|+1e-16), diagonal-to-
 * diagonal edges of a diagonal block halved) and w = exp(-beta1*d) (phylo_hmrf.py:585).  Implies set_grid. */
PHMRF_API int phmrf_block_build_grid_graph(phmrf_block_t b, int H, int W, int diagonal, int num_neighbor, double beta1);
/* Read back the device adjacency (tests): *D = row width, nbr_out int32 [n,D] (-1 padded), wgt_out f32 [n,D];
 * either output may be NULL. */
PHMRF_API int phmrf_block_get_adjacency(phmrf_block_t b, int* D, int32_t* nbr_out, float* wgt_out);

/* labels: host int32 [n], each in [0,K).  (reference: init_labels = labels_local[id1:id2],
 * phylo_hmrf.py:479; float input is cast by the Python layer.) */
PHMRF_API int phmrf_block_set_labels(phmrf_block_t b, const int32_t* labels);
PHMRF_API int phmrf_block_get_labels(phmrf_block_t b, int32_t* labels);
/* Device-side label snapshots (slot 0..3), e.g. labels_local / t_labels (base.py:419, :426) without a
 * host round trip. */
PHMRF_API int phmrf_block_save_labels(phmrf_block_t b, int slot);
PHMRF_API int phmrf_block_restore_labels(phmrf_block_t b, int slot);
PHMRF_API int phmrf_block_get_saved_labels(phmrf_block_t b, int slot, int32_t* labels);

/* The warm start of a labelling under new parameters.  The reference starts from labels_local, the labels of the EM
 * iteration with the lowest cost so far (phylo_hmrf.py:479, base.py:416-420), however old; the block's current labels are
 * the previous E-step's result.  Both are scored under the resident logprob (phmrf_emission first) and, with choose != 0,
 * the one with the lower energy becomes the current labelling: the solve then starts at or below the reference's start.
 * choose == 0: only the two energies (*e_current, *e_saved; the row tiles of a split block decide on their sums).
 * With all three output pointers NULL the evaluations and the choice (taken on the device) are only queued on the block's
 * stream: no host synchronisation.  When the snapshot IS the current labelling (saved or restored since the last change)
 * nothing is evaluated: *e_current = +inf, *e_saved = 0. */
PHMRF_API int phmrf_block_warm_start(phmrf_block_t b, double beta, int slot, int choose, double* e_current, double* e_saved,
                                     int* took_saved);

/* ---- b1: emission ---------------------------------------------------------------------------- */
/* logprob[i,k] = log N(x_i; means[k], covars[k]) for all nodes of the block, left resident in HBM.
 * means float64 [K,S], covars float64 [K,S,S].  Cholesky (with sklearn's +1e-7*I retry) and
 * L^-1 are done in float64 on the host, the per-node arithmetic in float32 on the device. */
PHMRF_API int phmrf_emission(phmrf_block_t b, const double* means, const double* covars);
PHMRF_API int phmrf_block_get_logprob(phmrf_block_t b, double* logprob /* [n,K] */);
/* Install externally computed log-likelihoods (drop-in cut_general_graph: unary = -logprob). */
PHMRF_API int phmrf_block_set_logprob(phmrf_block_t b, const double* logprob /* [n,K] */);

/* Stateless device-pointer form of b1 (all pointers are device memory, f32):
 * packed = phmrf_emission_pack() output copied to the device. */
PHMRF_API int phmrf_emission_pack_size(int S, int K, int64_t* n_floats);
PHMRF_API int phmrf_emission_pack(int S, int K, const double* means, const double* covars, float* packed_host);
PHMRF_API int phmrf_emission_dev(const float* X_dev, int64_t n, int S, int K, const float* packed_dev,
                                 float* logprob_dev, void* hip_stream);

/* ---- b2: MRF labelling ------------------------------------------------------------------------ */
typedef struct phmrf_solve_opts {
  int max_rounds;      /* <=0: default 64.  One round = every ACTIVE move type once: component moves (first round, after
                          rounds that moved the labelling at large, verification rounds), one strip fusion pass and one
                          strip alpha-expansion launch per orientation, coarse alpha-expansions while the solve moves at
                          large; chain moves and the ICM sweep run in verification rounds only on grid blocks            */
  int use_chains;      /* 1: exact 1-D chain moves when geometry is known                                      */
  int use_components;  /* 1: whole-component relabel moves                                                     */
  int init_mode;       /* 0: start from the block's current labels (reference warm start, phylo_hmrf.py:479)
                          1: start from argmax_k logprob                                                        */
  int use_strips;      /* 1: exact 5-row strip fusion moves (needs the grid)                                   */
  int use_expansion;   /* 1: strip alpha-expansion sweeps (one per label and orientation) whenever the cheaper
                          moves have gone quiet; the solve ends when such a sweep is quiet too                  */
  int min_changed;     /* a round that changes at most this many labels counts as quiet (default 0)             */
  int use_coarse;      /* 1: coarse alpha-expansions (super-cells of 2 x 2, 4 x 4 and 8 x 8 nodes switch to a label as a whole,
                          solved exactly on 5 x 63 windows of super-cells by the strip kernel; needs the grid).  They
                          run while the labelling is still moving at large (the previous round changed at least 25 %
                          of the labels: a cold start), and in the verification rounds of a solve that has moved at
                          least 12.5 % of the labels in all (a far-off start); such a solve does not stop on the
                          tolerance before they have had a last say                                                */
  int energy_tol_ppb;  /* > 0: stop as soon as a round lowers the energy by less than this many parts per billion
                          of |E| (no verification round).  Meanwhile the move types whose LAST RUNS changed the fewest
                          labels are rested (a type that did not run in a round keeps the count of its last run) as long
                          as, at the round's average gain per changed label, all rested types together were worth at most
                          a quarter of the tolerance -- a heuristic about labels, not a bound on the energy a stop
                          leaves behind (what the stops do leave is measured against gco in the parity tests).
                          0: run to the exact fixed point.  The host side's default is 10000 (1e-5: the level at which two
                          solves of the same inputs differ; 1e-6 costs 8 % more E-step time for 5e-6 of energy, the exact
                          fixed point 25 x for 1.7e-4: profiles/r6_late_round_economies.txt)                          */
  int coarse_start;    /* 1: a cold start (init_mode 1) of a grid block begins COARSE-TO-FINE: the labelling problem of its
                          4 x 4 super-cells (unary terms and crossing pair weights summed: an ordinary Potts problem on a
                          grid of n / 16 nodes) is solved by this same solver -- recursively while it is large --, its
                          labels are copied down to the cells, and the fine moves start from there instead of from
                          argmax_k logprob.  Only the START of the solve changes (8-neighbour grid blocks of >= 1,024
                          nodes that are not row tiles; otherwise ignored).  0 (default): off -- measured in round 5 on
                          the synthetic Hi-C workloads: the prolongated coarse labelling starts HIGHER than argmax + one
                          ICM sweep and the solve is slower (HISTORY.md 3.1), so nothing in the fit or the bench uses it */
} phmrf_solve_opts;

typedef struct phmrf_solve_result {
  double energy;        /* E_float = sum_i -logprob[i,l_i] + beta * sum_(i,j) w_ij [l_i != l_j]  */
  double energy_unary;
  double energy_pair;
  double energy_init;   /* same for the starting labels */
  int rounds;           /* rounds executed */
  int converged;        /* 1: ended at a fixed point of all move types (a quiet verification round) or inside the energy
                           tolerance; 0: ended by max_rounds or by the launch budget of one solve (60,000 kernel launches:
                           the change stamps are 16-bit launch ticks) -- the labels are valid and the energy never
                           went up, but more rounds could still lower it                                         */
  int64_t changed;      /* total label changes applied */
} phmrf_solve_result;

/* Minimise E_float by energy-non-increasing moves from the current labels; labels stay on the device
 * (phmrf_block_get_labels to fetch).  opts NULL = defaults, res may be NULL. */
PHMRF_API int phmrf_mrf_solve(phmrf_block_t b, double beta, const phmrf_solve_opts* opts, phmrf_solve_result* res);
/* ABI 124 (round 6): the same solve of SEVERAL independent blocks from the calling thread alone -- a round of every block is
 * queued on the block's own stream; the thread then looks at the streams in turn and, as a block's round drains, collects it,
 * decides it and queues that block's next round at once: the blocks' kernels overlap on the GPU as they do with one host thread
 * per block (the reference: one process per block, base.py:357-362), without the threads.  Block for block the labelling
 * phmrf_mrf_solve gives.  List the largest blocks first. */
PHMRF_API int phmrf_mrf_solve_group(phmrf_block_t* blocks, int n_blocks, double beta, const phmrf_solve_opts* opts);
/* The same solve in pieces (phmrf_mrf_solve is exactly begin; {launch; collect; decide} while *status == 0; end):
 *   begin    resets the solve's state (stamps, memos, schedule); want_init_energy: also evaluate the starting energy
 *   launch   queues one round's kernels and the read-back of its change counters and energy on the block's stream
 *   collect  waits for them: counters[128] (changes per move type, the slots of phmrf_mrf_solve) and
 *            energy[2] = (unary sum, pair sum without beta) after the round
 *   decide   runs the schedule on (counters, energy) -- this block's, or the SUMS over the row tiles of one block that
 *            live in different blocks / on different GPUs, which then all take the same decisions --;
 *            *status: 0 another round, 1 converged, 2 stopped by max_rounds or the launch budget
 *   end      fills res (may be NULL) and ends the solve.
 * Replaces, with phmrf_block_set_tile, the reference's one-process-per-block loop (base.py:357-372) for blocks that are
 * larger than one GPU's share. */
PHMRF_API int phmrf_mrf_solve_begin(phmrf_block_t b, double beta, const phmrf_solve_opts* opts, int want_init_energy);
PHMRF_API int phmrf_mrf_solve_round_launch(phmrf_block_t b);
PHMRF_API int phmrf_mrf_solve_round_collect(phmrf_block_t b, uint64_t* counters /*[128]*/, double* energy /*[2]*/);
PHMRF_API int phmrf_mrf_solve_round_decide(phmrf_block_t b, const uint64_t* counters /*[128]*/, const double* energy /*[2]*/,
                                           int* status);
PHMRF_API int phmrf_mrf_solve_end(phmrf_block_t b, phmrf_solve_result* res);

/* ---- row tiles: one grid block on several GPUs ------------------------------------------------------------------
 * The reference makes a block smaller only by the centromere split into independent pieces (utility.py:381-393).  Here a
 * grid block is cut into row tiles; a tile is a block of its own that stores the rows it owns plus one HALO row per
 * neighbouring tile (top / bottom != 0: there is a neighbour above / below; the first / last stored row is its halo).
 * Rows r0 .. r1 of an upper-triangular block are the first rows of a smaller upper triangle (diagonal=1, H < W), of a
 * full block a full block, so a tile has the geometry every kernel already handles.  After phmrf_block_set_tile
 *   - phmrf_mrf_energy, the solve's energies, phmrf_posterior_stats and phmrf_kmeans_* count the OWNED rows only (their
 *     sums over the tiles are the whole block's); sched_n = the whole block's node count (the schedule's thresholds);
 *   - phmrf_block_tile_pins(n_top, n_bottom) freezes the first n_top / last n_bottom rows (0, 1 or 2; the halo row counts):
 *     a pinned node's unary terms become +1e9 for every label but its own, so no move type relabels it; the real terms
 *     come back when the pin is lifted (phmrf_emission lifts all).  The two rows at a cut are pinned in turn, round by
 *     round, so that neighbouring tiles never move two adjacent nodes in the same round (tile.hip);
 *   - phmrf_block_tile_get_boundary / _put_halo move the one label row per cut that the neighbour needs (host buffers:
 *     the caller ships them with whatever transport it has -- torch.distributed in phylo_hmrf_amd/tiles.py).         */
PHMRF_API int phmrf_block_set_tile(phmrf_block_t b, int top, int bottom, int64_t sched_n);
PHMRF_API int phmrf_block_tile_pins(phmrf_block_t b, int n_top, int n_bottom);
PHMRF_API int phmrf_block_tile_get_boundary(phmrf_block_t b, uint8_t* top_out, uint8_t* bottom_out);
PHMRF_API int phmrf_block_tile_put_halo(phmrf_block_t b, const uint8_t* top_in, const uint8_t* bottom_in);
/* One colour-ordered ICM sweep / one sweep of one chain family / one component-move pass (exposed
 * for move-level parity tests against oracle/mrf_moves.py).  family: 0 rows, 1 columns,
 * 2 diagonals, 3 anti-diagonals. */
PHMRF_API int phmrf_mrf_icm_sweep(phmrf_block_t b, double beta, int64_t* changed);
PHMRF_API int phmrf_mrf_chain_sweep(phmrf_block_t b, double beta, int family, int64_t* changed);
PHMRF_API int phmrf_mrf_component_pass(phmrf_block_t b, double beta, int64_t* changed);
/* ABI 123 (round 6): one alpha-expansion of the WHOLE graph -- every node keeps its label or takes alpha -- by an exact
 * minimum s-t cut on the device (maxflow.hip: lock-free push-relabel with global relabelling, capacities quantised with the
 * largest term at 2^24 as gco's finest quantisation).  The move gco's expansion() makes (GCoptimization.cpp:1120-1280) on a
 * general graph (GCoptimization.h:551-597); phmrf_mrf_solve runs it for every label on blocks without grid geometry, where
 * the strip moves do not exist.  Works on any block with a graph; *changed = labels that took alpha. */
PHMRF_API int phmrf_mrf_graph_expansion(phmrf_block_t b, double beta, int alpha, int64_t* changed);
/* Optional scheduling hint: queue, on the block's stream, the part of the next component pass that depends on the labels
 * alone (the connected components of equal label).  The EM driver calls it between two E-steps, where the GPU would
 * otherwise wait for the host's M-step (/root/reference/base.py:399: the reference's M-step follows its E-step the same
 * way); the next solve's first component pass checks ON THE DEVICE that the labels are still the ones prepared for and
 * recomputes otherwise, so results never depend on whether, or when, this was called.  No-op before the block has
 * labels and a graph; PHMRF_ERR_STATE while a solve is in progress. */
PHMRF_API int phmrf_block_prepare_components(phmrf_block_t b);
/* One pass of exact strip fusion moves.  orient 0: strips of 5 grid rows, 1: of 5 grid columns; shift_r in [0,5],
 * shift_c in [0,63] move the fixed separator rows/columns; alpha >= 0: every node may keep its label or take
 * alpha (strip alpha-expansion); alpha < 0: every node may keep its label or take its best alternative label. */
PHMRF_API int phmrf_mrf_strip_pass(phmrf_block_t b, double beta, int orient, int shift_r, int shift_c, int alpha,
                                   int64_t* changed);
/* Every strip alpha-expansion of the labels named by the bits of label_mask on one cut, in ascending label order, in ONE
 * launch (strip_cols_kernel: a wave owns a strip, stages it once and runs the labels back to back; an exact filter
 * settles most (strip, label) pairs without the DP).  Same result, label for label, as the phmrf_mrf_strip_pass calls
 * of those labels in that order.  Replaces the K swap-cycle inner loops of gco's swap()
 * (GCoptimization.cpp:1282-1394) inside phmrf_mrf_solve. */
PHMRF_API int phmrf_mrf_strip_multi_pass(phmrf_block_t b, double beta, int orient, int shift_r, int shift_c,
                                         uint64_t label_mask, int64_t* changed);
/* One coarse alpha-expansion (coarse.hip): super-cells of scale x scale nodes (scale in {2, 4, 8}; super-cell of node
 * (i, j) = ((i + offset) / scale, (j + offset) / scale), 0 <= offset < scale) keep their labels or switch to alpha as a
 * whole; the two-label problem of the super-cells is solved by one strip pass per orientation (shift_r, shift_c as in
 * phmrf_mrf_strip_pass).  Never raises the energy. */
PHMRF_API int phmrf_mrf_coarse_pass(phmrf_block_t b, double beta, int scale, int offset, int alpha, int shift_r, int shift_c,
                                    int64_t* changed);
/* The coarse problem itself (tests): *nc super-cells; D_out[nc] = cost of switching a super-cell alone (beta folded in),
 * lam_out[nc,4] = weights of its forward pairs (E, SW, S, SE; without beta).  Outputs may be NULL. */
PHMRF_API int phmrf_block_coarse_problem(phmrf_block_t b, double beta, int scale, int offset, int alpha, int64_t* nc,
                                         float* D_out, float* lam_out);
PHMRF_API int phmrf_mrf_energy(phmrf_block_t b, double beta, double* e_total, double* e_unary, double* e_pair);

/* ---- b3: posteriors, costs, sufficient statistics ------------------------------------------- */
/* stats_out: host float64 [K + K*S + K*S*S] = post | obs | obs*obs.T (phylo_hmrf.py:311-314).
 * costs_out: host float64 [4] = UN-normalised sums over the block's nodes of
 *   [0] sum_i sum_{e in inc(i)} V[l_other,l_i]*w'_e      (n * pairwise_cost,            :438-447)
 *   [1] sum_i -log(ppn[i,l_i] + 1e-16)                   (n * pairwise_cost_normalize,  :386,:392)
 *   [2] sum_i -logprob[i,l_i]                            (n * unary_cost,               :385-390)
 *   [3] [1] + [2]                                        (n * cost1,                    :394)
 * w'_e = w_e when estimate_type == 3 else 1 (:431-434, :460-462).
 * posteriors_out: optional host float64 [n,K] (NULL to skip the download). */
PHMRF_API int phmrf_posterior_stats(phmrf_block_t b, double beta, int estimate_type, double* stats_out,
                                    double* costs_out, double* posteriors_out);
/* Same, results left in device memory (float64 [K+K*S+K*S*S+4], stats then costs) for an RCCL
 * all-reduce on the caller's stream; no host synchronisation. */
PHMRF_API int phmrf_posterior_stats_dev(phmrf_block_t b, double beta, int estimate_type, double* out_dev);

/* ---- initialisation --------------------------------------------------------- */
/* One Lloyd step of k-means on the block's device-resident observations.  The reference initialises the states
 * with sklearn's MiniBatchKMeans on the host (phylo_hmrf.py:234-238); this keeps X where it is.  Every node goes to
 * its nearest centre (squared Euclidean distance, lowest index on ties).
 *   out[0 .. K*S)        per-cluster sums of x        out[K*S .. K*S+K)  cluster sizes        out[K*S+K]  inertia
 * write_labels != 0: the assignment becomes the block's labels (the reference's init_label, :241-244).            */
PHMRF_API int phmrf_kmeans_step(phmrf_block_t b, const double* centers /* [K,S] */, int write_labels,
                                double* out /* [K*S + K + 1] */);

/* The same step with the per-cluster second moments behind the other outputs (S <= 8):
 *   out[K*S+K+1 .. +K*S*S)  sum over the cluster's nodes of x x^T.
 * With the sums and the counts that is everything the reference's initialisation takes from the observations after
 * clustering: the per-cluster OU fit works on a cluster's mean and X^T X / n (phylo_hmrf.py:1246-1325, :1427-1498) and
 * the first covariance is the global one (:258) -- so no host pass over the observations is left.                   */
PHMRF_API int phmrf_kmeans_moments(phmrf_block_t b, const double* centers /* [K,S] */, int write_labels,
                                   double* out /* [K*S + K + 1 + K*S*S] */);

/* ---- measurement ----------------------------------------------------------------------------- */
/* Accumulated device time (ms, hipEvent on the block's stream) and launch count per kernel class
 * since the last reset: 0 emission, 1 icm, 2 chain, 3 component, 4 energy, 5 posterior_stats, 6 strip (the strip
 * alpha-expansions: strip_cols_kernel, every listed label of a cut in one launch), 7 propose, 8 coarse (coarse
 * alpha-expansions: coarsen_kernel + strip_kernel on the super-cell grid + coarse_apply_kernel), 9 fusion (the fusion passes
 * of a solve: fusion_cols_kernel; the single-label strip passes of the API: strip_kernel). */
#define PHMRF_NUM_KERNEL_CLASSES 10
PHMRF_API int phmrf_block_enable_timing(phmrf_block_t b, int enable);
/* Which classes get event pairs while timing is on (bit k = class k; default all).  Two event records per launch group
 * cost host time and queue slots: a measurement that wants one kernel's durations without perturbing the rest of the run
 * times that class only (bench.py's timed region).  Launch counts and the device-counted work cover every class always. */
PHMRF_API int phmrf_block_set_timing_classes(phmrf_block_t b, uint32_t class_mask);
/* capacity = the length of the caller's arrays; min(capacity, PHMRF_NUM_KERNEL_CLASSES) entries are written */
PHMRF_API int phmrf_block_get_timing(phmrf_block_t b, int capacity, double* ms /*[capacity]*/, int64_t* launches /*[capacity]*/);
/* ... and the part of both that belongs to launches of the FIRST round of a solve (see phmrf_block_get_work_first) */
PHMRF_API int phmrf_block_get_timing_first(phmrf_block_t b, int capacity, double* ms /*[capacity]*/, int64_t* launches /*[capacity]*/);
PHMRF_API int phmrf_block_reset_timing(phmrf_block_t b);
/* Work the strip kernels (class 6) actually did since the last reset, counted ON THE DEVICE (inside a solve a strip
 * whose inputs did not change since its last quiet run is skipped after a look at its stamps and counts nothing):
 *   out[0] strips staged (one unit = one strip under one move: alpha-expansion or fusion)
 *   out[1] their strip cells (the nodes a unit re-decides: 5 x columns)
 *   out[2] grid cells staged (strip + fixed rim, 7 x (columns + 2): what a unit reads from HBM)
 *   out[3] DP steps walked (one 64-state step = one cell of one unit)   out[4] strip launches
 *   strip_cols_kernel (all labels of a strip in one wave) counts a (strip, label) pair in out[0] and, instead of
 *   out[1]:  out[5] the strip cells it swept (once per strip visit, whatever the number of labels),
 *            out[6] cells x labels examined (one label's unary term per cell)
 *   out[7] nodes whose fusion proposal was recomputed (the proposal kernels skip node tiles without a new change stamp)  */
PHMRF_API int phmrf_block_get_work(phmrf_block_t b, int64_t* out /*[8]*/);
/* The part of that work done in the FIRST round of every solve since the reset (same slots): a warm-started solve's first
 * round is the full sweep of both orientations, the later rounds revisit the strips whose inputs changed -- the two kinds of
 * launch a roofline figure has to keep apart (bench.py: roofline.full_sweep / roofline.mop_up). */
PHMRF_API int phmrf_block_get_work_first(phmrf_block_t b, int64_t* out /*[8]*/);
/* ABI 122 (round 6): the same ten-entry form of both (first_round_only != 0: the _first part); min(capacity, 10) entries.
 * Entries 0..7 as above.  Inside a solve strip_scan_kernel looks at a strip's stamps, memo row and SEED MASKS before the
 * expansion launch (the per-node masks propose_grid_kernel writes: bit a = an improving expansion of label a could start
 * here); a (strip, label) pair without a seed is settled there, exactly as the filter's first pass would settle it, and is
 * NOT in out[0] / out[6]:
 *   out[8] cells x labels settled by the seed masks (no unary term read)
 *   out[9] cells of the strips whose every listed label was settled that way (never staged: not in out[5])  */
PHMRF_API int phmrf_block_get_work_ex(phmrf_block_t b, int first_round_only, int capacity, int64_t* out /*[capacity]*/);
/* The timed intervals of one kernel class on a time base common to all blocks of the calling thread's device
 * (phmrf_time_base_reset marks t = 0; call it before the timed region): out = [start_ms, end_ms] pairs, at most
 * `capacity` of them; *count = how many there are.  Blocks run on their own streams, so their intervals overlap:
 * merging them gives the time during which at least one kernel of the class was running. */
PHMRF_API int phmrf_time_base_reset(void);
PHMRF_API int phmrf_block_get_intervals(phmrf_block_t b, int kclass, double* out, int64_t capacity, int64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* PHMRF_H_ */
