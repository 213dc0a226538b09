// Shared declarations of libphmrf (gfx950 only; no portability layer).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/phmrf.h"

namespace phmrf {

// ---- error plumbing: status int + thread-local message, nothing throws across the C ABI ---------
void set_error(const std::string& msg);
int fail(int status, const std::string& msg);

#define PHMRF_HIP(expr)                                                                     \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return ::phmrf::fail(PHMRF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

#define PHMRF_CHECK(cond, status, msg) \
  do {                                 \
    if (!(cond)) return ::phmrf::fail((status), (msg)); \
  } while (0)

#define PHMRF_TRY(expr)          \
  do {                           \
    int _s = (expr);             \
    if (_s != PHMRF_OK) return _s; \
  } while (0)

// ---- kernel classes for the built-in timers (phmrf.h) -------------------------------------------
enum KernelClass { KC_EMISSION = 0, KC_ICM = 1, KC_CHAIN = 2, KC_COMPONENT = 3, KC_ENERGY = 4, KC_POSTERIOR = 5, KC_STRIP = 6, KC_PROPOSE = 7, KC_COARSE = 8, KC_FUSION = 9 };

// A chain family (grid rows / columns / diagonals / anti-diagonals): `nodes` lists node ids chain after
// chain in chain order.  Chains of one colour share no edge.  Every chain is cut into SEGMENTS of at most
// 63 consecutive nodes separated by single nodes that stay fixed during the sweep (so simultaneous segment
// moves never share an edge); two phases shift the cut points by 32 so every node gets optimised.
struct ChainFamily {
  int32_t* nodes = nullptr;            // device [n]
  int32_t* seg_start[2][3] = {};       // device [nseg]: position in `nodes` of the segment's first node
  int32_t* seg_len[2][3] = {};         // device [nseg]: 1..63
  uint16_t* memo[2][3] = {};           // device [nseg]: tick of the segment's last quiet run (0 = none), per solve
                                       //   (slices of phmrf_block::chain_memo, set up by phmrf_mrf_solve)
  int nseg[2][3] = {};
  int n_chains = 0;
  int n_colours = 0;
  int max_len = 0;
};

}  // namespace phmrf

// One syntenic block resident on one GPU.  Layout in HBM (all row-major, contiguous):
//   X        f32 [n, S]        observations (species contact vectors)
//   logprob  f32 [n, K]        emission log-likelihoods (unary cost = -logprob)
//   labels   u8  [n]           current labelling
//   nbr      i32 [n, D]        ELL adjacency, neighbours ascending, -1 padded; D = max degree rounded up to 4
//   wgt      f32 [n, D]        edge weights w_ij (0 in padding)
struct phmrf_solve_state;
struct phmrf_block {
  int64_t n = 0;
  int S = 0, K = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;

  float* X = nullptr;
  float* logprob = nullptr;
  uint8_t* labels = nullptr;
  uint8_t* labels_tmp = nullptr;
  uint8_t* labels_eval = nullptr;           // device [n]: the labels at a solve's last energy evaluation (launch_energy_delta)
  int eval_tick = -1;                       //   ... and the launch tick then (-1: no evaluation in this solve yet)
  float* sgain = nullptr;                   // device [n]: cost of switching a node alone to its fusion proposal (launch_propose)
  unsigned long long* seed = nullptr;       // device [n]: bit a = "an improving expansion of label a could start at this node" (a superset
                                            //   of the strip filter's seed test, formed by propose_grid_kernel; strip_scan_kernel ORs it over a strip)
  unsigned long long* scan_out = nullptr;   // device [scan_slots][2]: per strip slot of the launch that follows (todo before / after the seed masks)
  int64_t scan_slots = 0;
  uint8_t* saved[4] = {nullptr, nullptr, nullptr, nullptr};
  int labels_are_slot = 0;                  // bit k: the current labels equal snapshot k (set by save / restore; 0 after anything that may write labels)
  bool has_X = false, has_logprob = false, has_labels = false, has_graph = false, has_grid = false;

  int D = 0;
  int64_t E = 0;
  int32_t* nbr = nullptr;
  float* wgt = nullptr;

  // ICM colour classes: node ids grouped by colour
  int n_colours = 0;
  int32_t* colour_nodes = nullptr;          // device [n]
  std::vector<int64_t> colour_ptr;          // host [n_colours+1]

  // grid geometry (optional) and chain families
  int H = 0, W = 0, diagonal = 0, num_neighbor = 0;
  bool grid_complete = false;               // every edge of the stencil is in the edge list (always so for device-built graphs)
  std::vector<phmrf::ChainFamily> families;

  // component-move scratch
  int32_t* comp = nullptr;                  // device [n] component root per node
  float* comp_tab = nullptr;                // device [n, K] per-root sums
  long long* comp_tab64 = nullptr;          // ... in deterministic mode: the same sums in 2^-16 fixed point (integer atomics)
  bool deterministic = false;               // PHMRF_DETERMINISTIC=1 at block creation: order-independent reductions
  int32_t* comp_best = nullptr;             // device [n]
  float* comp_gain = nullptr;               // device [n]
  uint8_t* cc_seen = nullptr;               // device [n]: the labels the prepared components were computed from
  int* cc_stale = nullptr;                  // device: 1 once cc_compare_kernel found other labels than cc_seen
  bool cc_prepared = false;                 // launch_component_prepare has run and no component pass has used it yet
  // grid-native inputs of the strip kernels
  float4* fwd_w = nullptr;                  // device [n]: weights of the four forward grid edges (E, SW, S, SE)
  float* uT = nullptr;                      // device [K][n]: unary planes, uT[k][i] = -logprob[i][k]
  float* uT_raw = nullptr;                  //   ... the allocation: uT = uT_raw + UT_PAD (slack at both ends)
  bool uT_valid = false;                    //   ... current with logprob
  bool unary_pins = false;                  // a coarse child problem: unary terms >= 1e29 pin a cell (strip_kernel looks first)
  // coarse alpha-expansions (coarse.hip): child blocks holding the two-label problem of the super-cells, side 2, 4, 8
  phmrf_block* coarse[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [level * 4 + slot in a batch of labels]
  unsigned int* coarse_flag = nullptr;      // device: set by coarse_apply_kernel when a label of the batch has moved
  // coarse-to-fine start of a cold solve (c2f.hip): the block of C2F_SCALE x C2F_SCALE super-cells, a full block of its own
  phmrf_block* c2f = nullptr;
  int c2f_Hc = 0, c2f_Wc = 0;
  unsigned long long* coarse_lab = nullptr;       // device [3][64]: labels changed per coarse scale and label in the round (the schedule rests labels)
  unsigned long long* coarse_lab_host = nullptr;  //   ... its pinned host mirror
  char* coarse_arena = nullptr;             // ONE allocation behind the twelve child problems (labels, unary planes, weights, counters)
  // alpha-expansion of a graph that is no grid by a minimum cut (maxflow.hip): reverse-arc slots and the flow's state
  uint8_t* mf_rev = nullptr;                // device [n, D]: slot of arc (u -> v) in v's adjacency row
  float* mf_theta = nullptr;                // device [n]: a node's switch cost
  int32_t* mf_cap = nullptr;                // device [n, D]: residual capacities of the arcs
  int32_t* mf_tcap = nullptr;               // device [n]: residual capacity of the arc to the sink
  long long* mf_exc = nullptr;              // device [n]: excess
  int32_t* mf_hgt = nullptr;                // device [n]: height / BFS level
  int32_t* mf_flags = nullptr;              // device [4]
  int32_t* mf_flags_host = nullptr;         //   ... pinned mirror
  int prop_tick = -1;                       // tick of the last proposal launch of this solve (-1: none)
  int seed_tick = -1;                       // ... of the last one that also wrote the seed masks: seed[i] is current while stamp[i] <= seed_tick
  int geom_phase = 0;                       // which of the three expansion cuts the next solve starts on (cycles across solves)
  // change stamps: stamp[i] = tick of the launch that last changed the label of node i OR OF ONE OF ITS NEIGHBOURS
  // (0 = not since the solve began);
  // memo[orient][geom][strip][alpha] = tick of the last strip alpha-expansion of that strip that found nothing to do.
  // A strip whose cells and border have no stamp newer than its memo would see identical inputs: skipped.
  uint16_t* stamp = nullptr;                // device [n]
  uint16_t* memo = nullptr;                 // device [2][3][memo_strips][K+1]  (slot K: the fusion pass)
  uint16_t* chain_memo = nullptr;           // device: the chain families' segment memos (ChainFamily::memo slices)
  size_t chain_memo_count = 0;
  int64_t memo_strips = 0;
  int tick = 0;                             // host launch counter inside one solve (0 = stamping off)
  int counter_slot = 0;                     // which of counters[128] the next move launches add their changes to

  float* emis_params = nullptr;             // device packed emission parameters
  float* posteriors = nullptr;              // device [n, K], allocated on demand
  double* accum = nullptr;                  // device small f64 accumulator area
  double* accum_host = nullptr;             // pinned mirror
  unsigned long long* counters = nullptr;   // device [128]
  unsigned long long* counters_host = nullptr;
  unsigned long long* work_acc = nullptr;   // device [WORK_BANKS][WORK_SLOTS], zeroed and read back with the counters
  unsigned long long* work_host = nullptr;  // pinned mirror

  // timing: event pairs recorded on the block's stream, resolved lazily (no host sync inside the measured loop)
  bool timing = false;
  unsigned int timing_mask = ~0u;           // which kernel classes get event pairs while timing is on (launch counts: all)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  struct Pending { int kclass; hipEvent_t a, b; bool first; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> free_events;
  hipEvent_t cur_start = nullptr;
  double ms[PHMRF_NUM_KERNEL_CLASSES] = {};
  int64_t launches[PHMRF_NUM_KERNEL_CLASSES] = {};
  double ms_first[PHMRF_NUM_KERNEL_CLASSES] = {};           // ... of the launches in the first round of a solve
  int64_t launches_first[PHMRF_NUM_KERNEL_CLASSES] = {};
  int64_t work_first[10] = {};              //   ... the part of it done in the first round of each solve
  int64_t work[10] = {};                    // strips staged, their cells, staged cells, DP steps, strip launches, ... (phmrf_block_get_work_ex)
  struct Interval { int kclass; float t0, t1; };
  std::vector<Interval> intervals;          // resolved timed intervals on the library's common time base (ms)

  // ---- row tiles (tile.hip, phmrf_block_set_tile): this block is rows [r0 - top, r1 + bottom) of a larger grid block whose
  // other rows live in other blocks (other GPUs).  The first / last stored row is then a HALO: a copy of the neighbour
  // tile's boundary row, never moved here, not counted in energies, costs and statistics.
  int tile_top = 0, tile_bot = 0;           // 1: there is a neighbour tile above / below (the first / last row is its halo)
  int64_t own0 = 0, own1 = -1;              // nodes [own0, own1) are the ones this block counts (own1 < 0: all n)
  int own_r0 = 0, own_r1 = -1;              // ... = grid rows [own_r0, own_r1)
  int64_t sched_n = 0;                      // node count the solver's schedule thresholds refer to (0: n; tiles: the whole block's)
  // pinned rows: the two rows at a cut are frozen in turn (round parity) so that the tiles on either side never move two
  // adjacent nodes at once.  A pinned node keeps its label because its unary terms say so: u(k) = PIN_COST for k != label.
  float* pin_save[2] = {nullptr, nullptr};  // device: the real logprob rows of the two top / two bottom rows, [rows][K]
  uint8_t* pin_label[2] = {nullptr, nullptr};   // device: their labels when the pins were last set (a pinned label must not move)
  uint8_t* xfer = nullptr;                  // device staging for the halo rows (2 rows)
  uint8_t* xfer_host = nullptr;             // pinned mirror
  int64_t xfer_cap = 0;
  bool boundary_queued = false;             // the outgoing rows' copy into xfer_host is queued / has arrived (tile_queue_boundary)
  int64_t pin_first[2] = {0, 0}, pin_count[2] = {0, 0};   // node range of the two top / two bottom rows
  int64_t pin_split[2] = {0, 0};            // nodes of the FIRST of the two rows (top region) / of the first of the two bottom rows
  int pin_rows[2] = {0, 0};                 // how many rows of each region are pinned right now (top: counted from the top, bottom: from the bottom)
  bool pin_saved = false;                   // pin_save holds the rows of the current logprob (set by tile_pins after an emission)
  struct phmrf_solve_state* ss = nullptr;   // a solve in progress (phmrf_mrf_solve_begin .. _end)
};

// the label solver between phmrf_mrf_solve_begin and _end (api.hip)
struct phmrf_solve_state {
  phmrf_solve_opts o;
  double beta = 0;
  float bf = 0;
  bool have_init_energy = false;
  double eu0 = 0, ep0 = 0;
  int64_t total = 0;
  int rounds = 0, converged = 0;
  bool chains = false, strips = false, expansions = false, coarse = false, graph_expansions = false;
  int n_fam = 0;
  int64_t tol = 0;
  int64_t sched_n = 0;
  std::vector<int> slots;
  int64_t last_changed = 0;            // labels changed by the previous round
  int64_t coarse_changed[3] = {0, 0, 0};  // ... by the coarse scales in their last run
  bool coarse_ran[3] = {false, false, false};
  unsigned long long coarse_lab_mask[3] = {~0ull, ~0ull, ~0ull};   // the labels a coarse scale still runs (bit per label)
  bool coarse_all[3] = {true, true, true};                         //   ... this round it ran all of them
  bool force_coarse = false;           // the tolerance wants to stop, but the coarse scales have not had their say
  bool coarse_checked = false;
  std::vector<char> active, ran;
  // labels changed by each move type in its LAST RUN (a type that did not run in a round -- chain families and ICM outside
  // verification rounds, the component pass in mop-up rounds, coarse scales that are off -- keeps that count: the resting
  // budget below must not read "did not run" as "changed nothing"); -1: has not run in this solve
  std::vector<long long> last_count;
  bool all_active = true, verifying = false;
  double eu_carry = 0, ep_carry = 0;   // (unary, pair without beta) at the last evaluation
  double e_prev = 0;
  int geom = 0;
  bool prev_moving = false;            // the previous round moved >= 1/64 of the labels
  // between round_launch and round_decide
  bool launched = false, collected = false, incremental = false, snapshot = false;
  int status = 0;                      // 0: another round; 1: converged; 2: out of rounds / launch budget
};

namespace phmrf {

// Development knobs (environment variables that switch kernels' phases off, force slow paths for A/B timing or change
// the schedule's shortcuts) exist only in builds with -DPHMRF_DEV (libphmrf_dev.so, tools/variant.sh).  The product library
// reads PHMRF_DETERMINISTIC (a feature, include/phmrf.h) and PHMRF_SOLVE_TRACE (prints; changes no result) and nothing else.
#ifdef PHMRF_DEV
#define PHMRF_DEV_ENV(name) (getenv(name))
#else
#define PHMRF_DEV_ENV(name) (static_cast<const char*>(nullptr))
#endif
constexpr int C2F_SCALE = 4;          // super-cells of the coarse-to-fine start (c2f.hip)
constexpr int UT_PAD = 64;            // floats of slack before and after the unary planes (strip.hip: launch_unary_planes)
constexpr int WORK_SLOTS = 9;         // see phmrf_block_get_work / _ex (slots 7, 8: what the seed masks settled)
constexpr int WORK_BANKS = 256;       // work_acc[WORK_BANKS][WORK_SLOTS]: strips staged, their cells, staged cells (with rim), DP
                                      // steps of the strip kernels, one bank per workgroup id mod 256 (no hot address)
constexpr int ACCUM_DOUBLES = 8192;  // >= K*(1+S+S*S)+16 for every (K,S) the posterior / statistics kernel supports
                                     // (K <= 64, S <= 8: 4,688); phmrf_posterior_stats rejects anything larger

// RAII-free helpers for the timers: call tic before the launches of one class, toc after.
void tic(phmrf_block* b, int kclass);
void toc(phmrf_block* b, int kclass, int n_launches);

inline int tile_threads(int K) { return K <= 40 ? 256 : 128; }
inline int padded_k(int K) { return (K % 2 == 0) ? K + 1 : K; }  // odd LDS row stride: conflict-free row-per-lane access

// kernels (defined in the .hip files) -------------------------------------------------------------
int launch_emission(const float* X, int64_t n, int S, int K, const float* packed, float* logprob, float* uT, hipStream_t st);
int launch_argmax_labels(const phmrf_block* b);
int launch_icm_colour(const phmrf_block* b, float beta, int colour);
int launch_energy(const phmrf_block* b, float beta, double* accum_at = nullptr);  // -> accum[4]=unary, accum[5]=pair (caller zeroes); accum_at: elsewhere
int launch_choose_labels(const phmrf_block* b, const uint8_t* saved, const double* acc_cur, const double* acc_sav, double beta);
bool energy_delta_available(const phmrf_block* b);
bool energy_diff_available(const phmrf_block* b);
int launch_energy_diff(const phmrf_block* b, const uint8_t* other);   // -> accum[4], [5] += E(labels) - E(other) (unary, pair)
int launch_energy_delta(const phmrf_block* b, double* accum_at = nullptr);   // -> accum[4], [5] (or accum_at[0], [1]) += the change since labels_eval / eval_tick
int launch_posterior_stats(const phmrf_block* b, float beta, int estimate_type, bool write_posteriors);
int launch_chain_colour(const phmrf_block* b, float beta, int family, int colour, int phase);
int launch_component_pass(phmrf_block* b, float beta);
int launch_component_prepare(phmrf_block* b);
int launch_grid_graph(const phmrf_block* b, int H, int W, int diagonal, int nn, double beta1);
int launch_propose(phmrf_block* b, float beta);  // best alternative label per node -> labels_tmp
int launch_strip_pass(const phmrf_block* b, float beta, int orient, int shift_r, int shift_c, int alpha,
                      int geom = -1);   // geom 0..2: one of the three fixed cuts (enables the memo of quiet strips)
int launch_strip_multi(const phmrf_block* b, float beta, int orient, int shift_r, int shift_c, unsigned long long label_mask,
                       int geom = -1);   // all listed alpha-expansions of the cut, one wave per strip
int launch_kmeans_step(const phmrf_block* b, const float* centers_dev, bool write_labels, double* acc_dev, bool outer = false);
int launch_fwd_weights(phmrf_block* b);                                             // ELL -> fwd_w (grid blocks)
int64_t coarse_nodes(const phmrf_block* b, int s, int off);                        // nodes of the coarse grid (s, off)
int launch_coarsen(const phmrf_block* b, phmrf_block* child, int s, int off, int alpha, float beta);
int launch_coarsen_batch(const phmrf_block* b, phmrf_block* const* children, const int* alphas, int nl, int s, int off, float beta,
                         const unsigned int* rebuild, int since, bool reset);
int launch_coarse_apply(const phmrf_block* b, const phmrf_block* child, int s, int off, int alpha, const unsigned long long* gate = nullptr,
                        unsigned int* moved_flag = nullptr, unsigned long long* changed_label = nullptr);
int launch_c2f_graph(const phmrf_block* b, phmrf_block* child, int Hc, int Wc, int s);     // c2f.hip
int launch_c2f_logprob(const phmrf_block* b, phmrf_block* child, int Wc, int s);
int launch_c2f_prolong(const phmrf_block* b, const phmrf_block* child, int Wc, int s);
int launch_graph_expansion(phmrf_block* b, float beta, int alpha);                  // maxflow.hip: one alpha-expansion of a general graph
int launch_unary_planes(phmrf_block* b);                                            // logprob -> uT
int64_t strip_scan_slots(const phmrf_block* b);                                     // capacity of scan_out (strip_scan_kernel)
// row tiles (tile.hip)
constexpr float PIN_COST = 1.0e9f;      // unary term of every label but its own at a pinned node (<< f32 max: sums of a strip's
                                        // or a component's terms stay finite; >> any real term: no move takes it)
int launch_tile_pins(phmrf_block* b, int region, int64_t first, int64_t count, int64_t pinned_first, int64_t pinned_count,
                     int64_t was_first, int64_t was_count, bool save_first);
int launch_put_labels(phmrf_block* b, int64_t first, int64_t count, const uint8_t* src_dev);

// ---- grid geometry for the kernels that find a node's neighbours by arithmetic (device code) ---------------------
// Diagonal blocks hold the upper triangle row-major: row i starts at start(i) = i W - i (i - 1) / 2 with column i.
__device__ __forceinline__ void grid_coords(int64_t v, int W, int diagonal, int* i, int* j) {
  if (!diagonal) {
    *i = (int)(v / W);
    *j = (int)(v - (int64_t)(*i) * W);
    return;
  }
  const double bq = 2.0 * W + 1.0;
  int r = (int)((bq - sqrt(bq * bq - 8.0 * (double)v)) * 0.5);     // the largest i with start(i) <= v, up to rounding
  r = r < 0 ? 0 : (r > W - 1 ? W - 1 : r);
  while (r > 0 && (int64_t)r * W - ((int64_t)r * (r - 1)) / 2 > v) --r;
  while (r + 1 < W && (int64_t)(r + 1) * W - ((int64_t)(r + 1) * r) / 2 <= v) ++r;
  *i = r;
  *j = r + (int)(v - ((int64_t)r * W - ((int64_t)r * (r - 1)) / 2));
}
// node of (i, j) = grid_row_base(i) + j
__device__ __forceinline__ int64_t grid_row_base(int i, int W, int diagonal) {
  return diagonal ? (int64_t)i * W - ((int64_t)i * (i - 1)) / 2 - i : (int64_t)i * W;
}
// f(c, w) for every grid neighbour c of node v = (i, j) with the weight w of the edge, in the order of the adjacency
// rows (ascending ids: NW N NE W E SW S SE).  The weights come from the forward-edge records (E, SW, S, SE of a node):
// the node's own for its forward edges, its backward neighbours' for the others.
template <class F>
__device__ __forceinline__ void grid_for_each_neighbour(int64_t v, int i, int j, int H, int W, int diagonal,
                                                        const float4* __restrict__ fwd_w, F&& f) {
  const int64_t up = grid_row_base(i - 1, W, diagonal), dn = grid_row_base(i + 1, W, diagonal);
  const int jlo_dn = diagonal ? i + 1 : 0;            // first column of the row below (the row above starts further left)
  const float4 own = fwd_w[v];
  if (i > 0) {
    if (j - 1 >= 0) { const int64_t c = up + j - 1; f(c, fwd_w[c].w); }          // NW holds the edge as its SE
    { const int64_t c = up + j; f(c, fwd_w[c].z); }                                // N: its S
    if (j + 1 < W) { const int64_t c = up + j + 1; f(c, fwd_w[c].y); }           // NE: its SW
  }
  if (j - 1 >= (diagonal ? i : 0)) { const int64_t c = v - 1; f(c, fwd_w[c].x); }  // W: its E
  if (j + 1 < W) f(v + 1, own.x);                                                   // E
  if (i + 1 < H) {
    if (j - 1 >= jlo_dn) f(dn + j - 1, own.y);                                      // SW
    if (j >= jlo_dn) f(dn + j, own.z);                                              // S
    if (j + 1 < W) f(dn + j + 1, own.w);                                            // SE
  }
}

// The same as arrays, every load issued before any use (absent neighbours: node v itself with weight 0): ids c[8] and
// weights w[8] in adjacency order.  For kernels whose work per neighbour is a dependent scatter -- with the callback
// form each neighbour's loads wait for the one before.
__device__ __forceinline__ void grid_gather_neighbours(int64_t v, int i, int j, int H, int W, int diagonal,
                                                       const float4* __restrict__ fwd_w, int64_t (&c)[8], float (&w)[8]) {
  const int64_t up = grid_row_base(i - 1, W, diagonal), dn = grid_row_base(i + 1, W, diagonal);
  const int jlo_dn = diagonal ? i + 1 : 0;
  const bool has[8] = {i > 0 && j - 1 >= 0, i > 0, i > 0 && j + 1 < W, j - 1 >= (diagonal ? i : 0), j + 1 < W,
                       i + 1 < H && j - 1 >= jlo_dn, i + 1 < H && j >= jlo_dn, i + 1 < H && j + 1 < W};
  const int64_t id[8] = {up + j - 1, up + j, up + j + 1, v - 1, v + 1, dn + j - 1, dn + j, dn + j + 1};
#pragma unroll
  for (int d = 0; d < 8; ++d) c[d] = has[d] ? id[d] : v;
  const float4 own = fwd_w[v];
  const float4 f0 = fwd_w[c[0]], f1 = fwd_w[c[1]], f2 = fwd_w[c[2]], f3 = fwd_w[c[3]];
  w[0] = has[0] ? f0.w : 0.f;      // NW holds the edge as its SE
  w[1] = has[1] ? f1.z : 0.f;      // N: its S
  w[2] = has[2] ? f2.y : 0.f;      // NE: its SW
  w[3] = has[3] ? f3.x : 0.f;      // W: its E
  w[4] = has[4] ? own.x : 0.f;
  w[5] = has[5] ? own.y : 0.f;
  w[6] = has[6] ? own.z : 0.f;
  w[7] = has[7] ? own.w : 0.f;
}

}  // namespace phmrf
