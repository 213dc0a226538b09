// gfx950 kernels of the COARSE alpha-expansion: the 2-D block move of strip.hip applied to super-cells.
//
// Why: a strip is 5 grid rows tall.  A region that should change label as a whole but is taller than that in both
// directions (an entire 20 x 30 rectangle of one of two near-identical states that the current labelling has merged
// into its surroundings) cannot be reached by strips: every 5-row slice of it pays two long new boundaries for a thin
// gain, while a 10-row slab already wins.  gco's alpha-beta swap re-partitions such a region in one global cut
// (GCoptimization.cpp:1372-1394); the fine moves alone leave it standing (measured on the 2,001,000-node block of
// BASELINE configs[1]: +7e-4 of energy, all of it in ~100 such regions).
//
// What: tile the grid into s x s SUPER-CELLS (s = 2, 4, 8; shifted by an offset that changes from round to round).  For a
// label alpha every super-cell either keeps its labels or switches ALL its nodes to alpha.  That is again a binary
// pairwise problem on an 8-neighbour grid -- of the super-cells -- and because alpha-expansion tables are submodular
// for a Potts model it can be written with one switch cost D_A per super-cell and one non-negative weight lambda_AB per
// pair of adjacent super-cells:
//     dE(x) = sum_A D_A x_A + beta * sum_AB lambda_AB [x_A != x_B],            x_A in {keep, switch}
// Every fine edge (i, j) with table t00 = w[l_i != l_j], t01 = w[l_i != alpha] (j switches), t10 = w[alpha != l_j]
// (i switches), t11 = 0 contributes: inside one super-cell -t00 to its D; across two super-cells
// lambda = (t10 + t01 - t00) / 2 >= 0 to their pair, t10 - t00 - lambda to the D of i's super-cell and t01 - t00 - lambda
// to the D of j's.  The coarse problem IS a two-label Potts block (unary 0 / D_A, weights lambda), so it is handed to
// the SAME strip kernel (strip.hip, K = 2, alpha = 1, labels all 0) on a small child block: exact on 5 x 63 windows of
// super-cells = 10..20 rows x 126..252 columns of nodes.  Its energy never goes above 0 = "nobody switches", so the
// fine labelling's energy never goes up.  (Model: oracle/mrf_moves.coarse_problem / coarse_expansion.)

#include <algorithm>

#include "common.h"

namespace phmrf {
namespace {

struct CoarseGeom {
  int H, W, diagonal;    // fine grid
  int s, off;            // super-cell side; super-cell index of fine (i, j) = ((i + off) / s, (j + off) / s)
  int Hc, Wc;            // coarse grid
};

__device__ __forceinline__ int fine_node(const CoarseGeom& g, int i, int j) {
  if (i < 0 || i >= g.H || j < 0 || j >= g.W) return -1;
  if (g.diagonal) {
    if (i > j) return -1;
    return i * g.W - (i * (i - 1)) / 2 + (j - i);
  }
  return i * g.W + j;
}

__device__ __forceinline__ int coarse_node(const CoarseGeom& g, int I, int J) {
  if (g.diagonal) return I * g.Wc - (I * (I - 1)) / 2 + (J - I);
  return I * g.Wc + J;
}

__device__ __forceinline__ float comp4(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// D (switch cost, beta folded in) and the four forward coarse weights (E, SW, S, SE; without beta, like fwd_w) of every
// super-cell.  A wavefront covers 64 consecutive grid columns of the s grid rows of one coarse row: lane <-> column, so
// every load is a coalesced row segment (labels, forward weights and unary planes of the row itself and of the rows
// above / below), and the s x s nodes of a super-cell are s adjacent lanes x s loop trips: their sums meet in an
// s-lane butterfly.  A cross edge is seen from both of its super-cells: each adds its own unary share, the one the
// pair's slot belongs to adds lambda.  No atomics.
// NL labels per pass (round 3): the labels, the forward-edge records of the node and of its four backward neighbours and
// the neighbours' labels -- two thirds of the kernel's time, and the same for every label -- are read once for up to four
// labels; each label's problem goes to a child block of its own.  Per label the arithmetic is what the one-label pass
// does, in the same order, so a problem built in a batch and one rebuilt alone (`rebuild`: only when a move since the
// batch has changed the labelling, coarse_apply_kernel's flag) are the same numbers.
struct CoarseOut {
  float* c_uT[4];
  float4* c_fwd[4];
  uint8_t* c_labels[4];
  int alpha[4];      // (slots past nl repeat the first label: computed, not written)
  int nl;
  // the batch's pass also resets what its labels count with: each child's switch counter (the gate of its apply pass and
  // of strip_kernel's look) and the block's "a label of this batch has moved" flag -- five 8-byte memsets per batch were
  // five launches per batch, 9,000 of the 42,000 launches of a cold EM iteration of the whole-genome workload
  unsigned long long* c_counter[4];
  unsigned int* moved_flag;
};

// Loads first (round 4): the labels of the node and of its eight neighbours, its own forward-edge record, the ONE weight
// of each backward neighbour's record that belongs to the shared edge and the unary terms are all requested before the
// first of them is used -- an absent neighbour reads the node itself and gets weight 0, which adds exact zeros where the
// first version of this kernel skipped the edge behind a branch (and waited for one load after the other: 373 - 423 us
// per pass on the 12.4 M-node block for ~0.5 GB, a quarter of what HBM allows).  The sums are formed in the same order.
template <int SCALE, int NL>
__device__ __forceinline__ void coarsen_row(const CoarseGeom& g, int I, int64_t n, int64_t nc, const uint8_t* __restrict__ labels,
                                            const float* __restrict__ uT, const float4* __restrict__ fwd_w, const CoarseOut& out,
                                            float beta) {
  const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x) - g.off;      // (j + off) % SCALE == lane % SCALE
  const int J = (int)(blockIdx.x * blockDim.x + threadIdx.x) / SCALE;
  constexpr int FI[4] = {0, 1, 1, 1};
  constexpr int FJ[4] = {1, -1, 0, 1};
  const float* __restrict__ fwd_s = reinterpret_cast<const float*>(fwd_w);
  float D[NL], lam[NL][4];
#pragma unroll
  for (int q = 0; q < NL; ++q) {
    D[q] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) lam[q][e] = 0.f;
  }
  float wcross = 0.f;                       // total weight of the fine edges that leave the super-cell
#pragma unroll
  for (int di = 0; di < SCALE; ++di) {
    const int i = I * SCALE - g.off + di;
    const int node = fine_node(g, i, j);
    if (node < 0) continue;
    // ---- addresses, then every load of this node ----
    int fnn[4], bnn[4], fdI[4], fdJ[4], bdI[4], bdJ[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ni = i + FI[e], nj = j + FJ[e];
      fnn[e] = fine_node(g, ni, nj);
      // (row offsets of the neighbours' super-cells are known per unrolled trip: i + off = I * SCALE + di)
      fdI[e] = (di + FI[e]) / SCALE;
      fdJ[e] = (nj + g.off) / SCALE - J;
      const int mi = i - FI[e], mj = j - FJ[e];
      bdI[e] = (di - FI[e] < 0) ? -1 : 0;
      bdJ[e] = (mj + g.off) / SCALE - J;
      // (a backward neighbour inside this super-cell is counted from the holder, not here)
      bnn[e] = (mi < 0 || mj < 0 || (bdI[e] == 0 && bdJ[e] == 0)) ? -1 : fine_node(g, mi, mj);
    }
    const int li = labels[node];
    const float4 fw = fwd_w[node];
    int lf[4], lb[4];
    float wb[4], ua[NL];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lf[e] = labels[fnn[e] >= 0 ? fnn[e] : node];
      lb[e] = labels[bnn[e] >= 0 ? bnn[e] : node];
      wb[e] = fwd_s[4 * (int64_t)(bnn[e] >= 0 ? bnn[e] : node) + e];
    }
#pragma unroll
    for (int q = 0; q < NL; ++q) ua[q] = uT[(int64_t)out.alpha[q] * n + node];
    const float ucur = uT[(int64_t)li * n + node];
    // ---- the sums, in the order of the first version ----
#pragma unroll
    for (int q = 0; q < NL; ++q)
      if (li != out.alpha[q]) D[q] += ua[q] - ucur;
#pragma unroll
    for (int e = 0; e < 4; ++e) {           // edges this node holds
      const float w = fnn[e] >= 0 ? comp4(fw, e) : 0.f;
      const int lj = lf[e];
      const int dI = fdI[e], dJ = fdJ[e];
      const bool inside = dI == 0 && dJ == 0;
      if (!inside) wcross += w;
      // The pair table of label alpha, t00 = w [l_i != l_j], t01 = w [l_i != alpha], t10 = w [alpha != l_j], is the same for
      // every alpha that neither end carries: lambda = (w + w - t00) / 2 and the D term w - t00 - lambda are formed once per
      // edge; with alpha at this end both are exactly 0, with alpha at the other end lambda is 0 and the term is -w --
      // the values the per-label expressions give, bit for bit, for a third fewer instructions per label.
      const float t00 = (li != lj) ? w : 0.f;
      const float lvG = 0.5f * (w + w - t00);
      const float dG = w - t00 - lvG;
#pragma unroll
      for (int q = 0; q < NL; ++q) {
        const int alpha = out.alpha[q];
        if (inside) {
          D[q] -= beta * t00;
        } else {
          const bool mine = li == alpha, other = lj == alpha;
          const float lv = (mine || other) ? 0.f : lvG;
          const float dt = mine ? 0.f : (other ? -w : dG);
          D[q] += beta * dt;
          if (dI == 0 && dJ == 1) lam[q][0] += lv;
          else if (dI == 1) lam[q][2 + dJ] += lv;           // SW (dJ -1) -> 1, S -> 2, SE -> 3
          // (dI == 0, dJ == -1): the pair's slot is the E slot of the other super-cell, which adds it below
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {           // edges held by the four backward neighbours
      const float w = bnn[e] >= 0 ? wb[e] : 0.f;
      const int lj = lb[e];                           // holder = "i" of the table, this node = "j"
      const int dI = bdI[e], dJ = bdJ[e];
      wcross += w;
      const float t00 = (lj != li) ? w : 0.f;               // (as above; here t01 belongs to the holder, t10 to this node)
      const float lvG = 0.5f * (w + w - t00);
      const float dG = w - t00 - lvG;
#pragma unroll
      for (int q = 0; q < NL; ++q) {
        const int alpha = out.alpha[q];
        const bool mine = li == alpha, other = lj == alpha;
        const float lv = (mine || other) ? 0.f : lvG;
        const float dt = mine ? 0.f : (other ? -w : dG);
        D[q] += beta * dt;
        if (dI == 0 && dJ == 1) lam[q][0] += lv;            // the holder sits in my E neighbour (a fine SW edge)
      }
    }
  }
  // the SCALE columns of a super-cell are SCALE adjacent lanes (fixed summation tree: deterministic)
#pragma unroll
  for (int m = 1; m < SCALE; m <<= 1) {
    wcross += __shfl_xor(wcross, m, 64);
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      D[q] += __shfl_xor(D[q], m, 64);
#pragma unroll
      for (int e = 0; e < 4; ++e) lam[q][e] += __shfl_xor(lam[q][e], m, 64);
    }
  }
  if ((threadIdx.x & (SCALE - 1)) != 0 || J >= g.Wc || (g.diagonal && J < I)) return;
  const int c = coarse_node(g, I, J);
#pragma unroll
  for (int q = 0; q < NL; ++q) {
    if (q >= out.nl) continue;
    out.c_uT[q][c] = 0.f;
    // A super-cell whose switch cost exceeds everything its pairs could give back (lambda <= w on every cross edge) is in
    // no optimal switch set: taking it out of any set lowers the energy.  It is pinned (strip.hip: unary >= 1e29 = no
    // proposal), and a strip of pinned super-cells costs the strip kernel one look at the unary plane.
    out.c_uT[q][nc + c] = (D[q] > beta * wcross * 1.0001f + 1e-6f) ? 1.0e30f : D[q];
    out.c_fwd[q][c] = make_float4(lam[q][0], lam[q][1], lam[q][2], lam[q][3]);
    out.c_labels[q][c] = 0;
  }
}

// rebuild != nullptr: the problem of ONE label of a batch, built again after a label before it in the batch has moved
// (*rebuild, raised by coarse_apply_kernel) -- and then only where the labelling has changed: a wavefront whose nodes carry
// no change stamp later than `since` (the tick of the batch's pass; the stamps are dilated, so a change next to a super-cell
// marks nodes inside it) would write the numbers that are there already, and returns after reading its stamps
// (2 bytes per node instead of ~40).  The rows of the coarse grid are strided over the launch's gridDim.y, so that a pass
// that has nothing to do is a small launch.
template <int SCALE, int NL>
__global__ __launch_bounds__(256) void coarsen_kernel(CoarseGeom g, int64_t n, int64_t nc, const uint8_t* __restrict__ labels,
                                                      const float* __restrict__ uT, const float4* __restrict__ fwd_w,
                                                      CoarseOut out, float beta, const unsigned int* __restrict__ rebuild,
                                                      const uint16_t* __restrict__ stamp, int since) {
  if (rebuild && *rebuild == 0u) return;        // nothing has moved since the batch built this label's problem
  if (!rebuild && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    if (out.moved_flag) *out.moved_flag = 0u;
    for (int q = 0; q < out.nl; ++q)
      if (out.c_counter[q]) *out.c_counter[q] = 0ull;
  }
  for (int I = blockIdx.y; I < g.Hc; I += gridDim.y) {
    // (an upper-triangular block: the columns of this workgroup hold no node from this coarse row on)
    if (g.diagonal && I * SCALE > (int)(blockIdx.x * blockDim.x + blockDim.x - 1)) break;
    if (stamp) {
      const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x) - g.off;
      bool dirty = false;
#pragma unroll
      for (int di = 0; di < SCALE; ++di) {
        const int node = fine_node(g, I * SCALE - g.off + di, j);
        if (node >= 0) dirty = dirty || (int)stamp[node] > since;
      }
      if (!__ballot(dirty)) continue;           // (wave-uniform: the butterfly below stays inside the wave)
    }
    coarsen_row<SCALE, NL>(g, I, n, nc, labels, uT, fwd_w, out, beta);
  }
}

// One thread per fine node: take alpha where the node's super-cell switched.
__global__ __launch_bounds__(256) void coarse_apply_kernel(CoarseGeom g, const uint8_t* __restrict__ c_labels, int alpha,
                                                           uint8_t* __restrict__ labels, uint16_t* __restrict__ stamp, int tick,
                                                           const int32_t* __restrict__ nbr, int D,
                                                           unsigned long long* __restrict__ changed,
                                                           const unsigned long long* __restrict__ gate,
                                                           unsigned int* __restrict__ moved_flag,
                                                           unsigned long long* __restrict__ changed_label) {
  if (gate && *gate == 0ull) return;            // the child's passes switched no super-cell: nothing to take over
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  // (row groups strided over gridDim.y: the launch that finds the gate down -- most of them -- is a small one)
  for (int i = blockIdx.y * blockDim.y + threadIdx.y; i < g.H; i += gridDim.y * blockDim.y) {
    if (g.diagonal && i > (int)(blockIdx.x * blockDim.x + blockDim.x - 1)) break;     // (no node below the diagonal)
    bool moved = false;
    const int node = fine_node(g, i, j);
    if (node >= 0) {
      const int c = coarse_node(g, (i + g.off) / g.s, (j + g.off) / g.s);
      if (c_labels[c] && labels[node] != alpha) {
        labels[node] = (uint8_t)alpha;
        if (stamp) {
          stamp[node] = (uint16_t)tick;
          const int32_t* nb = nbr + (int64_t)node * D;
          for (int x = 0; x < D; ++x)
            if (nb[x] >= 0) stamp[nb[x]] = (uint16_t)tick;
        }
        moved = true;
      }
    }
    const unsigned long long m = __ballot(moved);
    if ((threadIdx.x & 63) == 0 && m) {
      atomicAdd(changed, (unsigned long long)__popcll(m));
      if (changed_label) atomicAdd(changed_label, (unsigned long long)__popcll(m));     // (per scale and label: the schedule rests labels)
      if (moved_flag) *moved_flag = 1u;         // the labelling has changed: later labels of the batch rebuild their problems
    }
  }
}

}  // namespace

static CoarseGeom make_coarse_geom(const phmrf_block* b, int s, int off) {
  CoarseGeom g;
  g.H = b->H;
  g.W = b->W;
  g.diagonal = b->diagonal;
  g.s = s;
  g.off = off;
  g.Hc = (b->H - 1 + off) / s + 1;
  g.Wc = (b->W - 1 + off) / s + 1;
  return g;
}

int64_t coarse_nodes(const phmrf_block* b, int s, int off) {
  const CoarseGeom g = make_coarse_geom(b, s, off);
  // (upper-triangular blocks: rows 0 .. Hc-1 of the Wc x Wc triangle -- Hc < Wc for a row tile of a diagonal block)
  return g.diagonal ? (int64_t)g.Hc * g.Wc - (int64_t)g.Hc * (g.Hc - 1) / 2 : (int64_t)g.Hc * g.Wc;
}

// Point the child blocks at the coarse grid of (s, off) and fill their unary planes / forward weights / labels for up to
// four labels in one pass (alphas[q] < 0: unused).  rebuild != nullptr: the pass runs only if *rebuild != 0 (device).
int launch_coarsen_batch(const phmrf_block* b, phmrf_block* const* children, const int* alphas, int nl, int s, int off, float beta,
                         const unsigned int* rebuild, int since, bool reset) {
  const CoarseGeom g = make_coarse_geom(b, s, off);
  CoarseOut out;
  for (int q = 0; q < 4; ++q) {
    phmrf_block* child = children[q < nl ? q : 0];
    if (q < nl) {
      child->H = g.Hc;
      child->W = g.Wc;
      child->diagonal = g.diagonal;
      child->n = coarse_nodes(b, s, off);
    }
    out.c_uT[q] = child->uT;
    out.c_fwd[q] = child->fwd_w;
    out.c_labels[q] = child->labels;
    out.alpha[q] = q < nl ? alphas[q] : alphas[0];
    out.c_counter[q] = (q < nl && reset) ? child->counters : nullptr;
  }
  out.moved_flag = reset ? b->coarse_flag : nullptr;
  out.nl = nl;
  const int64_t nc = children[0]->n;
  // a rebuild (one label, gated on the device) strides the coarse rows over <= ~2048 workgroups; `since` >= 0 with change
  // stamps at hand (inside a solve): only the wavefronts whose nodes changed after that tick do the work
  const int gx = (g.Wc * s + 255) / 256;
  const int gy = rebuild ? std::max(1, std::min(g.Hc, 2048 / gx)) : g.Hc;
  const dim3 blk(256), grd(gx, gy);
  const uint16_t* stamp = (rebuild && since >= 0 && b->tick && b->stamp) ? b->stamp : nullptr;
#define PHMRF_LAUNCH_COARSEN(S_, NL_)                                                                                     \
  hipLaunchKernelGGL((coarsen_kernel<S_, NL_>), grd, blk, 0, b->stream, g, b->n, nc, b->labels, b->uT, b->fwd_w, out, beta, \
                     rebuild, stamp, since)
#define PHMRF_LAUNCH_COARSEN_S(S_)                                                                                        \
  {                                                                                                                       \
    if (nl == 1) PHMRF_LAUNCH_COARSEN(S_, 1);                                                                             \
    else if (nl == 2) PHMRF_LAUNCH_COARSEN(S_, 2);                                                                        \
    else PHMRF_LAUNCH_COARSEN(S_, 4);                                                                                     \
  }
  if (s == 2) PHMRF_LAUNCH_COARSEN_S(2)
  else if (s == 4) PHMRF_LAUNCH_COARSEN_S(4)
  else PHMRF_LAUNCH_COARSEN_S(8)
#undef PHMRF_LAUNCH_COARSEN_S
#undef PHMRF_LAUNCH_COARSEN
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

int launch_coarsen(const phmrf_block* b, phmrf_block* child, int s, int off, int alpha, float beta) {
  phmrf_block* one[1] = {child};
  return launch_coarsen_batch(b, one, &alpha, 1, s, off, beta, nullptr, -1, false);
}

int launch_coarse_apply(const phmrf_block* b, const phmrf_block* child, int s, int off, int alpha, const unsigned long long* gate,
                        unsigned int* moved_flag, unsigned long long* changed_label) {
  const CoarseGeom g = make_coarse_geom(b, s, off);
  const int gx = (g.W + 63) / 64;
  const dim3 blk(64, 4), grd(gx, std::max(1, std::min((g.H + 3) / 4, (gate ? 2048 : (1 << 20)) / gx)));
  hipLaunchKernelGGL(coarse_apply_kernel, grd, blk, 0, b->stream, g, child->labels, alpha, b->labels,
                     b->tick ? b->stamp : nullptr, b->tick, b->nbr, b->D, b->counters + b->counter_slot, gate, moved_flag, changed_label);
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

}  // namespace phmrf
