// Row tiles of one grid block on several GPUs (phmrf_block_set_tile, phmrf_block_tile_pins, phmrf_block_put_label_rows).
//
// The reference's only way to make a block smaller is the centromere split into independent pieces (utility.py:381-393),
// each run as its own process (base.py:357-362).  Here a block that is larger than a GPU's share of the nodes is cut into
// ROW TILES: tile t stores the grid rows it owns plus one HALO row per neighbouring tile (a copy of that tile's boundary
// row).  All rows of a tile are ordinary nodes of an ordinary block -- same kernels, same geometry (rows r0 .. r1 of an
// upper-triangular block are themselves the first rows of a smaller upper triangle) -- and what keeps the tiles' moves from
// colliding is decided per round, by the unary terms alone:
//
//   * at the cut between tile T (above) and tile T+1 (below), with c the last row T owns, the rows c and c+1 are PINNED in
//     turn: rounds of even parity pin row c (T moves rows < c, T+1 moves rows > c), rounds of odd parity pin row c+1 (T moves
//     rows <= c, T+1 moves rows > c+1).  Whatever the two tiles move in the same round shares no edge, so every round lowers
//     the energy of the whole block by the sum of what the tiles gained -- as strips of one pass do inside a block.
//   * a pinned node keeps its label because every other label costs PIN_COST: each move type (strip DP and its filter,
//     fusion, components, coarse super-cells, chains, ICM) reads the unary terms and turns the move down by itself; no
//     kernel knows about tiles.  The real rows are kept in pin_save and put back when the pin is lifted.
//   * after a round T sends row c and receives row c+1 (and T+1 the other way round): <= 5 KB per cut and round at 50 kb.
//
// Energies, costs and statistics are counted over the rows a tile OWNS (phmrf_block::own_*): their sums over the tiles are
// the whole block's.

#include "common.h"

namespace phmrf {
namespace {

// One thread per (node of the region, label).  region = the two rows at a cut; [pin0, pin0 + pinc) the nodes pinned from
// now on; [was0, was0 + wasc) those that were pinned until now (their labels must not have moved: checked).
__global__ __launch_bounds__(256) void tile_pins_kernel(float* __restrict__ logprob, float* __restrict__ uT, int64_t n, int K,
                                                        const uint8_t* __restrict__ labels, float* __restrict__ save,
                                                        uint8_t* __restrict__ pin_label, int64_t first, int64_t count,
                                                        int64_t pin0, int64_t pinc, int64_t was0, int64_t wasc, int save_first,
                                                        uint16_t* __restrict__ stamp, int tick,
                                                        unsigned long long* __restrict__ violations) {
  const int64_t total = count * K;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = t / K;
    const int k = (int)(t - q * K);
    const int64_t v = first + q;
    float real;
    if (save_first) {
      real = logprob[v * K + k];
      save[q * K + k] = real;
    } else {
      real = save[q * K + k];
    }
    const int c = labels[v];
    const bool pinned = v >= pin0 && v < pin0 + pinc;
    const float val = (pinned && k != c) ? -PIN_COST : real;
    logprob[v * K + k] = val;
    if (uT) uT[(int64_t)k * n + v] = -val;
    if (k == 0) {
      if (!save_first && v >= was0 && v < was0 + wasc && pin_label[q] != (uint8_t)c) atomicAdd(violations, 1ull);
      pin_label[q] = (uint8_t)c;
      if (stamp) stamp[v] = (uint16_t)tick;
    }
  }
}

// labels[first .. first + count) <- src; a node whose label changes is stamped with its neighbours (the stamps are dilated)
// and the energy snapshot follows, so that the change counts as an input of the next round and not as one of its moves.
__global__ __launch_bounds__(256) void put_labels_kernel(uint8_t* __restrict__ labels, uint8_t* __restrict__ labels_eval,
                                                         const uint8_t* __restrict__ src, int64_t first, int64_t count, int H, int W,
                                                         int diagonal, uint16_t* __restrict__ stamp, int tick,
                                                         uint8_t* __restrict__ pin_label0, int64_t pf0, int64_t pc0,
                                                         uint8_t* __restrict__ pin_label1, int64_t pf1, int64_t pc1) {
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = first + q;
    const uint8_t l = src[q];
    if (labels[v] == l) continue;
    labels[v] = l;
    if (labels_eval) labels_eval[v] = l;
    if (pin_label0 && v >= pf0 && v < pf0 + pc0) pin_label0[v - pf0] = l;
    if (pin_label1 && v >= pf1 && v < pf1 + pc1) pin_label1[v - pf1] = l;
    if (stamp) {
      int i, j;
      grid_coords(v, W, diagonal, &i, &j);
      const int64_t up = grid_row_base(i - 1, W, diagonal), dn = grid_row_base(i + 1, W, diagonal);
      const int jlo_dn = diagonal ? i + 1 : 0;
      stamp[v] = (uint16_t)tick;
      if (i > 0) {
        if (j - 1 >= 0) stamp[up + j - 1] = (uint16_t)tick;
        stamp[up + j] = (uint16_t)tick;
        if (j + 1 < W) stamp[up + j + 1] = (uint16_t)tick;
      }
      if (j - 1 >= (diagonal ? i : 0)) stamp[v - 1] = (uint16_t)tick;
      if (j + 1 < W) stamp[v + 1] = (uint16_t)tick;
      if (i + 1 < H) {
        if (j - 1 >= jlo_dn) stamp[dn + j - 1] = (uint16_t)tick;
        if (j >= jlo_dn) stamp[dn + j] = (uint16_t)tick;
        if (j + 1 < W) stamp[dn + j + 1] = (uint16_t)tick;
      }
    }
  }
}

}  // namespace

// region 0: the two top rows, 1: the two bottom rows.  [pinned_first, +pinned_count) is pinned from now on (a sub-range of the
// region, possibly empty), the rest of the region gets its real terms back.
int launch_tile_pins(phmrf_block* b, int region, int64_t first, int64_t count, int64_t pinned_first, int64_t pinned_count,
                     int64_t was_first, int64_t was_count, bool save_first) {
  if (count <= 0) return PHMRF_OK;
  const int64_t total = count * b->K;
  int64_t g64 = (total + 255) / 256;
  const int grid = (int)(g64 > 4096 ? 4096 : g64);
  hipLaunchKernelGGL(tile_pins_kernel, dim3(grid), dim3(256), 0, b->stream, b->logprob, (b->uT && b->uT_valid) ? b->uT : nullptr, b->n,
                     b->K, b->labels, b->pin_save[region], b->pin_label[region], first, count, pinned_first, pinned_count, was_first,
                     was_count, save_first ? 1 : 0, b->tick ? b->stamp : nullptr, b->tick,
                     // (the violation count lives in accum slot 6: solve_begin zeroes it, every round's read-back carries
                     //  it -- energy_round_launch --, solve_round_collect tests it; the counter bank would not do: it is
                     //  zeroed at the start of every round, after the pins of that round have been set)
                     reinterpret_cast<unsigned long long*>(b->accum + 6));
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

int launch_put_labels(phmrf_block* b, int64_t first, int64_t count, const uint8_t* src_dev) {
  if (count <= 0) return PHMRF_OK;
  int64_t g64 = (count + 255) / 256;
  const int grid = (int)(g64 > 4096 ? 4096 : g64);
  hipLaunchKernelGGL(put_labels_kernel, dim3(grid), dim3(256), 0, b->stream, b->labels,
                     (b->tick && b->eval_tick >= 0) ? b->labels_eval : nullptr, src_dev, first, count, b->H, b->W, b->diagonal,
                     b->tick ? b->stamp : nullptr, b->tick, b->pin_label[0], b->pin_first[0], b->pin_count[0], b->pin_label[1],
                     b->pin_first[1], b->pin_count[1]);
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

}  // namespace phmrf
