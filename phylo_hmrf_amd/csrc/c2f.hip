// Coarse-to-fine START of a cold solve (round 5).
//
// The reference's labelling call starts gco's swap from whatever labels it is handed (GCoptimization.cpp:1282-1305; the
// call site phylo_hmrf.py:496-498 hands it labels_local, the k-means labels in the first EM iteration).  The GPU solver's
// cold start -- argmax_k logprob, then rounds of fine moves with coarse alpha-expansions on top -- spends four fifths of its
// time getting the large-scale layout right.  A labelling that is CONSTANT on s x s super-cells has the energy of an
// ordinary Potts labelling of the coarse grid:
//
//     E(L) = sum_C [ sum_{i in C} u_i(L_C) ]  +  beta sum_{C ~ C'} [ sum_{(i,j) crossing C|C'} w_ij ] [L_C != L_C']
//
// (edges inside a super-cell join equal labels and cost nothing; every edge of the 8-neighbour stencil that leaves a
// super-cell ends in one of its eight neighbours).  So the coarse problem is a grid block of its own -- n / s^2 nodes, K
// labels, unary terms = sums of the cells' terms, pair weights = sums of the crossing weights -- and it is solved by the SAME
// solver (phmrf_mrf_solve on a child block, itself started coarse-to-fine while it is large), prolongated, and the fine
// solve starts from there.  Nothing but the START of the fine solve changes: every move the fine solve makes is still
// energy-non-increasing, and the parity tests against gco judge the result.
//
// Kernels: the coarse graph (once per block: the graph is constant over a fit), the coarse unary sums (every cold solve),
// the prolongation.  All sums run over a super-cell's cells in a fixed order: no atomics, the same bits every time.

#include "common.h"

namespace phmrf {
namespace {

// ELL row of every coarse node: its (up to) eight neighbours in adjacency order (NW N NE W E SW S SE), compacted, with the
// sum of the fine weights that cross between the two super-cells
__global__ __launch_bounds__(256) void c2f_graph_kernel(int64_t nc, int Hc, int Wc, int H, int W, int diagonal, int s,
                                                        const float4* __restrict__ fwd_w, int32_t* __restrict__ nbr_c,
                                                        float* __restrict__ wgt_c) {
  for (int64_t C = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; C < nc; C += (int64_t)gridDim.x * blockDim.x) {
    int I, J;
    grid_coords(C, Wc, diagonal, &I, &J);
    float w8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int di = 0; di < s; ++di) {
      const int i = I * s + di;
      if (i >= H) break;
      for (int dj = 0; dj < s; ++dj) {
        const int j = J * s + dj;
        if (j >= W || (diagonal && j < i)) continue;
        const int64_t v = grid_row_base(i, W, diagonal) + j;
        int64_t c[8];
        float w[8];
        grid_gather_neighbours(v, i, j, H, W, diagonal, fwd_w, c, w);
        constexpr int DI[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
        constexpr int DJ[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          // (an absent neighbour carries weight 0 and may point anywhere)
          const int i2 = i + DI[d], j2 = j + DJ[d];
          const int I2 = (i2 < 0 ? -1 : i2 / s), J2 = (j2 < 0 ? -1 : j2 / s);
          const int dI = I2 - I, dJ = J2 - J;
          if ((dI | dJ) == 0) continue;
          const int q = (dI + 1) * 3 + (dJ + 1);               // 0..8 without 4 -> adjacency order
          w8[q > 4 ? q - 1 : q] += w[d];
        }
      }
    }
    constexpr int DI[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
    constexpr int DJ[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int I2 = I + DI[q], J2 = J + DJ[q];
      if (I2 < 0 || I2 >= Hc || J2 < 0 || J2 >= Wc || (diagonal && I2 > J2)) continue;
      nbr_c[C * 8 + cnt] = (int32_t)(grid_row_base(I2, Wc, diagonal) + J2);
      wgt_c[C * 8 + cnt] = w8[q];
      ++cnt;
    }
    for (; cnt < 8; ++cnt) {
      nbr_c[C * 8 + cnt] = -1;
      wgt_c[C * 8 + cnt] = 0.f;
    }
  }
}

// logprob_c[C][k] = sum over the cells of super-cell C of logprob[i][k]: one thread per (C, k), the K threads of a
// super-cell read K consecutive floats of every cell
__global__ __launch_bounds__(256) void c2f_logprob_kernel(int64_t nc, int K, int Wc, int H, int W, int diagonal, int s,
                                                          const float* __restrict__ logprob, float* __restrict__ logprob_c) {
  const int64_t total = nc * K;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t C = t / K;
    const int k = (int)(t - C * K);
    int I, J;
    grid_coords(C, Wc, diagonal, &I, &J);
    float acc = 0.f;
    for (int di = 0; di < s; ++di) {
      const int i = I * s + di;
      if (i >= H) break;
      const int64_t rb = grid_row_base(i, W, diagonal);
      for (int dj = 0; dj < s; ++dj) {
        const int j = J * s + dj;
        if (j >= W || (diagonal && j < i)) continue;
        acc += logprob[(rb + j) * K + k];
      }
    }
    logprob_c[t] = acc;
  }
}

// labels[i] = labels_c[super-cell of i]
__global__ __launch_bounds__(256) void c2f_prolong_kernel(int64_t n, int Wc, int W, int diagonal, int s,
                                                          const uint8_t* __restrict__ labels_c, uint8_t* __restrict__ labels) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    int i, j;
    grid_coords(v, W, diagonal, &i, &j);
    labels[v] = labels_c[grid_row_base(i / s, Wc, diagonal) + j / s];
  }
}

int grid_of(int64_t items) {
  const int64_t g = (items + 255) / 256;
  return (int)(g > 256 * 32 ? 256 * 32 : (g < 1 ? 1 : g));
}

}  // namespace

// the child's adjacency rows from the parent's forward-edge records (the child's nbr / wgt must hold 8 entries per node)
int launch_c2f_graph(const phmrf_block* b, phmrf_block* child, int Hc, int Wc, int s) {
  hipLaunchKernelGGL(c2f_graph_kernel, dim3(grid_of(child->n)), dim3(256), 0, b->stream, child->n, Hc, Wc, b->H, b->W, b->diagonal,
                     s, b->fwd_w, child->nbr, child->wgt);
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

int launch_c2f_logprob(const phmrf_block* b, phmrf_block* child, int Wc, int s) {
  hipLaunchKernelGGL(c2f_logprob_kernel, dim3(grid_of(child->n * b->K)), dim3(256), 0, b->stream, child->n, b->K, Wc, b->H, b->W,
                     b->diagonal, s, b->logprob, child->logprob);
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

int launch_c2f_prolong(const phmrf_block* b, const phmrf_block* child, int Wc, int s) {
  hipLaunchKernelGGL(c2f_prolong_kernel, dim3(grid_of(b->n)), dim3(256), 0, b->stream, b->n, Wc, b->W, b->diagonal, s,
                     child->labels, b->labels);
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

}  // namespace phmrf
