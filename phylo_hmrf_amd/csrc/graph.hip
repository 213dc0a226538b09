// On-device construction of a grid block's neighbour graph (reference: utility.py:1871-2053,
// edge_weightlist_grid3_undirected_unsym / _undirected, and w = exp(-beta1 * d), phylo_hmrf.py:585).
//
//   d_ij = |x_i - x_j|^2 / (|x_i| |x_j| + 1e-16)            utility.py:1935-1939 / :2028-2032
//   diagonal blocks: d halved between two nodes that both lie on the diagonal   utility.py:1942-1953
//   stencil: 8-neighbour = right / lower-right / lower / lower-left (+ their mirrors), restricted to the upper
//   triangle for diagonal blocks; anything else = right / lower                utility.py:1898-1905, :1917-1920
//
// One thread per node writes its ELL row (neighbour ids ascending, which is the row-major order of the eight
// offsets).  This removes the host edge list (E x 3 float64) and its PCIe upload from the E-step set-up.

#include "common.h"

namespace phmrf {
namespace {

__device__ __forceinline__ void node_coords(int64_t id, int W, int diagonal, int* i, int* j) {
  if (!diagonal) {
    *i = (int)(id / W);
    *j = (int)(id - (int64_t)(*i) * W);
    return;
  }
  // start(i) = i*W - i(i-1)/2 ; largest i with start(i) <= id
  const double b = 2.0 * W + 1.0;
  int r = (int)((b - sqrt(b * b - 8.0 * (double)id)) * 0.5);
  if (r < 0) r = 0;
  if (r > W - 1) r = W - 1;
  while (r > 0 && (int64_t)r * W - ((int64_t)r * (r - 1)) / 2 > id) --r;
  while (r + 1 < W && (int64_t)(r + 1) * W - ((int64_t)(r + 1) * r) / 2 <= id) ++r;
  *i = r;
  *j = r + (int)(id - ((int64_t)r * W - ((int64_t)r * (r - 1)) / 2));
}

__device__ __forceinline__ int64_t node_of(int i, int j, int H, int W, int diagonal) {
  if (i < 0 || i >= H || j < 0 || j >= W) return -1;
  if (diagonal) {
    if (i > j) return -1;
    return (int64_t)i * W - ((int64_t)i * (i - 1)) / 2 + (j - i);
  }
  return (int64_t)i * W + j;
}

template <int S>
__global__ void grid_graph_kernel(const float* __restrict__ X, int64_t n, int H, int W, int diagonal, int nn,
                                  double beta1, int32_t* __restrict__ nbr, float* __restrict__ wgt) {
  const int DI[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
  const int DJ[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    int i, j;
    node_coords(v, W, diagonal, &i, &j);
    double x[S], nx = 0.0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      x[s] = (double)X[v * S + s];
      nx += x[s] * x[s];
    }
    nx = sqrt(nx);
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (nn != 8 && DI[q] != 0 && DJ[q] != 0) continue;
      const int64_t u = node_of(i + DI[q], j + DJ[q], H, W, diagonal);
      if (u < 0) continue;
      double d = 0.0, ny = 0.0;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const double y = (double)X[u * S + s];
        d += (x[s] - y) * (x[s] - y);
        ny += y * y;
      }
      d = d / (nx * sqrt(ny) + 1e-16);
      if (diagonal && i == j && (i + DI[q]) == (j + DJ[q])) d *= 0.5;
      nbr[v * 8 + cnt] = (int32_t)u;
      wgt[v * 8 + cnt] = (float)exp(-beta1 * d);
      ++cnt;
    }
    for (; cnt < 8; ++cnt) {
      nbr[v * 8 + cnt] = -1;
      wgt[v * 8 + cnt] = 0.f;
    }
  }
}

// fwd_w[v] = weights of v's four FORWARD grid edges: x = E (i, j+1), y = SW (i+1, j-1), z = S (i+1, j), w = SE (i+1, j+1);
// 0 where the edge does not exist.  Every undirected grid edge is held exactly once, by its upper / left end, so the
// strip kernels read 16 B per cell instead of an ELL row of ids and weights.
__global__ void fwd_weights_kernel(int64_t n, int H, int W, int diagonal, int D, const int32_t* __restrict__ nbr,
                                   const float* __restrict__ wgt, float4* __restrict__ fwd_w) {
  const int FI[4] = {0, 1, 1, 1};
  const int FJ[4] = {1, -1, 0, 1};
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    int i, j;
    node_coords(v, W, diagonal, &i, &j);
    float out[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t u = node_of(i + FI[q], j + FJ[q], H, W, diagonal);
      if (u < 0) continue;
      for (int x = 0; x < D; ++x)
        if (nbr[v * D + x] == (int32_t)u) out[q] = wgt[v * D + x];
    }
    fwd_w[v] = make_float4(out[0], out[1], out[2], out[3]);
  }
}

}  // namespace

int launch_fwd_weights(phmrf_block* b) {
  if (!b->fwd_w) PHMRF_HIP(hipMalloc(reinterpret_cast<void**>(&b->fwd_w), (size_t)b->n * sizeof(float4)));
  int64_t g64 = (b->n + 255) / 256;
  const int grid = (int)(g64 > 256 * 16 ? 256 * 16 : g64);
  hipLaunchKernelGGL(fwd_weights_kernel, dim3(grid), dim3(256), 0, b->stream, b->n, b->H, b->W, b->diagonal, b->D, b->nbr,
                     b->wgt, b->fwd_w);
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

int launch_grid_graph(const phmrf_block* b, int H, int W, int diagonal, int nn, double beta1) {
  int64_t g64 = (b->n + 255) / 256;
  const int grid = (int)(g64 > 256 * 16 ? 256 * 16 : g64);
  switch (b->S) {
#define PHMRF_CASE(S_)                                                                                               \
  case S_:                                                                                                           \
    hipLaunchKernelGGL((grid_graph_kernel<S_>), dim3(grid), dim3(256), 0, b->stream, b->X, b->n, H, W, diagonal, nn, \
                       beta1, b->nbr, b->wgt);                                                                       \
    break;
    PHMRF_CASE(1) PHMRF_CASE(2) PHMRF_CASE(3) PHMRF_CASE(4) PHMRF_CASE(5) PHMRF_CASE(6) PHMRF_CASE(7) PHMRF_CASE(8)
    PHMRF_CASE(9) PHMRF_CASE(10) PHMRF_CASE(11) PHMRF_CASE(12) PHMRF_CASE(13) PHMRF_CASE(14) PHMRF_CASE(15) PHMRF_CASE(16)
#undef PHMRF_CASE
    default: return fail(PHMRF_ERR_UNSUPPORTED, "S must be in [1,16]");
  }
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

}  // namespace phmrf
