// gfx950 kernels of the E-step: emission, argmax init, colour-ordered ICM, energy, posterior + statistics.
//
// Common shape ("node tile"): a workgroup of TB threads owns TB nodes.  The K-wide row of each node
// (logprob[i, 0:K]) is brought into LDS with coalesced 16-byte loads (K/4 lanes per row), laid out
// with an ODD row stride Kp so that "lane r reads word k of row r" is bank-conflict free; afterwards
// each lane works on its own node with plain dynamic LDS indexing (neighbour-label histogram,
// argmin / softmax over k).  All of these kernels are HBM-bound (<= 20 flop/B); no MFMA.

#include <cstdlib>

#include "common.h"

namespace phmrf {

namespace {

constexpr int ACC_COST = 0;    // accum[0..3]  cost numerators
constexpr int ACC_ENERGY = 4;  // accum[4..5]  unary, pair
constexpr int ACC_STATS = 8;   // accum[8..]   post | obs | obs*obs.T

// The two energy sums of a workgroup.  Deterministic mode (PHMRF_DETERMINISTIC=1): as 2^-20 fixed-point integers in the
// same 8-byte slots -- integer atomics commute, so the round-by-round energies the solver decides on are the same in
// every run (the host converts back, api.hip energy_now).
constexpr double ENERGY_FIX = 1048576.0;
__device__ __forceinline__ void energy_flush(double* accum, double tu, double tp, int det) {
  if (det) {
    atomicAdd(reinterpret_cast<unsigned long long*>(accum + ACC_ENERGY), (unsigned long long)__double2ll_rn(tu * ENERGY_FIX));
    atomicAdd(reinterpret_cast<unsigned long long*>(accum + ACC_ENERGY + 1), (unsigned long long)__double2ll_rn(tp * ENERGY_FIX));
  } else {
    atomicAdd(accum + ACC_ENERGY, tu);
    atomicAdd(accum + ACC_ENERGY + 1, tp);
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum over the workgroup, result valid in thread 0.  red: LDS scratch of >= 4 doubles.
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) t += red[i];
  }
  return t;
}

// Cooperative load of `rows` K-wide rows into the LDS tile: tile[r*Kp + c] = scale * src_row(r)[c] + add*tile.
// VEC floats per lane per access (VEC | K): consecutive lanes read consecutive 4*VEC-byte pieces of a row.
template <int VEC, bool ACCUM>
__device__ __forceinline__ void rows_to_tile(const float* __restrict__ mat, const int32_t* __restrict__ nodes,
                                             int64_t base, int rows, int K, int Kp, float* tile, float scale,
                                             float tile_scale) {
  const int KV = K / VEC;
  const int total = rows * KV;
  auto ld = [&](int q, float (&v)[VEC]) {
    const int r = q / KV;
    const int c = (q - r * KV) * VEC;
    const int64_t node = nodes ? (int64_t)nodes[base + r] : base + r;
    const float* src = mat + node * K + c;
    if (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4*>(src);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else if (VEC == 2) {
      const float2 t = *reinterpret_cast<const float2*>(src);
      v[0] = t.x; v[1] = t.y;
    } else {
      v[0] = src[0];
    }
  };
  auto st = [&](int q, const float (&v)[VEC]) {
    const int r = q / KV;
    const int c = (q - r * KV) * VEC;
    float* dst = tile + r * Kp + c;
#pragma unroll
    for (int j = 0; j < VEC; ++j) dst[j] = ACCUM ? fmaf(tile_scale, dst[j], scale * v[j]) : scale * v[j];
  };
  // four pieces per lane and trip, their loads side by side (one load -> one LDS store per trip is a memory round trip
  // per piece and lane)
  const int step = blockDim.x;
  int q = threadIdx.x;
  for (; q + 3 * step < total; q += 4 * step) {
    float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
    ld(q, v0); ld(q + step, v1); ld(q + 2 * step, v2); ld(q + 3 * step, v3);
    st(q, v0); st(q + step, v1); st(q + 2 * step, v2); st(q + 3 * step, v3);
  }
  for (; q < total; q += step) {
    float v[VEC];
    ld(q, v);
    st(q, v);
  }
}

// Cooperative store of a contiguous [rows, K] chunk from the LDS tile.
template <int VEC>
__device__ __forceinline__ void tile_to_rows(float* __restrict__ out, int rows, int K, int Kp, const float* tile) {
  const int KV = K / VEC;
  const int total = rows * KV;
  for (int q = threadIdx.x; q < total; q += blockDim.x) {
    const int r = q / KV;
    const int c = (q - r * KV) * VEC;
    const float* s = tile + r * Kp + c;
    float* d = out + (int64_t)r * K + c;
    if (VEC == 4) {
      *reinterpret_cast<float4*>(d) = make_float4(s[0], s[1], s[2], s[3]);
    } else if (VEC == 2) {
      *reinterpret_cast<float2*>(d) = make_float2(s[0], s[1]);
    } else {
      d[0] = s[0];
    }
  }
}

inline int vec_of(int K) { return (K % 4 == 0) ? 4 : (K % 2 == 0 ? 2 : 1); }

inline int grid_for(int64_t work_items, int tb, int max_blocks = 256 * 16) {
  int64_t g = (work_items + tb - 1) / tb;
  if (g > max_blocks) g = max_blocks;
  if (g < 1) g = 1;
  return (int)g;
}

// -------------------------------------------------------------------------------------------------
// b1 emission: logprob[i,k] = c_k - 0.5 * || Linv_k (x_i - mu_k) ||^2
// (phylo_hmrf.py:266-268; sklearn-0.18 log_multivariate_normal_density 'full': Cholesky + triangular solve.
//  The solve is replaced by a product with the host-inverted factor, S(S+1)/2 FMAs per state.)
// packed per state: mu[S] | Linv lower triangle row-major [S(S+1)/2] | c
// -------------------------------------------------------------------------------------------------
template <int S, int VEC>
__global__ __launch_bounds__(256) void emission_kernel(const float* __restrict__ X, int64_t n, int K, int Kp,
                                                       const float* __restrict__ P, float* __restrict__ out,
                                                       float* __restrict__ uT) {
  extern __shared__ float tile[];
  constexpr int PS = S + S * (S + 1) / 2 + 1;
  const int TB = blockDim.x;
  for (int64_t base = (int64_t)blockIdx.x * TB; base < n; base += (int64_t)gridDim.x * TB) {
    const int64_t i = base + threadIdx.x;
    if (i < n) {
      float x[S];
      if (S % 4 == 0) {
#pragma unroll
        for (int s = 0; s < S; s += 4) {
          const float4 t = *reinterpret_cast<const float4*>(X + i * S + s);
          x[s] = t.x; x[s + 1] = t.y; x[s + 2] = t.z; x[s + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int s = 0; s < S; ++s) x[s] = X[i * S + s];
      }
      float* row = tile + threadIdx.x * Kp;
      for (int k = 0; k < K; ++k) {
        const float* p = P + k * PS;  // wave-uniform address: scalar loads
        float d[S];
#pragma unroll
        for (int s = 0; s < S; ++s) d[s] = x[s] - p[s];
        float q = 0.f;
        int idx = S;
#pragma unroll
        for (int r = 0; r < S; ++r) {
          float y = 0.f;
#pragma unroll
          for (int c = 0; c <= r; ++c) y = fmaf(p[idx++], d[c], y);
          q = fmaf(y, y, q);
        }
        const float lp = fmaf(-0.5f, q, p[idx]);
        row[k] = lp;
        if (uT) uT[(int64_t)k * n + i] = -lp;        // the label-major unary planes of the strip moves, while the value is at hand
      }
    }
    __syncthreads();
    const int64_t rem = n - base;
    const int rows = rem < TB ? (int)rem : TB;
    tile_to_rows<VEC>(out + base * K, rows, K, Kp, tile);
    __syncthreads();
  }
}

template <int S>
int launch_emission_s(const float* X, int64_t n, int K, const float* packed, float* logprob, float* uT, hipStream_t st) {
  const int TB = tile_threads(K), Kp = padded_k(K);
  const size_t lds = (size_t)TB * Kp * sizeof(float);
  const int grid = grid_for(n, TB);
  switch (vec_of(K)) {
    case 4: hipLaunchKernelGGL((emission_kernel<S, 4>), dim3(grid), dim3(TB), lds, st, X, n, K, Kp, packed, logprob, uT); break;
    case 2: hipLaunchKernelGGL((emission_kernel<S, 2>), dim3(grid), dim3(TB), lds, st, X, n, K, Kp, packed, logprob, uT); break;
    default: hipLaunchKernelGGL((emission_kernel<S, 1>), dim3(grid), dim3(TB), lds, st, X, n, K, Kp, packed, logprob, uT); break;
  }
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

// -------------------------------------------------------------------------------------------------
// argmax_k logprob -> labels (init_mode 1)
// -------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logprob, int64_t n, int K, int Kp,
                                                     uint8_t* __restrict__ labels) {
  extern __shared__ float tile[];
  const int TB = blockDim.x;
  for (int64_t base = (int64_t)blockIdx.x * TB; base < n; base += (int64_t)gridDim.x * TB) {
    const int64_t rem = n - base;
    const int rows = rem < TB ? (int)rem : TB;
    rows_to_tile<VEC, false>(logprob, nullptr, base, rows, K, Kp, tile, 1.f, 0.f);
    __syncthreads();
    if ((int)threadIdx.x < rows) {
      const float* row = tile + threadIdx.x * Kp;
      float best = row[0];
      int bk = 0;
      for (int k = 1; k < K; ++k) {
        const float v = row[k];
        if (v > best) { best = v; bk = k; }
      }
      labels[base + threadIdx.x] = (uint8_t)bk;
    }
    __syncthreads();
  }
}

// -------------------------------------------------------------------------------------------------
// ICM, one colour class: every node of the class minimises
//     cost_k = -logprob[i,k] - beta * sum_{j in N(i), l_j == k} w_ij      (+ const)
// over k, neighbours fixed (no two nodes of a class are adjacent).  Strict improvement only; lowest
// label wins ties (same rule as oracle/mrf_moves.icm_sweep).
// -------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void icm_kernel(const float* __restrict__ logprob, const int32_t* __restrict__ nodes,
                                                  int64_t count, int K, int Kp, int D,
                                                  const int32_t* __restrict__ nbr, const float* __restrict__ wgt,
                                                  uint8_t* __restrict__ labels, float beta,
                                                  unsigned long long* __restrict__ changed,
                                                  uint16_t* __restrict__ stamp, int tick) {
  extern __shared__ float tile[];
  const int TB = blockDim.x;
  unsigned int my_changed = 0;
  for (int64_t base = (int64_t)blockIdx.x * TB; base < count; base += (int64_t)gridDim.x * TB) {
    const int64_t rem = count - base;
    const int rows = rem < TB ? (int)rem : TB;
    rows_to_tile<VEC, false>(logprob, nodes, base, rows, K, Kp, tile, -1.f, 0.f);
    __syncthreads();
    if ((int)threadIdx.x < rows) {
      const int64_t node = nodes ? (int64_t)nodes[base + threadIdx.x] : base + threadIdx.x;
      float* row = tile + threadIdx.x * Kp;
      const int32_t* nb = nbr + node * D;
      const float* wg = wgt + node * D;
      for (int j = 0; j < D; j += 4) {
        const int4 c = *reinterpret_cast<const int4*>(nb + j);
        const float4 w = *reinterpret_cast<const float4*>(wg + j);
        if (c.x >= 0) row[labels[c.x]] -= beta * w.x;
        if (c.y >= 0) row[labels[c.y]] -= beta * w.y;
        if (c.z >= 0) row[labels[c.z]] -= beta * w.z;
        if (c.w >= 0) row[labels[c.w]] -= beta * w.w;
      }
      const int cur = labels[node];
      float best = row[cur];
      int bk = cur;
      for (int k = 0; k < K; ++k) {
        const float v = row[k];
        if (v < best) { best = v; bk = k; }
      }
      if (bk != cur) {
        labels[node] = (uint8_t)bk;
        if (stamp) {          // dilated change stamp: the node and every neighbour see new inputs from this tick on
          stamp[node] = (uint16_t)tick;
          for (int j = 0; j < D; ++j)
            if (nb[j] >= 0) stamp[nb[j]] = (uint16_t)tick;
        }
        ++my_changed;
      }
    }
    __syncthreads();
  }
  // one atomic per wave
  unsigned int s = my_changed;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(changed, (unsigned long long)s);
}

// -------------------------------------------------------------------------------------------------
// a-E energy: unary = sum_i -logprob[i,l_i];  pair = 0.5 * sum_i sum_{j in N(i)} w_ij [l_i != l_j]
// (beta applied by the host).  f64 accumulation.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void energy_kernel(const float* __restrict__ logprob, int64_t n, int K, int D,
                                                     const int32_t* __restrict__ nbr, const float* __restrict__ wgt,
                                                     const uint8_t* __restrict__ labels, double* __restrict__ accum, int det) {
  __shared__ double red[8];
  double eu = 0.0, ep = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = labels[i];
    eu -= (double)logprob[i * K + l];
    const int32_t* nb = nbr + i * D;
    const float* wg = wgt + i * D;
    float s = 0.f;
    for (int j = 0; j < D; j += 4) {
      const int4 c = *reinterpret_cast<const int4*>(nb + j);
      const float4 w = *reinterpret_cast<const float4*>(wg + j);
      if (c.x >= 0 && labels[c.x] != l) s += w.x;
      if (c.y >= 0 && labels[c.y] != l) s += w.y;
      if (c.z >= 0 && labels[c.z] != l) s += w.z;
      if (c.w >= 0 && labels[c.w] != l) s += w.w;
    }
    ep += (double)s;
  }
  const double tu = block_sum(eu, red);
  const double tp = block_sum(ep, red);
  if (threadIdx.x == 0) energy_flush(accum, tu, 0.5 * tp, det);
}

// The same energy on a grid block, from the forward-edge records the strip kernels use (fwd_w: E, SW, S, SE -- every
// undirected edge once, held by its upper / left end) and four neighbour labels found by geometry: 21 B per node where
// the adjacency form reads 64 B of ids and weights and gathers eight labels.  One thread per cell of the H x W square
// (upper-triangular blocks: the cells below the diagonal idle).
__global__ __launch_bounds__(256) void energy_grid_kernel(const float* __restrict__ logprob, const float* __restrict__ uT,
                                                          int64_t n, int K, int H, int W,
                                                          int diagonal, const float4* __restrict__ fwd_w,
                                                          const uint8_t* __restrict__ labels, double* __restrict__ accum, int det,
                                                          int row0, int row1) {
  // (rows [row0, row1): a row tile counts the rows it owns -- unary terms and forward edges; its halo rows below supply labels)
  __shared__ double red[8];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  double eu = 0.0, ep = 0.0;
  // four rows of the thread's column per trip, their loads issued side by side: the chain label -> unary term -> neighbour
  // labels is latency, not bandwidth, and one row per trip left the memory system idle most of the time.  The unary term
  // comes from the label-major planes when they are current (neighbouring nodes mostly share a label, so the reads
  // coalesce, where the node-major rows cost a 64-byte sector per node).
  constexpr int UR = 4;
  const int stride = gridDim.y * 4;
  // (an upper-triangular block: the 64 columns of this workgroup hold no node below row 64 blockIdx.x + 63 -- until round 5
  //  every workgroup walked all H rows, half of the trips over nothing)
  const int row_end = (diagonal && blockIdx.x * 64 + 64 < row1) ? blockIdx.x * 64 + 64 : row1;
  for (int i0 = row0 + blockIdx.y * 4 + (threadIdx.x >> 6); i0 < row_end; i0 += stride * UR) {
    int64_t node[UR];
    int lab[UR];
    bool on[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int i = i0 + u * stride;
      on[u] = i < row1 && j < W && !(diagonal && j < i);
      const int64_t row = diagonal ? (int64_t)i * W - ((int64_t)i * (i - 1)) / 2 - i : (int64_t)i * W;      // node = row + j
      node[u] = on[u] ? row + j : 0;
      lab[u] = labels[node[u]];
    }
    float un[UR];
    float4 w[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      un[u] = uT ? uT[(int64_t)lab[u] * n + node[u]] : -logprob[node[u] * K + lab[u]];
      w[u] = fwd_w[node[u]];
    }
    // the four forward neighbours' labels of all rows, side by side (an absent neighbour reads the node itself: equal
    // label, no contribution)
    int ln[UR][4];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int i = i0 + u * stride;
      const int64_t row2 = diagonal ? (int64_t)(i + 1) * W - ((int64_t)(i + 1) * i) / 2 - (i + 1) : (int64_t)(i + 1) * W;
      const int jlo = diagonal ? i + 1 : 0;
      const bool below = on[u] && i + 1 < H;
      const int64_t me = node[u];
      ln[u][0] = labels[(on[u] && j + 1 < W) ? me + 1 : me];                             // E
      ln[u][1] = labels[(below && j - 1 >= jlo) ? row2 + j - 1 : me];                      // SW
      ln[u][2] = labels[(below && j >= jlo) ? row2 + j : me];                              // S
      ln[u][3] = labels[(below && j + 1 < W) ? row2 + j + 1 : me];                         // SE
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      if (!on[u]) continue;
      const int l = lab[u];
      eu += (double)un[u];
      float s = 0.f;
      if (ln[u][0] != l) s += w[u].x;
      if (ln[u][1] != l) s += w[u].y;
      if (ln[u][2] != l) s += w[u].z;
      if (ln[u][3] != l) s += w[u].w;
      ep += (double)s;
    }
  }
  const double tu = block_sum(eu, red);
  const double tp = block_sum(ep, red);
  if (threadIdx.x == 0) energy_flush(accum, tu, tp, det);
}

// The CHANGE of that energy since the last evaluation, from the nodes the moves have touched since: a node whose change
// stamp is newer than `since` has changed its label or is the neighbour of one that has (the stamps are dilated by one
// ring), and an edge's term can only differ where an end has changed -- so the touched nodes' unary terms and forward
// edges, new labelling minus the snapshot `prev` taken at the last evaluation, are the whole difference, term by term exact
// in f32 and summed in f64.  Waves whose nodes are all untouched read two bytes per node; a mop-up round of a warm solve
// touches a few per cent of the block.  Same thread layout as energy_grid_kernel.
__global__ __launch_bounds__(256) void energy_delta_grid_kernel(const float* __restrict__ uT, int64_t n, int H, int W, int diagonal,
                                                                const float4* __restrict__ fwd_w,
                                                                const uint8_t* __restrict__ labels,
                                                                const uint8_t* __restrict__ prev,
                                                                const uint16_t* __restrict__ stamp, int since,
                                                                double* __restrict__ accum, int det) {
  __shared__ double red[8];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  double eu = 0.0, ep = 0.0;
  constexpr int UR = 4;
  const int stride = gridDim.y * 4;
  const int row_end = (diagonal && blockIdx.x * 64 + 64 < H) ? blockIdx.x * 64 + 64 : H;      // (see energy_grid_kernel)
  for (int i0 = blockIdx.y * 4 + (threadIdx.x >> 6); i0 < row_end; i0 += stride * UR) {
    int64_t node[UR];
    bool on[UR];
    bool any = false;
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int i = i0 + u * stride;
      const bool in = i < H && j < W && !(diagonal && j < i);
      const int64_t row = diagonal ? (int64_t)i * W - ((int64_t)i * (i - 1)) / 2 - i : (int64_t)i * W;      // node = row + j
      node[u] = in ? row + j : 0;
      on[u] = in && (int)stamp[node[u]] > since;
      any = any || on[u];
    }
    if (!__any(any)) continue;
    int lab[UR], labp[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      lab[u] = labels[node[u]];
      labp[u] = prev[node[u]];
    }
    float un[UR], unp[UR];
    float4 w[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      un[u] = uT[(int64_t)lab[u] * n + node[u]];
      unp[u] = uT[(int64_t)labp[u] * n + node[u]];
      w[u] = fwd_w[node[u]];
    }
    int ln[UR][4], lp[UR][4];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int i = i0 + u * stride;
      const int64_t row2 = diagonal ? (int64_t)(i + 1) * W - ((int64_t)(i + 1) * i) / 2 - (i + 1) : (int64_t)(i + 1) * W;
      const int jlo = diagonal ? i + 1 : 0;
      const bool below = on[u] && i + 1 < H;
      const int64_t me = node[u];
      const int64_t c0 = (on[u] && j + 1 < W) ? me + 1 : me;                               // E
      const int64_t c1 = (below && j - 1 >= jlo) ? row2 + j - 1 : me;                      // SW
      const int64_t c2 = (below && j >= jlo) ? row2 + j : me;                              // S
      const int64_t c3 = (below && j + 1 < W) ? row2 + j + 1 : me;                         // SE
      ln[u][0] = labels[c0]; ln[u][1] = labels[c1]; ln[u][2] = labels[c2]; ln[u][3] = labels[c3];
      lp[u][0] = prev[c0]; lp[u][1] = prev[c1]; lp[u][2] = prev[c2]; lp[u][3] = prev[c3];
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      if (!on[u]) continue;
      const int l = lab[u], q = labp[u];
      eu += (double)un[u] - (double)unp[u];
      float s = 0.f;
      s += (ln[u][0] != l ? w[u].x : 0.f) - (lp[u][0] != q ? w[u].x : 0.f);
      s += (ln[u][1] != l ? w[u].y : 0.f) - (lp[u][1] != q ? w[u].y : 0.f);
      s += (ln[u][2] != l ? w[u].z : 0.f) - (lp[u][2] != q ? w[u].z : 0.f);
      s += (ln[u][3] != l ? w[u].w : 0.f) - (lp[u][3] != q ? w[u].w : 0.f);
      ep += (double)s;
    }
  }
  const double tu = block_sum(eu, red);
  const double tp = block_sum(ep, red);
  if (threadIdx.x == 0 && (tu != 0.0 || tp != 0.0)) energy_flush(accum, tu, tp, det);
}

// E(labels) - E(other) of two labellings of a grid block, from the nodes where they DIFFER: an edge's term can only differ
// where an end differs, and every edge is held by its upper / left end, so a node contributes if it or one of its four
// forward neighbours carries different labels in the two -- its unary terms (if it differs itself) and its forward edges,
// term by term exact in f32 and summed in f64.  Reads two label bytes per node and neighbour (cached), 24 B more where the
// labellings differ: the warm start of an E-step (phmrf_block_warm_start: labels_local against the previous result, which
// differ in a few per cent of the nodes) needs the SIGN of this difference, not two full energy passes.  Same thread layout as
// energy_grid_kernel; rows [row0, row1) are counted (a row tile's owned rows).
__global__ __launch_bounds__(256) void energy_diff_grid_kernel(const float* __restrict__ uT, int64_t n, int H, int W, int diagonal,
                                                               const float4* __restrict__ fwd_w,
                                                               const uint8_t* __restrict__ labels,
                                                               const uint8_t* __restrict__ other, double* __restrict__ accum,
                                                               int det, int row0, int row1) {
  __shared__ double red[8];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  double eu = 0.0, ep = 0.0;
  constexpr int UR = 4;
  const int stride = gridDim.y * 4;
  const int row_end = (diagonal && blockIdx.x * 64 + 64 < row1) ? blockIdx.x * 64 + 64 : row1;   // (see energy_grid_kernel)
  for (int i0 = row0 + blockIdx.y * 4 + (threadIdx.x >> 6); i0 < row_end; i0 += stride * UR) {
    int64_t me[UR], c[UR][4];
    bool in[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int i = i0 + u * stride;
      in[u] = i < row1 && j < W && !(diagonal && j < i);
      const int64_t row = diagonal ? (int64_t)i * W - ((int64_t)i * (i - 1)) / 2 - i : (int64_t)i * W;      // node = row + j
      const int64_t row2 = diagonal ? (int64_t)(i + 1) * W - ((int64_t)(i + 1) * i) / 2 - (i + 1) : (int64_t)(i + 1) * W;
      const int jlo = diagonal ? i + 1 : 0;
      const bool below = in[u] && i + 1 < H;
      me[u] = in[u] ? row + j : 0;
      c[u][0] = (in[u] && j + 1 < W) ? me[u] + 1 : me[u];                       // E
      c[u][1] = (below && j - 1 >= jlo) ? row2 + j - 1 : me[u];                 // SW
      c[u][2] = (below && j >= jlo) ? row2 + j : me[u];                         // S
      c[u][3] = (below && j + 1 < W) ? row2 + j + 1 : me[u];                    // SE
    }
    int la[UR], lb[UR], na[UR][4], nb[UR][4];
    bool on[UR];
    bool any = false;
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      la[u] = labels[me[u]];
      lb[u] = other[me[u]];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        na[u][q] = labels[c[u][q]];
        nb[u][q] = other[c[u][q]];
      }
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      on[u] = in[u] && (la[u] != lb[u] || na[u][0] != nb[u][0] || na[u][1] != nb[u][1] || na[u][2] != nb[u][2] || na[u][3] != nb[u][3]);
      any = any || on[u];
    }
    if (!__any(any)) continue;
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      if (!on[u]) continue;
      const float4 w = fwd_w[me[u]];
      if (la[u] != lb[u]) eu += (double)uT[(int64_t)la[u] * n + me[u]] - (double)uT[(int64_t)lb[u] * n + me[u]];
      float s = 0.f;
      // (an absent neighbour is the node itself: equal labels in both, and its weight is 0)
      s += (na[u][0] != la[u] ? w.x : 0.f) - (nb[u][0] != lb[u] ? w.x : 0.f);
      s += (na[u][1] != la[u] ? w.y : 0.f) - (nb[u][1] != lb[u] ? w.y : 0.f);
      s += (na[u][2] != la[u] ? w.z : 0.f) - (nb[u][2] != lb[u] ? w.z : 0.f);
      s += (na[u][3] != la[u] ? w.w : 0.f) - (nb[u][3] != lb[u] ? w.w : 0.f);
      ep += (double)s;
    }
  }
  const double tu = block_sum(eu, red);
  const double tp = block_sum(ep, red);
  if (threadIdx.x == 0 && (tu != 0.0 || tp != 0.0)) energy_flush(accum, tu, tp, det);
}

// -------------------------------------------------------------------------------------------------
// b3 posteriors + costs + sufficient statistics, fused (phylo_hmrf.py:334-355, :374-396, :311-314).
//   pp[i,k]   = beta * sum_{e in inc(i)} w'_e [l_other != k]          (w' = w if estimate_type==3 else 1)
//             = beta * (Wtot' - h[i,k]);   isolated node: beta*[k != l_i]            (:421-423)
//   post[i,k] = softmax_k(logprob - pp)   (max-shifted; the reference does not shift, :342-345)
//   ppn[i,k]  = softmax_k(-pp)
//   costs     : see phmrf.h
//   stats     : post | obs | obs*obs.T = Gamma^T [1 | x | x (x) x], accumulated in f64 across tiles,
//               one atomic flush per workgroup at the end (grid is capped, blocks stride over tiles)
// -------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// GRID: the block is the reference's 8-neighbour stencil on a grid: the neighbours are found by geometry and their
// weights come from the forward-edge records (16 B per node, each edge stored once) instead of the explicit adjacency
// rows (64 B per node) -- the same neighbours in the same order, so the two forms agree bit for bit.
template <int S, int VEC, bool WRITE_POST, bool GRID>
__global__ __launch_bounds__(256, 5) void posterior_kernel(const float* __restrict__ X, const float* __restrict__ logprob,
                                                        int64_t n, int K, int Kp, int D,
                                                        const int32_t* __restrict__ nbr, const float* __restrict__ wgt,
                                                        const uint8_t* __restrict__ labels, float beta, int use_w,
                                                        float* __restrict__ post_out, double* __restrict__ accum,
                                                        int gH, int gW, int gdiag, const float4* __restrict__ fwd_w,
                                                        int64_t n_first) {
  // (nodes [n_first, n): a row tile counts the nodes it owns; their neighbours may lie outside the range)
  extern __shared__ float lds[];
  constexpr int M = 1 + S + S * (S + 1) / 2;      // features [1 | x | x_s x_t, s <= t]: x x^T is symmetric
  // ... and every one of them is a product of two entries of xa = [1 | x]: only xa goes to LDS (S + 1 floats per node
  // instead of M), the product is formed where the statistics read it
  constexpr int Mp = ((S + 1) % 2 == 0) ? S + 2 : S + 1;
  constexpr int NMB = (M + 15) / 16;
  const int TB = blockDim.x;
  float* tile = lds;            // [TB][Kp]
  float* feat = lds + TB * Kp;  // [TB][Mp]: xa
  __shared__ double red[8];

  const int KM = K * M;
  // statistics of this workgroup, f64, in LDS behind the two tiles (8-byte aligned: TB * (Kp + Mp) is even)
  double* sacc = reinterpret_cast<double*>(lds + TB * (Kp + Mp));
  for (int o = threadIdx.x; o < KM; o += TB) sacc[o] = 0.0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double c_pair = 0.0, c_pcn = 0.0, c_un = 0.0;
  // the two factors of my feature column (lane & 15) in each 16-wide block of features: m = 0 -> (0, 0) = 1 * 1,
  // 1 <= m <= S -> (0, m) = 1 * x_{m-1}, pairs s <= t in the order of the flush below -> (s + 1, t + 1)
  int fa[NMB], fb[NMB];
#pragma unroll
  for (int q = 0; q < NMB; ++q) {
    const int m = q * 16 + (lane & 15);
    int a = 0, bq = 0;
    if (m >= 1 && m <= S) {
      bq = m;
    } else if (m > S && m < M) {
      int r = m - 1 - S, ps = 0;
      while (r >= S - ps) { r -= S - ps; ++ps; }
      a = ps + 1;
      bq = ps + r + 1;
    }
    fa[q] = a;
    fb[q] = bq;
  }
  __syncthreads();

  for (int64_t base = n_first + (int64_t)blockIdx.x * TB; base < n; base += (int64_t)gridDim.x * TB) {
    const int64_t rem = n - base;
    const int rows = rem < TB ? (int)rem : TB;
    const int64_t i = base + threadIdx.x;
    const bool live = (int)threadIdx.x < rows;
    float* row = tile + threadIdx.x * Kp;
    float wtot = 0.f;
    int li = 0;
    // phase 1: neighbour-label histogram h[k] in the node's own LDS row; pairwise / ppn / unary costs
    if (live) {
      li = labels[i];
      for (int k = 0; k < K; ++k) row[k] = 0.f;
      const int32_t* nb = nbr + i * D;
      const float* wg = wgt + i * D;
      float diff = 0.f;
      int deg = 0;
      if (GRID) {
        int gi, gj;
        grid_coords(i, gW, gdiag, &gi, &gj);
        int64_t cc[8];
        float ww[8];
        grid_gather_neighbours(i, gi, gj, gH, gW, gdiag, fwd_w, cc, ww);      // (an absent neighbour is the node itself)
        int ll[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) ll[t] = labels[cc[t]];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (cc[t] != i) {
            const float w = use_w ? ww[t] : 1.f;
            const int l = ll[t];
            row[l] += w;
            wtot += w;
            if (l != li) diff += w;
            ++deg;
          }
        }
      } else if (D == 8) {               // the reference's stencil: the whole row and the labels behind it side by side
        const int4 c0 = *reinterpret_cast<const int4*>(nb), c1 = *reinterpret_cast<const int4*>(nb + 4);
        const float4 w0 = *reinterpret_cast<const float4*>(wg), w1 = *reinterpret_cast<const float4*>(wg + 4);
        const int cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        int ll[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) ll[t] = labels[cc[t] >= 0 ? (int64_t)cc[t] : i];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (cc[t] >= 0) {
            const float w = use_w ? ww[t] : 1.f;
            const int l = ll[t];
            row[l] += w;
            wtot += w;
            if (l != li) diff += w;
            ++deg;
          }
        }
      } else
      for (int j = 0; j < D; j += 4) {
        const int4 c = *reinterpret_cast<const int4*>(nb + j);
        const float4 wv = *reinterpret_cast<const float4*>(wg + j);
        const int cc[4] = {c.x, c.y, c.z, c.w};
        const float ww[4] = {wv.x, wv.y, wv.z, wv.w};
        int ll[4];                       // (the four labels side by side; absent entries read the node's own)
#pragma unroll
        for (int t = 0; t < 4; ++t) ll[t] = labels[cc[t] >= 0 ? (int64_t)cc[t] : i];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (cc[t] >= 0) {
            const float w = use_w ? ww[t] : 1.f;
            const int l = ll[t];
            row[l] += w;
            wtot += w;
            if (l != li) diff += w;
            ++deg;
          }
        }
      }
      c_pair += (double)(beta * diff);
      if (!deg) {  // isolated (:421-423): pp = V[l_i,:]  <=>  h = onehot(l_i), Wtot = 1
        row[li] = 1.f;
        wtot = 1.f;
      }
      // -pp_k = beta*(h_k - wtot) <= 0 with max 0 reached only if some h_k == wtot; shift by the max anyway
      float mb = -3.0e38f;
      for (int k = 0; k < K; ++k) mb = fmaxf(mb, beta * (row[k] - wtot));
      float s2 = 0.f;
      for (int k = 0; k < K; ++k) s2 += __expf(beta * (row[k] - wtot) - mb);
      const float ppn = __expf(beta * (row[li] - wtot) - mb) / s2;
      c_pcn -= (double)logf(ppn + 1e-16f);
      c_un -= (double)logprob[i * K + li];
    }
    __syncthreads();
    // phase 2: tile = logprob + beta*h   (coalesced row loads, K/VEC lanes per row)
    // (measured: issuing these loads and the observation vector before phase 1 -- 24 more registers -- made the kernel 12 %
    //  slower; smaller tiles for more workgroups per CU: 128 rows +13 %, 64 rows +33 %; a form of this kernel that reads
    //  the node's K terms from the label-major planes into registers -- no LDS transposition, both soft-maxes unrolled
    //  on registers, two barriers per tile: half the vector instructions, 142 registers, 1.11 instead of 0.89 ms on the
    //  12.4 M-node block -- the kernel issues at ~0.19 vector instructions per cycle and SIMD as it is, and three waves
    //  per SIMD do not hide a longer chain of loads per tile)
    rows_to_tile<VEC, true>(logprob, nullptr, base, rows, K, Kp, tile, 1.f, beta);
    __syncthreads();
    // phase 3: posteriors = softmax_k(tile_k - beta*wtot) in place; per-node features [1 | x | x x^T]
    if (live) {
      float m = -3.0e38f;
      for (int k = 0; k < K; ++k) m = fmaxf(m, row[k]);
      float s1 = 0.f;
      for (int k = 0; k < K; ++k) {
        const float e = __expf(row[k] - m);
        row[k] = e;
        s1 += e;
      }
      const float inv = 1.f / s1;
      for (int k = 0; k < K; ++k) row[k] *= inv;
      float x[S];
      if (S % 4 == 0) {
#pragma unroll
        for (int s = 0; s < S; s += 4) {
          const float4 t = *reinterpret_cast<const float4*>(X + i * S + s);
          x[s] = t.x; x[s + 1] = t.y; x[s + 2] = t.z; x[s + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int s = 0; s < S; ++s) x[s] = X[i * S + s];
      }
      float* f = feat + threadIdx.x * Mp;
      f[0] = 1.f;
#pragma unroll
      for (int s = 0; s < S; ++s) f[1 + s] = x[s];
    }
    __syncthreads();
    if (WRITE_POST) tile_to_rows<VEC>(post_out + base * K, rows, K, Kp, tile);
    // phase 4: stats[k][m] += sum_r gamma[r][k] * feat[r][m] -- a (K x rows) x (rows x M) product per tile, on the matrix
    // cores in exact f32, in 16 x 16 blocks (v_mfma_f32_16x16x4_f32: A lane l = gamma^T[k = l & 15][r = l >> 4], B lane l =
    // feat[r = l >> 4][m = l & 15], four results per lane at column l & 15, rows 4 (l >> 4) + reg): K = 20 states and the
    // M = 15 features of S = 4 are 2 x 1 blocks, where the 32 x 32 form took twice the matrix-core time for a block that
    // is 40 % padding.  Each wave sums its own 64 rows of the tile in f32 and adds the blocks to the workgroup's f64
    // accumulators in LDS.
    {
      const int r0 = wave * 64 + (lane >> 4), c = lane & 15;
      for (int kb = 0; kb < K; kb += 16)
#pragma unroll
        for (int qb = 0; qb < NMB; ++qb) {
          const int mb = qb * 16;
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
          const bool ka = kb + c < K, ma = mb + c < M;
          const int ia = fa[qb], ib = fb[qb];
#pragma unroll 8
          for (int st = 0; st < 16; ++st) {
            const int r = r0 + 4 * st;
            const float a = (ka && r < rows) ? tile[r * Kp + kb + c] : 0.f;
            const float bq = (ma && r < rows) ? feat[r * Mp + ia] * feat[r * Mp + ib] : 0.f;
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, d, 0, 0, 0);
          }
          if (ma && wave * 64 < rows) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int k = kb + 4 * (lane >> 4) + q;
              if (k < K) atomicAdd(sacc + k * M + mb + c, (double)d[q]);
            }
          }
        }
    }
    __syncthreads();
  }
  // flush: statistics (layout post[K] | obs[K,S] | obsobsT[K,S,S]) and the three cost numerators
  __syncthreads();
  for (int o = threadIdx.x; o < KM; o += TB) {
    const int k = o / M, m = o - k * M;
    if (m == 0) {
      atomicAdd(accum + ACC_STATS + k, sacc[o]);
    } else if (m <= S) {
      atomicAdd(accum + ACC_STATS + K + k * S + (m - 1), sacc[o]);
    } else {                                  // pair (s, t), s <= t, in the order the features were written; mirrored
      int q = m - 1 - S, ps = 0;
      while (q >= S - ps) { q -= S - ps; ++ps; }
      const int pt = ps + q;
      atomicAdd(accum + ACC_STATS + K + K * S + k * S * S + ps * S + pt, sacc[o]);
      if (pt != ps) atomicAdd(accum + ACC_STATS + K + K * S + k * S * S + pt * S + ps, sacc[o]);
    }
  }
  const double t0 = block_sum(c_pair, red);
  const double t1 = block_sum(c_pcn, red);
  const double t2 = block_sum(c_un, red);
  if (threadIdx.x == 0) {
    atomicAdd(accum + ACC_COST + 0, t0);
    atomicAdd(accum + ACC_COST + 1, t1);
    atomicAdd(accum + ACC_COST + 2, t2);
    atomicAdd(accum + ACC_COST + 3, t1 + t2);
  }
}

template <int S>
int launch_posterior_s(const phmrf_block* b, float beta, int estimate_type, bool write_post) {
  constexpr int M = 1 + S + S * (S + 1) / 2;      // features [1 | x | x_s x_t, s <= t]: x x^T is symmetric
  constexpr int Mp = ((S + 1) % 2 == 0) ? S + 2 : S + 1;      // LDS holds [1 | x] per node
  const int K = b->K, Kp = padded_k(K);
  static const int tb_env = PHMRF_DEV_ENV("PHMRF_POST_TB") ? atoi(PHMRF_DEV_ENV("PHMRF_POST_TB")) : 0;     // development: tile rows
  int TB = (tb_env == 64 || tb_env == 128 || tb_env == 256) ? tb_env : 256;
  const size_t acc_bytes = (size_t)K * M * sizeof(double);            // the workgroup's f64 statistics
  // (the tile prefers 64 KB, which leaves two workgroups per CU; K and S at their limits -- K = 64, S = 8: 72 KB at 64 rows --
  //  take a larger share of the CU's 160 KB instead of being refused)
  while (TB > 64 && (size_t)TB * (Kp + Mp) * sizeof(float) + acc_bytes > 64 * 1024 - 256) TB >>= 1;
  PHMRF_CHECK((size_t)TB * (Kp + Mp) * sizeof(float) + acc_bytes <= 158 * 1024, PHMRF_ERR_UNSUPPORTED,
              "posterior_stats: K and S too large for the LDS tile");
  const size_t lds = (size_t)TB * (Kp + Mp) * sizeof(float) + acc_bytes;
  static const int cap_env = PHMRF_DEV_ENV("PHMRF_POST_GRID") ? atoi(PHMRF_DEV_ENV("PHMRF_POST_GRID")) : 0;   // development: grid cap
  // (grid cap swept on the 12.4 M-node block: 2048 -> 925 us, 1024 -> 1069, 768 = three resident workgroups per CU -> 885, 512 -> 1111)
  // (round 3, packed features: 39 KB at K = 20, S = 4 = four resident workgroups per CU: 768 -> 790 us, 1024 -> 680, 1280 -> 810;
  //  then only [1 | x] in LDS: 29 KB, and 84 registers under __launch_bounds__(256, 5) = five per CU: 1024 -> 700, 1280 -> 650)
  const int64_t n_first = b->own1 >= 0 ? b->own0 : 0, n_last = b->own1 >= 0 ? b->own1 : b->n;
  const int grid = grid_for(n_last - n_first, TB, cap_env > 0 ? cap_env : (lds <= 32 * 1024 ? 256 * 5 : (lds <= 40 * 1024 ? 256 * 4 : (lds <= 53 * 1024 ? 256 * 3 : 256 * 8))) * (256 / TB));
  const int use_w = estimate_type == 3 ? 1 : 0;
  // the grid form needs the 8-neighbour stencil's forward-edge records (phmrf_block_set_grid / build_grid_graph)
  const bool grid_form = b->has_grid && b->grid_complete && b->fwd_w && b->D == 8 && b->num_neighbor == 8;
#define PHMRF_LAUNCH_POST_G(VEC_, WP_, G_)                                                                          \
  {                                                                                                                 \
    if (lds > 64 * 1024 - 256)                                                                                      \
      PHMRF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&posterior_kernel<S, VEC_, WP_, G_>),             \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
    hipLaunchKernelGGL((posterior_kernel<S, VEC_, WP_, G_>), dim3(grid), dim3(TB), lds, b->stream, b->X, b->logprob, \
                       n_last, K, Kp, b->D, b->nbr, b->wgt, b->labels, beta, use_w, b->posteriors, b->accum, b->H, b->W, \
                       b->diagonal, b->fwd_w, n_first);                                                             \
  }
#define PHMRF_LAUNCH_POST(VEC_, WP_)                                                                                \
  if (grid_form) PHMRF_LAUNCH_POST_G(VEC_, WP_, true) else PHMRF_LAUNCH_POST_G(VEC_, WP_, false)
  const int v = vec_of(K);
  if (write_post) {
    if (v == 4) { PHMRF_LAUNCH_POST(4, true); } else if (v == 2) { PHMRF_LAUNCH_POST(2, true); } else { PHMRF_LAUNCH_POST(1, true); }
  } else {
    if (v == 4) { PHMRF_LAUNCH_POST(4, false); } else if (v == 2) { PHMRF_LAUNCH_POST(2, false); } else { PHMRF_LAUNCH_POST(1, false); }
  }
#undef PHMRF_LAUNCH_POST
#undef PHMRF_LAUNCH_POST_G
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

}  // namespace

// ---- launchers --------------------------------------------------------------------------------------
int launch_emission(const float* X, int64_t n, int S, int K, const float* packed, float* logprob, float* uT, hipStream_t st) {
  switch (S) {
#define PHMRF_CASE(S_) case S_: return launch_emission_s<S_>(X, n, K, packed, logprob, uT, st);
    PHMRF_CASE(1) PHMRF_CASE(2) PHMRF_CASE(3) PHMRF_CASE(4) PHMRF_CASE(5) PHMRF_CASE(6) PHMRF_CASE(7) PHMRF_CASE(8)
    PHMRF_CASE(9) PHMRF_CASE(10) PHMRF_CASE(11) PHMRF_CASE(12) PHMRF_CASE(13) PHMRF_CASE(14) PHMRF_CASE(15) PHMRF_CASE(16)
#undef PHMRF_CASE
  }
  return fail(PHMRF_ERR_UNSUPPORTED, "emission: S must be in [1,16]");
}

int launch_argmax_labels(const phmrf_block* b) {
  const int K = b->K, TB = tile_threads(K), Kp = padded_k(K);
  const size_t lds = (size_t)TB * Kp * sizeof(float);
  const int grid = grid_for(b->n, TB);
  switch (vec_of(K)) {
    case 4: hipLaunchKernelGGL((argmax_kernel<4>), dim3(grid), dim3(TB), lds, b->stream, b->logprob, b->n, K, Kp, b->labels); break;
    case 2: hipLaunchKernelGGL((argmax_kernel<2>), dim3(grid), dim3(TB), lds, b->stream, b->logprob, b->n, K, Kp, b->labels); break;
    default: hipLaunchKernelGGL((argmax_kernel<1>), dim3(grid), dim3(TB), lds, b->stream, b->logprob, b->n, K, Kp, b->labels); break;
  }
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

int launch_icm_colour(const phmrf_block* b, float beta, int colour) {
  const int K = b->K, TB = tile_threads(K), Kp = padded_k(K);
  const int64_t lo = b->colour_ptr[colour], hi = b->colour_ptr[colour + 1];
  const int64_t count = hi - lo;
  if (count <= 0) return PHMRF_OK;
  const size_t lds = (size_t)TB * Kp * sizeof(float);
  const int grid = grid_for(count, TB);
  const int32_t* nodes = b->colour_nodes + lo;
  switch (vec_of(K)) {
    case 4: hipLaunchKernelGGL((icm_kernel<4>), dim3(grid), dim3(TB), lds, b->stream, b->logprob, nodes, count, K, Kp, b->D, b->nbr, b->wgt, b->labels, beta, b->counters + b->counter_slot, b->tick ? b->stamp : nullptr, b->tick); break;
    case 2: hipLaunchKernelGGL((icm_kernel<2>), dim3(grid), dim3(TB), lds, b->stream, b->logprob, nodes, count, K, Kp, b->D, b->nbr, b->wgt, b->labels, beta, b->counters + b->counter_slot, b->tick ? b->stamp : nullptr, b->tick); break;
    default: hipLaunchKernelGGL((icm_kernel<1>), dim3(grid), dim3(TB), lds, b->stream, b->logprob, nodes, count, K, Kp, b->D, b->nbr, b->wgt, b->labels, beta, b->counters + b->counter_slot, b->tick ? b->stamp : nullptr, b->tick); break;
  }
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

// labels <- saved where the snapshot's energy (acc_sav: unary, pair sums) is not above the current labels' (acc_cur): the
// choice of phmrf_block_warm_start, taken on the device so that the host does not have to wait for the two evaluations
__global__ __launch_bounds__(256) void choose_labels_kernel(uint8_t* __restrict__ labels, const uint8_t* __restrict__ saved, int64_t n,
                                                            const double* __restrict__ acc_cur, const double* __restrict__ acc_sav,
                                                            double beta, int det) {
  double cu, cp, su, sp;
  if (det) {
    cu = (double)reinterpret_cast<const long long*>(acc_cur)[0] / ENERGY_FIX;
    cp = (double)reinterpret_cast<const long long*>(acc_cur)[1] / ENERGY_FIX;
    su = (double)reinterpret_cast<const long long*>(acc_sav)[0] / ENERGY_FIX;
    sp = (double)reinterpret_cast<const long long*>(acc_sav)[1] / ENERGY_FIX;
  } else {
    cu = acc_cur[0]; cp = acc_cur[1]; su = acc_sav[0]; sp = acc_sav[1];
  }
  if (cu + beta * cp < su + beta * sp) return;             // the current labels are the better start
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += (int64_t)gridDim.x * blockDim.x * 16) {
    if (i + 16 <= n && ((reinterpret_cast<uintptr_t>(labels + i) | reinterpret_cast<uintptr_t>(saved + i)) & 15) == 0) {
      *reinterpret_cast<uint4*>(labels + i) = *reinterpret_cast<const uint4*>(saved + i);
    } else {
      for (int64_t q = i; q < n && q < i + 16; ++q) labels[q] = saved[q];
    }
  }
}

int launch_choose_labels(const phmrf_block* b, const uint8_t* saved, const double* acc_cur, const double* acc_sav, double beta) {
  int64_t g64 = (b->n / 16 + 255) / 256 + 1;
  const int grid = (int)(g64 > 4096 ? 4096 : g64);
  hipLaunchKernelGGL(choose_labels_kernel, dim3(grid), dim3(256), 0, b->stream, b->labels, saved, b->n, acc_cur, acc_sav, beta,
                     b->deterministic ? 1 : 0);
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

// E(current labels) - E(other) into the block's energy slots (caller zeroes them); grid blocks with unary planes only
bool energy_diff_available(const phmrf_block* b) {
  return b->has_grid && b->fwd_w && b->H > 0 && b->W > 0 && b->uT && b->uT_valid && b->n >= (1 << 16);
}

int launch_energy_diff(const phmrf_block* b, const uint8_t* other) {
  const int gx = (b->W + 63) / 64;
  int gy = (b->H + 3) / 4;
  const int cap = 2048 / gx + 1;
  if (gy > cap) gy = cap;
  const int row0 = b->own_r1 >= 0 ? b->own_r0 : 0, row1 = b->own_r1 >= 0 ? b->own_r1 : b->H;
  hipLaunchKernelGGL(energy_diff_grid_kernel, dim3(gx, gy), dim3(256), 0, b->stream, b->uT, b->n, b->H, b->W, b->diagonal, b->fwd_w,
                     b->labels, other, b->accum, b->deterministic ? 1 : 0, row0, row1);
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

// accum_at: where the two sums go (nullptr: the block's accum area, slots ACC_ENERGY, ACC_ENERGY + 1)
int launch_energy(const phmrf_block* b, float beta, double* accum_at) {
  (void)beta;
  double* const acc = accum_at ? accum_at - ACC_ENERGY : b->accum;
  if (b->has_grid && b->fwd_w && b->H > 0 && b->W > 0) {
    const int gx = (b->W + 63) / 64;
    int gy = (b->H + 3) / 4;
    const int cap = 2048 / gx + 1;          // ~2048 workgroups: each sums many rows before its two f64 atomics
    if (gy > cap) gy = cap;
    const int row0 = b->own_r1 >= 0 ? b->own_r0 : 0, row1 = b->own_r1 >= 0 ? b->own_r1 : b->H;
    hipLaunchKernelGGL(energy_grid_kernel, dim3(gx, gy), dim3(256), 0, b->stream, b->logprob,
                       (b->uT && b->uT_valid) ? b->uT : nullptr, b->n, b->K, b->H, b->W, b->diagonal, b->fwd_w, b->labels, acc,
                       b->deterministic ? 1 : 0, row0, row1);
    PHMRF_HIP(hipGetLastError());
    return PHMRF_OK;
  }
  const int grid = grid_for(b->n, 256, 256 * 8);
  hipLaunchKernelGGL(energy_kernel, dim3(grid), dim3(256), 0, b->stream, b->logprob, b->n, b->K, b->D, b->nbr, b->wgt,
                     b->labels, acc, b->deterministic ? 1 : 0);
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

// -> accum[ACC_ENERGY .. +1] += the change of (unary, pair) since the snapshot `labels_eval` / tick `eval_tick` (grid blocks
// with current unary planes only: energy_delta_available)
bool energy_delta_available(const phmrf_block* b) {
  return b->has_grid && b->fwd_w && b->H > 0 && b->W > 0 && b->uT && b->uT_valid && b->stamp && b->tick > 0 && b->labels_eval &&
         b->eval_tick >= 0;
}

int launch_energy_delta(const phmrf_block* b, double* accum_at) {
  double* const acc = accum_at ? accum_at - ACC_ENERGY : b->accum;
  const int gx = (b->W + 63) / 64;
  int gy = (b->H + 3) / 4;
  const int cap = 2048 / gx + 1;
  if (gy > cap) gy = cap;
  hipLaunchKernelGGL(energy_delta_grid_kernel, dim3(gx, gy), dim3(256), 0, b->stream, b->uT, b->n, b->H, b->W, b->diagonal,
                     b->fwd_w, b->labels, b->labels_eval, b->stamp, b->eval_tick, acc, b->deterministic ? 1 : 0);
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

int launch_posterior_stats(const phmrf_block* b, float beta, int estimate_type, bool write_posteriors) {
  switch (b->S) {
#define PHMRF_CASE(S_) case S_: return launch_posterior_s<S_>(b, beta, estimate_type, write_posteriors);
    PHMRF_CASE(1) PHMRF_CASE(2) PHMRF_CASE(3) PHMRF_CASE(4) PHMRF_CASE(5) PHMRF_CASE(6) PHMRF_CASE(7) PHMRF_CASE(8)
#undef PHMRF_CASE
  }
  return fail(PHMRF_ERR_UNSUPPORTED, "posterior_stats: S must be in [1,8]");
}

}  // namespace phmrf
