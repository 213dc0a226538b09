// alpha-expansion on a graph that is no grid, by an exact minimum cut on the GPU (round 6).
//
// The boundary the build replaces is pygco.cut_general_graph on a GENERAL graph (phylo_hmrf.py:496-498,
// GCoptimization.h:551-597); the reference itself only ever builds the contact-map stencil (utility.py:1871-2053), for
// which the strip moves exist.  Off the grid the exact block moves of the solver were 1-D (paths) and whole components;
// gco's swap reaches minima that need moves over node sets with cycles in them.  This file gives the general graph the
// move gco's expansion() makes (GCoptimization.cpp:1120-1280): every node keeps its label or takes alpha, decided for
// ALL nodes at once by a minimum s-t cut -- not a port of gco's max-flow (BK augmenting paths: sequential) but the
// lock-free push-relabel of Hong / He, which suits a GPU: one thread per node, integer atomics, heights may be stale.
//
// The binary problem.  x_i = 1: node i takes alpha.  For the Potts model
//   E(x) = sum_i theta_i x_i + sum_{arcs i->j} c_ij x_i (1 - x_j) + const,
//   theta_i = u_i(alpha) - u_i(l_i) - beta sum_{j: l_j = alpha} w_ij - beta/2 sum_{j active} w_ij [l_i != l_j],
//   c_ij = c_ji = beta w_ij (1 - [l_i != l_j] / 2)                       (active = the nodes with l != alpha)
// (the symmetric reparametrisation of the pair tables A = w [l_i != l_j], B = C = w, D = 0: submodular because Potts is a
// metric).  theta_i < 0 is a source arc s->i of capacity -theta_i, theta_i > 0 a sink arc i->t.  Capacities are
// quantised to integers with the largest term at 2^24 -- gco's own finest quantisation (GCO_MAX_ENERGYTERM = 1e7,
// GCoptimization.h:139-143) -- so pushes are exact and the algorithm ends; a node's switch cost is rounded UP by one
// quantum so that ties keep the label.
//
// The flow.  push_relabel_kernel: every active node with excess and height < n pushes to the sink first, then to its
// lowest residual neighbour if that is lower, else lifts itself above it (atomics on the neighbour's excess and on the
// reverse arc only).  Excess that no sink can take is trapped among the nodes that WILL switch; instead of lifting them
// to n one step at a time, a global relabelling every PR_SWEEPS sweeps sets every node's height to its residual distance
// to the sink (level-synchronous BFS; most nodes have a sink arc, so the BFS only has to walk into the pockets) and n where
// there is no path.  The move: the nodes that cannot reach the sink take alpha.  Energy non-increasing up to the
// quantisation (1e-7 of the largest term per node); the solver's round energies are f64 and judge it.

#include <algorithm>
#include <vector>

#include "common.h"

namespace phmrf {
namespace {

constexpr int PR_SWEEPS = 24;          // push-relabel sweeps between two global relabellings
constexpr int BFS_LEVELS = 12;         // BFS levels queued per look at the "changed" flag
constexpr int MAX_ROUNDS = 64;         // (sweeps + relabelling) rounds before an expansion is given up (no move)
constexpr float CAP_SCALE_TOP = 16777216.f;      // 2^24

struct MfPtrs {
  int64_t n;
  int K, D;
  const float* logprob;
  const int32_t* nbr;
  const float* wgt;
  const uint8_t* rev;
  uint8_t* labels;
  float* theta;
  int32_t* cap;
  int32_t* tcap;
  long long* exc;
  int32_t* hgt;
  int32_t* flags;        // [0] max |term| as float bits, [1] BFS changed, [2] active nodes, [3] labels changed
};

__global__ __launch_bounds__(256) void mf_theta_kernel(MfPtrs p, float beta, int alpha) {
  float mx = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = p.labels[i];
    float th = 0.f;
    if (l != alpha) {
      th = -p.logprob[i * p.K + alpha] + p.logprob[i * p.K + l];
      const int32_t* nb = p.nbr + i * p.D;
      const float* wg = p.wgt + i * p.D;
      for (int a = 0; a < p.D; ++a) {
        const int j = nb[a];
        if (j < 0) continue;
        const int lj = p.labels[j];
        const float w = beta * wg[a];
        if (lj == alpha) th -= w;
        else if (lj != l) th -= 0.5f * w;
        mx = fmaxf(mx, w);
      }
    }
    p.theta[i] = th;
    mx = fmaxf(mx, fabsf(th));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(p.flags, __float_as_int(mx));      // (non-negative floats order like ints)
}

__global__ __launch_bounds__(256) void mf_quantise_kernel(MfPtrs p, float beta, int alpha) {
  const float top = __int_as_float(p.flags[0]);
  const float scale = top > 0.f ? CAP_SCALE_TOP / top : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = p.labels[i];
    const bool act = l != alpha;
    const int32_t* nb = p.nbr + i * p.D;
    const float* wg = p.wgt + i * p.D;
    for (int a = 0; a < p.D; ++a) {
      const int j = nb[a];
      int c = 0;
      if (act && j >= 0) {
        const int lj = p.labels[j];
        if (lj != alpha) c = (int)(scale * beta * wg[a] * (lj != l ? 0.5f : 1.f));
      }
      p.cap[i * p.D + a] = c;
    }
    // the switch cost rounded up by one quantum: a tie keeps the label
    const long long q = act ? (long long)ceilf(scale * p.theta[i]) + 1ll : 0ll;
    p.tcap[i] = q > 0 ? (int32_t)(q > 2000000000ll ? 2000000000ll : q) : 0;
    p.exc[i] = q < 0 ? -q : 0ll;
    p.hgt[i] = act ? 0 : (int32_t)p.n;
  }
}

// one sweep of the lock-free push-relabel (Hong 2008; He & Hong 2010): correct with stale heights, atomics on the excess of
// the receiving node and on the reverse arc only
__global__ __launch_bounds__(256) void mf_push_relabel_kernel(MfPtrs p) {
  const int32_t nn = (int32_t)p.n;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < p.n; u += (int64_t)gridDim.x * blockDim.x) {
    long long e = p.exc[u];
    const int32_t hu = p.hgt[u];
    if (e <= 0 || hu >= nn) continue;
    int32_t tc = p.tcap[u];
    if (tc > 0) {                                            // the sink has height 0: always downhill
      const long long d = e < (long long)tc ? e : (long long)tc;
      p.tcap[u] = tc - (int32_t)d;
      e = atomicAdd(reinterpret_cast<unsigned long long*>(p.exc + u), (unsigned long long)(-d)) - d;
      if (e <= 0) continue;
    }
    const int32_t* nb = p.nbr + u * p.D;
    int32_t* cu = p.cap + u * p.D;
    int32_t hmin = 0x7fffffff;
    int amin = -1;
    for (int a = 0; a < p.D; ++a) {
      const int v = nb[a];
      if (v < 0 || cu[a] <= 0) continue;
      const int32_t hv = p.hgt[v];
      if (hv < hmin) {
        hmin = hv;
        amin = a;
      }
    }
    if (amin < 0) {
      p.hgt[u] = nn;                                         // no residual arc at all: the excess stays here
    } else if (hu > hmin) {
      const int v = nb[amin];
      const int32_t c = cu[amin];
      const long long d = e < (long long)c ? e : (long long)c;
      cu[amin] = c - (int32_t)d;
      atomicAdd(p.cap + (int64_t)v * p.D + p.rev[u * p.D + amin], (int32_t)d);
      atomicAdd(reinterpret_cast<unsigned long long*>(p.exc + u), (unsigned long long)(-d));
      atomicAdd(reinterpret_cast<unsigned long long*>(p.exc + v), (unsigned long long)d);
    } else {
      p.hgt[u] = hmin + 1 < nn ? hmin + 1 : nn;
    }
  }
}

// global relabelling: heights = residual distances to the sink.  lab lives in hgt: UNSEEN until a level reaches the node.
constexpr int32_t UNSEEN = 0x7ffffff0;

__global__ __launch_bounds__(256) void mf_bfs_init_kernel(MfPtrs p, int alpha) {
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < p.n; u += (int64_t)gridDim.x * blockDim.x) {
    const bool act = p.labels[u] != alpha;
    p.hgt[u] = !act ? (int32_t)p.n : (p.tcap[u] > 0 ? 1 : UNSEEN);
  }
}

__global__ __launch_bounds__(256) void mf_bfs_level_kernel(MfPtrs p, int level, int gate_level) {
  // (a level whose predecessor changed nothing has nothing to do: flags[1] holds the last level that labelled a node)
  if (level > gate_level && p.flags[1] < level - 1) return;
  bool any = false;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < p.n; u += (int64_t)gridDim.x * blockDim.x) {
    if (p.hgt[u] != UNSEEN) continue;
    const int32_t* nb = p.nbr + u * p.D;
    const int32_t* cu = p.cap + u * p.D;
    for (int a = 0; a < p.D; ++a) {
      const int v = nb[a];
      if (v >= 0 && cu[a] > 0 && p.hgt[v] == level - 1) {
        p.hgt[u] = level;
        any = true;
        break;
      }
    }
  }
  if (any) atomicMax(p.flags + 1, level);
}

__global__ __launch_bounds__(256) void mf_bfs_finish_kernel(MfPtrs p) {
  int cnt = 0;
  const int32_t nn = (int32_t)p.n;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < p.n; u += (int64_t)gridDim.x * blockDim.x) {
    int32_t h = p.hgt[u];
    if (h == UNSEEN) p.hgt[u] = h = nn;                      // no residual path to the sink
    if (h < nn && p.exc[u] > 0) ++cnt;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(p.flags + 2, cnt);
}

__global__ __launch_bounds__(256) void mf_apply_kernel(MfPtrs p, int alpha, unsigned long long* __restrict__ changed,
                                                       uint16_t* __restrict__ stamp, int tick) {
  unsigned int mine = 0u;
  const int32_t nn = (int32_t)p.n;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < p.n; u += (int64_t)gridDim.x * blockDim.x) {
    if (p.labels[u] == alpha || p.hgt[u] < nn) continue;     // (an active node that cannot reach the sink takes alpha)
    p.labels[u] = (uint8_t)alpha;
    ++mine;
    if (stamp) {
      stamp[u] = (uint16_t)tick;
      const int32_t* nb = p.nbr + u * p.D;
      for (int a = 0; a < p.D; ++a)
        if (nb[a] >= 0) stamp[nb[a]] = (uint16_t)tick;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(changed, (unsigned long long)mine);
}

template <class T>
int ensure(T** p, size_t count) {
  if (*p) return PHMRF_OK;
  PHMRF_HIP(hipMalloc(reinterpret_cast<void**>(p), (count ? count : 1) * sizeof(T)));
  return PHMRF_OK;
}

}  // namespace

// the slot of arc (u -> v) in v's adjacency row, for every arc: once per graph, on the host
int graph_expansion_setup(phmrf_block* b) {
  if (b->mf_rev) return PHMRF_OK;
  const int64_t n = b->n;
  const int D = b->D;
  std::vector<int32_t> nbr((size_t)n * D);
  PHMRF_HIP(hipMemcpyAsync(nbr.data(), b->nbr, nbr.size() * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
  PHMRF_HIP(hipStreamSynchronize(b->stream));
  std::vector<uint8_t> rev((size_t)n * D, 0);
  for (int64_t u = 0; u < n; ++u)
    for (int a = 0; a < D; ++a) {
      const int32_t v = nbr[(size_t)u * D + a];
      if (v < 0) continue;
      const int32_t* rv = &nbr[(size_t)v * D];
      int slot = -1;
      for (int c = 0; c < D; ++c)
        if (rv[c] == (int32_t)u) {
          slot = c;
          break;
        }
      if (slot < 0) return fail(PHMRF_ERR_INVALID, "internal: adjacency is not symmetric");
      rev[(size_t)u * D + a] = (uint8_t)slot;
    }
  PHMRF_TRY(ensure(&b->mf_rev, (size_t)n * D));
  PHMRF_HIP(hipMemcpyAsync(b->mf_rev, rev.data(), rev.size(), hipMemcpyHostToDevice, b->stream));
  PHMRF_HIP(hipStreamSynchronize(b->stream));
  PHMRF_TRY(ensure(&b->mf_theta, (size_t)n));
  PHMRF_TRY(ensure(&b->mf_cap, (size_t)n * D));
  PHMRF_TRY(ensure(&b->mf_tcap, (size_t)n));
  PHMRF_TRY(ensure(&b->mf_exc, (size_t)n));
  PHMRF_TRY(ensure(&b->mf_hgt, (size_t)n));
  PHMRF_TRY(ensure(&b->mf_flags, (size_t)4));
  if (!b->mf_flags_host) PHMRF_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->mf_flags_host), 4 * sizeof(int32_t)));
  return PHMRF_OK;
}

// one alpha-expansion of the whole graph; the labels that changed are added to b->counters[b->counter_slot].
// Host-synchronous (a few small read-backs per expansion): general graphs are off the hot path.
int launch_graph_expansion(phmrf_block* b, float beta, int alpha) {
  PHMRF_TRY(graph_expansion_setup(b));
  const int64_t n = b->n;
  int64_t g64 = (n + 255) / 256;
  const int grid = (int)(g64 > 256 * 32 ? 256 * 32 : g64);
  hipStream_t st = b->stream;
  MfPtrs p{n, b->K, b->D, b->logprob, b->nbr, b->wgt, b->mf_rev, b->labels, b->mf_theta, b->mf_cap, b->mf_tcap, b->mf_exc,
           b->mf_hgt, b->mf_flags};
  PHMRF_HIP(hipMemsetAsync(b->mf_flags, 0, 4 * sizeof(int32_t), st));
  hipLaunchKernelGGL(mf_theta_kernel, dim3(grid), dim3(256), 0, st, p, beta, alpha);
  hipLaunchKernelGGL(mf_quantise_kernel, dim3(grid), dim3(256), 0, st, p, beta, alpha);
  bool done = false;
  for (int round = 0; round < MAX_ROUNDS && !done; ++round) {
    for (int sw = 0; sw < PR_SWEEPS; ++sw) hipLaunchKernelGGL(mf_push_relabel_kernel, dim3(grid), dim3(256), 0, st, p);
    // global relabelling: BFS from the sink, BFS_LEVELS levels per look at the flag
    PHMRF_HIP(hipMemsetAsync(b->mf_flags + 1, 0, 2 * sizeof(int32_t), st));
    hipLaunchKernelGGL(mf_bfs_init_kernel, dim3(grid), dim3(256), 0, st, p, alpha);
    int level = 2;
    for (;;) {
      const int gate = level;
      for (int q = 0; q < BFS_LEVELS; ++q, ++level) hipLaunchKernelGGL(mf_bfs_level_kernel, dim3(grid), dim3(256), 0, st, p, level, gate);
      PHMRF_HIP(hipMemcpyAsync(b->mf_flags_host, b->mf_flags, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      PHMRF_HIP(hipStreamSynchronize(st));
      if (b->mf_flags_host[1] < level - 1 || level > n + 2) break;        // the last queued level labelled nothing
    }
    hipLaunchKernelGGL(mf_bfs_finish_kernel, dim3(grid), dim3(256), 0, st, p);
    PHMRF_HIP(hipMemcpyAsync(b->mf_flags_host, b->mf_flags, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PHMRF_HIP(hipStreamSynchronize(st));
    done = b->mf_flags_host[2] == 0;                                       // no node with excess can still reach the sink
  }
  PHMRF_HIP(hipGetLastError());
  if (!done) return PHMRF_OK;                                             // (given up: no move; the other move types go on)
  hipLaunchKernelGGL(mf_apply_kernel, dim3(grid), dim3(256), 0, st, p, alpha, b->counters + b->counter_slot,
                     b->tick ? b->stamp : nullptr, b->tick);
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

}  // namespace phmrf
