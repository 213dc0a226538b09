// gfx950 kernels of the label solver's block moves (replacing gco's alpha-beta swap, phylo_hmrf.py:496-498):
//
//  chain_kernel      exact minimisation over ALL labellings of a 1-D segment (<= 63 consecutive nodes of a grid
//                    row / column / diagonal), every other node fixed: Viterbi with the Potts min-trick,
//                      m_t[k] = theta_t[k] + min(m_{t-1}[k], min_j m_{t-1}[j] + beta*w_{t-1,t}).
//                    One wavefront per segment.  Phase 1 maps lane <-> node (theta rows built in LDS exactly
//                    like the ICM tile), phase 2 maps lane <-> label: the min over labels is a DPP wave
//                    reduction, argmin / jump decisions are 64-bit ballots kept one step per lane
//                    (one v_cndmask per value), and the backtrack runs on scalars (v_readlane).
//  component pass    connected components of equal label (one-pass union-find with atomicCAS hooks), then for
//                    every component C and label k the exact energy change of relabelling all of C to k,
//                      dE = sum_{i in C} (u_i(k) - u_i(cur)) - beta * sum_{boundary edges to label k} w,
//                    and the best strictly-improving move of every component that beats all adjacent candidates.
//
// Both are energy non-increasing by construction (simultaneous moves never share an edge).

#include <cstdlib>

#include "common.h"

namespace phmrf {
namespace {

inline int vec_of(int K) { return (K % 4 == 0) ? 4 : (K % 2 == 0 ? 2 : 1); }

// ---- wave-wide min over 64 lanes with DPP; result broadcast through an SGPR -------------------------
#define PHMRF_DPP_MIN(v, ctrl, rmask)                                                                              \
  v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v),                   \
                                                                     __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false)))

__device__ __forceinline__ float wave_min_f32(float v) {
  PHMRF_DPP_MIN(v, 0xB1, 0xf);   // quad_perm [1,0,3,2]
  PHMRF_DPP_MIN(v, 0x4E, 0xf);   // quad_perm [2,3,0,1]
  PHMRF_DPP_MIN(v, 0x141, 0xf);  // row_half_mirror
  PHMRF_DPP_MIN(v, 0x140, 0xf);  // row_mirror          -> every 16-lane row holds its min
  PHMRF_DPP_MIN(v, 0x142, 0xa);  // row_bcast15 into rows 1,3
  PHMRF_DPP_MIN(v, 0x143, 0xc);  // row_bcast31 into rows 2,3 -> lane 63 holds the wave min
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

constexpr float BIG = 3.0e38f;

// -------------------------------------------------------------------------------------------------
// chain segments
// -------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void chain_kernel(const float* __restrict__ logprob, const int32_t* __restrict__ order,
                                                    const int32_t* __restrict__ seg_start,
                                                    const int32_t* __restrict__ seg_len, int nseg, int K, int Kp, int D,
                                                    const int32_t* __restrict__ nbr, const float* __restrict__ wgt,
                                                    uint8_t* __restrict__ labels, float beta,
                                                    unsigned long long* __restrict__ changed, int debug,
                                                    uint16_t* __restrict__ stamp, int tick, uint16_t* __restrict__ memo) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int WPB = blockDim.x >> 6;
  float* tile = lds + wave * 64 * Kp;  // [64][Kp], private to this wave
  unsigned int my_changed = 0;
  const int KV = K / VEC;

  // waves are independent (each has its own LDS tile): wave-level fences only, no block barrier
#define PHMRF_WAVE_SYNC()                                   \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
  } while (0)
  for (int seg = blockIdx.x * WPB + wave; seg < nseg; seg += gridDim.x * WPB) {
    const bool active = true;
    const int p0 = seg_start[seg];
    int len = seg_len[seg];
    // memo: a segment none of whose nodes (nor their neighbours: the stamps are dilated) changed since its last run
    // that found nothing to do sees identical inputs and is skipped.
    if (memo) {
      const int last_quiet = memo[seg];
      int st = (lane < len) ? (int)stamp[order[p0 + lane]] : 0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const int o2 = __shfl_xor(st, off, 64);
        st = o2 > st ? o2 : st;
      }
      if (last_quiet && st < last_quiet) continue;
    }
    const bool ran = len > 0;
    (void)active;
    // phase 1a: theta rows <- -logprob rows (K/VEC lanes per row, coalesced)
    for (int q = lane; q < (debug == 4 ? 0 : len * KV); q += 64) {
      const int r = q / KV;
      const int c = (q - r * KV) * VEC;
      const float* src = logprob + (int64_t)order[p0 + r] * K + c;
      float* dst = tile + r * Kp + c;
      if (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(src);
        dst[0] = -t.x; dst[1] = -t.y; dst[2] = -t.z; dst[3] = -t.w;
      } else if (VEC == 2) {
        const float2 t = *reinterpret_cast<const float2*>(src);
        dst[0] = -t.x; dst[1] = -t.y;
      } else {
        dst[0] = -src[0];
      }
    }
    PHMRF_WAVE_SYNC();
    // phase 1b: lane t subtracts beta*w for every neighbour outside the segment-chain (fixed labels);
    //           the edge to the chain successor becomes the link weight
    const int node = lane < len ? order[p0 + lane] : -1;
    int pv = __shfl_up(node, 1, 64);
    int nx = __shfl_down(node, 1, 64);
    if (lane == 0) pv = -1;
    if (lane >= len - 1) nx = -1;
    float link = 0.f;
    int old = 0;
    if (lane < len && debug != 3) {
      float* row = tile + lane * Kp;
      const int32_t* nb = nbr + (int64_t)node * D;
      const float* wg = wgt + (int64_t)node * D;
      for (int j0 = 0; j0 < D; j0 += 4) {          // 4 neighbour ids + weights per 16-byte load, 4 label gathers in flight
        const int4 cv = *reinterpret_cast<const int4*>(nb + j0);
        const float4 wv = *reinterpret_cast<const float4*>(wg + j0);
        const int cs[4] = {cv.x, cv.y, cv.z, cv.w};
        const float ws[4] = {wv.x, wv.y, wv.z, wv.w};
        int ls[4];
#pragma unroll
        for (int z = 0; z < 4; ++z) ls[z] = (cs[z] >= 0 && cs[z] != nx && cs[z] != pv) ? (int)labels[cs[z]] : -1;
#pragma unroll
        for (int z = 0; z < 4; ++z) {
          if (cs[z] < 0) continue;
          if (cs[z] == nx) { link = ws[z]; continue; }
          if (ls[z] >= 0) row[ls[z]] -= beta * ws[z];
        }
      }
      old = labels[node];
    }
    PHMRF_WAVE_SYNC();
    // phase 2: forward pass, lane <-> label
    const int len2 = (debug == 1 || debug == 3 || debug == 4) ? 0 : len;
    float m = (lane < K && len > 0) ? tile[lane] : BIG;
    unsigned int jm_lo = 0, jm_hi = 0, am_v = 0;  // lane t holds the decisions of step t
    for (int t = 1; t < len2; ++t) {
      const float th = lane < K ? tile[t * Kp + lane] : 0.f;
      const float mmin = wave_min_f32(m);
      const unsigned long long eq = __ballot(m == mmin);
      const int am = __ffsll((long long)eq) - 1;
      const float lk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, link), t - 1));
      const float alt = mmin + beta * lk;
      const bool jp = (lane < K) && (alt < m);
      const unsigned long long jm = __ballot(jp);
      m = lane < K ? th + (jp ? alt : m) : BIG;
      if (lane == t) {  // "write lane t": uniform value into one lane
        jm_lo = (unsigned int)(jm & 0xffffffffu);
        jm_hi = (unsigned int)(jm >> 32);
        am_v = (unsigned int)am;
      }
    }
    // backtrack on scalars
    int newl = old;
    if (len2 > 0 && debug != 2) {
      const float mmin = wave_min_f32(m);
      int cur = __ffsll((long long)__ballot(m == mmin)) - 1;
      for (int t = len - 1; t >= 1; --t) {
        if (lane == t) newl = cur;
        const unsigned long long jm = ((unsigned long long)__builtin_amdgcn_readlane(jm_hi, t) << 32) |
                                      (unsigned long long)__builtin_amdgcn_readlane(jm_lo, t);
        if ((jm >> cur) & 1ull) cur = (int)__builtin_amdgcn_readlane(am_v, t);
      }
      if (lane == 0) newl = cur;
    }
    const bool seg_changed = __any(lane < len && newl != old);
    if (memo && ran && lane == 0) memo[seg] = seg_changed ? (uint16_t)0 : (uint16_t)tick;
    if (lane < len && newl != old) {
      labels[node] = (uint8_t)newl;
      if (stamp) {
        stamp[node] = (uint16_t)tick;
        const int32_t* nb2 = nbr + (int64_t)node * D;
        for (int j = 0; j < D; ++j)
          if (nb2[j] >= 0) stamp[nb2[j]] = (uint16_t)tick;
      }
      ++my_changed;
    }
    PHMRF_WAVE_SYNC();
  }
  unsigned int s = my_changed;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0 && s) atomicAdd(changed, (unsigned long long)s);
}

// an adjacency row into registers: two 16-byte loads when D == 8 (the reference's stencil), so that the loop over the
// row is not one memory round trip per entry
__device__ __forceinline__ void load_row(const int32_t* __restrict__ nbr, int64_t i, int D, int (&nb)[8]) {
  if (D == 8) {
    const int4 a = *reinterpret_cast<const int4*>(nbr + i * 8), b = *reinterpret_cast<const int4*>(nbr + i * 8 + 4);
    nb[0] = a.x; nb[1] = a.y; nb[2] = a.z; nb[3] = a.w; nb[4] = b.x; nb[5] = b.y; nb[6] = b.z; nb[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) nb[j] = j < D ? nbr[i * D + j] : -1;
  }
}

// -------------------------------------------------------------------------------------------------
// connected components of equal label
// -------------------------------------------------------------------------------------------------
// Union-find in one pass (the scheme of ECL-CC): every tree hangs larger ids under smaller ones, so the root of a
// component is its smallest node id.  init: parent = first smaller same-label neighbour (or self); union: for every
// edge to a smaller same-label neighbour, hook the larger root under the smaller with atomicCAS (retry on the value
// the CAS returns); flatten: parent = root.  Reads that miss a concurrent hook only see an OLDER forest (a node that
// looks like a root but no longer is one makes the CAS fail and retry), never a merged one, so the result is exact.
// init: horizontal runs.  Node i is "linked left" when i-1 is one of its neighbours and carries the same label (in a
// grid block: the left cell of the same row).  Inside a 64-node wave chunk the head of a run is found from one ballot
// (nearest lane at or below mine that is not linked left); a run that continues from the previous chunk points at
// i-1 from the chunk's first lane.  This resolves long runs in O(1) per node instead of O(run) finds.
__global__ __launch_bounds__(256) void cc_init_kernel(int32_t* __restrict__ comp, int64_t n, int D,
                                                      const int32_t* __restrict__ nbr, const uint8_t* __restrict__ labels,
                                                      const int* __restrict__ gate) {
  if (gate && *gate == 0) return;      // (this labelling is already in place: launch_component_prepare)
  const int lane = threadIdx.x & 63;
  for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; base < n; base += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = base + lane;
    bool linked = false;
    if (i < n && i > 0) {
      const int l = labels[i], lw = labels[i - 1];
      if (D <= 8) {
        int nb[8];
        load_row(nbr, i, D, nb);
        bool adj = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) adj = adj || nb[j] == (int)(i - 1);
        linked = adj && lw == l;
      } else {
        const int32_t* nb = nbr + i * D;
        for (int j = 0; j < D; ++j)
          if (nb[j] == (int)(i - 1)) linked = lw == l;
      }
    }
    const unsigned long long starts = ~__ballot(linked);              // lanes that begin a run (or are out of range)
    const unsigned long long below = starts & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int head = 63 - __clzll((long long)below);                   // nearest run start at or below my lane; -1 if none
    if (i < n) comp[i] = head >= 0 ? (int)(base + head) : (lane == 0 ? (int)i : (int)base);
    // head < 0: the run started in an earlier chunk.  Lane 0 then points at i-1, every other lane of the run at lane 0.
    if (i < n && head < 0) comp[i] = lane == 0 ? (int)(i - 1) : (int)base;
  }
}

__device__ __forceinline__ int cc_find(int32_t* comp, int x) {
  int p = comp[x];
  while (p != x) {
    const int gp = comp[p];
    if (gp != p) comp[x] = gp;   // path halving (x is not a root and never becomes one again)
    x = p;
    p = gp;
  }
  return x;
}

// grid != 0 (stencil graphs): a node that is linked left shares its up-left and up neighbours with its left run-mate
// (they are that mate's up and up-right), so it only has to union with its LARGEST remaining smaller neighbour
// (up-right in an 8-neighbourhood, up in a 4-neighbourhood).  This removes two thirds of the finds and most of the
// atomics that would all hit the same pair of roots.
__device__ __forceinline__ void cc_hook(int32_t* comp, int& ri, int i, int c) {
  if (ri < 0) ri = cc_find(comp, i);
  int rc = cc_find(comp, c);
  while (ri != rc) {
    int hi = ri > rc ? ri : rc, lo = ri > rc ? rc : ri;
    const int old = atomicCAS(comp + hi, hi, lo);
    if (old == hi) {            // hooked: both ends now share the root lo
      ri = lo;
      rc = lo;
    } else {                    // hi was no longer a root: continue from where it points
      if (hi == ri) ri = cc_find(comp, old); else rc = cc_find(comp, old);
    }
  }
}

__global__ void cc_union_kernel(int32_t* __restrict__ comp, int64_t n, int D, const int32_t* __restrict__ nbr,
                                const uint8_t* __restrict__ labels, int grid, const int* __restrict__ gate) {
  if (gate && *gate == 0) return;      // (this labelling is already in place: launch_component_prepare)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = labels[i];
    if (D <= 8) {
      // the row and the labels of its smaller entries in registers, all loads side by side (the general form below
      // walks the row entry by entry: one memory round trip each)
      int nb[8], ln[8];
      load_row(nbr, i, D, nb);
#pragma unroll
      for (int j = 0; j < 8; ++j) ln[j] = (nb[j] >= 0 && nb[j] < (int)i) ? (int)labels[nb[j]] : -1;
      bool linked = false;
      int last = -1, n_above = 0, lab_last = -1, lab_before = -1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = nb[j];
        if (c >= 0 && c == (int)i - 1) linked = ln[j] == l;
        if (c >= 0 && c < (int)i - 1) {              // rows are ascending: the largest smaller neighbour below i-1
          lab_before = lab_last;
          lab_last = ln[j];
          last = j;
          ++n_above;
        }
      }
      const bool only_last = grid && linked;
      if (only_last && grid == 8) {                  // (see the general form for why)
        if (n_above < 3) continue;
        if (lab_before == l) continue;
      }
      const int first = only_last ? (last < 0 ? 8 : last) : 0;
      int ri = -1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = nb[j];
        if (j >= first && c >= 0 && c < (int)i - 1 && ln[j] == l) cc_hook(comp, ri, (int)i, c);
      }
      continue;
    }
    const int32_t* nb = nbr + i * D;
    bool linked = false;
    int last = -1, n_above = 0;
    for (int j = 0; j < D; ++j) {
      const int c = nb[j];
      if (c >= 0 && c == (int)i - 1) linked = labels[c] == l;
      if (c >= 0 && c < (int)i - 1) {              // rows are ascending: the largest smaller neighbour below i-1
        last = j;
        ++n_above;
      }
    }
    const bool only_last = grid && linked;
    if (only_last && grid == 8) {
      // 8-neighbour stencil: `last` is my up-right cell when the entry before it is its left neighbour (my up cell),
      // otherwise it IS my up cell (no up-right at the right border).  My up cell is also the up-right cell of my left
      // run-mate: if it carries the label, the mate (or, by induction, an earlier cell of the run) already joins the
      // run to that upper run, and my up-right cell then belongs to the same upper run.  So a linked cell only has
      // to act where an upper run STARTS at its up-right cell.
      // (a linked cell always has its up-left and up cells above it; a third one is the up-right cell)
      if (n_above < 3) continue;
      if (labels[nb[last - 1]] == l) continue;
    }
    int ri = -1;
    for (int j = only_last ? (last < 0 ? D : last) : 0; j < D; ++j) {
      const int c = nb[j];
      if (c < 0 || c >= (int)i - 1) { if (c >= (int)i - 1) break; continue; }
      if (labels[c] != l) continue;
      cc_hook(comp, ri, (int)i, c);
    }
  }
}

// The same two kernels on a grid block whose adjacency is the complete 8-neighbour stencil: the neighbours by geometry
// (one byte of label each, rows of the label image the wave shares) instead of a 32-byte adjacency row per node.  Same
// hooks as above, hence the same forest up to the order of the atomics and the same roots (a root is the smallest id).
__global__ __launch_bounds__(256) void cc_init_grid_kernel(int32_t* __restrict__ comp, int64_t n, int W, int diagonal,
                                                           const uint8_t* __restrict__ labels, const int* __restrict__ gate) {
  if (gate && *gate == 0) return;      // (this labelling is already in place: launch_component_prepare)
  const int lane = threadIdx.x & 63;
  for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; base < n; base += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = base + lane;
    bool linked = false;
    if (v < n && v > 0) {
      int i, j;
      grid_coords(v, W, diagonal, &i, &j);
      linked = j - 1 >= (diagonal ? i : 0) && labels[v - 1] == labels[v];
    }
    const unsigned long long starts = ~__ballot(linked);
    const unsigned long long below = starts & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int head = 63 - __clzll((long long)below);
    if (v < n) comp[v] = head >= 0 ? (int)(base + head) : (lane == 0 ? (int)(v - 1) : (int)base);
  }
}

__global__ void cc_union_grid_kernel(int32_t* __restrict__ comp, int64_t n, int W, int diagonal,
                                     const uint8_t* __restrict__ labels, const int* __restrict__ gate) {
  if (gate && *gate == 0) return;      // (this labelling is already in place: launch_component_prepare)
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    int i, j;
    grid_coords(v, W, diagonal, &i, &j);
    if (i == 0) continue;                                   // no row above: the runs of init are the components so far
    const int64_t up = grid_row_base(i - 1, W, diagonal);
    const bool has_w = j - 1 >= (diagonal ? i : 0), has_nw = j - 1 >= 0, has_ne = j + 1 < W;
    const int l = labels[v];
    const int lw = labels[has_w ? v - 1 : v], lnw = labels[has_nw ? up + j - 1 : v], ln = labels[up + j];
    const int lne = labels[has_ne ? up + j + 1 : v];
    int ri = -1;
    if (has_w && lw == l) {       // linked left: only where an upper run STARTS at my up-right cell (see cc_union_kernel)
      if (has_ne && ln != l && lne == l) cc_hook(comp, ri, (int)v, (int)(up + j + 1));
      continue;
    }
    if (has_nw && lnw == l) cc_hook(comp, ri, (int)v, (int)(up + j - 1));
    if (ln == l) cc_hook(comp, ri, (int)v, (int)(up + j));
    if (has_ne && lne == l) cc_hook(comp, ri, (int)v, (int)(up + j + 1));
  }
}

// flatten, and clear what the table / decide / block kernels accumulate into: only the ROOT rows of the move table are
// ever used, so only those are zeroed (instead of a memset of n*K floats per pass)
// Deterministic mode (PHMRF_DETERMINISTIC=1): the per-component sums are accumulated as 2^-16 fixed-point integers --
// integer addition is associative, so the table, the moves it decides and hence the labelling no longer depend on the
// order in which the atomics land.  (f32 atomics, the default: the labelling of two runs can differ in a few nodes.)
constexpr float TAB_FIX = 65536.f;
__device__ __forceinline__ void tab_add(float* tab, long long* tab64, int64_t idx, float acc) {
  if (tab64) atomicAdd(reinterpret_cast<unsigned long long*>(tab64 + idx), (unsigned long long)__float2ll_rn(acc * TAB_FIX));
  else atomicAdd(tab + idx, acc);
}

__global__ void cc_flatten_kernel(int32_t* __restrict__ comp, int64_t n, float* __restrict__ tab, int K,
                                  long long* __restrict__ tab64, const int* __restrict__ gate) {
  if (gate && *gate == 0) return;      // (this labelling is already in place: launch_component_prepare)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int c = comp[i];
    int p = comp[c];
    while (p != c) {
      c = p;
      p = comp[c];
    }
    comp[i] = c;
    if (c == (int)i) {
      if (tab64) {
        long long* row = tab64 + i * K;
        for (int k = 0; k < K; ++k) row[k] = 0ll;
      } else {
        float* row = tab + i * K;
        for (int k = 0; k < K; ++k) row[k] = 0.f;
      }
    }
  }
}

// The labelling the prepared components were computed from against the labelling now: *stale = 1 where they differ (the
// prepare step left it 0), and the gated kernels above run after all.  16 labels per thread and trip.
__global__ __launch_bounds__(256) void cc_compare_kernel(const uint8_t* __restrict__ labels, const uint8_t* __restrict__ seen,
                                                         int64_t n, int* __restrict__ stale) {
  bool differ = false;
  const int64_t n16 = n >> 4;
  const uint4* a = reinterpret_cast<const uint4*>(labels);
  const uint4* c = reinterpret_cast<const uint4*>(seen);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n16; q += (int64_t)gridDim.x * blockDim.x) {
    const uint4 x = a[q], y = c[q];
    differ = differ || x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w;
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n16 << 4) + threadIdx.x; i < n; i += blockDim.x) differ = differ || labels[i] != seen[i];
  if (__any(differ) && (threadIdx.x & 63) == 0) *stale = 1;
}

// -------------------------------------------------------------------------------------------------
// per-component move table: tab[root,k] = sum_{i in C} -logprob[i,k] - beta * sum_{boundary edges to label k} w
// Node tile like ICM; the tile rows are then summed over runs of equal root (consecutive node ids mostly
// share a component) before the atomics, so a large component receives one add per run, not per node.
// -------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void comp_table_kernel(const float* __restrict__ logprob, int64_t n, int K, int Kp, int D,
                                                         const int32_t* __restrict__ nbr, const float* __restrict__ wgt,
                                                         const uint8_t* __restrict__ labels,
                                                         const int32_t* __restrict__ comp, float beta,
                                                         float* __restrict__ tab, long long* __restrict__ tab64) {
  extern __shared__ float lds[];
  const int TB = blockDim.x;
  float* tile = lds;                                          // [TB][Kp]
  int32_t* roots = reinterpret_cast<int32_t*>(lds + TB * Kp);  // [TB]
  const int KV = K / VEC;
  for (int64_t base = (int64_t)blockIdx.x * TB; base < n; base += (int64_t)gridDim.x * TB) {
    const int64_t rem = n - base;
    const int rows = rem < TB ? (int)rem : TB;
    for (int q = threadIdx.x; q < rows * KV; q += TB) {
      const int r = q / KV;
      const int c = (q - r * KV) * VEC;
      const float* src = logprob + (base + r) * K + c;
      float* dst = tile + r * Kp + c;
      if (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(src);
        dst[0] = -t.x; dst[1] = -t.y; dst[2] = -t.z; dst[3] = -t.w;
      } else if (VEC == 2) {
        const float2 t = *reinterpret_cast<const float2*>(src);
        dst[0] = -t.x; dst[1] = -t.y;
      } else {
        dst[0] = -src[0];
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < rows) {
      const int64_t i = base + threadIdx.x;
      roots[threadIdx.x] = comp[i];
      float* row = tile + threadIdx.x * Kp;
      const int32_t* nb = nbr + i * D;
      const float* wg = wgt + i * D;
      const int li = labels[i];
      for (int j = 0; j < D; ++j) {
        const int c = nb[j];
        if (c < 0) continue;
        // a neighbour belongs to another component exactly when it carries another label (components are the connected
        // sets of equal label): one byte from a cache line the wave shares instead of a 4-byte root id per neighbour
        const int lc = labels[c];
        if (lc != li) row[lc] -= beta * wg[j];
      }
    }
    __syncthreads();
    // column k, row slice s: walk the slice, flush one atomic per run of equal root
    const int nslice = TB / K;
    const int k = threadIdx.x % K, s = threadIdx.x / K;
    if (s < nslice) {
      const int per = (rows + nslice - 1) / nslice;
      const int r0 = s * per, r1 = (r0 + per < rows) ? r0 + per : rows;
      if (r0 < r1) {
        int root = roots[r0];
        float acc = 0.f;
        for (int r = r0; r < r1; ++r) {
          const int rr = roots[r];
          if (rr != root) {
            tab_add(tab, tab64, (int64_t)root * K + k, acc);
            acc = 0.f;
            root = rr;
          }
          acc += tile[r * Kp + k];
        }
        tab_add(tab, tab64, (int64_t)root * K + k, acc);
      }
    }
    __syncthreads();
  }
}

// The same table on a grid block from the tables of the strip moves: the unary terms from the label-major planes
// (column k of the tile is a coalesced read, no transposition), the boundary weights from the forward-edge records by
// geometry; every load of a thread is issued before its uses (planes four at a time, the eight neighbours gathered
// first).  The neighbours come in the order of the adjacency rows, so the two kernels agree up to the order of the atomics.
__global__ __launch_bounds__(256) void comp_table_grid_kernel(const float* __restrict__ uT, int64_t n, int K, int Kp, int H,
                                                              int W, int diagonal, const float4* __restrict__ fwd_w,
                                                              const uint8_t* __restrict__ labels,
                                                              const int32_t* __restrict__ comp, float beta,
                                                              float* __restrict__ tab, long long* __restrict__ tab64) {
  extern __shared__ float lds[];
  const int TB = blockDim.x;
  float* tile = lds;                                          // [TB][Kp]
  int32_t* roots = reinterpret_cast<int32_t*>(lds + TB * Kp);  // [TB]
  for (int64_t base = (int64_t)blockIdx.x * TB; base < n; base += (int64_t)gridDim.x * TB) {
    const int64_t rem = n - base;
    const int rows = rem < TB ? (int)rem : TB;
    if ((int)threadIdx.x < rows) {
      const int64_t v = base + threadIdx.x;
      float* row = tile + threadIdx.x * Kp;
      int i, j;
      grid_coords(v, W, diagonal, &i, &j);
      int64_t nc[8];
      float nw[8];
      grid_gather_neighbours(v, i, j, H, W, diagonal, fwd_w, nc, nw);
      int nl[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) nl[d] = labels[nc[d]];
      const int li = labels[v];
      roots[threadIdx.x] = comp[v];
      int k0 = 0;
      for (; k0 + 4 <= K; k0 += 4) {
        const float a0 = uT[(int64_t)k0 * n + v], a1 = uT[(int64_t)(k0 + 1) * n + v];
        const float a2 = uT[(int64_t)(k0 + 2) * n + v], a3 = uT[(int64_t)(k0 + 3) * n + v];
        row[k0] = a0; row[k0 + 1] = a1; row[k0 + 2] = a2; row[k0 + 3] = a3;
      }
      for (; k0 < K; ++k0) row[k0] = uT[(int64_t)k0 * n + v];
#pragma unroll
      for (int d = 0; d < 8; ++d)
        if (nl[d] != li) row[nl[d]] -= beta * nw[d];      // (an absent neighbour is the node itself: same label, skipped)
    }
    __syncthreads();
    // column k, row slice s: walk the slice, flush one atomic per run of equal root
    const int nslice = TB / K;
    const int k = threadIdx.x % K, s = threadIdx.x / K;
    if (s < nslice) {
      const int per = (rows + nslice - 1) / nslice;
      const int r0 = s * per, r1 = (r0 + per < rows) ? r0 + per : rows;
      if (r0 < r1) {
        int root = roots[r0];
        float acc = 0.f;
        for (int r = r0; r < r1; ++r) {
          const int rr = roots[r];
          if (rr != root) {
            tab_add(tab, tab64, (int64_t)root * K + k, acc);
            acc = 0.f;
            root = rr;
          }
          acc += tile[r * Kp + k];
        }
        tab_add(tab, tab64, (int64_t)root * K + k, acc);
      }
    }
    __syncthreads();
  }
}

// best strictly-improving label per component (computed at the root node); gain = dE < 0 or 0
__global__ void comp_decide_kernel(const float* __restrict__ tab, int64_t n, int K, const uint8_t* __restrict__ labels,
                                   const int32_t* __restrict__ comp, int32_t* __restrict__ best,
                                   float* __restrict__ gain, const long long* __restrict__ tab64) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float g = 0.f;
    int bk = -1;
    if (comp[i] == (int32_t)i && tab64) {            // deterministic mode: fixed-point sums
      const long long* row = tab64 + i * K;
      const int cur = labels[i];
      const float tc = (float)row[cur] * (1.f / TAB_FIX);
      float bv = tc;
      for (int k = 0; k < K; ++k) {
        const float v = (float)row[k] * (1.f / TAB_FIX);
        if (v < bv) { bv = v; bk = k; }
      }
      const float margin = 1e-5f * fabsf(tc) + 1e-6f;
      if (bk >= 0 && bv < tc - margin) g = bv - tc; else bk = -1;
    } else if (comp[i] == (int32_t)i) {
      const float* row = tab + i * K;
      const int cur = labels[i];
      const float tc = row[cur];
      float bv = tc;
      int k = 0;
      for (; k + 4 <= K; k += 4) {                  // (four loads side by side; same scan order)
        const float v0 = row[k], v1 = row[k + 1], v2 = row[k + 2], v3 = row[k + 3];
        if (v0 < bv) { bv = v0; bk = k; }
        if (v1 < bv) { bv = v1; bk = k + 1; }
        if (v2 < bv) { bv = v2; bk = k + 2; }
        if (v3 < bv) { bv = v3; bk = k + 3; }
      }
      for (; k < K; ++k) {
        const float v = row[k];
        if (v < bv) { bv = v; bk = k; }
      }
      // float atomics sum in arbitrary order: demand a margin well above their rounding noise
      const float margin = 1e-5f * fabsf(tc) + 1e-6f;
      if (bk >= 0 && bv < tc - margin) g = bv - tc; else bk = -1;
    }
    if (comp[i] == (int32_t)i) {        // (best / gain are read at roots only: comp_block, comp_apply)
      best[i] = bk;
      gain[i] = g;
    }
  }
}

// a candidate component is blocked when an adjacent candidate has a better (more negative) gain; ties by root id
__global__ void comp_block_kernel(int64_t n, int D, const int32_t* __restrict__ nbr, const int32_t* __restrict__ comp,
                                  const float* __restrict__ gain, int32_t* __restrict__ best) {
  // (a blocked component loses its decision: best[root] = -1 -- nothing here reads `best`, and comp_apply then needs the
  //  root's decision alone, not a second flag behind the same root: one dependent read fewer per node)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = comp[i];
    const float gi = gain[ci];
    if (gi >= 0.f) continue;
    if (D <= 8) {
      int nb[8], cj[8];
      load_row(nbr, i, D, nb);
#pragma unroll
      for (int j = 0; j < 8; ++j) cj[j] = nb[j] >= 0 ? comp[nb[j]] : ci;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (cj[j] == ci) continue;
        const float gj = gain[cj[j]];
        if (gj < gi || (gj == gi && cj[j] < ci)) best[ci] = -1;
      }
      continue;
    }
    const int32_t* nb = nbr + i * D;
    for (int j = 0; j < D; ++j) {
      const int c = nb[j];
      if (c < 0) continue;
      const int cj = comp[c];
      if (cj == ci) continue;
      const float gj = gain[cj];
      if (gj < gi || (gj == gi && cj < ci)) best[ci] = -1;
    }
  }
}

__global__ void comp_apply_kernel(int64_t n, const int32_t* __restrict__ comp, const int32_t* __restrict__ best,
                                  uint8_t* __restrict__ labels,
                                  unsigned long long* __restrict__ changed, uint16_t* __restrict__ stamp, int tick,
                                  const int32_t* __restrict__ nbr, int D) {
  unsigned int mine = 0;
  // four nodes per thread and trip, their two dependent reads (root, its decision -- comp_block has withdrawn the decisions of
  // blocked components) side by side: the kernel is two memory round trips per node and nothing else
  constexpr int U = 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += U * stride) {
    int ci[U], bk[U];
    bool go[U];
#pragma unroll
    for (int u = 0; u < U; ++u) ci[u] = i0 + u * stride < n ? comp[i0 + u * stride] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) bk[u] = ci[u] >= 0 ? best[ci[u]] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) go[u] = bk[u] >= 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!go[u]) continue;
      const int64_t i = i0 + u * stride;
      labels[i] = (uint8_t)bk[u];
      if (stamp) {
        stamp[i] = (uint16_t)tick;
        if (D <= 8) {
          int nb[8];
          load_row(nbr, i, D, nb);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (nb[j] >= 0) stamp[nb[j]] = (uint16_t)tick;
        } else {
          const int32_t* nb2 = nbr + i * D;
          for (int j = 0; j < D; ++j)
            if (nb2[j] >= 0) stamp[nb2[j]] = (uint16_t)tick;
        }
      }
      ++mine;
    }
  }
  unsigned int s = mine;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(changed, (unsigned long long)s);
}

inline int grid1d(int64_t n, int tb = 256, int cap = 256 * 16) {
  int64_t g = (n + tb - 1) / tb;
  if (g > cap) g = cap;
  return g < 1 ? 1 : (int)g;
}

template <typename T>
int ensure(T** p, size_t count) {
  if (*p) return PHMRF_OK;
  PHMRF_HIP(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
  return PHMRF_OK;
}

}  // namespace

static int chain_debug() {   // timing experiments only (PHMRF_CHAIN_DEBUG=1: phase 1 only, 2: no backtrack, 3: phase 1a only, 4: 1b only)
  static int v = -1;
  if (v < 0) {
    const char* e = PHMRF_DEV_ENV("PHMRF_CHAIN_DEBUG");
    v = e ? atoi(e) : 0;
  }
  return v;
}

int launch_chain_colour(const phmrf_block* b, float beta, int family, int colour, int phase) {
  const ChainFamily& f = b->families[family];
  const int nseg = f.nseg[phase][colour];
  if (nseg <= 0) return PHMRF_OK;
  const int K = b->K, Kp = padded_k(K);
  // one-wave workgroups: the waves are independent (a wave owns a segment), so this is the finest dispatch grain and no
  // finished wave's slot waits for its siblings (measured on the strip kernels: -20 % stream time)
  const int TB = 64, WPB = TB / 64;
  const size_t lds = (size_t)WPB * 64 * Kp * sizeof(float);
  int grid = (nseg + WPB - 1) / WPB;
  if (grid > 256 * 64) grid = 256 * 64;
#define PHMRF_LAUNCH_CHAIN(VEC_)                                                                                      \
  hipLaunchKernelGGL((chain_kernel<VEC_>), dim3(grid), dim3(TB), lds, b->stream, b->logprob, f.nodes,                  \
                     f.seg_start[phase][colour], f.seg_len[phase][colour], nseg, K, Kp, b->D, b->nbr, b->wgt, b->labels, \
                     beta, b->counters + b->counter_slot, chain_debug(), b->tick ? b->stamp : nullptr, b->tick, \
                     (b->tick && f.memo[phase][colour]) ? f.memo[phase][colour] : nullptr)
  switch (vec_of(K)) {
    case 4: PHMRF_LAUNCH_CHAIN(4); break;
    case 2: PHMRF_LAUNCH_CHAIN(2); break;
    default: PHMRF_LAUNCH_CHAIN(1); break;
  }
#undef PHMRF_LAUNCH_CHAIN
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

// The part of the component pass that depends on the labels alone -- the connected components of equal label, flattened,
// with the root rows of the move table cleared -- queued ahead of the solve that will use it (phmrf_block_prepare_components:
// between two E-steps the GPU waits for the host's M-step).  It keeps a copy of the labels it saw; the next component pass
// compares them with the labels it finds (cc_compare_kernel, on the device) and runs the three kernels again only if they
// differ -- the warm start may have installed another labelling, a tile may have received a halo row --, so the result is
// that of the unprepared pass whatever happened in between.
static int launch_cc(phmrf_block* b, const int* gate) {
  const int64_t n = b->n;
  const int K = b->K, D = b->D;
  hipStream_t st = b->stream;
  const int g = grid1d(n);
  long long* const tab64 = b->deterministic ? b->comp_tab64 : nullptr;
  static const bool cc_rows = PHMRF_DEV_ENV("PHMRF_CC_ROWS") != nullptr;       // development: the adjacency-row form on grid blocks
  if (b->has_grid && b->grid_complete && b->num_neighbor == 8 && D == 8 && !cc_rows) {
    hipLaunchKernelGGL(cc_init_grid_kernel, dim3(g), dim3(256), 0, st, b->comp, n, b->W, b->diagonal, b->labels, gate);
    hipLaunchKernelGGL(cc_union_grid_kernel, dim3(g), dim3(256), 0, st, b->comp, n, b->W, b->diagonal, b->labels, gate);
  } else {
    hipLaunchKernelGGL(cc_init_kernel, dim3(g), dim3(256), 0, st, b->comp, n, D, b->nbr, b->labels, gate);
    hipLaunchKernelGGL(cc_union_kernel, dim3(g), dim3(256), 0, st, b->comp, n, D, b->nbr, b->labels, b->has_grid ? b->num_neighbor : 0, gate);
  }
  hipLaunchKernelGGL(cc_flatten_kernel, dim3(g), dim3(256), 0, st, b->comp, n, b->comp_tab, K, tab64, gate);
  return PHMRF_OK;
}

static int ensure_component_buffers(phmrf_block* b) {
  const int64_t n = b->n;
  PHMRF_TRY(ensure(&b->comp, (size_t)n));
  if (b->deterministic) PHMRF_TRY(ensure(&b->comp_tab64, (size_t)n * b->K));
  else PHMRF_TRY(ensure(&b->comp_tab, (size_t)n * b->K));
  PHMRF_TRY(ensure(&b->comp_best, (size_t)n));
  PHMRF_TRY(ensure(&b->comp_gain, (size_t)n));
  return PHMRF_OK;
}

int launch_component_prepare(phmrf_block* b) {
  PHMRF_TRY(ensure_component_buffers(b));
  PHMRF_TRY(ensure(&b->cc_seen, (size_t)b->n + 16));
  PHMRF_TRY(ensure(&b->cc_stale, (size_t)1));
  PHMRF_TRY(launch_cc(b, nullptr));
  PHMRF_HIP(hipMemcpyAsync(b->cc_seen, b->labels, (size_t)b->n, hipMemcpyDeviceToDevice, b->stream));
  PHMRF_HIP(hipMemsetAsync(b->cc_stale, 0, sizeof(int), b->stream));
  PHMRF_HIP(hipGetLastError());
  b->cc_prepared = true;
  return PHMRF_OK;
}

int launch_component_pass(phmrf_block* b, float beta) {
  const int64_t n = b->n;
  const int K = b->K, Kp = padded_k(K), D = b->D;
  PHMRF_TRY(ensure_component_buffers(b));
  long long* const tab64 = b->deterministic ? b->comp_tab64 : nullptr;
  hipStream_t st = b->stream;
  const int g = grid1d(n);
  if (b->cc_prepared) {            // (once: whatever this pass moves makes the prepared labelling history)
    b->cc_prepared = false;
    hipLaunchKernelGGL(cc_compare_kernel, dim3(grid1d(n, 256 * 16, 2048)), dim3(256), 0, st, b->labels, b->cc_seen, n, b->cc_stale);
    PHMRF_TRY(launch_cc(b, b->cc_stale));
  } else {
    PHMRF_TRY(launch_cc(b, nullptr));
  }
  const bool grid_tables = b->has_grid && b->num_neighbor == 8 && D == 8 && b->fwd_w && b->uT && b->uT_valid;
  if (grid_tables) {
    const int TB = tile_threads(K);
    const size_t lds = (size_t)TB * Kp * sizeof(float) + (size_t)TB * sizeof(int32_t);
    hipLaunchKernelGGL(comp_table_grid_kernel, dim3(grid1d(n, TB)), dim3(TB), lds, st, b->uT, n, K, Kp, b->H, b->W, b->diagonal,
                       b->fwd_w, b->labels, b->comp, beta, b->comp_tab, tab64);
  } else {
    const int TB = tile_threads(K);
    const size_t lds = (size_t)TB * Kp * sizeof(float) + (size_t)TB * sizeof(int32_t);
    const int grid = grid1d(n, TB);
#define PHMRF_LAUNCH_TAB(VEC_)                                                                                       \
  hipLaunchKernelGGL((comp_table_kernel<VEC_>), dim3(grid), dim3(TB), lds, st, b->logprob, n, K, Kp, D, b->nbr, b->wgt, \
                     b->labels, b->comp, beta, b->comp_tab, tab64)
    switch (vec_of(K)) {
      case 4: PHMRF_LAUNCH_TAB(4); break;
      case 2: PHMRF_LAUNCH_TAB(2); break;
      default: PHMRF_LAUNCH_TAB(1); break;
    }
#undef PHMRF_LAUNCH_TAB
  }
  hipLaunchKernelGGL(comp_decide_kernel, dim3(g), dim3(256), 0, st, b->comp_tab, n, K, b->labels, b->comp, b->comp_best,
                     b->comp_gain, tab64);
  hipLaunchKernelGGL(comp_block_kernel, dim3(g), dim3(256), 0, st, n, D, b->nbr, b->comp, b->comp_gain, b->comp_best);
  hipLaunchKernelGGL(comp_apply_kernel, dim3(g), dim3(256), 0, st, n, b->comp, b->comp_best, b->labels,
                     b->counters + b->counter_slot, b->tick ? b->stamp : nullptr, b->tick, b->nbr, D);
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

}  // namespace phmrf
