// gfx950 kernel of the 2-D block move: EXACT binary fusion on strips.
//
// A strip is STRIP_H = 5 grid rows (or columns) by up to 63 cells along the other axis.  Every node of the strip
// chooses between its current label l_i and a proposal p_i (p_i = a constant alpha: an alpha-expansion restricted
// to the strip; p_i = the node's best alternative label: a fusion move that shifts region fronts); all nodes
// outside the strip are fixed.  The binary problem is solved exactly by a profile ("broken line") dynamic
// programme over the cells in column-major order: the state is the choice bit of the last STRIP_H+1 = 6 cells
// (newest in bit 0), i.e. 64 states = ONE WAVEFRONT with one state per lane.  A cell step is
//     new[s'] = min_{d in {0,1}} old[(s' >> 1) | (d << 5)] + cost(cell, bit s'&1, neighbour bits in the old state)
// where the four already-visited 8-neighbours of the cell sit at fixed profile positions (up: bit 0,
// left-up: bit 5 = the bit being dropped, left: bit 4, left-down: bit 3).  The two predecessor values come from
// two ds_bpermute shuffles, the decision bits are a 64-bit ballot per step kept in LDS, the backtrack is scalar.
// No submodularity is needed (the DP is exact for any 2x2 tables), so fusion proposals are as valid as expansions.
// Strips of one pass are separated by one fixed row and one fixed column, so simultaneous moves share no edge:
// the pass never raises the energy.   (Model: oracle/mrf_moves.strip_fusion.)

#include "common.h"

namespace phmrf {
namespace {

constexpr int SH = 5;          // strip rows
constexpr int SL = 63;         // strip columns per segment
constexpr int NCELL_MAX = SH * SL;
constexpr int REC = 8;         // dwords per cell record in LDS
constexpr int CELL_PAD = 320;  // records per wave
constexpr float BIG = 1.0e30f;

struct StripGeom {
  int H, W, diagonal, orient, shift_r, shift_c, Hs, Ws, nbands, nsegs;
};

__device__ __forceinline__ int strip_node(const StripGeom& g, int sr, int sc) {
  if (sr < 0 || sr >= g.Hs || sc < 0 || sc >= g.Ws) return -1;
  const int i = g.orient ? sc : sr, j = g.orient ? sr : sc;
  if (g.diagonal) {
    if (i > j) return -1;
    return i * g.W - (i * (i - 1)) / 2 + (j - i);
  }
  return i * g.W + j;
}

#define PHMRF_DPP_MIN(v, ctrl, rmask)                                                                            \
  v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v),                 \
                                                                     __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false)))

__device__ __forceinline__ float wave_min_f32(float v) {
  PHMRF_DPP_MIN(v, 0xB1, 0xf);
  PHMRF_DPP_MIN(v, 0x4E, 0xf);
  PHMRF_DPP_MIN(v, 0x141, 0xf);
  PHMRF_DPP_MIN(v, 0x140, 0xf);
  PHMRF_DPP_MIN(v, 0x142, 0xa);
  PHMRF_DPP_MIN(v, 0x143, 0xc);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__global__ __launch_bounds__(256) void strip_kernel(StripGeom g, const float* __restrict__ logprob, int K, int D,
                                                    const int32_t* __restrict__ nbr, const float* __restrict__ wgt,
                                                    uint8_t* __restrict__ labels, const uint8_t* __restrict__ prop,
                                                    int alpha, float beta, unsigned long long* __restrict__ changed) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int WPB = blockDim.x >> 6;
  float* rec = lds + wave * (CELL_PAD * REC + CELL_PAD * 2);
  int* reci = reinterpret_cast<int*>(rec);
  unsigned long long* bal = reinterpret_cast<unsigned long long*>(rec + CELL_PAD * REC);
  const int nstrips = g.nbands * g.nsegs;
  unsigned int my_changed = 0;

  for (int s0 = blockIdx.x * WPB; s0 < nstrips; s0 += gridDim.x * WPB) {
    const int strip = s0 + wave;
    const bool active = strip < nstrips;
    const int bnd = active ? strip / g.nsegs : 0;
    const int seg = active ? strip - bnd * g.nsegs : 0;
    const int rs0 = bnd * (SH + 1) - g.shift_r;
    const int cs0 = seg * 64 - g.shift_c;
    const int ca = cs0 > 0 ? cs0 : 0;
    const int cb = (cs0 + SL < g.Ws) ? cs0 + SL : g.Ws;
    const int ncols = (active && cb > ca) ? cb - ca : 0;
    const int ncell = ncols * SH;

    // ---- phase 1: lane <-> cell: unary costs against the fixed outside, weights / label relations to the four
    //      already-visited in-strip neighbours.  A cell whose unary loss exceeds the total weight of its edges can
    //      never be part of an improving move (adding it to ANY switched set raises the energy), so it is pinned
    //      (c1 = BIG) and the DP only has to span the pinned-free range [t_lo, t_hi + SH + 1].
    int t_lo = NCELL_MAX, t_hi = -1;
    for (int t0 = 0; t0 < ncell; t0 += 64) {
      const int t = t0 + lane;
      bool sw = false;
      if (t < ncell) {
      const int cc = t / SH, rr = t - cc * SH;
      const int sr = rs0 + rr, sc = ca + cc;
      const int node = strip_node(g, sr, sc);
      float c0 = 0.f, c1 = BIG, w4[4] = {0.f, 0.f, 0.f, 0.f};
      int bits = 0;
      if (node >= 0) {
        int ids[8];
        const bool up = rr > 0, dn = rr < SH - 1, lf = cc > 0, rt = cc < ncols - 1;
        ids[0] = up ? strip_node(g, sr - 1, sc) : -1;              // up
        ids[1] = (up && lf) ? strip_node(g, sr - 1, sc - 1) : -1;  // left-up
        ids[2] = lf ? strip_node(g, sr, sc - 1) : -1;              // left
        ids[3] = (dn && lf) ? strip_node(g, sr + 1, sc - 1) : -1;  // left-down
        ids[4] = dn ? strip_node(g, sr + 1, sc) : -1;              // forward neighbours (handled from their side)
        ids[5] = (up && rt) ? strip_node(g, sr - 1, sc + 1) : -1;
        ids[6] = rt ? strip_node(g, sr, sc + 1) : -1;
        ids[7] = (dn && rt) ? strip_node(g, sr + 1, sc + 1) : -1;
        const int l = labels[node];
        const int p = prop ? (int)prop[node] : alpha;
        const bool can = p != l;
        c0 = -logprob[(int64_t)node * K + l];
        c1 = can ? -logprob[(int64_t)node * K + p] : BIG;
        const int32_t* nb = nbr + (int64_t)node * D;
        const float* wg = wgt + (int64_t)node * D;
        float wtot = 0.f;
        const float du = c1 - c0;
        for (int j = 0; j < D; ++j) {
          const int c = nb[j];
          if (c < 0) continue;
          const float w = beta * wg[j];
          wtot += w;
          int q = 8;
#pragma unroll
          for (int z = 0; z < 8; ++z)
            if (c == ids[z]) q = z;
          if (q < 4) {
            w4[q] = w;
          } else if (q == 8) {
            const int lc = labels[c];
            if (l != lc) c0 += w;
            if (can && p != lc) c1 += w;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (ids[q] >= 0) {
            const int lj = labels[ids[q]];
            const int pj = prop ? (int)prop[ids[q]] : alpha;
            const int nib = (l != lj ? 1 : 0) | (l != pj ? 2 : 0) | (p != lj ? 4 : 0) | (p != pj ? 8 : 0);
            bits |= nib << (4 * q);
          }
        }
        sw = can && !(du > wtot);
        if (!sw) c1 = BIG;
      }
      float* r = rec + t * REC;
      r[0] = c0; r[1] = c1; r[2] = w4[0]; r[3] = w4[1]; r[4] = w4[2]; r[5] = w4[3];
      reci[t * REC + 6] = bits;
      reci[t * REC + 7] = node;
      }
      const unsigned long long swm = __ballot(sw);
      if (swm) {
        const int first = t0 + __ffsll((long long)swm) - 1;
        const int last = t0 + 63 - __clzll((long long)swm);
        t_lo = first < t_lo ? first : t_lo;
        t_hi = last > t_hi ? last : t_hi;
      }
    }
    const bool need = t_hi >= 0;
    int t_end = t_hi + SH + 1;
    if (t_end > ncell - 1) t_end = ncell - 1;
    __syncthreads();

    // ---- phase 2: lane <-> state (6 profile bits)
    float m = lane == 0 ? 0.f : BIG;      // every cell before t_lo keeps its label: profile 000000
    const int b = lane & 1, pl = lane >> 1;
    const int bu = pl & 1, bl = (pl >> 4) & 1, bld = (pl >> 3) & 1;
    for (int t = need ? t_lo : ncell; t <= t_end; ++t) {
      const float4 ra = *reinterpret_cast<const float4*>(rec + t * REC);
      const float4 rb = *reinterpret_cast<const float4*>(rec + t * REC + 4);
      const int bits = __builtin_bit_cast(int, rb.z);
      float base = b ? ra.y : ra.x;
      if ((bits >> (0 + b * 2 + bu)) & 1) base += ra.z;
      if ((bits >> (8 + b * 2 + bl)) & 1) base += rb.x;
      if ((bits >> (12 + b * 2 + bld)) & 1) base += rb.y;
      const float lu0 = ((bits >> (4 + b * 2)) & 1) ? ra.w : 0.f;
      const float lu1 = ((bits >> (5 + b * 2)) & 1) ? ra.w : 0.f;
      const float o0 = __shfl(m, pl, 64);
      const float o1 = __shfl(m, pl + 32, 64);
      const float a0 = o0 + base + lu0;
      const float a1 = o1 + base + lu1;
      const bool take1 = a1 < a0;
      m = take1 ? a1 : a0;
      const unsigned long long bt = __ballot(take1);
      if (lane == 0) bal[t] = bt;
    }
    __syncthreads();

    // ---- backtrack (wave-uniform scalars); the choice bit of cell t replaces the neighbour-relation word
    if (need) {
      const float mmin = wave_min_f32(m);
      int s = __ffsll((long long)__ballot(m == mmin)) - 1;
      for (int t = t_end; t >= t_lo; --t) {
        const int x = s & 1;
        const int d = (int)((bal[t] >> s) & 1ull);
        s = (s >> 1) | (d << 5);
        if (lane == 0) reci[t * REC + 6] = x;
      }
    }
    __syncthreads();

    // ---- phase 3: lane <-> cell: apply
    for (int t = (need ? t_lo : ncell) + lane; t <= t_end; t += 64) {
      const int node = reci[t * REC + 7];
      if (node >= 0 && reci[t * REC + 6]) {
        labels[node] = prop ? prop[node] : (uint8_t)alpha;
        ++my_changed;
      }
    }
    __syncthreads();
  }
  unsigned int s = my_changed;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0 && s) atomicAdd(changed, (unsigned long long)s);
}

// best alternative label per node: argmin_{k != l_i} ( -logprob[i,k] - beta * sum_{j in N(i), l_j == k} w_ij )
template <int VEC>
__global__ __launch_bounds__(256) void propose_kernel(const float* __restrict__ logprob, int64_t n, int K, int Kp, int D,
                                                      const int32_t* __restrict__ nbr, const float* __restrict__ wgt,
                                                      const uint8_t* __restrict__ labels, float beta,
                                                      uint8_t* __restrict__ prop) {
  extern __shared__ float tile[];
  const int TB = blockDim.x;
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
      float* row = tile + threadIdx.x * Kp;
      const int32_t* nb = nbr + i * D;
      const float* wg = wgt + i * D;
      for (int j = 0; j < D; ++j) {
        const int c = nb[j];
        if (c >= 0) row[labels[c]] -= beta * wg[j];
      }
      const int cur = labels[i];
      float best = 3.0e38f;
      int bk = cur;
      for (int k = 0; k < K; ++k) {
        const float v = row[k];
        if (k != cur && v < best) { best = v; bk = k; }
      }
      prop[i] = (uint8_t)bk;
    }
    __syncthreads();
  }
}

inline int vec_of(int K) { return (K % 4 == 0) ? 4 : (K % 2 == 0 ? 2 : 1); }

}  // namespace

int launch_propose(const phmrf_block* b, float beta) {
  const int K = b->K, TB = tile_threads(K), Kp = padded_k(K);
  const size_t lds = (size_t)TB * Kp * sizeof(float);
  int64_t g64 = (b->n + TB - 1) / TB;
  const int grid = (int)(g64 > 256 * 16 ? 256 * 16 : g64);
#define PHMRF_LAUNCH_PROP(VEC_)                                                                                     \
  hipLaunchKernelGGL((propose_kernel<VEC_>), dim3(grid), dim3(TB), lds, b->stream, b->logprob, b->n, K, Kp, b->D, b->nbr, \
                     b->wgt, b->labels, beta, b->labels_tmp)
  switch (vec_of(K)) {
    case 4: PHMRF_LAUNCH_PROP(4); break;
    case 2: PHMRF_LAUNCH_PROP(2); break;
    default: PHMRF_LAUNCH_PROP(1); break;
  }
#undef PHMRF_LAUNCH_PROP
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

// alpha < 0: fusion with the proposals in b->labels_tmp (launch_propose); alpha >= 0: strip alpha-expansion
int launch_strip_pass(const phmrf_block* b, float beta, int orient, int shift_r, int shift_c, int alpha) {
  StripGeom g;
  g.H = b->H;
  g.W = b->W;
  g.diagonal = b->diagonal;
  g.orient = orient;
  g.shift_r = shift_r;
  g.shift_c = shift_c;
  g.Hs = orient ? b->W : b->H;
  g.Ws = orient ? b->H : b->W;
  g.nbands = (g.Hs + shift_r + SH) / (SH + 1);
  g.nsegs = (g.Ws + shift_c + 63) / 64;
  const int nstrips = g.nbands * g.nsegs;
  if (nstrips <= 0) return PHMRF_OK;
  const int TB = 256, WPB = 4;
  const size_t lds = (size_t)WPB * (CELL_PAD * REC + CELL_PAD * 2) * sizeof(float);
  int grid = (nstrips + WPB - 1) / WPB;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(strip_kernel, dim3(grid), dim3(TB), lds, b->stream, g, b->logprob, b->K, b->D, b->nbr, b->wgt,
                     b->labels, alpha < 0 ? b->labels_tmp : nullptr, alpha, beta, b->counters);
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

}  // namespace phmrf
