// gfx950 kernel of the 2-D block move: EXACT binary fusion on strips.
//
// A strip is STRIP_H = 5 grid rows (or columns) by up to 63 cells along the other axis.  Every node of the strip
// chooses between its current label l_i and a proposal p_i (p_i = a constant alpha: an alpha-expansion restricted
// to the strip; p_i = the node's best alternative label: a fusion move that shifts region fronts); all nodes
// outside the strip are fixed.  The binary problem is solved exactly by a profile ("broken line") dynamic
// programme over the cells in column-major order: the state is the choice bit of the last STRIP_H+1 = 6 cells,
// i.e. 64 states = ONE WAVEFRONT with one state per lane.  Per pass of 64 cells the cost of a cell for every
// combination of (own bit, three neighbour bits) is tabulated in a 5 KB LDS slab; a DP step is then two LDS reads that
// do not depend on the DP state, one in-register lane exchange (DPP / v_permlane16_swap / v_permlane32_swap -- see
// dp_step) for the two predecessor values, a compare whose mask IS the decision ballot, and an add.  The decision
// ballots stay in registers (lane t mod 64 of a per-pass register pair); the backtrack is scalar.
// No submodularity is needed (the DP is exact for any 2x2 tables), so fusion proposals are as valid as expansions.
// Strips of one pass are separated by one fixed row and one fixed column, so simultaneous moves share no edge:
// the pass never raises the energy.   (Model: oracle/mrf_moves.strip_fusion.)

#include <cstdlib>
#include <utility>

#include "common.h"

namespace phmrf {
namespace {

constexpr int SH = 5;          // strip rows
constexpr int SL = 63;         // strip columns per segment
constexpr int NCELL_MAX = SH * SL;
constexpr float BIG = 1.0e30f;

struct StripGeom {
  int H, W, diagonal, orient, shift_r, shift_c, Hs, Ws, nbands, nsegs;
  int xcd;      // 1: orientation 1's strips in the XCD-aware order (strip_of_slot); 0: in their own order (development A/B)
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

__device__ __forceinline__ float comp4f(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// lane of the wavefront (workgroups are whole waves here), without threadIdx: inside an out-of-line device function the
// work-item id is a library call
__device__ __forceinline__ int wave_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// an LDS object from its 32-bit LDS address: pointers handed to an out-of-line function arrive as generic pointers and
// every access through them is a flat_load / flat_store (slow, and it waits on both counters); through the address-space
// cast the compiler recovers ds_read / ds_write
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // four consecutive floats at any 4-byte boundary
template <class T>
__device__ __forceinline__ T* lds_object(unsigned int lds_address) {
  return (T*)(__attribute__((address_space(3))) T*)(unsigned long)lds_address;
}
// ... and a global-memory pointer argument of an out-of-line function as such (global_load instead of flat_load)
template <class T>
using global_ptr = __attribute__((address_space(1))) T*;
template <class T>
__device__ __forceinline__ global_ptr<T> as_global(T* p) {
  return (global_ptr<T>)p;
}
template <class T>
__device__ __forceinline__ unsigned int lds_address_of(T* p) {
  return (unsigned int)(unsigned long)(__attribute__((address_space(3))) T*)p;
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

// State <-> lane.  The DP exchanges values between the two states that differ in ONE profile bit q.  DPP offers xor 1
// and xor 2 (quad_perm), xor 7 (row_half_mirror) and xor 15 (row_mirror) as single moves, v_permlane16/32_swap give
// xor 16 / xor 32 -- but not xor 4 or xor 8.  So the state is not the lane id: lane = s0*1 ^ s1*2 ^ s2*7 ^ s3*15 ^
// s4*16 ^ s5*32 (an invertible GF(2) map), which makes "flip bit q" one of those six lane permutations for every q.
__device__ __forceinline__ int state_of_lane(int lane) {
  const int l2 = (lane >> 2) & 1, l3 = (lane >> 3) & 1;
  return ((lane ^ l2) & 1) | ((((lane >> 1) ^ l2) & 1) << 1) | ((l2 ^ l3) << 2) | (lane & 0x38);
}
__device__ __forceinline__ int lane_of_state(int st) {
  const int s2 = (st >> 2) & 1, s3 = (st >> 3) & 1;
  const int l2 = s2 ^ s3;
  return ((st ^ l2) & 1) | ((((st >> 1) ^ l2) & 1) << 1) | (l2 << 2) | (st & 0x38);
}

// value of the state that differs from mine in bit Q
template <int Q>
__device__ __forceinline__ float xor_exchange(float v) {
  const int iv = __builtin_bit_cast(int, v);
  // (old = 0 with bound_ctrl: every lane has a valid source, and the move can be folded into the consuming v_add)
  if (Q == 0) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xf, 0xf, true));    // lane ^ 1
  if (Q == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xf, 0xf, true));    // lane ^ 2
  if (Q == 2) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, iv, 0x141, 0xf, 0xf, true));   // lane ^ 7
  if (Q == 3) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, iv, 0x140, 0xf, 0xf, true));   // lane ^ 15
  if (Q == 4) {  // lane ^ 16: odd rows of vdst <-> even rows of vsrc
    auto r = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
    return __builtin_bit_cast(float, (int)((wave_lane() & 16) ? r[0] : r[1]));
  }
  auto r = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);   // lane ^ 32
  return __builtin_bit_cast(float, (int)((wave_lane() & 32) ? r[0] : r[1]));
}

// ---- per-cell cost tables (LDS, one pass of 64 cells at a time) ---------------------------------------------------
// Cell record in LDS: for each of the 16 combinations (own bit b, up bu, left bl, left-down bld) the pair
//   { cost + lu[b][self], cost + lu[b][other] },   cost = c_b + wu*N_u[b][bu] + wl*N_l[b][bl] + wld*N_ld[b][bld],
//   lu[b][d] = wlu*N_lu[b][d]  (d = choice of the left-up cell leaving the profile; "self" d == b, "other" d == 1-b)
// = 32 floats, padded to a stride of 36 words (conflict-free for ds_write_b128).  A DP step reads ONE float2.
constexpr int TAB = 36;
// phase-1 staging record per cell (same LDS slab, before the tables are built): the four forward grid weights of the
// cell (times beta) and its packed labels (l | p << 8 | present << 16); stride 5 words: conflict-free.
// (the packed labels travel as float bit patterns: every access of the slab is a float access, so that the type-based
// alias rules cannot order a label read after the table stores that reuse the slab)
constexpr int REC = 5;
constexpr int EH = 7;                   // rows of the staged rectangle: the strip's 5 plus the fixed row above and below
constexpr int ECELLS = EH * (63 + 2);   // ... times its columns plus the fixed column left and right
constexpr int SLAB = 64 * TAB;           // floats per wave (9216 B): the tables of one pass; >= the staging area
static_assert(SLAB >= ECELLS * REC && SLAB % 4 == 0, "LDS slab too small");

// w where bit k of `bits` is set, else 0 (a sign-extended one-bit field as an AND mask: two vector instructions, no compare)
__device__ __forceinline__ float bit_masked(float w, int bits, int k) {
  const int m = (int)((unsigned int)bits << (31 - k)) >> 31;
  return __builtin_bit_cast(float, __builtin_bit_cast(int, w) & m);
}

__device__ __forceinline__ void build_table(float* tab, int lane, float c0, float c1, float wu, float wlu, float wl,
                                            float wld, int bits) {
  float4* dst = reinterpret_cast<float4*>(tab + lane * TAB);
  // cost(b, bu, bl, bld) = ((c_b + wu N_u[b][bu]) + wl N_l[b][bl]) + wld N_ld[b][bld], built as a tree over the three
  // neighbours (2 + 4 + 8 additions per b instead of 3 per combination; the same sums in the same order), then the two
  // left-up variants lu[b][self], lu[b][other]
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const float cb = b ? c1 : c0;
    const float lus = bit_masked(wlu, bits, b ? 7 : 4), luo = bit_masked(wlu, bits, b ? 6 : 5);
    float cA[2], cAB[2][2];
#pragma unroll
    for (int bu = 0; bu < 2; ++bu) cA[bu] = cb + bit_masked(wu, bits, 0 + b * 2 + bu);
#pragma unroll
    for (int bu = 0; bu < 2; ++bu)
#pragma unroll
      for (int bl = 0; bl < 2; ++bl) cAB[bu][bl] = cA[bu] + bit_masked(wl, bits, 8 + b * 2 + bl);
    const float C0 = bit_masked(wld, bits, 12 + b * 2), C1 = bit_masked(wld, bits, 12 + b * 2 + 1);
#pragma unroll
    for (int bu = 0; bu < 2; ++bu)
#pragma unroll
      for (int bl = 0; bl < 2; ++bl) {
        const float x0 = cAB[bu][bl] + C0, x1 = cAB[bu][bl] + C1;      // bld = 0, 1: combinations idx, idx + 1
        dst[b * 4 + bu * 2 + bl] = make_float4(x0 + lus, x0 + luo, x1 + lus, x1 + luo);
      }
  }
}

// One cell step of the profile DP.  The profile is kept in ROTATING positions: cell t owns bit (t mod 6) of the state
// index (= lane id), so stepping to cell t overwrites the bit of cell t-6 (the left-up neighbour, about to leave the
// profile) and new[s] = base[s] + min over that old bit d of (old[s with bit q := d] + lu[b][d]): the two candidates
// live in lanes s and s ^ (1<<q) -- ONE in-register lane exchange.  Neighbours: up = t-1 at bit q-1, left = t-5 at
// q+1, left-down = t-4 at q+2; the lane's table index (b, bu, bl, bld) is therefore a compile-time function of
// (lane, q).  base / lu come from two LDS reads that do not depend on the DP state (issued one step ahead).
template <int Q>
__device__ __forceinline__ int tab_offset(int st) {   // byte offsets of this state's entries for position Q
  constexpr int QU = (Q + 5) % 6, QL = (Q + 1) % 6, QLD = (Q + 2) % 6;
  const int b = (st >> Q) & 1, bu = (st >> QU) & 1, bl = (st >> QL) & 1, bld = (st >> QLD) & 1;
  return (b * 8 + bu * 4 + bl * 2 + bld) * 8;
}

template <int Q>
__device__ __forceinline__ void dp_step(float& m, unsigned long long& took, float t_self, float t_other,
                                        unsigned long long* decision) {
  // the two predecessors of state s differ in bit q (the choice d of cell t-6, which leaves the profile): "self" is this
  // lane's own value (d equal to the new cell's bit b), "other" the value of the state with bit q flipped; the table
  // entries already hold the cell's cost plus the left-up term of either case.  Ties keep "self".
  const float e_self = m + t_self;
  const float e_other = xor_exchange<Q>(m) + t_other;
  const bool take_other = e_other < e_self;
  m = __builtin_fminf(e_self, e_other);          // off the compare: the dependent chain is exchange -> add -> min
  *decision = __ballot(take_other);
  // took bit 0: state 000000 was once reached more cheaply from a state with a switched cell than along the all-keep
  // path.  Both continue identically with "keep", so this happens iff a strictly improving move exists (a scalar OR).
  took |= *decision;
}

#define PHMRF_RL(x, l) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l))

// one uniform value into one lane of a VGPR as a plain select (v_cmp + v_cndmask, no exec juggling)
__device__ __forceinline__ void write_lane(unsigned int& dst, unsigned int value, int lane_sel) {
  dst = (wave_lane() == lane_sel) ? value : dst;
}

// One step at a compile-time position: the table addresses are a per-lane register plus an immediate offset.
template <int P, int TT>
__device__ __forceinline__ void dp_step_at(float& m, unsigned long long& took, int lane, const char* tabc) {
  constexpr int Q = (4 * P + TT) % 6;
  const char* rec = tabc + TT * (TAB * 4);
  const float2 tv = *reinterpret_cast<const float2*>(rec + tab_offset<Q>(state_of_lane(lane)));
  unsigned long long dec;
  dp_step<Q>(m, took, tv.x, tv.y, &dec);
}

template <int P, int... TT>
__device__ __forceinline__ void dp_steps_unrolled(float& m, unsigned long long& took, int lane, const char* tabc,
                                                  std::integer_sequence<int, TT...>) {
  (dp_step_at<P, TT>(m, took, lane, tabc), ...);
}

// All steps of pass P (cells t = 64 P + tt): the pass's 64 cell tables are built into the wave's LDS slab, then walked.
template <int P, bool RECORD>
__device__ __forceinline__ void dp_pass(float& m, unsigned long long& took, int lane, float* tab, float c0, float c1, float wu, float wlu, float wl,
                                        float wld, int bits, int t_lo, int t_end, unsigned int& dlo, unsigned int& dhi) {
  // RECORD: the decision ballots of the pass stay in registers, lane tt keeps the ballot of step tt
  dlo = 0u;
  dhi = 0u;
  if (P * 64 > t_end || P * 64 + 63 < t_lo) return;
  build_table(tab, lane, c0, c1, wu, wlu, wl, wld, bits);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // [t_lo, t_end] is wave-uniform and already widened to whole groups of 6 steps (cells outside the true range are
  // pinned or absent, so visiting them is harmless); the only per-step test left is the static tail of the pass.
  int tb0 = t_lo - P * 64;
  tb0 = tb0 < 0 ? 0 : tb0 - tb0 % 6;
  int tb1 = t_end - P * 64;
  tb1 = tb1 > 63 ? 63 : tb1;
  const char* tabc = reinterpret_cast<const char*>(tab);
  if (!RECORD) {
    // the common walk (no decisions kept): all 64 steps of the pass as straight-line code, so that no address
    // arithmetic is left in a step (cells outside [t_lo, t_end] are pinned or absent: visiting them changes nothing)
    dp_steps_unrolled<P>(m, took, lane, tabc, std::make_integer_sequence<int, 64>{});
    __builtin_amdgcn_wave_barrier();
    return;
  }
  for (int tb = tb0; tb <= tb1; tb += 6) {   // rotating position q = t mod 6 = (4 P + tt) mod 6: static in the 6-unroll
#define PHMRF_STEP(J)                                                                                                \
  {                                                                                                                  \
    const int tt = tb + J;                                                                                           \
    if (J < 4 || tt < 64) {                                                                                          \
      constexpr int Q = (4 * P + J) % 6;                                                                             \
      const char* rec = tabc + tt * (TAB * 4);                                                                       \
      const float2 tv = *reinterpret_cast<const float2*>(rec + tab_offset<Q>(state_of_lane(lane)));                  \
      unsigned long long dec;                                                                                        \
      dp_step<Q>(m, took, tv.x, tv.y, &dec);                                                                         \
      if (RECORD) {                                                                                                  \
        write_lane(dlo, (unsigned int)(dec & 0xffffffffull), tt);                                                    \
        write_lane(dhi, (unsigned int)(dec >> 32), tt);                                                              \
      }                                                                                                              \
    }                                                                                                                \
  }
    PHMRF_STEP(0) PHMRF_STEP(1) PHMRF_STEP(2) PHMRF_STEP(3) PHMRF_STEP(4) PHMRF_STEP(5)
#undef PHMRF_STEP
  }
  __builtin_amdgcn_wave_barrier();
}

// Backtrack of pass P on scalars: x_t = bit q of the state, then the state gets back the bit of cell t-6.
template <int P>
__device__ __forceinline__ void backtrack_pass(int& s, int t_lo, int t_end, unsigned int dlo, unsigned int dhi,
                                               unsigned int& xsel) {
  xsel = 0u;
  if (P * 64 > t_end || P * 64 + 63 < t_lo) return;
  int tb0 = t_lo - P * 64;
  tb0 = tb0 < 0 ? 0 : tb0 - tb0 % 6;
  int tb1 = t_end - P * 64;
  tb1 = tb1 > 63 ? 63 : tb1;
  tb1 = tb1 - tb1 % 6;                        // first step of the last group
  for (int tb = tb1; tb >= tb0; tb -= 6) {
#define PHMRF_BACK(J)                                                                                                \
  {                                                                                                                  \
    const int tt = tb + J;                                                                                           \
    if (J < 4 || tt < 64) {                                                                                          \
      constexpr int Q = (4 * P + J) % 6;                                                                             \
      const unsigned long long dec = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dhi, tt) << 32) | \
                                     (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dlo, tt);      \
      const int x = (s >> Q) & 1;                                                                                    \
      const int d = x ^ (int)((dec >> lane_of_state(s)) & 1ull);     /* ballot: "took the OTHER predecessor" */      \
      s = (s & ~(1 << Q)) | (d << Q);                                                                                \
      write_lane(xsel, (unsigned int)x, tt);                                                                         \
    }                                                                                                                \
  }
    PHMRF_BACK(5) PHMRF_BACK(4) PHMRF_BACK(3) PHMRF_BACK(2) PHMRF_BACK(1) PHMRF_BACK(0)
#undef PHMRF_BACK
  }
}

constexpr int NPASS = 5;   // ceil(315 / 64)

#ifndef PHMRF_STRIP_WPE
#define PHMRF_STRIP_WPE 4
#endif
#ifndef PHMRF_STRIP_WPB
#define PHMRF_STRIP_WPB 1       // waves per workgroup (the waves are independent: one-wave workgroups, finest dispatch grain)
#endif
template <int ORIENT>
__global__ __launch_bounds__(64 * PHMRF_STRIP_WPB, PHMRF_STRIP_WPE) void strip_kernel(StripGeom g, int64_t n, int K, int D,
                                                    const int32_t* __restrict__ nbr, const float4* __restrict__ fwd_w,
                                                    const float* __restrict__ uT, uint8_t* __restrict__ labels,
                                                    const uint8_t* __restrict__ prop, int alpha, float beta,
                                                    unsigned long long* __restrict__ changed, int debug,
                                                    uint16_t* __restrict__ stamp, uint16_t* __restrict__ memo, int tick,
                                                    unsigned long long* __restrict__ work) {
  __shared__ __attribute__((aligned(16))) float tabs[PHMRF_STRIP_WPB * SLAB];   // one 9.1 KB slab per wave: phase-1 staging, then the cost tables of the pass walked
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int WPB = blockDim.x >> 6;
  const int nstrips = g.nbands * g.nsegs;
  float* tab = tabs + wave * SLAB;
  unsigned int my_changed = 0;
  // work actually done by this wave (measurement: bench.py's roofline): strips staged, their cells, staged cells, DP
  // steps walked.  Kept in LDS (lane 0 adds, fire and forget): no register may stay live across the strip loop for it.
  __shared__ unsigned int wk[4];
  if (threadIdx.x < 4) wk[threadIdx.x] = 0u;
  __syncthreads();

  // waves are independent: every wave walks strips of the cut
  for (int strip_v = blockIdx.x * WPB + wave; strip_v < nstrips; strip_v += gridDim.x * WPB) {
    const int strip = __builtin_amdgcn_readfirstlane(strip_v);     // the wave's strip: its geometry in SGPRs
    const int bnd = strip / g.nsegs;
    const int seg = strip - bnd * g.nsegs;
    const int rs0 = bnd * (SH + 1) - g.shift_r;
    const int cs0 = seg * 64 - g.shift_c;
    const int ca = cs0 > 0 ? cs0 : 0;
    const int cb = (cs0 + SL < g.Ws) ? cs0 + SL : g.Ws;
    const int ncols = cb > ca ? cb - ca : 0;
    const int ncell = ncols * SH;
    // (a strip of the bounding rectangle that holds no node -- below the diagonal of an upper-triangular block: see strip_cols_kernel)
    if (g.diagonal && (ORIENT == 0 ? (rs0 > 0 ? rs0 : 0) > cb - 1 : ca > (rs0 + SH - 1 < g.Hs - 1 ? rs0 + SH - 1 : g.Hs - 1))) continue;

    // ---- memo test (inside a solve, fixed cuts): a (dilated) change stamp is renewed whenever the node or one of its
    //      neighbours changes label, so the newest stamp among the strip's cells covers the fixed border too.  If nothing
    //      changed since this very strip was last found quiet for this move, its inputs are identical -> nothing to do.
    if ((debug & 4) && lane == 0) atomicAdd(changed - alpha - 8 + 100, 1ull);   // strips seen (expansion slots only)
    uint16_t* my_memo = memo ? memo + (int64_t)strip * (K + 1) + (alpha >= 0 ? alpha : K) : nullptr;
    if (my_memo) {
      const int last_quiet = *my_memo;
      if (last_quiet) {
        int nw = 0;
        for (int e = lane; e < ncell; e += 64) {
          const int cc = e / SH, rr = e - cc * SH;
          const int node = strip_node(g, rs0 + rr, ca + cc);
          if (node >= 0) {
            const int st = stamp[node];
            nw = st > nw ? st : nw;
          }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const int o2 = __shfl_xor(nw, off, 64);
          nw = o2 > nw ? o2 : nw;
        }
        if (nw < last_quiet) continue;
      }
    }

    if (debug == 16) continue;           // timing experiments: launch + memo + mask only
    // ---- the coarse child problems (debug & 32: two labels, keep = 0 with unary term 0 everywhere, switch = alpha with the
    //      super-cell's switch cost D, which coarsen_kernel sets to >= 1e29 where a switch can never pay): step B's test
    //      "does any cell of the strip propose a switch" made on the unary plane alone, before the strip is staged.  And
    //      while no super-cell of the problem has switched yet (*changed == 0: the first pass, and the second after a quiet
    //      first -- 3,599 strips in 3,600), every cell in and around the strip is at "keep", every pair term is >= 0, and a
    //      switch set costs the sum of its D's plus its cut: with no D < 0 in the strip nothing beats all-keep (cost 0,
    //      which the DP's own tie rule prefers), so only strips that hold a NEGATIVE switch cost go on -- measured on the
    //      cold solve of the 12.4 M-node block: half of all child strips went through the DP over 276 of their 315 cells,
    //      one in 3,600 found a move.
    if ((debug & 32) && alpha >= 0 && !prop) {
      const bool fresh = *changed == 0ull;
      if (!fresh) {
        bool any = false;
        for (int e = lane; e < ncell; e += 64) {
          const int cc = e / SH, rr = e - cc * SH;
          const int node = strip_node(g, rs0 + rr, ca + cc);
          if (node >= 0) any = any || (labels[node] != alpha && uT[(int64_t)alpha * n + node] < 1.0e29f);
        }
        if (!__ballot(any)) {
          if (my_memo && lane == 0) *my_memo = (uint16_t)tick;
          continue;
        }
      } else {
        // the strip's cells with a negative switch cost, gathered one per lane (the slab is free until the strip is staged)
        const float* __restrict__ ua = uT + (int64_t)alpha * n;
        int total = 0;
        int lnode[NPASS], llab[NPASS];
        float lcost[NPASS];
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {            // every load of the look in flight at once (an absent cell reads node 0)
          const int e = p * 64 + lane;
          const int cc = e / SH, rr = e - cc * SH;
          lnode[p] = e < ncell ? strip_node(g, rs0 + rr, ca + cc) : -1;
        }
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
          llab[p] = labels[lnode[p] >= 0 ? lnode[p] : 0];
          lcost[p] = ua[lnode[p] >= 0 ? lnode[p] : 0];
        }
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
          const int e = p * 64 + lane;
          const bool neg = lnode[p] >= 0 && llab[p] != alpha && lcost[p] < 0.f;
          const unsigned long long m = __ballot(neg);
          const int slot = total + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
          if (neg && slot < 64) tab[slot] = __builtin_bit_cast(float, e);
          total += __popcll(m);
        }
        if (total == 0) {
          if (my_memo && lane == 0) *my_memo = (uint16_t)tick;
          continue;
        }
        // ---- a one-hop flow certificate for those cells (HISTORY.md 3.1 item 6 (v)).  With every cell at "keep", a switch set S costs
        //      sum_S D + cut(S).  Let every cell A with D_A < 0 ship f_AB <= lambda_AB to neighbours B with D_B > 0, each B
        //      taking in at most D_B in all.  If A can ship its whole deficit (sum_B f_AB >= -D_A) it is SETTLED, and a set S
        //      whose negative cells are all settled costs sum_S D + cut(S) >= -sum_{A in S, B not in S} f_AB + cut(S) >= 0 (what
        //      A ships to a B inside S is paid for by D_B; what it ships out of S is at most the cut).  The split: a cell with
        //      D_B > 0 offers each of its n_B neighbours with a negative cost min(lambda_AB, D_B / n_B) -- whichever strip those
        //      neighbours lie in; a pinned cell (D >= 1e29: its true cost exceeds all its pair weights) offers lambda_AB.  So a
        //      strip whose negative cells are all settled has nothing better than all-keep, which the DP's tie rule prefers:
        //      it stops here, and only the others -- one strip in seven at 2 x 2, one in thirty at 4 x 4 on the whole-genome
        //      blocks where three in four hold a negative cost -- are staged and walked.
        if (total <= 64) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          bool open_cell = false;
          if (lane < total) {
            const int e = __builtin_bit_cast(int, tab[lane]);
            const int cc = e / SH, rr = e - cc * SH;
            const int R = rs0 + rr, C = ca + cc;
            const int node = strip_node(g, R, C);
            const float4 fw = fwd_w[node];
            const float* __restrict__ fws = reinterpret_cast<const float*>(fwd_w);
            // switch costs of the 5 x 5 window around the cell: which are negative, and the eight inner ones themselves
            unsigned int negw = 0u;
            float dn[8];
            int nb[8];
#pragma unroll
            for (int dr = -2; dr <= 2; ++dr)
#pragma unroll
              for (int dc = -2; dc <= 2; ++dc) {
                if (dr == 0 && dc == 0) {
                  negw |= 1u << 12;
                  continue;
                }
                const int nw = strip_node(g, R + dr, C + dc);
                const float dv = ua[nw >= 0 ? nw : node];       // (an absent cell reads this one: no load waits behind a branch)
                if (nw >= 0 && dv < 0.f) negw |= 1u << ((dr + 2) * 5 + (dc + 2));
                if (dr >= -1 && dr <= 1 && dc >= -1 && dc <= 1) {
                  constexpr int dummy = 0;
                  (void)dummy;
                  const int k = (dr + 1) * 3 + (dc + 1);
                  const int k8 = k > 4 ? k - 1 : k;
                  dn[k8] = nw >= 0 ? dv : 0.f;
                  nb[k8] = nw;
                }
              }
            float recv = 0.f;
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
              const int k = k8 >= 4 ? k8 + 1 : k8;
              const int dr = k / 3 - 1, dc = k % 3 - 1;
              const int di = ORIENT ? dc : dr, dj = ORIENT ? dr : dc;       // grid direction of the edge (as in step B)
              const bool fwd = di > 0 || (di == 0 && dj > 0);
              const int comp = (di == 0) ? 0 : (fwd ? dj + 2 : 2 - dj);
              const float wraw = fwd ? comp4f(fw, comp) : fws[4 * (int64_t)(nb[k8] >= 0 ? nb[k8] : node) + comp];
              const float w = nb[k8] >= 0 ? beta * wraw : 0.f;
              // negative cells among B's eight neighbours (this cell is one of them)
              int nneg = 0;
#pragma unroll
              for (int er = -1; er <= 1; ++er)
#pragma unroll
                for (int ec = -1; ec <= 1; ++ec)
                  if (er != 0 || ec != 0) nneg += (int)((negw >> ((dr + er + 2) * 5 + (dc + ec + 2))) & 1u);
              const float capb = dn[k8] > 0.f ? dn[k8] : 0.f;
              recv += __builtin_fminf(w, capb / (float)nneg);
            }
            open_cell = !(recv >= -ua[node] * 1.0001f + 1.0e-6f);
          }
          const bool any_open = __ballot(open_cell) != 0ull;
          __builtin_amdgcn_wave_barrier();      // (the slab is about to be staged)
          if (!any_open) {
            if (my_memo && lane == 0) *my_memo = (uint16_t)tick;
            continue;
          }
        }
      }
    }
    if ((debug & 4) && lane == 0) atomicAdd(changed - alpha - 8 + 101, 1ull);   // strips reaching phase 1
    if (lane == 0) {
      atomicAdd(&wk[0], 1u);
      atomicAdd(&wk[1], (unsigned int)ncell);                  // nodes this unit re-decides
      atomicAdd(&wk[2], (unsigned int)(EH * (ncols + 2)));     // the staged rectangle: strip + fixed rim (what is read)
    }
    // ---- phase 1.  Step A: the strip's rectangle PLUS its fixed rim (7 x (ncols + 2) cells, lane <-> cell) is staged
    //      in LDS: per cell the four forward grid weights (times beta) and the packed labels -- 1 + 16 coalesced bytes
    //      per cell, each grid edge weight read once, by its upper/left end.  Step B: lane <-> strip cell
    //      (t = 64 p + lane, column-major: cc = t / 5, rr = t % 5) reads its eight neighbours back from LDS at
    //      compile-time offsets: the four already-visited in-strip ones give the pair tables, the rim ones are folded
    //      into the unary costs.
    float rc0[NPASS], rc1[NPASS], rwu[NPASS], rwlu[NPASS], rwl[NPASS], rwld[NPASS];
    int rbits[NPASS];
    int t_lo = NCELL_MAX, t_hi = -1;
    {
      constexpr int NEP = (ECELLS + 63) / 64;       // 8 lane-passes over the staged rectangle
      int enode[NEP], eidx[NEP];
#pragma unroll
      for (int q = 0; q < NEP; ++q) {               // all addresses first ...
        int er, ec;
        if (ORIENT == 0) {
          // strip columns are contiguous in memory: one staged row per lane-pass (64 consecutive nodes per load
          // instruction), the 65th column in the last pass
          int l2 = lane;
          asm volatile("" : "+v"(l2));              // (kept out of the loop-invariant set, see step B)
          er = q < EH ? q : l2;
          ec = q < EH ? l2 : 64;
          if (q >= EH && l2 >= EH) ec = 1 << 20;    // no cell
        } else {
          // strip rows are contiguous in memory: cells in column-major order (runs of 7 consecutive nodes)
          int e = q * 64 + lane;
          asm volatile("" : "+v"(e));
          ec = e / EH;
          er = e - ec * EH;
        }
        const bool have = ec < ncols + 2;
        eidx[q] = have ? ec * EH + er : -1;
        enode[q] = have ? strip_node(g, rs0 - 1 + er, ca - 1 + ec) : -1;
      }
      int elab[NEP], epl[NEP];
      float4 ef[NEP];
#pragma unroll
      for (int q = 0; q < NEP; ++q) {               // ... then every load of the wave in flight at once
        const int node = enode[q];
        elab[q] = 0;
        epl[q] = alpha;
        ef[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (node >= 0) {
          elab[q] = labels[node];
          if (prop) epl[q] = prop[node];
          ef[q] = fwd_w[node];
        }
      }
#pragma unroll
      for (int q = 0; q < NEP; ++q) {
        const int e = eidx[q];
        if (e >= 0) {
          const bool present = enode[q] >= 0;
          tab[e * REC + 0] = ef[q].x * beta;
          tab[e * REC + 1] = ef[q].y * beta;
          tab[e * REC + 2] = ef[q].z * beta;
          tab[e * REC + 3] = ef[q].w * beta;
          tab[e * REC + 4] = __builtin_bit_cast(float, (int)(present ? (elab[q] | (epl[q] << 8)) : 0));
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (debug == 8) continue;            // timing experiments: step A only
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      int t = p * 64 + lane;
      asm volatile("" : "+v"(t));     // not loop-invariant for the compiler: hoisting the per-pass cell coordinates
                                      // out of the strip loop would pin ~20 VGPRs for the whole kernel
      bool sw = false;
      float c0 = 0.f, c1 = BIG, w4[4] = {0.f, 0.f, 0.f, 0.f};
      int bits = 0, node = -1;
      if (t < ncell) {
        const int cc = t / SH, rr = t - cc * SH;
        node = strip_node(g, rs0 + rr, ca + cc);
        if (node >= 0) {
          const int e0 = (cc + 1) * EH + (rr + 1);
          const int lw = __builtin_bit_cast(int, tab[e0 * REC + 4]);
          const int l = lw & 255, pl = (lw >> 8) & 255;
          const bool can = pl != l;
          const float u0 = uT[(int64_t)l * n + node];
          const float u1 = can ? uT[(int64_t)pl * n + node] : BIG;
          float a0 = 0.f, a1 = 0.f;            // rim sums
#pragma unroll
          for (int d = 0; d < 8; ++d) {
            constexpr int DR[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
            constexpr int DC[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
            const int dr = DR[d], dc = DC[d];
            const int di = ORIENT ? dc : dr, dj = ORIENT ? dr : dc;       // grid direction of the edge
            // forward edges (E, SW, S, SE) are held by me, backward ones (W, NE, N, NW) by the neighbour, same component:
            // E/W -> 0, SW/NE -> 1, S/N -> 2, SE/NW -> 3
            const bool fwd = di > 0 || (di == 0 && dj > 0);
            const int comp = (di == 0) ? 0 : (fwd ? dj + 2 : 2 - dj);
            const int en = e0 + dc * EH + dr;
            const float w = fwd ? tab[e0 * REC + comp] : tab[en * REC + comp];
            const int nl = __builtin_bit_cast(int, tab[en * REC + 4]);
            const int lj = nl & 255, pj = (nl >> 8) & 255;
            const int r2 = rr + dr, c2 = cc + dc;
            const bool inside = r2 >= 0 && r2 < SH && c2 >= 0 && c2 < ncols;
            // visited-before neighbours: up (d 1), left-up (d 0), left (d 3), left-down (d 5)
            constexpr int QOF[8] = {1, 0, -1, 2, -1, 3, -1, -1};
            if (QOF[d] >= 0) {
              if (inside) {
                w4[QOF[d]] = w;
                const int nib = (l != lj ? 1 : 0) | (l != pj ? 2 : 0) | (pl != lj ? 4 : 0) | (pl != pj ? 8 : 0);
                bits |= nib << (4 * QOF[d]);
              }
            }
            if (!inside) {                       // fixed neighbour (an absent one has weight 0)
              if (l != lj) a0 += w;
              if (pl != lj) a1 += w;
            }
          }
          // (a proposal whose unary cost is "infinite" is no proposal: coarse.hip pins super-cells that way)
          const bool can2 = can && u1 < 1.0e29f;
          c0 = u0 + a0;
          c1 = can2 ? u1 + a1 : BIG;
          sw = can2;
        }
      }
      rc0[p] = c0; rc1[p] = c1; rwu[p] = w4[0]; rwlu[p] = w4[1]; rwl[p] = w4[2]; rwld[p] = w4[3];
      rbits[p] = bits;
      const unsigned long long swm = __ballot(sw);
      if (swm) {
        const int first = p * 64 + __ffsll((long long)swm) - 1;
        const int last = p * 64 + 63 - __clzll((long long)swm);
        t_lo = first < t_lo ? first : t_lo;
        t_hi = last > t_hi ? last : t_hi;
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the passes apart: interleaving them only costs registers
    }
    __builtin_amdgcn_wave_barrier();      // the slab is about to be reused for the cost tables
    if (t_hi < 0 || (debug & 3) == 1) {           // nothing can move in this strip (wave-uniform)
      if (my_memo && lane == 0) *my_memo = (uint16_t)tick;
      continue;
    }
    t_lo = __builtin_amdgcn_readfirstlane(t_lo);
    int t_end = __builtin_amdgcn_readfirstlane(t_hi) + SH + 1;
    {   // widen to whole groups of 6 steps counted from each pass start (the DP passes run in such groups)
      const int pe = t_end >> 6, re = t_end & 63;
      int r6 = re - re % 6 + 5;
      if (r6 > 63) r6 = 63;
      t_end = pe * 64 + r6;
      if (t_end > NPASS * 64 - 1) t_end = NPASS * 64 - 1;
      const int pl = t_lo >> 6, rl = t_lo & 63;
      t_lo = pl * 64 + (rl - rl % 6);
    }

    if (lane == 0) atomicAdd(&wk[3], (unsigned int)(t_end - t_lo + 1));
    if ((debug & 4) && lane == 0) {
      atomicAdd(changed - alpha - 8 + 102, 1ull);                                // strips reaching the DP
      atomicAdd(changed - alpha - 8 + 103, (unsigned long long)(t_end - t_lo + 1));   // DP steps
    }
    // ---- phase 2: lane <-> state.  Records are broadcast with v_readlane, decisions are one 64-bit ballot per step
    //      parked in lane (t mod 64) of a per-pass register pair.
    float m = lane == 0 ? 0.f : BIG;     // every cell before t_lo keeps its label: profile 000000
    unsigned long long took = 0ull;
    unsigned int dlo[NPASS], dhi[NPASS];
    dp_pass<0, false>(m, took, lane, tab, rc0[0], rc1[0], rwu[0], rwlu[0], rwl[0], rwld[0], rbits[0], t_lo, t_end, dlo[0], dhi[0]);
    dp_pass<1, false>(m, took, lane, tab, rc0[1], rc1[1], rwu[1], rwlu[1], rwl[1], rwld[1], rbits[1], t_lo, t_end, dlo[1], dhi[1]);
    dp_pass<2, false>(m, took, lane, tab, rc0[2], rc1[2], rwu[2], rwlu[2], rwl[2], rwld[2], rbits[2], t_lo, t_end, dlo[2], dhi[2]);
    dp_pass<3, false>(m, took, lane, tab, rc0[3], rc1[3], rwu[3], rwlu[3], rwl[3], rwld[3], rbits[3], t_lo, t_end, dlo[3], dhi[3]);
    dp_pass<4, false>(m, took, lane, tab, rc0[4], rc1[4], rwu[4], rwlu[4], rwl[4], rwld[4], rbits[4], t_lo, t_end, dlo[4], dhi[4]);

    if ((debug & 3) == 2) continue;
    // ---- final state: among the minimisers take the one whose SHIFT-encoded index (newest cell in bit 0, as in the
    //      move model) is lowest, so ties are broken exactly like oracle/mrf_moves.strip_fusion
    const int q_end = t_end % 6;
    int sidx = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) sidx |= ((state_of_lane(lane) >> ((q_end - j + 6) % 6)) & 1) << j;
    const float mmin = wave_min_f32(m);
    // the all-keep path is optimal (see dp_step): nothing to backtrack or apply -- the case of ~997 strips in 1000
    if (!(took & 1ull) && PHMRF_RL(m, 0) == mmin) {
      if (my_memo && lane == 0) *my_memo = (uint16_t)tick;
      continue;
    }
    if ((debug & 4) && lane == 0) atomicAdd(changed - alpha - 8 + 104, 1ull);   // DPs that found a move
    // a move exists (about 3 strips in 1000): walk the DP again, this time recording the decision ballots
    m = lane == 0 ? 0.f : BIG;
    dp_pass<0, true>(m, took, lane, tab, rc0[0], rc1[0], rwu[0], rwlu[0], rwl[0], rwld[0], rbits[0], t_lo, t_end, dlo[0], dhi[0]);
    dp_pass<1, true>(m, took, lane, tab, rc0[1], rc1[1], rwu[1], rwlu[1], rwl[1], rwld[1], rbits[1], t_lo, t_end, dlo[1], dhi[1]);
    dp_pass<2, true>(m, took, lane, tab, rc0[2], rc1[2], rwu[2], rwlu[2], rwl[2], rwld[2], rbits[2], t_lo, t_end, dlo[2], dhi[2]);
    dp_pass<3, true>(m, took, lane, tab, rc0[3], rc1[3], rwu[3], rwlu[3], rwl[3], rwld[3], rbits[3], t_lo, t_end, dlo[3], dhi[3]);
    dp_pass<4, true>(m, took, lane, tab, rc0[4], rc1[4], rwu[4], rwlu[4], rwl[4], rwld[4], rbits[4], t_lo, t_end, dlo[4], dhi[4]);
    const float mmin2 = wave_min_f32(m);   // the recorded walk starts at the window's first group, the common walk at
    //  the pass start: their values differ by the keep-costs of the cells in between -> the recorded walk's own minimum
    const float cand = (m == mmin2) ? (float)sidx : 127.f;
    const float best = wave_min_f32(cand);
    int s = state_of_lane(__ffsll((long long)__ballot(cand == best)) - 1);

    // ---- backtrack on scalars; the choice of cell t lands in lane (t mod 64) of xsel[pass]
    unsigned int xsel[NPASS];
    s = __builtin_amdgcn_readfirstlane(s);
    backtrack_pass<4>(s, t_lo, t_end, dlo[4], dhi[4], xsel[4]);
    backtrack_pass<3>(s, t_lo, t_end, dlo[3], dhi[3], xsel[3]);
    backtrack_pass<2>(s, t_lo, t_end, dlo[2], dhi[2], xsel[2]);
    backtrack_pass<1>(s, t_lo, t_end, dlo[1], dhi[1], xsel[1]);
    backtrack_pass<0>(s, t_lo, t_end, dlo[0], dhi[0], xsel[0]);

    // ---- phase 3: lane <-> cell: apply
    __builtin_amdgcn_wave_barrier();
    bool moved = false;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int t = p * 64 + lane;
      if (xsel[p] && t < ncell) {
        const int cc = t / SH, rr = t - cc * SH;
        const int node = strip_node(g, rs0 + rr, ca + cc);
        if (node < 0) continue;
        labels[node] = prop ? prop[node] : (uint8_t)alpha;
        if (stamp) {
          stamp[node] = (uint16_t)tick;
          const int32_t* nb2 = nbr + (int64_t)node * D;
          for (int j = 0; j < D; ++j)
            if (nb2[j] >= 0) stamp[nb2[j]] = (uint16_t)tick;
        }
        ++my_changed;
        moved = true;
      }
    }
    {
      const bool any_moved = __any(moved);
      if (my_memo && lane == 0) {
        *my_memo = any_moved ? (uint16_t)0 : (uint16_t)tick;
      }
    }
  }
  unsigned int s = my_changed;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0 && s) atomicAdd(changed, (unsigned long long)s);
  __syncthreads();                       // every wave of the workgroup gets here (the strip loop only `continue`s)
  if (work && threadIdx.x < 4) {         // one add per workgroup and counter, spread over WORK_BANKS addresses
    const unsigned int v = wk[threadIdx.x];
    if (v) atomicAdd(work + (blockIdx.x & (WORK_BANKS - 1)) * WORK_SLOTS + threadIdx.x, (unsigned long long)v);
  }
}

constexpr int PEEL_MAX = 16;      // sweeps before the DP takes over with whatever U is left (any U is sound; measured: 8 -> 16 sweeps = -30 % DP steps, -1.5 % E-step)

// ---------------------------------------------------------------------------------------------------------------
// strip_cols_kernel (round 3): the same launch -- every alpha-expansion of a strip in one wave, behind the exact
// filter -- with the FILTER in a lane <-> strip COLUMN layout and the rare DP fed from the staged slab.
//
// Filter (lane c owns the five cells of strip column c): the set U is five 64-bit scalars, one per strip row, bit c =
// column c.  "Is my neighbour (r + dr, c + dc) in U" is bit c of U[r + dr] shifted by dc: ten scalar shifts per sweep
// instead of a funnel shift, an AND with a row pattern and a pass-boundary fix-up per (pass, direction) -- a sweep is
// ~110 instructions where the cell-order layout of round 2's strip_multi_kernel (deleted in round 5) needed ~300.  Sweeps are Jacobi (five independent
// chains per wave).  A label that occurs nowhere in the strip's rectangle (most labels: a 64-bit presence mask formed at
// staging) needs no neighbour-label compare at all, and its first sweep is the compare of the single-site cost with a
// label-independent cap.  The label's unary terms are 5 row loads per wave, contiguous in orientation 0.
// Flagged (label, U) pairs wait in a small LDS buffer; then the wave turns to the DP in cell order (lane <-> cell
// t = 64 p + lane as before): records straight from the slab (which the filter never overwrites), only for the passes
// the window touches, tables built and walked in chunks of 18 cells (2.6 KB instead of a 9.2 KB slab per pass).
// The order of events: labels ascending, and after a move on the strip everything later
// is filtered again on the new labels -- so the launch still equals the single-label passes of the move model.
// the staged rectangle of strip_cols_kernel: the four forward weights of the cells of rows -1 .. 4 plus the one weight of the
// bottom rim row that the strip needs (orientation 1: the edge to strip row 4 of the next column): 24 + 1 words per column
// (odd stride: lane <-> column reads are conflict-free), and one label byte per cell of all seven rows: 7 KB where the
// 20-byte records of strip_kernel take 9.1
constexpr int SWC = 25;                      // words per column in the weight slab
constexpr int SLABW = (63 + 2) * SWC;        // floats
constexpr int SLABL = ((63 + 2) * EH + 3) / 4 * 4;   // label bytes
constexpr int CH = 18;                       // cells per table chunk: three groups of six DP steps
constexpr int NBUF = 8;                      // flagged labels buffered before the wave turns to the DP
// everything strip_cols_kernel keeps in LDS: 9.9 KB with one wave per workgroup (WV = 1: the full sweeps), 11 KB with WV = 4
// waves that share one staged strip and split its labels (round 6: the launches of a solve's late rounds, strip_cols_kernel)
template <int WV>
struct ColsLdsT {
  alignas(16) float tabch[CH * 36];          // DP tables of one chunk (2.6 KB; 36 = TAB)
  alignas(16) float slabw[SLABW];            // the staged rectangle, never overwritten: forward weights ...
  unsigned long long ubuf[WV * NBUF][SH];    // U of the flagged labels, one word per strip row (NBUF entries per wave)
  int abuf[WV * NBUF];                       // ... and which labels they are
  unsigned int wk[WORK_SLOTS];
  int nbuf[WV];                              // flagged labels per wave
  unsigned long long left[WV];               // WV > 1: the labels a wave did not get to (its buffer was full)
  unsigned char slabl[SLABL];                // ... and label bytes of the staged rectangle
};
using ColsLds = ColsLdsT<1>;

// ... and of fusion_cols_kernel: the same, plus the proposal byte of every staged cell
struct FusLds {
  ColsLds c;
  unsigned char slabp[SLABL];
};

template <int P, int C, int... TT>
__device__ __forceinline__ void dp_chunk_steps(float& m, unsigned long long& took, int lane, const char* tabc,
                                               std::integer_sequence<int, TT...>) {
  (([&] {
     constexpr int T = C * CH + TT;
     if constexpr (T < 64) {
       constexpr int Q = (4 * P + T) % 6;
       const float2 tv = *reinterpret_cast<const float2*>(tabc + TT * (TAB * 4) + tab_offset<Q>(state_of_lane(lane)));
       unsigned long long dec;
       dp_step<Q>(m, took, tv.x, tv.y, &dec);
     }
   }()),
   ...);
}

template <int P, int C, int... TT>
__device__ __forceinline__ void dp_chunk_steps_rec(float& m, unsigned long long& took, int lane, const char* tabc,
                                                   unsigned int& dlo, unsigned int& dhi, std::integer_sequence<int, TT...>) {
  (([&] {
     constexpr int T = C * CH + TT;
     if constexpr (T < 64) {
       constexpr int Q = (4 * P + T) % 6;
       const float2 tv = *reinterpret_cast<const float2*>(tabc + TT * (TAB * 4) + tab_offset<Q>(state_of_lane(lane)));
       unsigned long long dec;
       dp_step<Q>(m, took, tv.x, tv.y, &dec);
       write_lane(dlo, (unsigned int)(dec & 0xffffffffull), T);
       write_lane(dhi, (unsigned int)(dec >> 32), T);
     }
   }()),
   ...);
}

// chunk C of pass P: the chunk's cells build their tables (lanes C*18 .. C*18+17 hold their records), then every state
// walks them.  [t_lo, t_end] is wave-uniform and aligned to chunks.
#ifdef PHMRF_PHASE_DP
// development build: shader-clock cycles of the DP's sub-phases into the work counters (1 U conversion + unary loads,
// 2 records, 3 table builds, 4 walks, 5 everything after a move was found)
#define DPH(K_)                                                              \
  {                                                                          \
    const unsigned long long tn_ = __builtin_amdgcn_s_memtime();             \
    if (wave_lane() == 0) atomicAdd(&dp_wk[K_], (unsigned int)(tn_ - dp_t0) >> 4); \
    dp_t0 = tn_;                                                             \
  }
__device__ unsigned int* dp_wk_dummy;
#else
#define DPH(K_)
#endif
template <int P, int C, bool RECORD>
__device__ __forceinline__ void dp_chunk(float& m, unsigned long long& took, int lane, float* tabch, float c0, float c1,
                                         float wu, float wlu, float wl, float wld, int bits, int t_lo, int t_end,
                                         unsigned int& dlo, unsigned int& dhi, unsigned int live, unsigned int* dp_wk,
                                         unsigned long long& dp_t0) {
  constexpr int T0 = C * CH, T1 = (T0 + CH - 1 < 63) ? T0 + CH - 1 : 63;
  if (P * 64 + T0 > t_end || P * 64 + T1 < t_lo) return;
  // A chunk none of whose cells, nor the six before it, is in U: every cell it holds is pinned and the profile it starts
  // from is 000000 in every finite state -- walking it would only add the same constant to the one finite state.
  if (!((live >> (P * 4 + C)) & 1u)) return;
  if (lane >= T0 && lane <= T1) build_table(tabch, lane - T0, c0, c1, wu, wlu, wl, wld, bits);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  DPH(3)
  const char* tabc = reinterpret_cast<const char*>(tabch);
  if (RECORD) dp_chunk_steps_rec<P, C>(m, took, lane, tabc, dlo, dhi, std::make_integer_sequence<int, CH>{});
  else dp_chunk_steps<P, C>(m, took, lane, tabc, std::make_integer_sequence<int, CH>{});
  __builtin_amdgcn_wave_barrier();
  DPH(4)
}

template <int P, bool RECORD>
__device__ __forceinline__ void dp_pass_chunked(float& m, unsigned long long& took, int lane, float* tabch, float c0, float c1,
                                                float wu, float wlu, float wl, float wld, int bits, int t_lo, int t_end,
                                                unsigned int& dlo, unsigned int& dhi, unsigned int live, unsigned int* dp_wk,
                                                unsigned long long& dp_t0) {
  if (P * 64 > t_end || P * 64 + 63 < t_lo) return;
  dp_chunk<P, 0, RECORD>(m, took, lane, tabch, c0, c1, wu, wlu, wl, wld, bits, t_lo, t_end, dlo, dhi, live, dp_wk, dp_t0);
  dp_chunk<P, 1, RECORD>(m, took, lane, tabch, c0, c1, wu, wlu, wl, wld, bits, t_lo, t_end, dlo, dhi, live, dp_wk, dp_t0);
  dp_chunk<P, 2, RECORD>(m, took, lane, tabch, c0, c1, wu, wlu, wl, wld, bits, t_lo, t_end, dlo, dhi, live, dp_wk, dp_t0);
  dp_chunk<P, 3, RECORD>(m, took, lane, tabch, c0, c1, wu, wlu, wl, wld, bits, t_lo, t_end, dlo, dhi, live, dp_wk, dp_t0);
}

// ---- the exact filter on one strip, lane <-> strip column (shared by strip_cols_kernel and fusion_cols_kernel) -------
// cap_r (the discount sum over the in-strip neighbours that are still in U) lives in a register per row; a pass tests
// every row against its cap and SUBTRACTS the contributions of the cells it deletes from the caps of the rows around them
// (three selects and at most eight multiply-adds per row that lost a cell; a row that lost none costs nothing, and the pass
// that confirms the fixed point is twenty instructions).  Rows from the outside in (0, 4, 1, 3, 2): a strip's edge rows have
// the fewest in-strip neighbours, the peeling runs from them towards the middle row and gets there within one pass.  Any
// order and any stale (too large) cap is sound: U only ever loses cells that cannot be in a switching set.
//
// peel_drop<R>: the cells of row R in `del` leave U: their discounts come off the caps of the rows R - 1, R, R + 1
// (a cell of row r' is the d = 5 / 6 / 7 neighbour of row r' - 1, the d = 3 / 4 neighbour of row r', the d = 0 / 1 / 2
//  neighbour of row r' + 1; (dr, dc) of d: 0 (-1,-1) 1 (-1,0) 2 (-1,+1) 3 (0,-1) 4 (0,+1) 5 (+1,-1) 6 (+1,0) 7 (+1,+1))
template <int R>
__device__ __forceinline__ void peel_drop(float (&cap)[SH], const float (&v8)[SH][8], unsigned long long del) {
  if (del == 0ull) return;
  const float gl = __builtin_amdgcn_inverse_ballot_w64(del << 1) ? -1.f : 0.f;   // column c - 1 leaves
  const float gc = __builtin_amdgcn_inverse_ballot_w64(del) ? -1.f : 0.f;
  const float gr = __builtin_amdgcn_inverse_ballot_w64(del >> 1) ? -1.f : 0.f;   // column c + 1 leaves
  if (R > 0) {
    constexpr int A = R > 0 ? R - 1 : 0;
    cap[A] = __builtin_fmaf(gl, v8[A][5], cap[A]);
    cap[A] = __builtin_fmaf(gc, v8[A][6], cap[A]);
    cap[A] = __builtin_fmaf(gr, v8[A][7], cap[A]);
  }
  cap[R] = __builtin_fmaf(gl, v8[R][3], cap[R]);
  cap[R] = __builtin_fmaf(gr, v8[R][4], cap[R]);
  if (R < SH - 1) {
    constexpr int B = R < SH - 1 ? R + 1 : 0;
    cap[B] = __builtin_fmaf(gl, v8[B][0], cap[B]);
    cap[B] = __builtin_fmaf(gc, v8[B][1], cap[B]);
    cap[B] = __builtin_fmaf(gr, v8[B][2], cap[B]);
  }
}

#ifdef PHMRF_FILTER_STATS
// development: per-wave tallies of the filter (scalar): passes run, rows of a pass that lost a cell
#define PHMRF_FS_PARAMS , unsigned int& fs_drops_, unsigned int& fs_passes_
#define PHMRF_FS_ARGS , fs_drops_, fs_passes_
#else
#define PHMRF_FS_PARAMS
#define PHMRF_FS_ARGS
#endif
template <int R>
__device__ __forceinline__ void peel_row(const float (&sc)[SH], const float (&v8)[SH][8], float (&cap)[SH],
                                         unsigned long long (&U)[SH], unsigned long long& seeds, unsigned long long& gone
                                         PHMRF_FS_PARAMS) {
  // (a hair of slack for the f32 sums: the caps are carried by subtraction, so the slack has an absolute part that covers
  //  eight roundings at the largest cap a strip can have)
  const float capx = __builtin_fmaf(cap[R], 1.0001f, 2e-5f);
  const unsigned long long keep = __ballot(sc[R] <= capx);
  const unsigned long long sd = __ballot(sc[R] < 0.5f * capx);
  const unsigned long long del = U[R] & ~keep;
  U[R] &= keep;
  seeds |= U[R] & sd;
  gone |= del;
#ifdef PHMRF_FILTER_STATS
  if (del) ++fs_drops_;
#endif
  peel_drop<R>(cap, v8, del);
}

// -> true: no seed is left (no improving switching set exists on this strip); false: the DP has to decide, on U
__device__ __forceinline__ bool peel_strip(const float (&sc)[SH], const float (&v8)[SH][8], float (&cap)[SH],
                                           unsigned long long (&U)[SH], int peel_max PHMRF_FS_PARAMS) {
  for (int it = 0; it < peel_max; ++it) {
    unsigned long long seeds = 0ull, gone = 0ull;
#ifdef PHMRF_FILTER_STATS
    ++fs_passes_;
#endif
    peel_row<0>(sc, v8, cap, U, seeds, gone PHMRF_FS_ARGS);
    peel_row<4>(sc, v8, cap, U, seeds, gone PHMRF_FS_ARGS);
    peel_row<1>(sc, v8, cap, U, seeds, gone PHMRF_FS_ARGS);
    peel_row<3>(sc, v8, cap, U, seeds, gone PHMRF_FS_ARGS);
    peel_row<2>(sc, v8, cap, U, seeds, gone PHMRF_FS_ARGS);
    if (!seeds) return true;
    if (!gone) break;
  }
  return false;
}

// cell record of the DP for cell t of the strip, straight from the staged slab (strip_kernel's step B with a constant
// proposal alpha and the cells outside U pinned)
// FUSION: every cell has a proposal of its own (slabp) instead of the constant alpha.
template <int ORIENT, bool FUSION>
__device__ __forceinline__ void slab_record(const float* slabw, const unsigned char* slabl, const unsigned char* slabp, int t, int ncols,
                                            int ncell, int alpha, bool in_u, float u0, float u1, bool have, float& c0, float& c1,
                                            float (&w4)[4], int& bits) {
  c0 = 0.f;
  c1 = BIG;
  w4[0] = w4[1] = w4[2] = w4[3] = 0.f;
  bits = 0;
  if (!have) return;
  const int cc = t / SH, rr = t - cc * SH;
  const int e0 = (cc + 1) * EH + (rr + 1), w0 = (cc + 1) * SWC + (rr + 1) * 4;
  const int l = slabl[e0];
  const int pl = FUSION ? (int)slabp[e0] : alpha;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    constexpr int DR[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
    constexpr int DC[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
    const int dr = DR[d], dc = DC[d];
    const int di = ORIENT ? dc : dr, dj = ORIENT ? dr : dc;
    const bool fwd = di > 0 || (di == 0 && dj > 0);
    const int comp = (di == 0) ? 0 : (fwd ? dj + 2 : 2 - dj);
    const int en = e0 + dc * EH + dr, wn = w0 + dc * SWC + dr * 4;
    // (a backward neighbour in the bottom rim row -- orientation 1, left-down -- keeps its one needed weight in word 24)
    const int wb = (ORIENT == 1 && dr > 0 && !fwd) ? wn + comp - (rr == SH - 1 ? 1 : 0) : wn + comp;
    const float w = fwd ? slabw[w0 + comp] : slabw[wb];
    const int lj = slabl[en];
    const int r2 = rr + dr, c2 = cc + dc;
    const bool inside = r2 >= 0 && r2 < SH && c2 >= 0 && c2 < ncols;
    constexpr int QOF[8] = {1, 0, -1, 2, -1, 3, -1, -1};
    if (QOF[d] >= 0 && inside) {
      w4[QOF[d] >= 0 ? QOF[d] : 0] = w;
      const int pj = FUSION ? (int)slabp[en] : alpha;
      const int nib = (l != lj ? 1 : 0) | (l != pj ? 2 : 0) | (pl != lj ? 4 : 0) | (pl != pj ? 8 : 0);
      bits |= nib << (4 * (QOF[d] >= 0 ? QOF[d] : 0));
    }
    if (!inside) {
      if (l != lj) a0 += w;
      if (pl != lj) a1 += w;
    }
  }
  const bool can = in_u && l != pl && u1 < 1.0e29f;
  c0 = u0 + a0;
  c1 = can ? u1 + a1 : BIG;
}

// The DP of ONE flagged label on the wave's strip, lane <-> cell t = 64 p + lane (column-major), cells outside U pinned;
// everything it needs is in LDS (the staged slab, U as five row words).  Returns the number of cells it moved.
#ifndef PHMRF_DP_INLINE
#define PHMRF_DP_INLINE __noinline__
#endif
template <int ORIENT, bool FUSION, int WV = 1>
__device__ PHMRF_DP_INLINE unsigned int dp_flagged(StripGeom g, unsigned int lds, int kb, int lane,
                                                   int rs0, int ca, int ncols, int ncell, int alpha, int tick_a, int64_t n, int D,
                                                   const int32_t* nbr_, const float* uT_, uint8_t* labels_, uint16_t* stamp_,
                                                   uint16_t* mslot_, unsigned long long* changed_slot) {
    const global_ptr<const int32_t> nbr = as_global(nbr_);
    const global_ptr<const float> uT = as_global(uT_);
    const global_ptr<uint8_t> labels = as_global(labels_);
    const global_ptr<uint16_t> stamp = as_global(stamp_);
    const global_ptr<uint16_t> mslot = as_global(mslot_);       // this move's memo entry (label alpha's, or the fusion pass's)
    ColsLdsT<WV>* L = lds_object<ColsLdsT<WV>>(lds);
    const float* slabw = L->slabw;
    const unsigned char* slabl = L->slabl;
    const unsigned char* slabp = FUSION ? lds_object<FusLds>(lds)->slabp : L->slabl;
    float* tabch = L->tabch;
    const unsigned long long* ub = L->ubuf[kb];
    unsigned int* wk = L->wk;
    unsigned int* dp_wk = wk;
    unsigned long long dp_t0 = 0ull;
#ifdef PHMRF_PHASE_DP
    dp_t0 = __builtin_amdgcn_s_memtime();
#endif
    unsigned long long Ut[NPASS];
    bool inu[NPASS];
    int t_lo = NCELL_MAX, t_hi = -1;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      int t = p * 64 + lane;
      asm volatile("" : "+v"(t));
      const int cc = t / SH, rr = t - cc * SH;
      const unsigned long long ur = ub[rr];
      inu[p] = t < ncell && ((ur >> cc) & 1ull);
      Ut[p] = __ballot(inu[p]);
      if (Ut[p]) {
        const int first = p * 64 + __ffsll((long long)Ut[p]) - 1;
        const int last = p * 64 + 63 - __clzll((long long)Ut[p]);
        t_lo = first < t_lo ? first : t_lo;
        t_hi = last > t_hi ? last : t_hi;
      }
    }
    if (t_hi < 0) {
      if (mslot && lane == 0) *mslot = (uint16_t)tick_a;
      return 0u;
    }
    int t_end = t_hi + SH + 1;
    if (t_end > NPASS * 64 - 1) t_end = NPASS * 64 - 1;
    {   // widen to whole table chunks
      const int pe = t_end >> 6, re = t_end & 63;
      int r1 = (re / CH) * CH + CH - 1;
      if (r1 > 63) r1 = 63;
      t_end = pe * 64 + r1;
      const int pl = t_lo >> 6, rl = t_lo & 63;
      t_lo = pl * 64 + (rl / CH) * CH;
    }
#ifndef PHMRF_PHASE_CLOCK
    #ifndef PHMRF_FILTER_STATS
    if (lane == 0) atomicAdd(&wk[3], (unsigned int)(t_end - t_lo + 1));
#endif
#endif
#ifdef PHMRF_DP_COUNT
    if (lane == 0) {                     // development build: solver-trace counters (PHMRF_SOLVE_TRACE)
      atomicAdd(changed_slot - (FUSION ? 0 : alpha) - 8 + 100, 1ull);
      atomicAdd(changed_slot - (FUSION ? 0 : alpha) - 8 + 102, 1ull);
      atomicAdd(changed_slot - (FUSION ? 0 : alpha) - 8 + 103, (unsigned long long)(t_end - t_lo + 1));
    }
#endif
    // which table chunks hold, or directly follow, a cell of U (see dp_chunk)
    unsigned int live = 0u;
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int T0 = c * CH, T1 = (T0 + CH - 1 < 63) ? T0 + CH - 1 : 63;
        const int lo = T0 - SH - 1 > 0 ? T0 - SH - 1 : 0;
        unsigned long long hit = Ut[p] & ((~0ull >> (63 - T1)) & (~0ull << lo));
        if (T0 < SH + 1 && p > 0) hit |= Ut[p > 0 ? p - 1 : 0] >> (64 - (SH + 1 - T0));
        live |= (hit ? 1u : 0u) << (p * 4 + c);
      }
    // the unary terms of the passes the window touches (own label, alpha): all loads in flight together.  The records
    // themselves are formed pass by pass, right in front of the pass's table chunks: seven live registers, not 35.
    float ru0[NPASS], ru1[NPASS];
    unsigned int havem = 0u;                      // bit p: my cell of pass p is a node; bit 8 + p: ... and in U
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      ru0[p] = 0.f; ru1[p] = BIG;
      if (p * 64 > t_end || p * 64 + 63 < t_lo) continue;
      int t = p * 64 + lane;
      asm volatile("" : "+v"(t));
      const int cc = t / SH, rr = t - cc * SH;
      const int nd = t < ncell ? strip_node(g, rs0 + rr, ca + cc) : -1;
      const int e0 = (cc + 1) * EH + (rr + 1);
      const int l = nd >= 0 ? (int)slabl[e0] : 0;
      ru0[p] = uT[(int64_t)l * n + (nd >= 0 ? nd : 0)];
      const int pl = FUSION ? (nd >= 0 ? (int)slabp[e0] : 0) : alpha;
      ru1[p] = uT[(int64_t)pl * n + (nd >= 0 ? nd : 0)];
      havem |= (nd >= 0 ? 1u : 0u) << p;
      havem |= (inu[p] ? 1u : 0u) << (8 + p);
    }
    DPH(1)
#define PHMRF_DP_PASS(P_, REC_)                                                                                          \
  if (!(P_ * 64 > t_end || P_ * 64 + 63 < t_lo)) {                                                                     \
    int t = P_ * 64 + lane;                                                                                            \
    asm volatile("" : "+v"(t));                                                                                        \
    float w4[4], c0, c1;                                                                                               \
    int bits;                                                                                                          \
    slab_record<ORIENT, FUSION>(slabw, slabl, slabp, t, ncols, ncell, alpha, (havem >> (8 + P_)) & 1u, ru0[P_], ru1[P_], (havem >> P_) & 1u,   \
                        c0, c1, w4, bits);                                                                             \
    DPH(2)                                                                                                             \
    dp_pass_chunked<P_, REC_>(m, took, lane, tabch, c0, c1, w4[0], w4[1], w4[2], w4[3], bits, t_lo, t_end, dlo[P_],     \
                              dhi[P_], live, dp_wk, dp_t0);                                                               \
  }

    float m = lane == 0 ? 0.f : BIG;
    unsigned long long took = 0ull;
    unsigned int dlo[NPASS], dhi[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) dlo[p] = dhi[p] = 0u;
    PHMRF_DP_PASS(0, false) PHMRF_DP_PASS(1, false) PHMRF_DP_PASS(2, false) PHMRF_DP_PASS(3, false) PHMRF_DP_PASS(4, false)
    const float mmin = wave_min_f32(m);
    if (!(took & 1ull) && PHMRF_RL(m, 0) == mmin) {
      if (mslot && lane == 0) *mslot = (uint16_t)tick_a;
      return 0u;
    }
#ifdef PHMRF_DP_COUNT
    if (lane == 0) atomicAdd(changed_slot - (FUSION ? 0 : alpha) - 8 + 104, 1ull);
#endif
    DPH(4)
    // a move exists: walk again, this time recording the decision ballots
    const int q_end = t_end % 6;
    int sidx = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) sidx |= ((state_of_lane(lane) >> ((q_end - j + 6) % 6)) & 1) << j;
    m = lane == 0 ? 0.f : BIG;
    PHMRF_DP_PASS(0, true) PHMRF_DP_PASS(1, true) PHMRF_DP_PASS(2, true) PHMRF_DP_PASS(3, true) PHMRF_DP_PASS(4, true)
#undef PHMRF_DP_PASS
    const float mmin2 = wave_min_f32(m);
    const float cand = (m == mmin2) ? (float)sidx : 127.f;
    const float best = wave_min_f32(cand);
    int s = state_of_lane(__ffsll((long long)__ballot(cand == best)) - 1);
    unsigned int xsel[NPASS];
    s = __builtin_amdgcn_readfirstlane(s);
    backtrack_pass<4>(s, t_lo, t_end, dlo[4], dhi[4], xsel[4]);
    backtrack_pass<3>(s, t_lo, t_end, dlo[3], dhi[3], xsel[3]);
    backtrack_pass<2>(s, t_lo, t_end, dlo[2], dhi[2], xsel[2]);
    backtrack_pass<1>(s, t_lo, t_end, dlo[1], dhi[1], xsel[1]);
    backtrack_pass<0>(s, t_lo, t_end, dlo[0], dhi[0], xsel[0]);

    __builtin_amdgcn_wave_barrier();
    unsigned int my_changed = 0;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int t = p * 64 + lane;
      if (xsel[p] && t < ncell) {
        const int cc = t / SH, rr = t - cc * SH;
        const int node = strip_node(g, rs0 + rr, ca + cc);
        if (node >= 0) {
          const unsigned char nl = FUSION ? slabp[(cc + 1) * EH + (rr + 1)] : (unsigned char)alpha;
          labels[node] = nl;
          // (the staged labels follow: only this wave changes the cells of its strip during a pass, the rim belongs to
          //  nobody's strip -- after a move the slab is what a restaging would load, and the caller does not restage)
          const_cast<unsigned char*>(slabl)[(cc + 1) * EH + (rr + 1)] = nl;
          if (stamp) {
            stamp[node] = (uint16_t)tick_a;
            const global_ptr<const int32_t> nb2 = nbr + (int64_t)node * D;
            for (int j = 0; j < D; ++j)
              if (nb2[j] >= 0) stamp[nb2[j]] = (uint16_t)tick_a;
          }
          ++my_changed;
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) my_changed += __shfl_xor(my_changed, off, 64);
    if (lane == 0) {
      if (my_changed) atomicAdd(changed_slot, (unsigned long long)my_changed);
      if (mslot) *mslot = my_changed ? (uint16_t)0 : (uint16_t)tick_a;
    }
    DPH(5)
    return my_changed;
}

// The filter over the labels of `todo` on the wave's staged strip, lane <-> strip column.  Quiet labels get their memo
// entry; flagged ones go to ubuf / abuf (at most NBUF, then the caller turns to the DP).  Returns the labels not looked at
// yet; *nbuf_out (LDS) = how many are flagged.  Inlined into the kernel, with the DP out of line (dp_flagged): the two
// register sets -- ~80 registers of per-cell state here, the DP's records and tables there -- then never share one
// allocation (both inlined: 100+ VGPRs spilled), and an out-of-line filter would save and restore the 40 callee-saved
// VGPRs it needs on every call (measured: ~100 MB of scratch writes per launch on the chr1 block).
#ifndef PHMRF_FILTER_INLINE
#define PHMRF_FILTER_INLINE __forceinline__
#endif
template <int ORIENT, int WV = 1>
__device__ PHMRF_FILTER_INLINE unsigned long long filter_phase(StripGeom g, unsigned int lds, int lane, int rs0_, int ca_, int ncols_, int ncell_,
                                                             unsigned long long v0_, unsigned long long v1_, unsigned long long v2_,
                                                             unsigned long long v3_, unsigned long long v4_, unsigned long long todo_,
                                                             int64_t n, const float* uT_, uint16_t* mrow_, int tick0_, int peel_max_,
                                                             int wslot = 0) {
    const global_ptr<const float> uT = as_global(uT_);
    const global_ptr<uint16_t> mrow = as_global(mrow_);
    ColsLdsT<WV>* L = lds_object<ColsLdsT<WV>>(lds);
    const float* slabw = L->slabw;
    const unsigned char* slabl = L->slabl;
    unsigned long long (*ubuf)[SH] = L->ubuf + (WV > 1 ? wslot * NBUF : 0);      // (this wave's share of the buffer)
    int* abuf = L->abuf + (WV > 1 ? wslot * NBUF : 0);
    int* nbuf_out = &L->nbuf[WV > 1 ? wslot : 0];
    unsigned int* wk = L->wk;
#define PHMRF_UNI64(x) (((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)((x) >> 32)) << 32) | \
                        (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(x)))
    // (arguments arrive in vector registers; everything here is wave-uniform)
    const int rs0 = __builtin_amdgcn_readfirstlane(rs0_), ca = __builtin_amdgcn_readfirstlane(ca_);
    const int ncols = __builtin_amdgcn_readfirstlane(ncols_), ncell = __builtin_amdgcn_readfirstlane(ncell_);
    const int tick0 = __builtin_amdgcn_readfirstlane(tick0_), peel_max = __builtin_amdgcn_readfirstlane(peel_max_);
    g.H = __builtin_amdgcn_readfirstlane(g.H); g.W = __builtin_amdgcn_readfirstlane(g.W);
    g.diagonal = __builtin_amdgcn_readfirstlane(g.diagonal); g.orient = __builtin_amdgcn_readfirstlane(g.orient);
    g.Hs = __builtin_amdgcn_readfirstlane(g.Hs); g.Ws = __builtin_amdgcn_readfirstlane(g.Ws);
    unsigned long long valid[SH] = {PHMRF_UNI64(v0_), PHMRF_UNI64(v1_), PHMRF_UNI64(v2_), PHMRF_UNI64(v3_), PHMRF_UNI64(v4_)};
    unsigned long long todo = PHMRF_UNI64(todo_);
#undef PHMRF_UNI64
#ifdef PHMRF_PHASE_CLOCK
    // development build: cycles of the extraction (slot 1), the single-site costs (3), the sweeps (5 is the DP: the
    // caller's), sweeps run (0)
    unsigned int fpc[4] = {0u, 0u, 0u, 0u};
    unsigned long long fpt = __builtin_amdgcn_s_memtime();
#define FPH(K_)                                                           \
  {                                                                       \
    const unsigned long long tn_ = __builtin_amdgcn_s_memtime();          \
    fpc[K_] += (unsigned int)(tn_ - fpt);                                 \
    fpt = tn_;                                                            \
  }
#else
#define FPH(K_)
#endif
    // ---- extraction, lane <-> column: per cell the eight neighbour terms  v8 = w (2 - [l != l_d])  (= the discount
    //      of an in-strip edge; a neighbour labelled alpha differs from l, so the same register is the plain weight
    //      the single-site cost needs), the neighbour labels, the own-label unary term, the weight sum of the
    //      neighbours that share the label, and the cap over ALL in-strip neighbours (first sweep of absent labels)
    float v8[SH][8];
    unsigned int laba[SH], labb[SH];
    int ownl[SH];
    float ucur[SH], hs[SH], capf[SH];
    unsigned long long pres;
    int ndx[SH];                 // my column's nodes (absent cells: node 0, whose value is never used)
    {
      // (the column index through an opaque copy: otherwise the compiler keeps the whole extracted state in registers
      //  across the DP phase, to save the re-extraction on the path where the slab was not restaged -- and spills it)
      int lc = lane;
      asm volatile("" : "+v"(lc));
      const bool act = lc < ncols;
      unsigned long long pm = 0ull;
#pragma unroll
      for (int r = 0; r < SH; ++r) {
        const int e0 = (lc + 1) * EH + (r + 1);
        const bool have = (valid[r] >> lc) & 1ull;
        ownl[r] = have ? (int)slabl[act ? e0 : 0] : 0;
        ndx[r] = have ? strip_node(g, rs0 + r, ca + lc) : 0;
        ucur[r] = uT[(int64_t)ownl[r] * n + ndx[r]];
      }
#pragma unroll
      for (int r = 0; r < SH; ++r) {
        const bool have = (valid[r] >> lc) & 1ull;
        unsigned int la = 0u, lb = 0u;
        float h = 0.f, cf = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) v8[r][d] = 0.f;
        if (have) {
          const int e0 = (lc + 1) * EH + (r + 1), w0 = (lc + 1) * SWC + (r + 1) * 4;
          const int l = ownl[r];
          pm |= 1ull << (l & 63);
#pragma unroll
          for (int d = 0; d < 8; ++d) {
            constexpr int DR[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
            constexpr int DC[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
            const int dr = DR[d], dc = DC[d];
            const int di = ORIENT ? dc : dr, dj = ORIENT ? dr : dc;
            const bool fwd = di > 0 || (di == 0 && dj > 0);
            const int comp = (di == 0) ? 0 : (fwd ? dj + 2 : 2 - dj);
            const int en = e0 + dc * EH + dr, wn = w0 + dc * SWC + dr * 4;
            const int wb = (ORIENT == 1 && dr > 0 && !fwd && r == SH - 1) ? wn + comp - 1 : wn + comp;
            const float w = fwd ? slabw[w0 + comp] : slabw[wb];
            const int lj = slabl[en];
            const bool inside = (r + dr >= 0) && (r + dr < SH) && (lc + dc >= 0) && (lc + dc < ncols);
            const bool eq = lj == l;
            h += eq ? w : 0.f;
            const float v = eq ? w + w : w;
            v8[r][d] = v;
            cf += inside ? v : 0.f;
            pm |= 1ull << (lj & 63);
            if (d < 4) la |= (unsigned int)lj << (8 * d);
            else lb |= (unsigned int)lj << (8 * (d - 4));
          }
        }
        laba[r] = la; labb[r] = lb; hs[r] = h;
        capf[r] = cf;
        __builtin_amdgcn_sched_barrier(0);       // keep the rows apart: interleaving their LDS reads only costs registers
      }
      unsigned int p0 = (unsigned int)pm, p1 = (unsigned int)(pm >> 32);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        p0 |= (unsigned int)__shfl_xor((int)p0, off, 64);
        p1 |= (unsigned int)__shfl_xor((int)p1, off, 64);
      }
      pres = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)p1) << 32) |
             (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)p0);
    }
    
    // ---- where a label's unary terms of my column live: ONE wave-uniform base per label and per-lane 32-bit byte offsets
    //      (a load is  global_load_dword v, v_off, s[base] offset:imm:  no 64-bit address per lane and row).  node(i, j) =
    //      rowbase(i) + j with rowbase(i) = i W - i (i - 1) / 2 - i in an upper-triangular block, i W otherwise.
    //      Orientation 0: strip row r is grid row rs0 + r, lane c is grid column ca + c: one offset register per row.
    //      Orientation 1: strip row r is grid COLUMN rs0 + r, lane c is grid row ca + c: the five cells of a lane are
    //      consecutive floats -- one offset register and the instruction's immediate offset.
    //      Cells that do not exist read a neighbouring element (rows and lanes clamped into the block; a column index up to
    //      five beyond either end of its row, for which the planes carry UT_PAD floats of slack at both ends); what they load
    //      is never used.
    unsigned long long plane0 = reinterpret_cast<unsigned long long>(uT_), pstride = (unsigned long long)n * 4ull;
    unsigned int voffr[ORIENT == 0 ? SH : 1];
    {
      int lc = lane;
      asm volatile("" : "+v"(lc));
      lc = lc < ncols ? lc : ncols - 1;
      if (ORIENT == 0) {
        int i0 = rs0 < 0 ? 0 : (rs0 > g.H - 1 ? g.H - 1 : rs0);
        const int rb0 = (g.diagonal ? i0 * g.W - (i0 * (i0 - 1)) / 2 - i0 : i0 * g.W) + ca;
        plane0 += (unsigned long long)(long long)rb0 << 2;
#pragma unroll
        for (int r = 0; r < SH; ++r) {
          int i = rs0 + r;
          i = i < 0 ? 0 : (i > g.H - 1 ? g.H - 1 : i);
          const int rb = (g.diagonal ? i * g.W - (i * (i - 1)) / 2 - i : i * g.W) + ca;
          voffr[ORIENT == 0 ? r : 0] = (unsigned int)(rb - rb0 + lc) * 4u;
        }
      } else {
        const int i = ca + lc;
        const int rb = g.diagonal ? i * g.W - (i * (i - 1)) / 2 - i : i * g.W;
        const int rb0 = g.diagonal ? ca * g.W - (ca * (ca - 1)) / 2 - ca : ca * g.W;
        voffr[0] = (unsigned int)(rb - rb0) * 4u;
        plane0 += (unsigned long long)(long long)(rb0 + rs0) << 2;
      }
    }
    // (base and stride as values of their own: left as kernel arguments, the register allocator re-reads them from the
    //  kernel-argument segment inside the label loop -- an s_load and its wait per label)
    asm volatile("" : "+s"(plane0), "+s"(pstride));
#define PHMRF_LOAD_U1(ALPHA_)                                                                                          \
  {                                                                                                                    \
    unsigned long long pb_ = plane0 + (unsigned long long)(ALPHA_) * pstride;                                          \
    asm volatile("" : "+s"(pb_));                                                                                      \
    if (ORIENT == 1 && SH == 5) {                                                                                      \
      /* (orientation 1: a lane's five cells are consecutive floats -- one 16-byte load, which needs no more than the   \
          4-byte alignment it has, and one 4-byte load instead of five: 128 line look-ups per wave and label, not 320) */ \
      unsigned int vo_ = voffr[0];                                                                                     \
      asm volatile("" : "+v"(vo_));                                                                                    \
      const f32x4u q_ = *reinterpret_cast<const global_ptr<const f32x4u>>(pb_ + (unsigned long long)vo_);              \
      u1[SH - 1] = *reinterpret_cast<const global_ptr<const float>>(pb_ + (unsigned long long)vo_ + 16);               \
      u1[0] = q_.x; u1[1] = q_.y; u1[2] = q_.z; u1[3] = q_.w;                                                          \
    } else {                                                                                                           \
    _Pragma("unroll") for (int r = 0; r < SH; ++r) {                                                                   \
      /* (opaque, so that the zero extension of the lane offset is not hoisted out of the block that loads: base in     \
          SGPRs + zext(32-bit VGPR) + constant is what selects the saddr form) */                                       \
      unsigned int vo_ = voffr[ORIENT == 0 ? r : 0];                                                                   \
      asm volatile("" : "+v"(vo_));                                                                                    \
      u1[r] = *reinterpret_cast<const global_ptr<const float>>(pb_ + (unsigned long long)vo_ + (ORIENT == 0 ? 0 : 4 * r)); \
    }                                                                                                                  \
    }                                                                                                                  \
  }
    FPH(0)
    // ---- the filter over the labels still to do; flagged ones wait in ubuf
        int nbuf = 0;
    float u1[SH];
    int alpha_cur = __ffsll((long long)todo) - 1;
    PHMRF_LOAD_U1(alpha_cur)
    // (quiet labels and the pair count are kept on the scalar side and written once at the end: a global store per
    //  label put a store round trip into every label's s_waitcnt vmcnt(0), next to the prefetched unary terms)
    unsigned long long quiet_mask = 0ull;
    unsigned int n_pairs = 0u;
#ifdef PHMRF_FILTER_STATS
    // development: work slots 1 passes, 2 rows of a pass that lost a cell, 3 pairs whose label occurs in the rectangle,
    // 4 flagged pairs, 5 rows with a cell of the label (the caps' initial drops); slot 0 stays the pairs
    unsigned int fs_drops_ = 0u, fs_passes_ = 0u, fs_present = 0u, fs_flagged = 0u, fs_init = 0u;
#endif
    while (todo && nbuf < NBUF) {
      const int alpha = alpha_cur;
      todo &= todo - 1ull;
      ++n_pairs;
      const bool present = (pres >> alpha) & 1ull;
      float sc[SH];
      unsigned long long U[SH];
#pragma unroll
      for (int r = 0; r < SH; ++r) {
        float hist = 0.f;
        bool ne = true;
        if (present) {
#pragma unroll
          for (int d = 0; d < 8; ++d) {
            const int lj = (int)(((d < 4 ? laba[r] : labb[r]) >> (8 * (d & 3))) & 255u);
            hist += lj == alpha ? v8[r][d] : 0.f;
          }
          ne = ownl[r] != alpha;
        }
        sc[r] = (u1[r] - ucur[r]) + (hs[r] - hist);
        // (a cell whose unary term for alpha is "infinite" -- no proposal -- fails the first keep test; the DP tests it itself)
        U[r] = present ? (valid[r] & __ballot(ne)) : valid[r];
      }
        FPH(1)
      if (todo) {                                  // the next label's terms, in flight during the sweeps
        alpha_cur = __ffsll((long long)todo) - 1;
        PHMRF_LOAD_U1(alpha_cur)
      }
      // ---- the filter (peel_strip): the caps start from the label-independent sum over ALL in-strip neighbours, minus --
      //      for a label that occurs in the rectangle -- the cells that carry it
      FPH(3)
      float cap[SH];
#pragma unroll
      for (int r = 0; r < SH; ++r) cap[r] = capf[r];
#ifdef PHMRF_FILTER_STATS
      if (present) {
        ++fs_present;
        for (int r = 0; r < SH; ++r) fs_init += (valid[r] & ~U[r]) ? 1u : 0u;
      }
#endif
      if (present) {
        peel_drop<0>(cap, v8, valid[0] & ~U[0]);
        peel_drop<1>(cap, v8, valid[1] & ~U[1]);
        peel_drop<2>(cap, v8, valid[2] & ~U[2]);
        peel_drop<3>(cap, v8, valid[3] & ~U[3]);
        peel_drop<4>(cap, v8, valid[4] & ~U[4]);
      }
      const bool quiet = peel_strip(sc, v8, cap, U, peel_max PHMRF_FS_ARGS);
      FPH(2)
      if (quiet) {
        quiet_mask |= 1ull << alpha;
        continue;
      }
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < SH; ++r) ubuf[nbuf][r] = U[r];
        abuf[nbuf] = alpha;
      }
#ifdef PHMRF_FILTER_STATS
      ++fs_flagged;
#endif
      ++nbuf;
    }
#if defined(PHMRF_PHASE_CLOCK) && !defined(PHMRF_PHASE_DP)
    if (lane == 0) {
      atomicAdd(&wk[1], fpc[0] >> 4);      // extraction
      atomicAdd(&wk[3], fpc[1] >> 4);      // single-site costs
      atomicAdd(&wk[5], (fpc[2] + fpc[3]) >> 4);      // sweeps (+ prefetch issue)
    }
#endif
#undef FPH
#undef PHMRF_LOAD_U1
    if (mrow && ((quiet_mask >> lane) & 1ull)) mrow[lane] = (uint16_t)(tick0 + lane);      // lane <-> label
    if (lane == 0) {
      *nbuf_out = nbuf;
      atomicAdd(&wk[0], n_pairs);
#ifdef PHMRF_FILTER_STATS
      atomicAdd(&wk[1], fs_passes_);
      atomicAdd(&wk[2], fs_drops_);
      atomicAdd(&wk[3], fs_present);
      atomicAdd(&wk[4], fs_flagged);
      atomicAdd(&wk[5], fs_init);
#elif !defined(PHMRF_PHASE_CLOCK)
      atomicAdd(&wk[5], n_pairs * (unsigned int)ncell);
#endif
    }
    return todo;
}

// Which strip a workgroup of strip_cols_kernel takes.  Orientation 0: the workgroups' order is the strips' order (a strip's
// rows are contiguous in memory, neighbouring strips share nothing but a segment's end).  Orientation 1: a lane's five cells
// are 20 consecutive bytes of ONE grid row, a wave's load of a label's terms touches 63 lines of 128 bytes, and the strips of
// the bands next door (same segment: the next six grid columns) read the SAME lines -- 5.3 bands to a line.  Workgroups are
// dealt round-robin to the 8 XCDs, each with an L2 of its own (MI355X_MICROARCH.md, workgroup dispatch: blocks b and b + 8
// share an XCD; speed only, never correctness): in the strips' own order the bands that share lines are nsegs workgroups apart
// and land in different L2s, and every line came from HBM once per band (PMC, round 4: 293 MB fetched per launch against
// 109 MB in orientation 0).  So: groups of six adjacent bands are dealt to the XCD labels round-robin (an upper-triangular
// block's work grows with the band, neighbouring groups weigh the same), and within a label the workgroups run segment by
// segment through a group's six bands -- six consecutive workgroups of one XCD read one set of lines.  A bijection onto the
// strips plus padding workgroups that return at once; the strips of a launch are independent, so the labelling does not
// depend on the order.
#ifndef PHMRF_XCD_GROUP_BANDS
#define PHMRF_XCD_GROUP_BANDS 6
#endif
constexpr int XCD_GROUP_BANDS = PHMRF_XCD_GROUP_BANDS;
__host__ __device__ inline int strip_slots(int orient, int nbands, int nsegs, int xcd) {
  if (!orient || !xcd) return nbands * nsegs;
  const int ngroups = (nbands + XCD_GROUP_BANDS - 1) / XCD_GROUP_BANDS;
  return ((ngroups + 7) / 8) * 8 * XCD_GROUP_BANDS * nsegs;
}
template <int ORIENT>
__device__ __forceinline__ int strip_of_slot(const StripGeom& g, int q) {
  if (ORIENT == 1 && g.xcd) {
    const int per = XCD_GROUP_BANDS * g.nsegs;
    const int x = q & 7, j = q >> 3;
    const int gidx = j / per, rem = j - gidx * per;
    const int seg = rem / XCD_GROUP_BANDS, bi = rem - seg * XCD_GROUP_BANDS;
    const int bnd = (gidx * 8 + x) * XCD_GROUP_BANDS + bi;
    return bnd < g.nbands ? bnd * g.nsegs + seg : -1;
  }
  return q;
}

// ---------------------------------------------------------------------------------------------------------------
// strip_scan_kernel (round 6): what every strip of the launch that follows has to do, decided BEFORE that launch by a light
// kernel (no LDS slab, ~30 VGPRs, four strips per workgroup) instead of in the prologue of its 127-VGPR, 10 KB-LDS waves:
//   * the memo test: the strip's newest change stamp against the ticks of its labels' last quiet runs (as before);
//   * the SEED MASKS (expansions only): the OR over the strip's cells of the per-node masks propose_grid_kernel wrote.  A
//     label none of whose bits is set has no seed on this strip in the filter's first pass, hence in no pass: it is quiet
//     exactly as if the filter had run, gets its memo entry here, and the expansion kernel never loads its unary terms.  The
//     masks are used only where no cell of the strip carries a stamp later than the proposals' launch (seed_tick): a label
//     changed since then in or next to the strip makes every cell it touches carry one (dilated stamps).
// out[2 slot] = the labels the memo leaves (what the strip falls back to after a move of its own: the masks are stale
// then), out[2 slot + 1] = those that also have a seed -- 0: the launch's workgroup returns after one scalar load.
// FUSION: the fusion pass's memo entry (slot K) alone; out[2 slot] = out[2 slot + 1] = 1 when the pass has to look.
template <int ORIENT, bool FUSION>
__global__ __launch_bounds__(256) void strip_scan_kernel(StripGeom g, int K, unsigned long long label_mask,
                                                         const uint16_t* __restrict__ stamp, uint16_t* __restrict__ memo, int tick0,
                                                         const unsigned long long* __restrict__ seed, int seed_tick,
                                                         unsigned long long* __restrict__ out, unsigned long long* __restrict__ work) {
  __shared__ unsigned int acc[2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nstrips = g.nbands * g.nsegs;
  const int nslots = strip_slots(ORIENT, g.nbands, g.nsegs, g.xcd);
  if (threadIdx.x < 2) acc[threadIdx.x] = 0u;
  __syncthreads();
  for (int base = blockIdx.x * 4; base < nslots; base += gridDim.x * 4) {
    const int slot = __builtin_amdgcn_readfirstlane(base + wv);
    if (slot >= nslots) continue;
    unsigned long long tf = 0ull, td = 0ull;
    const int strip = strip_of_slot<ORIENT>(g, slot);
    if (strip >= 0 && strip < nstrips) {
      const int bnd = strip / g.nsegs;
      const int seg = strip - bnd * g.nsegs;
      const int rs0 = bnd * (SH + 1) - g.shift_r;
      const int cs0 = seg * 64 - g.shift_c;
      const int ca = cs0 > 0 ? cs0 : 0;
      const int cb = (cs0 + SL < g.Ws) ? cs0 + SL : g.Ws;
      const int ncols = cb > ca ? cb - ca : 0;
      const bool empty = ncols <= 0 ||
                         (g.diagonal && (ORIENT == 0 ? (rs0 > 0 ? rs0 : 0) > cb - 1 : ca > (rs0 + SH - 1 < g.Hs - 1 ? rs0 + SH - 1 : g.Hs - 1)));
      if (!empty) {
        int nd[SH];
#pragma unroll
        for (int r = 0; r < SH; ++r) nd[r] = lane < ncols ? strip_node(g, rs0 + r, ca + lane) : -1;
        const uint16_t* mrow = memo + (int64_t)strip * (K + 1);
        // every load of the wave before any use
        int st[SH];
        unsigned long long sm[SH];
#pragma unroll
        for (int r = 0; r < SH; ++r) st[r] = stamp[nd[r] >= 0 ? nd[r] : 0];
        const int lq = FUSION ? (int)mrow[K] : (lane < K ? (int)mrow[lane] : 0);
        if (!FUSION && seed) {
#pragma unroll
          for (int r = 0; r < SH; ++r) sm[r] = seed[nd[r] >= 0 ? nd[r] : 0];
        }
        int nw = 0;
        unsigned int have = 0u;
#pragma unroll
        for (int r = 0; r < SH; ++r) {
          nw = (nd[r] >= 0 && st[r] > nw) ? st[r] : nw;
          have |= nd[r] >= 0 ? 1u : 0u;
        }
        unsigned int s0 = 0u, s1 = 0u;
        if (!FUSION && seed) {
#pragma unroll
          for (int r = 0; r < SH; ++r) {
            s0 |= nd[r] >= 0 ? (unsigned int)sm[r] : 0u;
            s1 |= nd[r] >= 0 ? (unsigned int)(sm[r] >> 32) : 0u;
          }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const int o2 = __shfl_xor(nw, off, 64);
          nw = o2 > nw ? o2 : nw;
          if (!FUSION && seed) {
            s0 |= (unsigned int)__shfl_xor((int)s0, off, 64);
            s1 |= (unsigned int)__shfl_xor((int)s1, off, 64);
          }
        }
        const bool any_node = __ballot(have != 0u) != 0ull;
        if (any_node) {
          if (FUSION) {
            tf = td = (lq && nw < lq) ? 0ull : 1ull;
          } else {
            tf = label_mask & __ballot(lane < K && !(lq && nw < lq));
            td = tf;
            if (seed && nw <= seed_tick) {
              td = tf & (((unsigned long long)s1 << 32) | (unsigned long long)s0);
              const unsigned long long pruned = tf & ~td;
              // quiet without the filter: the memo entry the filter would have written (lane <-> label)
              if ((pruned >> lane) & 1ull) const_cast<uint16_t*>(mrow)[lane] = (uint16_t)(tick0 + lane);
              if (lane == 0 && pruned) {
                const unsigned int cells = (unsigned int)(ncols * SH);
                atomicAdd(&acc[0], (unsigned int)__popcll(pruned) * cells);     // label cells settled by the masks
                if (!td) atomicAdd(&acc[1], cells);                             // ... cells of strips settled entirely
              }
            }
          }
        }
      }
    }
    if (lane == 0) {
      out[2 * (int64_t)slot] = tf;
      out[2 * (int64_t)slot + 1] = td;
    }
  }
  __syncthreads();
  if (work && threadIdx.x < 2) {
    const unsigned int v = acc[threadIdx.x];
    if (v) atomicAdd(work + (blockIdx.x & (WORK_BANKS - 1)) * WORK_SLOTS + 7 + threadIdx.x, (unsigned long long)v);
  }
}

#ifndef PHMRF_COLS_WPE
#define PHMRF_COLS_WPE 4
#endif
template <int ORIENT, int WV = 1>
__global__ __launch_bounds__(64 * WV, PHMRF_COLS_WPE) void strip_cols_kernel(StripGeom g, int64_t n, int K, int D,
                                                                         const int32_t* __restrict__ nbr,
                                                                         const float4* __restrict__ fwd_w,
                                                                         const float* __restrict__ uT, uint8_t* __restrict__ labels,
                                                                         float beta, unsigned long long label_mask,
                                                                         unsigned long long* __restrict__ changed,
                                                                         uint16_t* __restrict__ stamp, uint16_t* __restrict__ memo,
                                                                         int tick0, unsigned long long* __restrict__ work, int peel_max,
                                                                         const unsigned long long* __restrict__ scan) {
  __shared__ ColsLdsT<WV> lds_pool;
  float* slabw = lds_pool.slabw;
  unsigned char* slabl = lds_pool.slabl;
  int* abuf = lds_pool.abuf;
  unsigned int* wk = lds_pool.wk;
  const unsigned int lds = lds_address_of(&lds_pool);
#ifdef PHMRF_COLS_LDS_PAD        // development: occupancy experiments
  __shared__ volatile float lds_pad[PHMRF_COLS_LDS_PAD / 4];
  lds_pad[(threadIdx.x * 61) % (PHMRF_COLS_LDS_PAD / 4)] = 1.f;
#endif
  const int lane = threadIdx.x & 63;
  const int nstrips = g.nbands * g.nsegs;
  if (threadIdx.x < WORK_SLOTS) wk[threadIdx.x] = 0u;
  __syncthreads();
#ifdef PHMRF_PHASE_CLOCK
  unsigned int phc[6] = {0u, 0u, 0u, 0u, 0u, 0u};
  unsigned long long pht = __builtin_amdgcn_s_memtime();
#define PH(K_)                                                            \
  {                                                                       \
    const unsigned long long tn_ = __builtin_amdgcn_s_memtime();          \
    phc[K_] += (unsigned int)(tn_ - pht);                                 \
    pht = tn_;                                                            \
  }
#else
#define PH(K_)
#endif

  const int nslots = strip_slots(ORIENT, g.nbands, g.nsegs, g.xcd);
  for (int slot_v = blockIdx.x; slot_v < nslots; slot_v += gridDim.x) {
    const int slot_u = __builtin_amdgcn_readfirstlane(slot_v);
    // (inside a solve strip_scan_kernel has looked at the strip's stamps, memo row and seed masks: two words per slot)
    unsigned long long todo = label_mask, todo_full = label_mask;
    if (scan) {
      todo_full = scan[2 * (int64_t)slot_u];
      todo = scan[2 * (int64_t)slot_u + 1];
      if (!todo) continue;
    }
    const int strip = strip_of_slot<ORIENT>(g, slot_u);
    if (strip < 0 || strip >= nstrips) continue;
    const int bnd = strip / g.nsegs;
    const int seg = strip - bnd * g.nsegs;
    const int rs0 = bnd * (SH + 1) - g.shift_r;
    const int cs0 = seg * 64 - g.shift_c;
    const int ca = cs0 > 0 ? cs0 : 0;
    const int cb = (cs0 + SL < g.Ws) ? cs0 + SL : g.Ws;
    const int ncols = cb > ca ? cb - ca : 0;
    const int ncell = ncols * SH;
    if (ncell <= 0) continue;
    // A strip of the grid's bounding rectangle that holds no node: in an upper-triangular block every strip below the
    // diagonal -- half of them.  (Until round 5 such a strip went through staging, extraction and the filter of every
    // label with empty masks: a third of a real strip's instructions for nothing.)
    if (g.diagonal && (ORIENT == 0 ? (rs0 > 0 ? rs0 : 0) > cb - 1 : ca > (rs0 + SH - 1 < g.Hs - 1 ? rs0 + SH - 1 : g.Hs - 1))) continue;
    PH(1)

    // ---- column layout: lane c <-> strip column c; the node of row r is nodec[r] (-1: none)
    unsigned long long valid[SH];
    uint16_t* mrow = memo ? memo + (int64_t)strip * (K + 1) : nullptr;
    {
      int nw = 0;
      int nd[SH];
#pragma unroll
      for (int r = 0; r < SH; ++r) nd[r] = lane < ncols ? strip_node(g, rs0 + r, ca + lane) : -1;
      if (mrow && !scan) {
#pragma unroll
        for (int r = 0; r < SH; ++r) {
          const int st = stamp[nd[r] >= 0 ? nd[r] : 0];
          nw = (nd[r] >= 0 && st > nw) ? st : nw;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const int o2 = __shfl_xor(nw, off, 64);
          nw = o2 > nw ? o2 : nw;
        }
        const int lq = lane < K ? (int)mrow[lane] : 0;
        todo &= __ballot(lane < K && !(lq && nw < lq));
        todo_full = todo;
      }
#pragma unroll
      for (int r = 0; r < SH; ++r) valid[r] = __ballot(nd[r] >= 0);
    }
    todo = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(todo >> 32)) << 32) |
           (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)todo);
    todo_full = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(todo_full >> 32)) << 32) |
                (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)todo_full);
    if (!todo || !(valid[0] | valid[1] | valid[2] | valid[3] | valid[4])) continue;
    auto stage_strip = [&]() {
      // ---- staging (as strip_kernel, step A): labels and forward weights of the strip's rectangle and rim -> LDS
      constexpr int NEP = (ECELLS + 63) / 64;
      int enode[NEP], eidx[NEP];
#pragma unroll
      for (int q = 0; q < NEP; ++q) {
        int er, ec;
        if (ORIENT == 0) {
          int l2 = lane;
          asm volatile("" : "+v"(l2));
          er = q < EH ? q : l2;
          ec = q < EH ? l2 : 64;
          if (q >= EH && l2 >= EH) ec = 1 << 20;
        } else {
          int e = q * 64 + lane;
          asm volatile("" : "+v"(e));
          ec = e / EH;
          er = e - ec * EH;
        }
        const bool have = ec < ncols + 2;
        eidx[q] = have ? ec * EH + er : -1;
        enode[q] = have ? strip_node(g, rs0 - 1 + er, ca - 1 + ec) : -1;
      }
      int elab[NEP];
      float4 ef[NEP];
#pragma unroll
      for (int q = 0; q < NEP; ++q) {
        const int node = enode[q];
        elab[q] = 0;
        ef[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (node >= 0) {
          elab[q] = labels[node];
          ef[q] = fwd_w[node];
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < NEP; ++q) {
        const int e = eidx[q];
        if (e >= 0) {
          const int ec = e / EH, er = e - ec * EH;
          float* wr = slabw + ec * SWC + er * 4;
          if (er < EH - 1) {
            wr[0] = ef[q].x * beta;
            wr[1] = ef[q].y * beta;
            wr[2] = ef[q].z * beta;
            wr[3] = ef[q].w * beta;
          } else {
            // the bottom rim row holds one edge the strip needs, and only in orientation 1: its cells' grid edge (+1, -1),
            // which runs to strip row 4 of the next column; it lives in the column's 25th word
            wr[0] = ef[q].y * beta;
          }
          slabl[e] = (unsigned char)(enode[q] >= 0 ? elab[q] : 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) {
#if !defined(PHMRF_PHASE_CLOCK) && !defined(PHMRF_FILTER_STATS)
        atomicAdd(&wk[2], (unsigned int)(EH * (ncols + 2)));
        atomicAdd(&wk[4], (unsigned int)ncell);
#endif
      }
    };
    bool staged = false;
    if constexpr (WV > 1) {
      // ---- WV waves on ONE strip (the launches of a solve's late rounds, where a handful of dirty strips is all there is and
      //      a launch takes as long as one wave needs to walk one strip's labels in sequence).  Wave 0 stages the strip; every
      //      wave runs the exact FILTER of every WV-th listed label on the staged labelling -- read-only work, so the verdicts
      //      are those of the sequential order as long as nothing has moved --; then wave 0 alone runs the DPs of the flagged
      //      labels in ascending order.  The first DP that moves a cell makes every later verdict stale: from there on wave 0
      //      continues as the one-wave kernel does, filtering every later label again (the memo entries the other waves
      //      wrote for those labels are overwritten by that second look: its stores come after theirs, fence + barrier).
      //      Label for label the launch is the one-wave launch.
      const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
      if (wave == 0) stage_strip();
      staged = true;
      __syncthreads();
      unsigned long long mine = 0ull;
      {
        unsigned long long t = todo;
        int i = 0;
        while (t) {
          const int a = __ffsll((long long)t) - 1;
          t &= t - 1ull;
          if ((i++ % WV) == wave) mine |= 1ull << a;
        }
      }
      unsigned long long left = 0ull;
      if (mine) {
        left = filter_phase<ORIENT, WV>(g, lds, lane, rs0, ca, ncols, ncell, valid[0], valid[1], valid[2], valid[3], valid[4], mine, n,
                                        uT, mrow, tick0, peel_max, wave);
        left = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(left >> 32)) << 32) |
               (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)left);
      } else if (lane == 0) {
        lds_pool.nbuf[wave] = 0;
      }
      if (lane == 0) lds_pool.left[wave] = left;
      __threadfence();
      __syncthreads();
      unsigned long long rest = 0ull;
      if (wave == 0) {
        // labels nobody has looked at (a wave's buffer of flagged labels was full: rare) bound the merged pass from above
        unsigned long long unl = 0ull;
#pragma unroll
        for (int w = 0; w < WV; ++w) unl |= lds_pool.left[w];
        unl = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unl >> 32)) << 32) |
              (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)unl);
        const int cut = unl ? __ffsll((long long)unl) - 1 : 64;
        int head[WV], cnt[WV];
#pragma unroll
        for (int w = 0; w < WV; ++w) {
          head[w] = 0;
          cnt[w] = __builtin_amdgcn_readfirstlane(lds_pool.nbuf[w]);
        }
        rest = cut < 64 ? (todo & ~((1ull << cut) - 1ull)) : 0ull;
        for (;;) {
          int bw = -1, ba = 64;
#pragma unroll
          for (int w = 0; w < WV; ++w)
            if (head[w] < cnt[w]) {
              const int a = __builtin_amdgcn_readfirstlane(lds_pool.abuf[w * NBUF + head[w]]);
              if (a < ba) {
                ba = a;
                bw = w;
              }
            }
          if (bw < 0 || ba >= cut) break;
          int kb = 0;
#pragma unroll
          for (int w = 0; w < WV; ++w)
            if (w == bw) kb = w * NBUF + head[w]++;
          const unsigned int my_changed = dp_flagged<ORIENT, false, WV>(g, lds, kb, lane, rs0, ca, ncols, ncell, ba, tick0 + ba, n, D, nbr, uT,
                                                                        labels, stamp, mrow ? mrow + ba : nullptr, changed + ba);
          if (my_changed) {
            __threadfence();
            rest = todo_full & ~((2ull << ba) - 1ull);
            break;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      todo = rest;                             // (waves 1 .. WV - 1: nothing; wave 0: what is left for the sequential loop)
    }

    while (todo) {
      if (!staged) {
        stage_strip();
        staged = true;
      }

      PH(2)
      todo = filter_phase<ORIENT, WV>(g, lds, lane, rs0, ca, ncols, ncell, valid[0], valid[1], valid[2], valid[3], valid[4], todo, n, uT,
                                      mrow, tick0, peel_max, 0);
      todo = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(todo >> 32)) << 32) |
             (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)todo);
      PH(0)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();

      // ---- the DP of the flagged labels, lane <-> cell t = 64 p + lane (column-major), cells outside U pinned
#ifdef PHMRF_COLS_NO_DP            // development: the filter alone (timing / register experiments; the labelling is wrong)
      const int nbuf = 0;
#else
      const int nbuf = __builtin_amdgcn_readfirstlane(lds_pool.nbuf[0]);
#endif
      for (int kb = 0; kb < nbuf; ++kb) {
        const int alpha = __builtin_amdgcn_readfirstlane(abuf[kb]);
        const unsigned int my_changed = dp_flagged<ORIENT, false, WV>(g, lds, kb, lane, rs0, ca, ncols, ncell, alpha, tick0 + alpha, n, D, nbr,
                                                                      uT, labels, stamp, mrow ? mrow + alpha : nullptr, changed + alpha);
        PH(4)
        if (my_changed) {
          // the labels of this strip have changed: everything later is filtered again on the new labelling (the slab's
          // labels were patched by the DP's apply step; its weights do not depend on the labels: no restaging)
          __threadfence();
#ifdef PHMRF_COLS_RESTAGE
          staged = false;
#endif
          // (every later label the memo left -- those the seed masks had settled, too: the masks are those of the labels
          //  before the move)
          todo = todo_full & ~((2ull << alpha) - 1ull);
          break;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if constexpr (WV > 1) __syncthreads();       // (the strip's LDS is free for the workgroup's next strip)
  }
#ifdef PHMRF_PHASE_CLOCK
  // slots: 0 pairs + 4096 x general sweeps, 1 extraction, 2 ids/memo/staging, 3 single-site costs, 4 DP, 5 sweeps
#ifndef PHMRF_PHASE_DP
  if (lane == 0) {
    wk[2] += (phc[1] + phc[2]) >> 4;
    wk[4] += phc[4] >> 4;
  }
#endif
#endif
#undef PH
  __syncthreads();
  if (work && threadIdx.x < WORK_SLOTS) {
    const unsigned int v = wk[threadIdx.x];
    if (v) atomicAdd(work + (blockIdx.x & (WORK_BANKS - 1)) * WORK_SLOTS + threadIdx.x, (unsigned long long)v);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// fusion_cols_kernel (round 3): the FUSION pass -- every cell keeps its label or takes its proposal (its best alternative
// label, launch_propose) -- with the exact filter of the expansions in front of the DP.
//
// A cell's single-site cost s_i (switch i alone to p_i) and, per in-strip edge, what the pair saves against the sum of the
// two single-site costs when both ends switch,  disc_ij = w ([p_i != l_j] + [l_i != p_j] - [l_i != l_j] - [p_i != p_j]),
// play the parts they play in strip_cols_kernel.  disc_ij can be negative here (the pair table of a fusion move need not
// be submodular); the filter uses max(0, disc_ij): every bound it relies on -- a member of an optimal switching set cannot
// leave at a profit, an improving set has a seed -- only gets weaker with a larger discount, so the deletion stays sound
// and the DP (exact for any 2 x 2 tables) decides what is left.  At the warm start of an EM iteration the proposals are
// near-ICM-optimal alternatives, s_i >= 0 almost everywhere, and most strips are settled by the filter: the pass no longer
// walks 315 DP steps per strip to find nothing.
#ifndef PHMRF_FUSION_WPE
#define PHMRF_FUSION_WPE 4
#endif
template <int ORIENT>
__global__ __launch_bounds__(64, PHMRF_FUSION_WPE) void fusion_cols_kernel(StripGeom g, int64_t n, int K, int D,
                                                                          const int32_t* __restrict__ nbr,
                                                                          const float4* __restrict__ fwd_w,
                                                                          const float* __restrict__ uT, uint8_t* __restrict__ labels,
                                                                          const uint8_t* __restrict__ prop,
                                                                          const float* __restrict__ sgain, float beta,
                                                                          unsigned long long* __restrict__ changed,
                                                                          uint16_t* __restrict__ stamp, uint16_t* __restrict__ memo,
                                                                          int tick, unsigned long long* __restrict__ work, int peel_max,
                                                                          const unsigned long long* __restrict__ scan) {
  __shared__ FusLds lds_pool;
  float* slabw = lds_pool.c.slabw;
  unsigned char* slabl = lds_pool.c.slabl;
  unsigned char* slabp = lds_pool.slabp;
  unsigned int* wk = lds_pool.c.wk;
  const unsigned int lds = lds_address_of(&lds_pool);
  const int lane = threadIdx.x & 63;
  const int nstrips = g.nbands * g.nsegs;
  if (threadIdx.x < WORK_SLOTS) wk[threadIdx.x] = 0u;
  __syncthreads();

  const int nslots = strip_slots(ORIENT, g.nbands, g.nsegs, g.xcd);      // (the XCD-aware order of orientation 1: strip_of_slot)
  for (int slot_v = blockIdx.x; slot_v < nslots; slot_v += gridDim.x) {
    const int slot_u = __builtin_amdgcn_readfirstlane(slot_v);
    if (scan && !scan[2 * (int64_t)slot_u + 1]) continue;      // (strip_scan_kernel: the memo entry is newer than every stamp)
    const int strip = strip_of_slot<ORIENT>(g, slot_u);
    if (strip < 0 || strip >= nstrips) continue;
    const int bnd = strip / g.nsegs;
    const int seg = strip - bnd * g.nsegs;
    const int rs0 = bnd * (SH + 1) - g.shift_r;
    const int cs0 = seg * 64 - g.shift_c;
    const int ca = cs0 > 0 ? cs0 : 0;
    const int cb = (cs0 + SL < g.Ws) ? cs0 + SL : g.Ws;
    const int ncols = cb > ca ? cb - ca : 0;
    const int ncell = ncols * SH;
    if (ncell <= 0) continue;
    if (g.diagonal && (ORIENT == 0 ? (rs0 > 0 ? rs0 : 0) > cb - 1 : ca > (rs0 + SH - 1 < g.Hs - 1 ? rs0 + SH - 1 : g.Hs - 1))) continue;   // no node (see strip_cols_kernel)

    // ---- column layout, memo test (slot K of the strip's memo row: the fusion pass)
    unsigned long long valid[SH];
    int ndx[SH];
    uint16_t* mslot = memo ? memo + (int64_t)strip * (K + 1) + K : nullptr;
#pragma unroll
    for (int r = 0; r < SH; ++r) ndx[r] = lane < ncols ? strip_node(g, rs0 + r, ca + lane) : -1;
    if (mslot && !scan) {
      const int last_quiet = *mslot;
      if (last_quiet) {
        int nw = 0;
#pragma unroll
        for (int r = 0; r < SH; ++r) {
          const int st = stamp[ndx[r] >= 0 ? ndx[r] : 0];
          nw = (ndx[r] >= 0 && st > nw) ? st : nw;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const int o2 = __shfl_xor(nw, off, 64);
          nw = o2 > nw ? o2 : nw;
        }
        if (nw < last_quiet) continue;
      }
    }
#pragma unroll
    for (int r = 0; r < SH; ++r) valid[r] = __ballot(ndx[r] >= 0);
    if (!(valid[0] | valid[1] | valid[2] | valid[3] | valid[4])) continue;

    // ---- staging: labels, proposals and forward weights of the strip's rectangle and rim -> LDS
    {
      constexpr int NEP = (ECELLS + 63) / 64;
      int enode[NEP], eidx[NEP];
#pragma unroll
      for (int q = 0; q < NEP; ++q) {
        int er, ec;
        if (ORIENT == 0) {
          int l2 = lane;
          asm volatile("" : "+v"(l2));
          er = q < EH ? q : l2;
          ec = q < EH ? l2 : 64;
          if (q >= EH && l2 >= EH) ec = 1 << 20;
        } else {
          int e = q * 64 + lane;
          asm volatile("" : "+v"(e));
          ec = e / EH;
          er = e - ec * EH;
        }
        const bool have = ec < ncols + 2;
        eidx[q] = have ? ec * EH + er : -1;
        enode[q] = have ? strip_node(g, rs0 - 1 + er, ca - 1 + ec) : -1;
      }
      int elab[NEP], epl[NEP];
      float4 ef[NEP];
#pragma unroll
      for (int q = 0; q < NEP; ++q) {
        const int node = enode[q];
        elab[q] = 0;
        epl[q] = 0;
        ef[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (node >= 0) {
          elab[q] = labels[node];
          epl[q] = prop[node];
          ef[q] = fwd_w[node];
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < NEP; ++q) {
        const int e = eidx[q];
        if (e >= 0) {
          const int ec = e / EH, er = e - ec * EH;
          float* wr = slabw + ec * SWC + er * 4;
          if (er < EH - 1) {
            wr[0] = ef[q].x * beta;
            wr[1] = ef[q].y * beta;
            wr[2] = ef[q].z * beta;
            wr[3] = ef[q].w * beta;
          } else {
            wr[0] = ef[q].y * beta;          // (the bottom rim row's one needed weight: see strip_cols_kernel)
          }
          slabl[e] = (unsigned char)(enode[q] >= 0 ? elab[q] : 0);
          slabp[e] = (unsigned char)(enode[q] >= 0 ? epl[q] : 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
#ifndef PHMRF_FILTER_STATS
      atomicAdd(&wk[0], 1u);
      atomicAdd(&wk[1], (unsigned int)ncell);                  // nodes this unit re-decides
      atomicAdd(&wk[2], (unsigned int)(EH * (ncols + 2)));     // the staged rectangle
#endif
    }

    // ---- extraction, lane <-> column: single-site costs, clipped pair discounts, the cells that have a proposal
    float v8[SH][8], sc[SH], cap[SH];
    unsigned long long U[SH];
    {
      const bool act = lane < ncols;
      float sg[SH];
      int pl_[SH], l_[SH];
#pragma unroll
      for (int r = 0; r < SH; ++r) {
        const int e0 = (lane + 1) * EH + (r + 1);
        const bool have = (valid[r] >> lane) & 1ull;
        l_[r] = have ? (int)slabl[act ? e0 : 0] : 0;
        pl_[r] = have ? (int)slabp[act ? e0 : 0] : 0;
        // the single-site cost of the proposal, as the proposal kernel found it (five row loads, contiguous in orientation
        // 0; no gathers into the unary planes -- those are left to the rare DP)
        sg[r] = sgain[have ? ndx[r] : 0];
      }
#pragma unroll
      for (int r = 0; r < SH; ++r) {
        const bool have = (valid[r] >> lane) & 1ull;
        float cf = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) v8[r][d] = 0.f;
        if (have && pl_[r] != l_[r]) {
          const int e0 = (lane + 1) * EH + (r + 1), w0 = (lane + 1) * SWC + (r + 1) * 4;
          const int l = l_[r], pl = pl_[r];
#pragma unroll
          for (int d = 0; d < 8; ++d) {
            constexpr int DR[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
            constexpr int DC[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
            const int dr = DR[d], dc = DC[d];
            if (r + dr < 0 || r + dr >= SH) continue;              // a rim neighbour is never in U
            const int di = ORIENT ? dc : dr, dj = ORIENT ? dr : dc;
            const bool fwd = di > 0 || (di == 0 && dj > 0);
            const int comp = (di == 0) ? 0 : (fwd ? dj + 2 : 2 - dj);
            const int en = e0 + dc * EH + dr, wn = w0 + dc * SWC + dr * 4;
            const float w = fwd ? slabw[w0 + comp] : slabw[wn + comp];
            const int lj = slabl[en], pj = slabp[en];
            const bool inside = (lane + dc >= 0) && (lane + dc < ncols);
            // what the pair saves when both ends switch, clipped at 0 (an absent or proposal-less neighbour is never in U)
            const float disc = w * ((pl != lj ? 1.f : 0.f) + (l != pj ? 1.f : 0.f) - (l != lj ? 1.f : 0.f) - (pl != pj ? 1.f : 0.f));
            const float dp = (inside && pj != lj) ? fmaxf(disc, 0.f) : 0.f;
            v8[r][d] = dp;
            cf += dp;
          }
        }
        sc[r] = sg[r];
        cap[r] = cf;
        U[r] = valid[r] & __ballot(have && pl_[r] != l_[r] && sg[r] < 1.0e29f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // (cap holds the discounts of all in-strip neighbours that have a proposal = exactly the neighbours in the first U)
#ifdef PHMRF_FILTER_STATS
    unsigned int fs_drops_ = 0u, fs_passes_ = 0u;
#endif
    const bool quiet = !(U[0] | U[1] | U[2] | U[3] | U[4]) || peel_strip(sc, v8, cap, U, peel_max PHMRF_FS_ARGS);
    if (quiet) {
      if (mslot && lane == 0) *mslot = (uint16_t)tick;
      continue;
    }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < SH; ++r) lds_pool.c.ubuf[0][r] = U[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    dp_flagged<ORIENT, true>(g, lds, 0, lane, rs0, ca, ncols, ncell, -1, tick, n, D, nbr, uT, labels, stamp, mslot, changed);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  if (work && threadIdx.x < WORK_SLOTS) {
    const unsigned int v = wk[threadIdx.x];
    if (v) atomicAdd(work + (blockIdx.x & (WORK_BANKS - 1)) * WORK_SLOTS + threadIdx.x, (unsigned long long)v);
  }
}

// uT[k][i] = -logprob[i][k]: one plane per label, so an expansion of label a reads its unary terms contiguously
__global__ __launch_bounds__(256) void unary_planes_kernel(const float* __restrict__ logprob, int64_t n, int K, int Kp,
                                                           float* __restrict__ uT) {
  extern __shared__ float tile[];
  const int TB = blockDim.x;
  for (int64_t base = (int64_t)blockIdx.x * TB; base < n; base += (int64_t)gridDim.x * TB) {
    const int64_t rem = n - base;
    const int rows = rem < TB ? (int)rem : TB;
    for (int q = threadIdx.x; q < rows * K; q += TB) {      // coalesced read of rows*K consecutive floats
      const int r = q / K, c = q - r * K;
      tile[r * Kp + c] = -logprob[base * K + q];
    }
    __syncthreads();
    if ((int)threadIdx.x < rows)
      for (int k = 0; k < K; ++k) uT[(int64_t)k * n + base + threadIdx.x] = tile[threadIdx.x * Kp + k];
    __syncthreads();
  }
}

// best alternative label per node: argmin_{k != l_i} ( -logprob[i,k] - beta * sum_{j in N(i), l_j == k} w_ij )
template <int VEC>
__global__ __launch_bounds__(256) void propose_kernel(const float* __restrict__ logprob, int64_t n, int K, int Kp, int D,
                                                      const int32_t* __restrict__ nbr, const float* __restrict__ wgt,
                                                      const uint8_t* __restrict__ labels, float beta,
                                                      uint8_t* __restrict__ prop, const uint16_t* __restrict__ stamp,
                                                      int since, unsigned long long* __restrict__ work, float* __restrict__ sgain) {
  extern __shared__ float tile[];
  const int TB = blockDim.x;
  const int KV = K / VEC;
  unsigned int done = 0u;                   // nodes whose proposal this workgroup recomputed (thread 0's count)
  for (int64_t base = (int64_t)blockIdx.x * TB; base < n; base += (int64_t)gridDim.x * TB) {
    const int64_t rem = n - base;
    const int rows = rem < TB ? (int)rem : TB;
    // since >= 0: the proposals of the previous launch are still in `prop`; a node's proposal depends on its own and
    // its neighbours' labels only, so a tile none of whose (dilated) stamps is newer than that launch is left alone
    if (since >= 0 && !__syncthreads_or((int)threadIdx.x < rows && (int)stamp[base + threadIdx.x] > since)) continue;
    done += (unsigned int)rows;
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
      // what switching the node alone to its proposal costs (>= 0 at an ICM fixed point): the fusion pass's filter reads it
      if (sgain) sgain[i] = bk != cur ? best - row[cur] : 1.0e30f;
    }
    __syncthreads();
  }
  if (work && threadIdx.x == 0 && done) atomicAdd(work + (blockIdx.x & (WORK_BANKS - 1)) * WORK_SLOTS + 6, (unsigned long long)done);
}

// The same proposals on a grid block, from the tables the strip moves use: the unary terms from the label-major planes
// (coalesced as they are, no transposition through LDS), the eight edge weights from the forward-edge table of the node
// and of its four backward neighbours (found by geometry) -- 98 B per node where the adjacency form reads 145 B.  The
// neighbours are visited in the order of the adjacency rows (NW N NE W E SW S SE), so the two kernels agree bit for bit.
__global__ __launch_bounds__(256) void propose_grid_kernel(const float* __restrict__ uT, int64_t n, int K, int Kp, int H, int W,
                                                           int diagonal, const float4* __restrict__ fwd_w,
                                                           const uint8_t* __restrict__ labels, float beta,
                                                           uint8_t* __restrict__ prop, const uint16_t* __restrict__ stamp,
                                                           int since, unsigned long long* __restrict__ work, float* __restrict__ sgain,
                                                           unsigned long long* __restrict__ seed) {
  extern __shared__ float tile[];
  const int TB = blockDim.x;
  unsigned int done = 0u;
  for (int64_t base = (int64_t)blockIdx.x * TB; base < n; base += (int64_t)gridDim.x * TB) {
    const int64_t rem = n - base;
    const int rows = rem < TB ? (int)rem : TB;
    // (every thread works in its own LDS row: the "nothing changed here" test is per WAVE -- 64 consecutive nodes -- which
    //  leaves far fewer nodes to recompute after a round of scattered changes than a test per 256-node tile)
    const bool mine = (int)threadIdx.x < rows;
    if (since >= 0 && !__any(mine && (int)stamp[base + threadIdx.x] > since)) continue;
    if (mine) ++done;
    if (mine) {
      const int64_t v = base + threadIdx.x;
      float* row = tile + threadIdx.x * Kp;
      // (loads first, uses after: a loop of load -> LDS store per label is one memory round trip per label and thread)
      int i, j;
      grid_coords(v, W, diagonal, &i, &j);
      int64_t nc[8];
      float nw[8];
      grid_gather_neighbours(v, i, j, H, W, diagonal, fwd_w, nc, nw);
      int nl[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) nl[d] = labels[nc[d]];
      const int cur = labels[v];
      int k0 = 0;
      for (; k0 + 4 <= K; k0 += 4) {
        const float a0 = uT[(int64_t)k0 * n + v], a1 = uT[(int64_t)(k0 + 1) * n + v];
        const float a2 = uT[(int64_t)(k0 + 2) * n + v], a3 = uT[(int64_t)(k0 + 3) * n + v];
        row[k0] = a0; row[k0 + 1] = a1; row[k0 + 2] = a2; row[k0 + 3] = a3;
      }
      for (; k0 < K; ++k0) row[k0] = uT[(int64_t)k0 * n + v];
#pragma unroll
      for (int d = 0; d < 8; ++d) row[nl[d]] -= beta * nw[d];      // (an absent neighbour subtracts 0 from some entry)
      float best = 3.0e38f;
      int bk = cur;
      for (int k = 0; k < K; ++k) {
        const float x = row[k];
        if (k != cur && x < best) { best = x; bk = k; }
      }
      prop[v] = (uint8_t)bk;
      if (sgain) sgain[v] = bk != cur ? best - row[cur] : 1.0e30f;
      if (seed) {
        // SEED MASK (round 6): bit a = "an improving expansion of label a on a strip could start at this node".  The strip
        // filter (peel_row) calls cell i a seed of label a when  s_i(a) < cap_i / 2,  s_i(a) = u_i(a) - u_i(l) + beta (h(l) - h(a))
        // = row[a] - row[l] here, cap_i = the discounts w (2 - [l != l_j]) of i's in-strip neighbours still in U.  cap_i is at
        // most V = sum over ALL eight neighbours, so  row[a] - row[l] < V / 2  is implied by the filter's test whatever the
        // strip, the cut and the pass; the margins (1.0002 V + 4e-5 against the filter's 1.0001 cap + 2e-5; 2e-6 of the
        // magnitudes against the <= 19 roundings of the two sums) keep that true in f32.  A (strip, label) pair none of
        // whose cells has the bit is quiet without a look at the label's unary terms (strip_scan_kernel).
        float vtot = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          const float bw = beta * nw[d];
          vtot += nl[d] == cur ? bw + bw : bw;
        }
        const float rc = row[cur];
        const float lim = rc + 0.5f * __builtin_fmaf(vtot, 1.0002f, 4e-5f) + 2e-6f * (fabsf(rc) + 2.f * vtot);
        unsigned int mlo = 0u, mhi = 0u;
        for (int k = 0; k < K && k < 32; ++k) {
          const float x = row[k];
          mlo |= (k != cur && __builtin_fmaf(fabsf(x), -2e-6f, x) < lim) ? (1u << k) : 0u;
        }
        for (int k = 32; k < K; ++k) {
          const float x = row[k];
          mhi |= (k != cur && __builtin_fmaf(fabsf(x), -2e-6f, x) < lim) ? (1u << (k - 32)) : 0u;
        }
        seed[v] = ((unsigned long long)mhi << 32) | (unsigned long long)mlo;
      }
    }
  }
  {   // nodes recomputed by this workgroup (per-thread counts -> one add per wave)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) done += __shfl_xor(done, off, 64);
    if (work && (threadIdx.x & 63) == 0 && done)
      atomicAdd(work + (blockIdx.x & (WORK_BANKS - 1)) * WORK_SLOTS + 6, (unsigned long long)done);
  }
}

inline int vec_of(int K) { return (K % 4 == 0) ? 4 : (K % 2 == 0 ? 2 : 1); }

}  // namespace

#ifdef PHMRF_DEV
constexpr int PAR_DIRTY_DEFAULT = 0;      // launch_strip_multi (development builds): four waves per strip below this many (estimated) dirty strips; 0 = never
#endif

// Development knobs (read from the environment) exist only in builds with -DPHMRF_DEV (tools/variant.sh): the product
// library has none that can change or break a labelling.
#ifdef PHMRF_DEV
static int strip_debug() {   // timing experiments only (PHMRF_STRIP_DEBUG=1: phase 1 only, 2: no backtrack/apply, +4: count strips)
  static int v = -1;
  if (v < 0) {
    const char* e = PHMRF_DEV_ENV("PHMRF_STRIP_DEBUG");
    v = e ? atoi(e) : 0;
  }
  return v;
}

static int peel_sweeps() {   // PHMRF_PEEL_SWEEPS=0 switches the filter off (timing experiments: every pair goes to the DP)
  static int v = -1;
  if (v < 0) {
    const char* e = PHMRF_DEV_ENV("PHMRF_PEEL_SWEEPS");
    v = e ? atoi(e) : PEEL_MAX;
    if (v < 0 || v > 64) v = PEEL_MAX;
  }
  return v;
}
static bool scan_enabled() {        // PHMRF_SCAN=1: strip_scan_kernel in front of every strip launch of a solve (a measured negative,
  static const bool on = PHMRF_DEV_ENV("PHMRF_SCAN") != nullptr || PHMRF_DEV_ENV("PHMRF_SEED_MASKS") != nullptr;   // DESIGN.md 3.2)
  return on;
}
static int mopup_grid() {           // PHMRF_MOPUP_GRID=n: workgroups of a strip launch in the rounds after a solve's first (0: one per slot)
  static int v = -2;
  if (v == -2) {
    const char* e = PHMRF_DEV_ENV("PHMRF_MOPUP_GRID");
    v = e ? atoi(e) : -1;
  }
  return v;
}
static bool seed_masks_enabled() {  // PHMRF_SEED_MASKS=1: the scan also uses the seed masks (a measured negative, DESIGN.md 3.2: kept
  static const bool on = PHMRF_DEV_ENV("PHMRF_SEED_MASKS") != nullptr;   // for the A/B -- the labellings must not differ)
  return on;
}
static int par_dirty_limit() {      // PHMRF_PAR_DIRTY=n: four waves per strip while the estimated dirty strips of a launch are <= n (0: never)
  static int v = -2;
  if (v == -2) {
    const char* e = PHMRF_DEV_ENV("PHMRF_PAR_DIRTY");
    v = e ? atoi(e) : PAR_DIRTY_DEFAULT;
  }
  return v;
}
#else
static constexpr int strip_debug() { return 0; }
static constexpr int peel_sweeps() { return PEEL_MAX; }
static constexpr bool scan_enabled() { return false; }
static constexpr int mopup_grid() { return -1; }
static constexpr bool seed_masks_enabled() { return false; }
#endif

int launch_propose(phmrf_block* b, float beta) {
  const int since = b->tick ? b->prop_tick : -1;
  if (!b->sgain) PHMRF_HIP(hipMalloc(reinterpret_cast<void**>(&b->sgain), (size_t)b->n * sizeof(float)));
  if (!b->seed && seed_masks_enabled()) PHMRF_HIP(hipMalloc(reinterpret_cast<void**>(&b->seed), (size_t)b->n * sizeof(unsigned long long)));
  const int K = b->K, TB = tile_threads(K), Kp = padded_k(K);
  const size_t lds = (size_t)TB * Kp * sizeof(float);
  int64_t g64 = (b->n + TB - 1) / TB;
  const int grid = (int)(g64 > 256 * 16 ? 256 * 16 : g64);
  if (b->has_grid && b->fwd_w && b->uT && b->uT_valid && b->D == 8) {
    hipLaunchKernelGGL(propose_grid_kernel, dim3(grid), dim3(TB), lds, b->stream, b->uT, b->n, K, Kp, b->H, b->W, b->diagonal,
                       b->fwd_w, b->labels, beta, b->labels_tmp, b->stamp, since, b->work_acc, b->sgain, b->seed);
    PHMRF_HIP(hipGetLastError());
    b->prop_tick = b->tick ? b->tick : -1;
    b->seed_tick = b->prop_tick;             // (the masks are those of the labelling at this tick; -1: not inside a solve)
    return PHMRF_OK;
  }
#define PHMRF_LAUNCH_PROP(VEC_)                                                                                     \
  hipLaunchKernelGGL((propose_kernel<VEC_>), dim3(grid), dim3(TB), lds, b->stream, b->logprob, b->n, K, Kp, b->D, b->nbr, \
                     b->wgt, b->labels, beta, b->labels_tmp, b->stamp, since, b->work_acc, b->sgain)
  switch (vec_of(K)) {
    case 4: PHMRF_LAUNCH_PROP(4); break;
    case 2: PHMRF_LAUNCH_PROP(2); break;
    default: PHMRF_LAUNCH_PROP(1); break;
  }
#undef PHMRF_LAUNCH_PROP
  PHMRF_HIP(hipGetLastError());
  b->prop_tick = b->tick ? b->tick : -1;
  b->seed_tick = -1;                         // (the adjacency form writes no seed masks)
  return PHMRF_OK;
}

static StripGeom make_geom(const phmrf_block* b, int orient, int shift_r, int shift_c) {
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
  static const bool plain_order = PHMRF_DEV_ENV("PHMRF_NO_XCD_MAP") != nullptr;      // development: A/B of the strips' order
  g.xcd = plain_order ? 0 : 1;
  return g;
}

// how many strip slots a launch on this block can have at most (either orientation, any cut): the size of scan_out
int64_t strip_scan_slots(const phmrf_block* b) {
  int64_t m = 0;
  for (int orient = 0; orient < 2; ++orient) {
    const int Hs = orient ? b->W : b->H, Ws = orient ? b->H : b->W;
    const int nbands = (Hs + 5 + SH) / (SH + 1), nsegs = (Ws + 63 + 63) / 64;
    const int64_t a = strip_slots(orient, nbands, nsegs, 1), c = (int64_t)nbands * nsegs;
    m = a > m ? a : m;
    m = c > m ? c : m;
  }
  return m;
}

int launch_unary_planes(phmrf_block* b) {
  if (b->uT_valid) return PHMRF_OK;
  if (!b->uT) {      // (UT_PAD floats of slack at both ends: strip_cols_kernel's row loads may run up to five columns past a row)
    PHMRF_HIP(hipMalloc(reinterpret_cast<void**>(&b->uT_raw), ((size_t)b->n * b->K + 2 * UT_PAD) * sizeof(float)));
    PHMRF_HIP(hipMemsetAsync(b->uT_raw, 0, ((size_t)b->n * b->K + 2 * UT_PAD) * sizeof(float), b->stream));
    b->uT = b->uT_raw + UT_PAD;
  }
  const int K = b->K, TB = 256, Kp = padded_k(K);
  const size_t lds = (size_t)TB * Kp * sizeof(float);
  int64_t g64 = (b->n + TB - 1) / TB;
  const int grid = (int)(g64 > 256 * 16 ? 256 * 16 : g64);
  hipLaunchKernelGGL(unary_planes_kernel, dim3(grid), dim3(TB), lds, b->stream, b->logprob, b->n, K, Kp, b->uT);
  PHMRF_HIP(hipGetLastError());
  b->uT_valid = true;
  return PHMRF_OK;
}

// alpha < 0: fusion with the proposals in b->labels_tmp (launch_propose); alpha >= 0: strip alpha-expansion of one
// label.  geom >= 0 names one of the fixed cuts whose memo of quiet runs applies (inside a solve).
int launch_strip_pass(const phmrf_block* b, float beta, int orient, int shift_r, int shift_c, int alpha, int geom) {
  const StripGeom g = make_geom(b, orient, shift_r, shift_c);
  const int nstrips = g.nbands * g.nsegs;
  if (nstrips <= 0) return PHMRF_OK;
  if (!b->fwd_w || !b->uT || !b->uT_valid) return fail(PHMRF_ERR_STATE, "strip moves need the grid tables (fwd_w, unary planes)");
  const int WPB = PHMRF_STRIP_WPB, TB = 64 * WPB;
  int grid = (nstrips + WPB - 1) / WPB;
  if (grid > (1 << 22)) grid = 1 << 22;          // one workgroup per strip (see launch_strip_multi)
  const bool use_memo = b->tick && geom >= 0 && b->memo && (int64_t)nstrips <= b->memo_strips;
  // the fusion pass of a solve (proposals in labels_tmp) runs behind the exact filter (fusion_cols_kernel); the
  // single-label passes of the API and the coarse child problems keep strip_kernel.
  if (alpha < 0) {
    const int fslots = strip_slots(orient, g.nbands, g.nsegs, g.xcd);
    int fgrid = fslots;
    if (fgrid > (1 << 22)) fgrid = 1 << 22;
    if (b->ss && b->ss->rounds > 0 && mopup_grid() > 0 && fgrid > mopup_grid()) fgrid = mopup_grid();
    uint16_t* const fmemo = use_memo ? b->memo + ((int64_t)(orient * 3 + geom) * b->memo_strips) * (b->K + 1) : nullptr;
    // inside a solve the strips' stamps and memo entries are looked at by a light kernel of their own (strip_scan_kernel)
    const bool use_scan = use_memo && b->scan_out && (int64_t)fslots <= b->scan_slots && scan_enabled();
    if (use_scan) {
#define PHMRF_LAUNCH_FSCAN(O_)                                                                                        \
  hipLaunchKernelGGL((strip_scan_kernel<O_, true>), dim3((fslots + 3) / 4), dim3(256), 0, b->stream, g, b->K, 0ull, b->stamp, fmemo, \
                     b->tick, static_cast<const unsigned long long*>(nullptr), -1, b->scan_out, b->work_acc)
      if (orient) PHMRF_LAUNCH_FSCAN(1);
      else PHMRF_LAUNCH_FSCAN(0);
#undef PHMRF_LAUNCH_FSCAN
      PHMRF_HIP(hipGetLastError());
    }
#define PHMRF_LAUNCH_FUSION(O_)                                                                                       \
  hipLaunchKernelGGL((fusion_cols_kernel<O_>), dim3(fgrid), dim3(64), 0, b->stream, g, b->n, b->K, b->D, b->nbr, b->fwd_w,  \
                     b->uT, b->labels, b->labels_tmp, b->sgain, beta, b->counters + b->counter_slot,                   \
                     b->tick ? b->stamp : nullptr, fmemo, b->tick, b->work_acc, peel_sweeps(),                        \
                     use_scan ? b->scan_out : static_cast<const unsigned long long*>(nullptr))
    if (orient) PHMRF_LAUNCH_FUSION(1);
    else PHMRF_LAUNCH_FUSION(0);
#undef PHMRF_LAUNCH_FUSION
    PHMRF_HIP(hipGetLastError());
    return PHMRF_OK;
  }
#ifdef PHMRF_DEV
  static const bool no_pin_look = PHMRF_DEV_ENV("PHMRF_NO_PIN_LOOK") != nullptr;      // development: A/B timing
  static const bool child_count = PHMRF_DEV_ENV("PHMRF_CHILD_COUNT") != nullptr;     // development: strips seen / staged / into the DP
#else
  constexpr bool no_pin_look = false, child_count = false;
#endif
  const int pin_look = b->unary_pins ? ((no_pin_look ? 0 : 32) | (child_count ? 4 : 0)) : 0;   // (coarse child problems)
#define PHMRF_LAUNCH_STRIP(O_)                                                                                        \
  hipLaunchKernelGGL((strip_kernel<O_>), dim3(grid), dim3(TB), 0, b->stream, g, b->n, b->K, b->D, b->nbr, b->fwd_w, b->uT, \
                     b->labels, alpha < 0 ? b->labels_tmp : nullptr, alpha, beta, b->counters + b->counter_slot,       \
                     ((alpha >= 0 && b->counter_slot == 8 + alpha) ? strip_debug() : (strip_debug() & 3)) | pin_look,  \
                     b->tick ? b->stamp : nullptr,                                                                     \
                     use_memo ? b->memo + ((int64_t)(orient * 3 + geom) * b->memo_strips) * (b->K + 1) : nullptr,      \
                     b->tick, b->work_acc)
  if (orient) PHMRF_LAUNCH_STRIP(1);
  else PHMRF_LAUNCH_STRIP(0);
#undef PHMRF_LAUNCH_STRIP
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

// every alpha-expansion of the labels in `label_mask` on the cut (orient, shift_r, shift_c), one wave per strip
// (strip_cols_kernel).  geom >= 0: the memo of quiet runs of that fixed cut applies (inside a solve).  The launch uses
// the ticks b->tick .. b->tick + K - 1 (one per label, so that a label's quiet run after another label's move on the
// same strip stays valid); the caller advances b->tick by K.
int launch_strip_multi(const phmrf_block* b, float beta, int orient, int shift_r, int shift_c, unsigned long long label_mask,
                       int geom) {
  const StripGeom g = make_geom(b, orient, shift_r, shift_c);
  const int nstrips = g.nbands * g.nsegs;
  if (nstrips <= 0 || !label_mask) return PHMRF_OK;
  if (!b->fwd_w || !b->uT || !b->uT_valid) return fail(PHMRF_ERR_STATE, "strip moves need the grid tables (fwd_w, unary planes)");
  const int TB = 64;
  // one workgroup per strip (orientation 1: per slot of the XCD-aware order, strip_of_slot) up to 4 M: the dispatcher hands a
  // free slot the next strip, which balances the uneven strips better than waves striding over them (measured against a cap
  // of 8 resident sets: -3 % on the rows cut)
  const int nslots = strip_slots(orient, g.nbands, g.nsegs, g.xcd);
  int grid = nslots;
  if (grid > (1 << 22)) grid = 1 << 22;
  if (b->ss && b->ss->rounds > 0 && mopup_grid() > 0 && grid > mopup_grid()) grid = mopup_grid();
  const bool use_memo = b->tick && geom >= 0 && b->memo && (int64_t)nstrips <= b->memo_strips;
  uint16_t* const mmemo = use_memo ? b->memo + ((int64_t)(orient * 3 + geom) * b->memo_strips) * (b->K + 1) : nullptr;
  // inside a solve: strip_scan_kernel first -- stamps against the memo, and the seed masks of the proposals' launch where
  // they are current (b->seed_tick: launch_propose on a grid block in this solve)
  const bool use_scan = use_memo && b->scan_out && (int64_t)nslots <= b->scan_slots && scan_enabled();
  if (use_scan) {
    const bool masks = b->seed && b->seed_tick >= 0 && seed_masks_enabled();
#define PHMRF_LAUNCH_SCAN(O_)                                                                                         \
  hipLaunchKernelGGL((strip_scan_kernel<O_, false>), dim3((nslots + 3) / 4), dim3(256), 0, b->stream, g, b->K, label_mask, b->stamp, \
                     mmemo, b->tick, masks ? b->seed : static_cast<const unsigned long long*>(nullptr), b->seed_tick,   \
                     b->scan_out, b->work_acc)
    if (orient) PHMRF_LAUNCH_SCAN(1);
    else PHMRF_LAUNCH_SCAN(0);
#undef PHMRF_LAUNCH_SCAN
    PHMRF_HIP(hipGetLastError());
  }
  // (round 6, development builds only: a measured negative) FOUR WAVES PER STRIP in a solve's late rounds,
  // PHMRF_PAR_DIRTY=n: when the previous round changed so few labels that the dirty strips (estimated from the round's change
  // count) are <= n, four waves share the staged strip and split the labels' filters (strip_cols_kernel, WV = 4).  Same
  // labelling, label for label -- and no faster: a four-wave workgroup takes a quarter of the GPU's strip slots.
  bool four_waves = false;
#ifdef PHMRF_DEV
  if (use_memo && b->ss && b->ss->rounds > 0 && par_dirty_limit() > 0) {
    const int64_t est_dirty = std::min<int64_t>((int64_t)nstrips, 2 * b->ss->last_changed);
    four_waves = est_dirty <= par_dirty_limit();
  }
#endif
#define PHMRF_LAUNCH_MULTI(O_, W_)                                                                                    \
  hipLaunchKernelGGL((strip_cols_kernel<O_, W_>), dim3(grid), dim3(TB * W_), 0, b->stream, g, b->n, b->K, b->D, b->nbr, b->fwd_w, \
                     b->uT, b->labels, beta, label_mask, b->counters + 8, b->tick ? b->stamp : nullptr, mmemo,          \
                     b->tick, b->work_acc, peel_sweeps(),                                                             \
                     use_scan ? b->scan_out : static_cast<const unsigned long long*>(nullptr))
#ifdef PHMRF_DEV
  if (four_waves) {
    if (orient) PHMRF_LAUNCH_MULTI(1, 4);
    else PHMRF_LAUNCH_MULTI(0, 4);
  } else
#endif
  {
    (void)four_waves;
    if (orient) PHMRF_LAUNCH_MULTI(1, 1);
    else PHMRF_LAUNCH_MULTI(0, 1);
  }
#undef PHMRF_LAUNCH_MULTI
  PHMRF_HIP(hipGetLastError());
  return PHMRF_OK;
}

}  // namespace phmrf
