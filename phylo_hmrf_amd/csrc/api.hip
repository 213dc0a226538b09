// C ABI of libphmrf (include/phmrf.h): host-side orchestration around the gfx950 kernels.

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <random>
#include <thread>

#include "common.h"

namespace phmrf {

static thread_local std::string g_error;

void set_error(const std::string& msg) { g_error = msg; }

int fail(int status, const std::string& msg) {
  g_error = msg;
  return status;
}

static hipEvent_t take_event(phmrf_block* b) {
  if (!b->free_events.empty()) {
    hipEvent_t e = b->free_events.back();
    b->free_events.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;   // the interval is then dropped (tic / toc), never mis-timed
  return e;
}

void tic(phmrf_block* b, int kclass) {
  if (!b->timing || !((b->timing_mask >> kclass) & 1u)) return;
  b->cur_start = take_event(b);
  if (b->cur_start && hipEventRecord(b->cur_start, b->stream) != hipSuccess) {
    b->free_events.push_back(b->cur_start);
    b->cur_start = nullptr;
  }
}

void toc(phmrf_block* b, int kclass, int n_launches) {
  const bool first = b->ss && b->ss->rounds == 0;           // (a launch of the first round of a solve: the full sweeps)
  b->launches[kclass] += n_launches;
  if (first) b->launches_first[kclass] += n_launches;
  if (!b->timing || !b->cur_start) return;
  hipEvent_t e = take_event(b);
  if (e && hipEventRecord(e, b->stream) == hipSuccess) {
    b->pending.push_back({kclass, b->cur_start, e, first});
  } else {                                                  // no event: this interval is not timed
    if (e) b->free_events.push_back(e);
    b->free_events.push_back(b->cur_start);
  }
  b->cur_start = nullptr;
}

// One time base per device for all blocks: intervals of different blocks (streams) can be laid on one time line
// (bench.py merges them: the time during which AT LEAST ONE kernel of a class was running).
static hipEvent_t g_time_base[64] = {};

static void resolve_timing(phmrf_block* b) {
  if (b->pending.empty()) return;
  (void)hipStreamSynchronize(b->stream);
  hipEvent_t base = (b->device >= 0 && b->device < 64) ? g_time_base[b->device] : nullptr;
  for (auto& p : b->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      b->ms[p.kclass] += ms;
      if (p.first) b->ms_first[p.kclass] += ms;
      float t0 = 0.f;
      if (base && hipEventElapsedTime(&t0, base, p.a) == hipSuccess) b->intervals.push_back({p.kclass, t0, t0 + ms});
    }
    b->free_events.push_back(p.a);
    b->free_events.push_back(p.b);
  }
  b->pending.clear();
}

namespace {

template <typename T>
int dev_alloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  PHMRF_HIP(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
  return PHMRF_OK;
}

template <typename T>
void dev_free(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

int upload(void* dst, const void* src, size_t bytes, hipStream_t st) {
  PHMRF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
  PHMRF_HIP(hipStreamSynchronize(st));
  return PHMRF_OK;
}

int download(void* dst, const void* src, size_t bytes, hipStream_t st) {
  PHMRF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
  PHMRF_HIP(hipStreamSynchronize(st));
  return PHMRF_OK;
}

int zero_accum(phmrf_block* b, size_t first, size_t count) {
  PHMRF_HIP(hipMemsetAsync(b->accum + first, 0, count * sizeof(double), b->stream));
  return PHMRF_OK;
}

int n_stats(const phmrf_block* b) { return b->K * (1 + b->S + b->S * b->S); }

void free_family(ChainFamily& f) {
  dev_free(f.nodes);
  for (int p = 0; p < 2; ++p)
    for (int c = 0; c < 3; ++c) {
      dev_free(f.seg_start[p][c]);
      dev_free(f.seg_len[p][c]);
      f.memo[p][c] = nullptr;          // slices of phmrf_block::chain_memo
    }
}

// (row, col) of node id under the block geometry
struct Geometry {
  int H, W, diagonal;
  std::vector<int64_t> row_start;  // diagonal: first node id of each row
  explicit Geometry(int H_, int W_, int diag) : H(H_), W(W_), diagonal(diag) {
    if (diagonal) {
      row_start.resize(H + 1);
      int64_t s = 0;
      for (int i = 0; i < H; ++i) {
        row_start[i] = s;
        s += W - i;
      }
      row_start[H] = s;
    }
  }
  int64_t count() const { return diagonal ? row_start[H] : (int64_t)H * W; }
  void coords(int64_t id, int* i, int* j) const {
    if (!diagonal) {
      *i = (int)(id / W);
      *j = (int)(id % W);
    } else {
      int r = (int)(std::upper_bound(row_start.begin(), row_start.end(), id) - row_start.begin()) - 1;
      *i = r;
      *j = r + (int)(id - row_start[r]);
    }
  }
  int64_t id(int i, int j) const { return diagonal ? row_start[i] + (j - i) : (int64_t)i * W + j; }
  bool valid(int i, int j) const { return i >= 0 && i < H && j >= 0 && j < W && (!diagonal || i <= j); }
};

// ICM colour classes and chain families of a grid block, by direct enumeration (O(n), no sorting).
int setup_grid_tables(phmrf_block* b, const Geometry& g, int num_neighbor) {
  const int64_t n = b->n;
  const int H = g.H, W = g.W;
  // ICM colours: 2x2 parity classes (oracle/mrf_moves.icm_colours)
  {
    std::vector<int64_t> cptr(5, 0);
    for (int i = 0; i < H; ++i)
      for (int j = g.diagonal ? i : 0; j < W; ++j) ++cptr[(i % 2) * 2 + (j % 2) + 1];
    for (int c = 0; c < 4; ++c) cptr[c + 1] += cptr[c];
    std::vector<int64_t> pos(cptr.begin(), cptr.end() - 1);
    std::vector<int32_t> cnodes(n);
    for (int i = 0; i < H; ++i)
      for (int j = g.diagonal ? i : 0; j < W; ++j) cnodes[pos[(i % 2) * 2 + (j % 2)]++] = (int32_t)g.id(i, j);
    PHMRF_TRY(upload(b->colour_nodes, cnodes.data(), cnodes.size() * sizeof(int32_t), b->stream));
    b->n_colours = 4;
    b->colour_ptr = cptr;
  }
  // chain families: 0 rows, 1 columns, 2 diagonals (j - i), 3 anti-diagonals (i + j)
  for (auto& f : b->families) free_family(f);
  b->families.clear();
  const int nfam = num_neighbor == 8 ? 4 : 2;
  for (int fam = 0; fam < nfam; ++fam) {
    const int ncol = fam < 2 ? 2 : 3;
    std::vector<int32_t> order;
    order.reserve(n);
    std::vector<int32_t> chain_ptr;
    std::vector<int> chain_colour;
    auto begin_chain = [&](int key) {
      chain_ptr.push_back((int32_t)order.size());
      chain_colour.push_back(((key % ncol) + ncol) % ncol);
    };
    if (fam == 0) {
      for (int i = 0; i < H; ++i) {
        const size_t before = order.size();
        for (int j = 0; j < W; ++j)
          if (g.valid(i, j)) {
            if (order.size() == before) begin_chain(i);
            order.push_back((int32_t)g.id(i, j));
          }
      }
    } else if (fam == 1) {
      for (int j = 0; j < W; ++j) {
        const size_t before = order.size();
        for (int i = 0; i < H; ++i)
          if (g.valid(i, j)) {
            if (order.size() == before) begin_chain(j);
            order.push_back((int32_t)g.id(i, j));
          }
      }
    } else if (fam == 2) {
      for (int d = -(H - 1); d <= W - 1; ++d) {
        const size_t before = order.size();
        for (int i = std::max(0, -d); i < H && i + d < W; ++i)
          if (g.valid(i, i + d)) {
            if (order.size() == before) begin_chain(d);
            order.push_back((int32_t)g.id(i, i + d));
          }
      }
    } else {
      for (int a = 0; a <= H + W - 2; ++a) {
        const size_t before = order.size();
        for (int i = std::max(0, a - (W - 1)); i < H && i <= a; ++i)
          if (g.valid(i, a - i)) {
            if (order.size() == before) begin_chain(a);
            order.push_back((int32_t)g.id(i, a - i));
          }
      }
    }
    chain_ptr.push_back((int32_t)order.size());
    PHMRF_CHECK((int64_t)order.size() == n, PHMRF_ERR_INVALID, "internal: chain enumeration does not cover the block");
    const int C = (int)chain_ptr.size() - 1;
    int max_len = 0;
    for (int c = 0; c < C; ++c) max_len = std::max(max_len, chain_ptr[c + 1] - chain_ptr[c]);
    ChainFamily f;
    f.n_chains = C;
    f.n_colours = ncol;
    f.max_len = max_len;
    PHMRF_TRY(dev_alloc(&f.nodes, (size_t)n));
    PHMRF_TRY(upload(f.nodes, order.data(), (size_t)n * sizeof(int32_t), b->stream));
    for (int phase = 0; phase < 2; ++phase)
      for (int col = 0; col < ncol; ++col) {
        std::vector<int32_t> ss, sl;
        for (int c = 0; c < C; ++c) {
          if (chain_colour[c] != col) continue;
          const int32_t p0 = chain_ptr[c], L = chain_ptr[c + 1] - chain_ptr[c];
          // separators (fixed nodes) sit at chain positions 63, 127, ... (phase 0) or 31, 95, ... (phase 1)
          int32_t start = 0;
          int32_t sep = phase ? 31 : 63;
          while (start < L) {
            const int32_t end = std::min<int32_t>(sep, L);
            if (end > start) {
              ss.push_back(p0 + start);
              sl.push_back(end - start);
            }
            start = sep + 1;
            sep += 64;
          }
        }
        f.nseg[phase][col] = (int)ss.size();
        PHMRF_TRY(dev_alloc(&f.seg_start[phase][col], ss.size()));
        PHMRF_TRY(dev_alloc(&f.seg_len[phase][col], sl.size()));
        if (!ss.empty()) {
          PHMRF_TRY(upload(f.seg_start[phase][col], ss.data(), ss.size() * sizeof(int32_t), b->stream));
          PHMRF_TRY(upload(f.seg_len[phase][col], sl.data(), sl.size() * sizeof(int32_t), b->stream));
        }
      }
    b->families.push_back(f);
  }
  b->H = H;
  b->W = W;
  b->diagonal = g.diagonal;
  b->num_neighbor = num_neighbor;
  b->has_grid = true;
  return launch_fwd_weights(b);          // grid-native edge weights of the strip kernels
}

// ---- path moves on a graph that is no grid (round 6) ------------------------------------------------------------------
// The exact 1-D move of chain_kernel -- all K labels of <= 63 consecutive nodes, everything else fixed -- needs nothing of
// a grid but two properties of its chains: inside a chain a node's only neighbours are its predecessor and its successor
// (an INDUCED path), and chains that move at the same time share no edge.  On a general graph both are had by
// construction: a path grows from a seed at either end by a neighbour that has exactly ONE neighbour among the nodes
// already taken by this colour's paths (the end it is attached to); a seed has none.  `cnt` counts those neighbours, so a
// path is induced and two paths of a colour are never adjacent.  Colours are filled one after the other, seeds and
// extensions prefer nodes no earlier colour has covered, until every node is covered or MAX_PATH_COLOURS are full (what is
// left keeps ICM and the component moves).  Four independent decompositions (fixed seeds: the tables are the same in every
// run) fill the two cut phases of two pairs of chain families: a node meets four different paths per round pair.
// Host-side, once per graph (O(colours x edges)); general graphs are not the reference's case -- its edge builders only
// emit the contact-map stencil (utility.py:1871-2053) -- but they are what pygco.cut_general_graph accepts
// (phylo_hmrf.py:496-498).
constexpr int MAX_PATH_COLOURS = 6;          // per decomposition: two ChainFamily objects of three colours

struct PathSet {
  std::vector<int32_t> nodes;                // path after path
  std::vector<int32_t> start, len, colour;   // per path
};

static void decompose_paths(int64_t n, int D, const std::vector<int32_t>& nbr, unsigned seed, PathSet* out) {
  std::mt19937 gen(seed);
  std::vector<char> covered(n, 0), inp(n, 0);
  std::vector<int32_t> cnt(n, 0), order;
  std::vector<int32_t> path, cand;
  int64_t n_cov = 0;
  for (int colour = 0; colour < MAX_PATH_COLOURS && n_cov < n; ++colour) {
    std::fill(cnt.begin(), cnt.end(), 0);
    std::fill(inp.begin(), inp.end(), 0);
    order.clear();
    for (int64_t v = 0; v < n; ++v)
      if (!covered[v]) order.push_back((int32_t)v);
    std::shuffle(order.begin(), order.end(), gen);
    auto take = [&](int32_t v) {
      inp[v] = 1;
      const int32_t* c = &nbr[(size_t)v * D];
      for (int x = 0; x < D && c[x] >= 0; ++x) ++cnt[c[x]];
    };
    for (int32_t s0 : order) {
      if (inp[s0] || cnt[s0] != 0) continue;
      path.assign(1, s0);
      take(s0);
      for (int side = 0; side < 2; ++side) {
        while ((int)path.size() < 63) {
          const int32_t e = side == 0 ? path.back() : path.front();
          const int32_t* c = &nbr[(size_t)e * D];
          cand.clear();
          bool any_new = false;
          for (int x = 0; x < D && c[x] >= 0; ++x)
            if (!inp[c[x]] && cnt[c[x]] == 1) {
              if (!covered[c[x]] && !any_new) {
                cand.clear();
                any_new = true;
              }
              if (!any_new || !covered[c[x]]) cand.push_back(c[x]);
            }
          if (cand.empty()) break;
          const int32_t v = cand[gen() % cand.size()];
          take(v);
          if (side == 0) path.push_back(v);
          else path.insert(path.begin(), v);
        }
      }
      out->start.push_back((int32_t)out->nodes.size());
      out->len.push_back((int32_t)path.size());
      out->colour.push_back(colour);
      for (int32_t v : path) {
        out->nodes.push_back(v);
        if (!covered[v]) {
          covered[v] = 1;
          ++n_cov;
        }
      }
    }
  }
}

int setup_path_families(phmrf_block* b) {
  const int64_t n = b->n;
  const int D = b->D;
  std::vector<int32_t> nbr((size_t)n * D);
  PHMRF_TRY(download(nbr.data(), b->nbr, nbr.size() * sizeof(int32_t), b->stream));
  PHMRF_HIP(hipStreamSynchronize(b->stream));
  for (auto& f : b->families) free_family(f);
  b->families.clear();
  for (int pair = 0; pair < 2; ++pair) {                 // decompositions (2 pair, 2 pair + 1) = cut phases 0 and 1 of families 2 pair, 2 pair + 1
    PathSet ps[2];
    for (int phase = 0; phase < 2; ++phase) decompose_paths(n, D, nbr, 0x9E3779B9u + 7919u * (unsigned)(2 * pair + phase), &ps[phase]);
    for (int half = 0; half < 2; ++half) {               // colours 3 half .. 3 half + 2 of both decompositions
      ChainFamily f;
      f.n_colours = 3;
      std::vector<int32_t> order;
      std::vector<int32_t> ss[2][3], sl[2][3];
      for (int phase = 0; phase < 2; ++phase)
        for (size_t q = 0; q < ps[phase].start.size(); ++q) {
          const int col = ps[phase].colour[q] - 3 * half;
          if (col < 0 || col > 2 || ps[phase].len[q] < 2) continue;      // (a path of one node is an ICM step)
          ss[phase][col].push_back((int32_t)order.size());
          sl[phase][col].push_back(ps[phase].len[q]);
          order.insert(order.end(), ps[phase].nodes.begin() + ps[phase].start[q],
                       ps[phase].nodes.begin() + ps[phase].start[q] + ps[phase].len[q]);
          f.max_len = std::max(f.max_len, (int)ps[phase].len[q]);
          ++f.n_chains;
        }
      if (order.empty()) order.push_back(0);
      PHMRF_TRY(dev_alloc(&f.nodes, order.size()));
      PHMRF_TRY(upload(f.nodes, order.data(), order.size() * sizeof(int32_t), b->stream));
      for (int phase = 0; phase < 2; ++phase)
        for (int col = 0; col < 3; ++col) {
          f.nseg[phase][col] = (int)ss[phase][col].size();
          PHMRF_TRY(dev_alloc(&f.seg_start[phase][col], ss[phase][col].size()));
          PHMRF_TRY(dev_alloc(&f.seg_len[phase][col], sl[phase][col].size()));
          if (!ss[phase][col].empty()) {
            PHMRF_TRY(upload(f.seg_start[phase][col], ss[phase][col].data(), ss[phase][col].size() * sizeof(int32_t), b->stream));
            PHMRF_TRY(upload(f.seg_len[phase][col], sl[phase][col].data(), sl[phase][col].size() * sizeof(int32_t), b->stream));
          }
        }
      PHMRF_HIP(hipStreamSynchronize(b->stream));         // (the host vectors go out of scope)
      b->families.push_back(f);
    }
  }
  return PHMRF_OK;
}

}  // namespace
}  // namespace phmrf

using namespace phmrf;

extern "C" {

// ---- library ------------------------------------------------------------------------------------
int phmrf_version(void) { return PHMRF_VERSION; }

const char* phmrf_last_error(void) { return g_error.c_str(); }

const char* phmrf_status_string(int status) {
  switch (status) {
    case PHMRF_OK: return "ok";
    case PHMRF_ERR_INVALID: return "invalid argument";
    case PHMRF_ERR_HIP: return "HIP runtime error";
    case PHMRF_ERR_NO_DEVICE: return "no GPU device";
    case PHMRF_ERR_UNSUPPORTED: return "unsupported configuration";
    case PHMRF_ERR_STATE: return "call order / missing input";
    case PHMRF_ERR_NOT_PD: return "covariance not positive definite";
  }
  return "unknown status";
}

int phmrf_device_count(int* count) {
  PHMRF_CHECK(count, PHMRF_ERR_INVALID, "count is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    return fail(PHMRF_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = c;
  return PHMRF_OK;
}

int phmrf_set_device(int device) {
  PHMRF_HIP(hipSetDevice(device));
  return PHMRF_OK;
}

// ---- block lifetime -----------------------------------------------------------------------------
static bool deterministic_env() {       // PHMRF_DETERMINISTIC=1: order-independent reductions (read at every block creation)
  const char* e = getenv("PHMRF_DETERMINISTIC");
  return e && e[0] == '1';
}

int phmrf_block_create(int64_t n, int S, int K, phmrf_block_t* out) {
  PHMRF_CHECK(out, PHMRF_ERR_INVALID, "out is NULL");
  *out = nullptr;
  PHMRF_CHECK(n > 0 && n < ((int64_t)1 << 31) - 64, PHMRF_ERR_INVALID, "n must be in [1, 2^31)");
  PHMRF_CHECK(S >= 1 && S <= 16, PHMRF_ERR_UNSUPPORTED, "S must be in [1,16]");
  PHMRF_CHECK(K >= 1 && K <= 64, PHMRF_ERR_UNSUPPORTED, "K must be in [1,64]");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(PHMRF_ERR_NO_DEVICE, "no HIP device visible");
  phmrf_block* b = new phmrf_block();
  b->n = n;
  b->S = S;
  b->K = K;
  b->deterministic = deterministic_env();
  int st = PHMRF_OK;
  auto guard = [&](int s) {
    if (s != PHMRF_OK && st == PHMRF_OK) st = s;
  };
  if (hipGetDevice(&b->device) != hipSuccess) guard(fail(PHMRF_ERR_HIP, "hipGetDevice failed"));
  if (st == PHMRF_OK && hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking) != hipSuccess)
    guard(fail(PHMRF_ERR_HIP, "hipStreamCreate failed"));
  b->stream = b->own_stream;
  if (st == PHMRF_OK) guard(dev_alloc(&b->X, (size_t)n * S));
  if (st == PHMRF_OK) guard(dev_alloc(&b->logprob, (size_t)n * K));
  if (st == PHMRF_OK) guard(dev_alloc(&b->labels, (size_t)n));
  if (st == PHMRF_OK) guard(dev_alloc(&b->labels_tmp, (size_t)n));
  if (st == PHMRF_OK) guard(dev_alloc(&b->accum, (size_t)ACCUM_DOUBLES));
  if (st == PHMRF_OK) guard(dev_alloc(&b->counters, (size_t)128));
  if (st == PHMRF_OK) guard(dev_alloc(&b->work_acc, (size_t)WORK_BANKS * WORK_SLOTS));
  if (st == PHMRF_OK && hipMemsetAsync(b->work_acc, 0, WORK_BANKS * WORK_SLOTS * sizeof(unsigned long long), b->own_stream) != hipSuccess)
    guard(fail(PHMRF_ERR_HIP, "hipMemset failed"));
  if (st == PHMRF_OK && hipHostMalloc(reinterpret_cast<void**>(&b->work_host), WORK_BANKS * WORK_SLOTS * sizeof(unsigned long long)) != hipSuccess)
    guard(fail(PHMRF_ERR_HIP, "hipHostMalloc failed"));
  if (st == PHMRF_OK) guard(dev_alloc(&b->emis_params, (size_t)K * (S + S * (S + 1) / 2 + 1)));
  if (st == PHMRF_OK && hipHostMalloc(reinterpret_cast<void**>(&b->accum_host), ACCUM_DOUBLES * sizeof(double)) != hipSuccess)
    guard(fail(PHMRF_ERR_HIP, "hipHostMalloc failed"));
  if (st == PHMRF_OK && hipHostMalloc(reinterpret_cast<void**>(&b->counters_host), 128 * sizeof(unsigned long long)) != hipSuccess)
    guard(fail(PHMRF_ERR_HIP, "hipHostMalloc failed"));
  if (st == PHMRF_OK && (hipEventCreate(&b->ev0) != hipSuccess || hipEventCreate(&b->ev1) != hipSuccess))
    guard(fail(PHMRF_ERR_HIP, "hipEventCreate failed"));
  if (st == PHMRF_OK && hipMemsetAsync(b->labels, 0, (size_t)n, b->stream) != hipSuccess)
    guard(fail(PHMRF_ERR_HIP, "hipMemset failed"));
  if (st != PHMRF_OK) {
    std::string keep = g_error;
    phmrf_block_destroy(b);
    g_error = keep;
    return st;
  }
  *out = b;
  return PHMRF_OK;
}

static void coarse_children_destroy(phmrf_block_t b);

int phmrf_block_destroy(phmrf_block_t b) {
  if (!b) return PHMRF_OK;
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  coarse_children_destroy(b);
  if (b->c2f) {
    (void)phmrf_block_destroy(b->c2f);
    b->c2f = nullptr;
  }
  dev_free(b->X);
  dev_free(b->logprob);
  dev_free(b->labels);
  dev_free(b->labels_tmp);
  dev_free(b->labels_eval);
  dev_free(b->coarse_flag);
  dev_free(b->coarse_lab);
  if (b->coarse_lab_host) (void)hipHostFree(b->coarse_lab_host);
  b->coarse_lab_host = nullptr;
  dev_free(b->sgain);
  dev_free(b->seed);
  dev_free(b->scan_out);
  dev_free(b->mf_rev);
  dev_free(b->mf_theta);
  dev_free(b->mf_cap);
  dev_free(b->mf_tcap);
  dev_free(b->mf_exc);
  dev_free(b->mf_hgt);
  dev_free(b->mf_flags);
  if (b->mf_flags_host) (void)hipHostFree(b->mf_flags_host);
  b->mf_flags_host = nullptr;
  for (int s = 0; s < 4; ++s) dev_free(b->saved[s]);
  dev_free(b->nbr);
  dev_free(b->wgt);
  dev_free(b->colour_nodes);
  for (auto& f : b->families) free_family(f);
  dev_free(b->comp);
  dev_free(b->comp_tab);
  dev_free(b->comp_tab64);
  dev_free(b->comp_best);
  dev_free(b->comp_gain);
  dev_free(b->cc_seen);
  dev_free(b->cc_stale);
  dev_free(b->stamp);
  dev_free(b->memo);
  dev_free(b->chain_memo);
  dev_free(b->fwd_w);
  if (b->uT_raw) {                          // (the planes own their allocation; a coarse child's live in its parent's arena)
    dev_free(b->uT_raw);
    b->uT = nullptr;
  }
  dev_free(b->emis_params);
  dev_free(b->posteriors);
  for (int r = 0; r < 2; ++r) {
    dev_free(b->pin_save[r]);
    dev_free(b->pin_label[r]);
  }
  dev_free(b->xfer);
  if (b->xfer_host) (void)hipHostFree(b->xfer_host);
  if (b->ss) {
    delete b->ss;
    b->ss = nullptr;
  }
  dev_free(b->accum);
  dev_free(b->counters);
  dev_free(b->work_acc);
  if (b->work_host) (void)hipHostFree(b->work_host);
  if (b->accum_host) (void)hipHostFree(b->accum_host);
  if (b->counters_host) (void)hipHostFree(b->counters_host);
  for (auto& p : b->pending) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for (auto e : b->free_events) (void)hipEventDestroy(e);
  if (b->cur_start) (void)hipEventDestroy(b->cur_start);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
  delete b;
  return PHMRF_OK;
}

int phmrf_block_set_stream(phmrf_block_t b, void* hip_stream) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_HIP(hipStreamSynchronize(b->stream));
  b->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : b->own_stream;
  for (int lv = 0; lv < 12; ++lv)
    if (b->coarse[lv]) b->coarse[lv]->stream = b->stream;
  return PHMRF_OK;
}

int phmrf_block_sync(phmrf_block_t b) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_HIP(hipStreamSynchronize(b->stream));
  return PHMRF_OK;
}

int phmrf_block_set_observations(phmrf_block_t b, const double* X) {
  PHMRF_CHECK(b && X, PHMRF_ERR_INVALID, "NULL argument");
  const size_t cnt = (size_t)b->n * b->S;
  std::vector<float> tmp(cnt);
  for (size_t i = 0; i < cnt; ++i) tmp[i] = (float)X[i];
  PHMRF_TRY(upload(b->X, tmp.data(), cnt * sizeof(float), b->stream));
  b->has_X = true;
  return PHMRF_OK;
}

int phmrf_block_set_observations_dev(phmrf_block_t b, const float* X_dev) {
  PHMRF_CHECK(b && X_dev, PHMRF_ERR_INVALID, "NULL argument");
  PHMRF_HIP(hipMemcpyAsync(b->X, X_dev, (size_t)b->n * b->S * sizeof(float), hipMemcpyDeviceToDevice, b->stream));
  b->has_X = true;
  return PHMRF_OK;
}

// ---- graph --------------------------------------------------------------------------------------
int phmrf_block_set_graph(phmrf_block_t b, int64_t E, const int64_t* edges, const double* w) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_CHECK(E >= 0 && (E == 0 || (edges && w)), PHMRF_ERR_INVALID, "edges / w is NULL");
  const int64_t n = b->n;
  std::vector<int32_t> deg(n, 0);
  for (int64_t e = 0; e < E; ++e) {
    const int64_t a = edges[2 * e], c = edges[2 * e + 1];
    PHMRF_CHECK(a >= 0 && a < n && c >= 0 && c < n, PHMRF_ERR_INVALID, "edge endpoint out of range");
    PHMRF_CHECK(a != c, PHMRF_ERR_INVALID, "self loop in edge list");
    PHMRF_CHECK(w[e] >= 0.0 && std::isfinite(w[e]), PHMRF_ERR_INVALID, "edge weights must be finite and >= 0");
    ++deg[a];
    ++deg[c];
  }
  int maxdeg = 0;
  for (int64_t i = 0; i < n; ++i) maxdeg = std::max(maxdeg, deg[i]);
  PHMRF_CHECK(maxdeg <= 64, PHMRF_ERR_UNSUPPORTED, "node degree > 64 is not supported");
  const int D = std::max(4, (maxdeg + 3) / 4 * 4);
  std::vector<int32_t> nbr((size_t)n * D, -1);
  std::vector<float> wgt((size_t)n * D, 0.f);
  std::fill(deg.begin(), deg.end(), 0);
  for (int64_t e = 0; e < E; ++e) {
    const int64_t a = edges[2 * e], c = edges[2 * e + 1];
    nbr[(size_t)a * D + deg[a]] = (int32_t)c;
    wgt[(size_t)a * D + deg[a]++] = (float)w[e];
    nbr[(size_t)c * D + deg[c]] = (int32_t)a;
    wgt[(size_t)c * D + deg[c]++] = (float)w[e];
  }
  // neighbours ascending (deterministic summation order == oracle/mrf_moves.Graph); rows are short
  for (int64_t i = 0; i < n; ++i) {
    int32_t* c = &nbr[(size_t)i * D];
    float* ww = &wgt[(size_t)i * D];
    for (int x = 1; x < deg[i]; ++x) {
      const int32_t cv = c[x];
      const float wv = ww[x];
      int y = x - 1;
      while (y >= 0 && c[y] > cv) {
        c[y + 1] = c[y];
        ww[y + 1] = ww[y];
        --y;
      }
      c[y + 1] = cv;
      ww[y + 1] = wv;
    }
    for (int x = 1; x < deg[i]; ++x)
      PHMRF_CHECK(c[x] != c[x - 1], PHMRF_ERR_INVALID, "duplicate edge in edge list");
  }
  // greedy colouring in index order (replaced by the parity colouring when a grid is declared)
  std::vector<int32_t> colour(n, -1);
  int ncol = 0;
  {
    std::vector<int> mark(66, -1);
    for (int64_t i = 0; i < n; ++i) {
      const int32_t* c = &nbr[(size_t)i * D];
      for (int x = 0; x < deg[i]; ++x)
        if (colour[c[x]] >= 0) mark[colour[c[x]]] = (int)(i & 0x7fffffff);
      int col = 0;
      while (mark[col] == (int)(i & 0x7fffffff)) ++col;
      colour[i] = col;
      ncol = std::max(ncol, col + 1);
    }
  }
  std::vector<int64_t> cptr(ncol + 1, 0);
  for (int64_t i = 0; i < n; ++i) ++cptr[colour[i] + 1];
  for (int c = 0; c < ncol; ++c) cptr[c + 1] += cptr[c];
  std::vector<int32_t> cnodes(n);
  {
    std::vector<int64_t> pos(cptr.begin(), cptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) cnodes[pos[colour[i]]++] = (int32_t)i;
  }
  dev_free(b->nbr);
  dev_free(b->wgt);
  dev_free(b->colour_nodes);
  PHMRF_TRY(dev_alloc(&b->nbr, (size_t)n * D));
  PHMRF_TRY(dev_alloc(&b->wgt, (size_t)n * D));
  PHMRF_TRY(dev_alloc(&b->colour_nodes, (size_t)n));
  PHMRF_TRY(upload(b->nbr, nbr.data(), nbr.size() * sizeof(int32_t), b->stream));
  PHMRF_TRY(upload(b->wgt, wgt.data(), wgt.size() * sizeof(float), b->stream));
  PHMRF_TRY(upload(b->colour_nodes, cnodes.data(), cnodes.size() * sizeof(int32_t), b->stream));
  b->D = D;
  b->E = E;
  b->n_colours = ncol;
  b->colour_ptr = cptr;
  b->has_graph = true;
  b->has_grid = false;
  for (auto& f : b->families) free_family(f);
  b->families.clear();
  dev_free(b->mf_rev);                       // (the arcs' reverse slots and the flow's arrays belong to the old graph)
  dev_free(b->mf_theta);
  dev_free(b->mf_cap);
  dev_free(b->mf_tcap);
  dev_free(b->mf_exc);
  dev_free(b->mf_hgt);
  return PHMRF_OK;
}

int phmrf_block_set_grid(phmrf_block_t b, int H, int W, int diagonal, int num_neighbor) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_CHECK(b->has_graph, PHMRF_ERR_STATE, "set_graph must precede set_grid");
  PHMRF_CHECK(H >= 1 && W >= 1, PHMRF_ERR_INVALID, "H, W must be >= 1");
  PHMRF_CHECK(num_neighbor == 8 || num_neighbor == 4, PHMRF_ERR_INVALID, "num_neighbor must be 8 or 4");
  PHMRF_CHECK(!diagonal || H <= W, PHMRF_ERR_INVALID, "a diagonal block is the first H <= W rows of a W x W upper triangle");
  Geometry g(H, W, diagonal);
  PHMRF_CHECK(g.count() == b->n, PHMRF_ERR_INVALID, "H, W, diagonal do not match the node count");
  const int64_t n = b->n;
  const int D = b->D;
  std::vector<int32_t> nbr((size_t)n * D);
  std::vector<float> wgt((size_t)n * D);
  PHMRF_TRY(download(nbr.data(), b->nbr, nbr.size() * sizeof(int32_t), b->stream));
  PHMRF_TRY(download(wgt.data(), b->wgt, wgt.size() * sizeof(float), b->stream));
  std::vector<int32_t> ci(n), cj(n);
  for (int64_t v = 0; v < n; ++v) {
    int i, j;
    g.coords(v, &i, &j);
    ci[v] = i;
    cj[v] = j;
  }
  int64_t present = 0, expected = 0;       // adjacency entries on file / entries of the complete stencil
  for (int64_t v = 0; v < n; ++v) {
    for (int x = 0; x < D; ++x) {
      const int32_t u = nbr[(size_t)v * D + x];
      if (u < 0) continue;
      const int di = std::abs(ci[u] - ci[v]), dj = std::abs(cj[u] - cj[v]);
      PHMRF_CHECK(di <= 1 && dj <= 1 && (di + dj) > 0, PHMRF_ERR_INVALID, "edge list joins nodes that are not grid neighbours");
      PHMRF_CHECK(num_neighbor == 8 || (di + dj) == 1, PHMRF_ERR_INVALID, "diagonal edge in a 4-neighbour block");
      ++present;
    }
    for (int di = -1; di <= 1; ++di)
      for (int dj = -1; dj <= 1; ++dj) {
        if ((di == 0 && dj == 0) || (num_neighbor == 4 && di != 0 && dj != 0)) continue;
        const int ni = ci[v] + di, nj = cj[v] + dj;
        if (ni < 0 || ni >= H || nj < 0 || nj >= W || (diagonal && ni > nj)) continue;
        ++expected;
      }
  }
  PHMRF_TRY(setup_grid_tables(b, g, num_neighbor));
  // kernels that find their neighbours by geometry and COUNT them (the posterior kernel: isolated nodes, estimate_type
  // != 3) need every stencil edge to be on file; weights alone are indifferent to a missing edge (it weighs 0)
  b->grid_complete = present == expected;
  return PHMRF_OK;
}

int phmrf_block_build_grid_graph(phmrf_block_t b, int H, int W, int diagonal, int num_neighbor, double beta1) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_CHECK(b->has_X, PHMRF_ERR_STATE, "observations must be set before the graph is built from them");
  PHMRF_CHECK(H >= 1 && W >= 1, PHMRF_ERR_INVALID, "H, W must be >= 1");
  PHMRF_CHECK(num_neighbor == 8 || num_neighbor == 4, PHMRF_ERR_INVALID, "num_neighbor must be 8 or 4");
  PHMRF_CHECK(!diagonal || H <= W, PHMRF_ERR_INVALID, "a diagonal block is the first H <= W rows of a W x W upper triangle");
  PHMRF_CHECK(beta1 >= 0.0 && std::isfinite(beta1), PHMRF_ERR_INVALID, "beta1 must be finite and >= 0");
  Geometry g(H, W, diagonal);
  PHMRF_CHECK(g.count() == b->n, PHMRF_ERR_INVALID, "H, W, diagonal do not match the node count");
  dev_free(b->nbr);
  dev_free(b->wgt);
  PHMRF_TRY(dev_alloc(&b->nbr, (size_t)b->n * 8));
  PHMRF_TRY(dev_alloc(&b->wgt, (size_t)b->n * 8));
  b->D = 8;
  PHMRF_TRY(launch_grid_graph(b, H, W, diagonal, num_neighbor, beta1));
  if (!b->colour_nodes) PHMRF_TRY(dev_alloc(&b->colour_nodes, (size_t)b->n));
  b->has_graph = true;
  b->E = 0;
  PHMRF_TRY(setup_grid_tables(b, g, num_neighbor));
  b->grid_complete = true;                 // built from the stencil: every edge is there
  return PHMRF_OK;
}

int phmrf_block_get_adjacency(phmrf_block_t b, int* D, int32_t* nbr_out, float* wgt_out) {
  PHMRF_CHECK(b && D, PHMRF_ERR_INVALID, "NULL argument");
  PHMRF_CHECK(b->has_graph, PHMRF_ERR_STATE, "graph not set");
  *D = b->D;
  if (nbr_out) PHMRF_TRY(download(nbr_out, b->nbr, (size_t)b->n * b->D * sizeof(int32_t), b->stream));
  if (wgt_out) PHMRF_TRY(download(wgt_out, b->wgt, (size_t)b->n * b->D * sizeof(float), b->stream));
  return PHMRF_OK;
}

// ---- labels -------------------------------------------------------------------------------------
int phmrf_block_set_labels(phmrf_block_t b, const int32_t* labels) {
  PHMRF_CHECK(b && labels, PHMRF_ERR_INVALID, "NULL argument");
  std::vector<uint8_t> tmp(b->n);
  for (int64_t i = 0; i < b->n; ++i) {
    PHMRF_CHECK(labels[i] >= 0 && labels[i] < b->K, PHMRF_ERR_INVALID, "label out of range [0,K)");
    tmp[i] = (uint8_t)labels[i];
  }
  PHMRF_TRY(upload(b->labels, tmp.data(), (size_t)b->n, b->stream));
  b->has_labels = true;
  b->labels_are_slot = 0;
  return PHMRF_OK;
}

static int fetch_labels(phmrf_block_t b, const uint8_t* src, int32_t* labels) {
  std::vector<uint8_t> tmp(b->n);
  PHMRF_TRY(download(tmp.data(), src, (size_t)b->n, b->stream));
  for (int64_t i = 0; i < b->n; ++i) labels[i] = tmp[i];
  return PHMRF_OK;
}

int phmrf_block_get_labels(phmrf_block_t b, int32_t* labels) {
  PHMRF_CHECK(b && labels, PHMRF_ERR_INVALID, "NULL argument");
  return fetch_labels(b, b->labels, labels);
}

int phmrf_block_save_labels(phmrf_block_t b, int slot) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_CHECK(slot >= 0 && slot < 4, PHMRF_ERR_INVALID, "slot must be in [0,4)");
  if (!b->saved[slot]) PHMRF_TRY(dev_alloc(&b->saved[slot], (size_t)b->n));
  PHMRF_HIP(hipMemcpyAsync(b->saved[slot], b->labels, (size_t)b->n, hipMemcpyDeviceToDevice, b->stream));
  b->labels_are_slot |= 1 << slot;
  return PHMRF_OK;
}

int phmrf_block_restore_labels(phmrf_block_t b, int slot) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_CHECK(slot >= 0 && slot < 4 && b->saved[slot], PHMRF_ERR_STATE, "label slot is empty");
  PHMRF_HIP(hipMemcpyAsync(b->labels, b->saved[slot], (size_t)b->n, hipMemcpyDeviceToDevice, b->stream));
  b->has_labels = true;
  b->labels_are_slot = 1 << slot;
  return PHMRF_OK;
}

int phmrf_block_get_saved_labels(phmrf_block_t b, int slot, int32_t* labels) {
  PHMRF_CHECK(b && labels, PHMRF_ERR_INVALID, "NULL argument");
  PHMRF_CHECK(slot >= 0 && slot < 4 && b->saved[slot], PHMRF_ERR_STATE, "label slot is empty");
  return fetch_labels(b, b->saved[slot], labels);
}

// ---- b1 emission --------------------------------------------------------------------------------
int phmrf_emission_pack_size(int S, int K, int64_t* n_floats) {
  PHMRF_CHECK(n_floats && S >= 1 && S <= 16 && K >= 1, PHMRF_ERR_INVALID, "bad S/K");
  *n_floats = (int64_t)K * (S + S * (S + 1) / 2 + 1);
  return PHMRF_OK;
}

int phmrf_emission_pack(int S, int K, const double* means, const double* covars, float* packed) {
  PHMRF_CHECK(means && covars && packed, PHMRF_ERR_INVALID, "NULL argument");
  PHMRF_CHECK(S >= 1 && S <= 16 && K >= 1, PHMRF_ERR_INVALID, "bad S/K");
  const int PS = S + S * (S + 1) / 2 + 1;
  std::vector<double> L(S * S), Li(S * S);
  for (int k = 0; k < K; ++k) {
    const double* cv = covars + (size_t)k * S * S;
    bool ok = false;
    for (int attempt = 0; attempt < 2 && !ok; ++attempt) {  // sklearn 0.18: retry once with +1e-7*I
      const double jitter = attempt ? 1e-7 : 0.0;
      ok = true;
      std::fill(L.begin(), L.end(), 0.0);
      for (int i = 0; i < S && ok; ++i)
        for (int j = 0; j <= i; ++j) {
          double s = cv[i * S + j] + (i == j ? jitter : 0.0);
          for (int t = 0; t < j; ++t) s -= L[i * S + t] * L[j * S + t];
          if (i == j) {
            if (!(s > 0.0) || !std::isfinite(s)) { ok = false; break; }
            L[i * S + i] = std::sqrt(s);
          } else {
            L[i * S + j] = s / L[j * S + j];
          }
        }
    }
    if (!ok) return fail(PHMRF_ERR_NOT_PD, "'covars' must be symmetric, positive-definite (state " + std::to_string(k) + ")");
    // Li = L^-1 (lower) by forward substitution; logdet = 2 sum log diag(L)
    std::fill(Li.begin(), Li.end(), 0.0);
    double logdet = 0.0;
    for (int i = 0; i < S; ++i) logdet += 2.0 * std::log(L[i * S + i]);
    for (int c = 0; c < S; ++c)
      for (int i = c; i < S; ++i) {
        double s = (i == c) ? 1.0 : 0.0;
        for (int t = c; t < i; ++t) s -= L[i * S + t] * Li[t * S + c];
        Li[i * S + c] = s / L[i * S + i];
      }
    float* p = packed + (size_t)k * PS;
    for (int s = 0; s < S; ++s) p[s] = (float)means[(size_t)k * S + s];
    int idx = S;
    for (int i = 0; i < S; ++i)
      for (int c = 0; c <= i; ++c) p[idx++] = (float)Li[i * S + c];
    p[idx] = (float)(-0.5 * (S * std::log(2.0 * M_PI) + logdet));
  }
  return PHMRF_OK;
}

int phmrf_emission_dev(const float* X_dev, int64_t n, int S, int K, const float* packed_dev, float* logprob_dev,
                       void* hip_stream) {
  PHMRF_CHECK(X_dev && packed_dev && logprob_dev, PHMRF_ERR_INVALID, "NULL argument");
  PHMRF_CHECK(n > 0 && K >= 1 && K <= 64, PHMRF_ERR_INVALID, "bad n/K");
  return launch_emission(X_dev, n, S, K, packed_dev, logprob_dev, nullptr, reinterpret_cast<hipStream_t>(hip_stream));
}

int phmrf_emission(phmrf_block_t b, const double* means, const double* covars) {
  PHMRF_CHECK(b && means && covars, PHMRF_ERR_INVALID, "NULL argument");
  PHMRF_CHECK(b->has_X, PHMRF_ERR_STATE, "observations not set");
  const int PS = b->S + b->S * (b->S + 1) / 2 + 1;
  std::vector<float> packed((size_t)b->K * PS);
  PHMRF_TRY(phmrf_emission_pack(b->S, b->K, means, covars, packed.data()));
  PHMRF_TRY(upload(b->emis_params, packed.data(), packed.size() * sizeof(float), b->stream));
  tic(b, KC_EMISSION);
  // (a block that has run strip moves before owns its unary planes: the kernel writes them along with logprob)
  PHMRF_TRY(launch_emission(b->X, b->n, b->S, b->K, b->emis_params, b->logprob, b->uT, b->stream));
  toc(b, KC_EMISSION, 1);
  b->has_logprob = true;
  b->uT_valid = b->uT != nullptr;
  b->pin_saved = false;                     // (every row is real again: whatever was pinned is not any more)
  b->pin_rows[0] = b->pin_rows[1] = 0;
  return PHMRF_OK;
}

int phmrf_block_get_logprob(phmrf_block_t b, double* logprob) {
  PHMRF_CHECK(b && logprob, PHMRF_ERR_INVALID, "NULL argument");
  PHMRF_CHECK(b->has_logprob, PHMRF_ERR_STATE, "logprob not computed");
  const size_t cnt = (size_t)b->n * b->K;
  std::vector<float> tmp(cnt);
  PHMRF_TRY(download(tmp.data(), b->logprob, cnt * sizeof(float), b->stream));
  for (size_t i = 0; i < cnt; ++i) logprob[i] = tmp[i];
  return PHMRF_OK;
}

int phmrf_block_set_logprob(phmrf_block_t b, const double* logprob) {
  PHMRF_CHECK(b && logprob, PHMRF_ERR_INVALID, "NULL argument");
  const size_t cnt = (size_t)b->n * b->K;
  std::vector<float> tmp(cnt);
  for (size_t i = 0; i < cnt; ++i) tmp[i] = (float)logprob[i];
  PHMRF_TRY(upload(b->logprob, tmp.data(), cnt * sizeof(float), b->stream));
  b->has_logprob = true;
  b->uT_valid = false;
  b->pin_saved = false;
  b->pin_rows[0] = b->pin_rows[1] = 0;
  return PHMRF_OK;
}

// ---- b2 MRF -------------------------------------------------------------------------------------
static int read_counter(phmrf_block_t b, int64_t* v) {
  PHMRF_HIP(hipMemcpyAsync(b->counters_host, b->counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
  PHMRF_HIP(hipStreamSynchronize(b->stream));
  *v = (int64_t)b->counters_host[0];
  return PHMRF_OK;
}

static int zero_counter(phmrf_block_t b) {
  PHMRF_HIP(hipMemsetAsync(b->counters, 0, 128 * sizeof(unsigned long long), b->stream));
  b->counter_slot = 0;
  return PHMRF_OK;
}

// strip-kernel work counters: device banks -> host totals (the stream must be synchronised by the caller afterwards)
static int work_fetch_async(phmrf_block_t b) {
  PHMRF_HIP(hipMemcpyAsync(b->work_host, b->work_acc, WORK_BANKS * WORK_SLOTS * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
  PHMRF_HIP(hipMemsetAsync(b->work_acc, 0, WORK_BANKS * WORK_SLOTS * sizeof(unsigned long long), b->stream));
  return PHMRF_OK;
}
static void work_fold(phmrf_block_t b, bool first_round = false) {
  static const int SLOT_OF[WORK_SLOTS] = {0, 1, 2, 3, 5, 6, 7, 8, 9};   // work[4] = launches (host-counted)
  for (int k = 0; k < WORK_BANKS; ++k)
    for (int q = 0; q < WORK_SLOTS; ++q) {
      const int64_t v = (int64_t)b->work_host[k * WORK_SLOTS + q];
      b->work[SLOT_OF[q]] += v;
      if (first_round) b->work_first[SLOT_OF[q]] += v;
    }
}

static int check_solvable(phmrf_block_t b) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_CHECK(b->has_graph, PHMRF_ERR_STATE, "graph not set");
  PHMRF_CHECK(b->has_logprob, PHMRF_ERR_STATE, "logprob not set (run phmrf_emission or phmrf_block_set_logprob)");
  return PHMRF_OK;
}

static int icm_sweep_nocount(phmrf_block_t b, float beta) {
  if (b->tick) ++b->tick;
  tic(b, KC_ICM);
  for (int c = 0; c < b->n_colours; ++c) PHMRF_TRY(launch_icm_colour(b, beta, c));
  toc(b, KC_ICM, b->n_colours);
  return PHMRF_OK;
}

static int chain_sweep_nocount(phmrf_block_t b, float beta, int family, int phase, bool timed = true) {
  const ChainFamily& f = b->families[family];
  if (b->tick) ++b->tick;
  if (timed) tic(b, KC_CHAIN);
  for (int c = 0; c < f.n_colours; ++c) PHMRF_TRY(launch_chain_colour(b, beta, family, c, phase));
  if (timed) toc(b, KC_CHAIN, f.n_colours);
  return PHMRF_OK;
}

static int energy_now(phmrf_block_t b, double beta, double* eu, double* ep) {
  PHMRF_TRY(zero_accum(b, 4, 2));
  tic(b, KC_ENERGY);
  PHMRF_TRY(launch_energy(b, (float)beta));
  toc(b, KC_ENERGY, 1);
  PHMRF_HIP(hipMemcpyAsync(b->accum_host + 4, b->accum + 4, 2 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  PHMRF_HIP(hipStreamSynchronize(b->stream));
  if (b->deterministic) {             // 2^-20 fixed-point integers in the two slots (kernels.hip energy_flush)
    long long q[2];
    std::memcpy(q, b->accum_host + 4, sizeof(q));
    *eu = (double)q[0] / 1048576.0;
    *ep = beta * ((double)q[1] / 1048576.0);
    return PHMRF_OK;
  }
  *eu = b->accum_host[4];
  *ep = beta * b->accum_host[5];
  return PHMRF_OK;
}

// The energy after a round of a solve, in two halves: energy_round_launch queues the evaluation behind the round's moves,
// energy_round_collect reads it once the stream has been synchronised.  The first evaluation of a solve is the full pass;
// later ones on a large grid block add the change since the previous evaluation, taken from the nodes the round's moves have
// stamped (energy_delta_grid_kernel) -- a mop-up round touches a few per cent of the block.  Each evaluation leaves a
// snapshot of the labels and its tick behind for the next.  (PHMRF_ENERGY_FULL=1: always the full pass;
// PHMRF_ENERGY_CHECK=1: both, compared.)  The carried values are (unary, pair without beta).
static const int ROUND_ENERGY_SLOT = 120;      // counters[120], [121]: the round's (unary, pair) energy sums (energy_round_launch)
static int energy_round_launch(phmrf_block_t b, bool* incremental_out, bool* snapshot_out) {
  static const bool always_full = PHMRF_DEV_ENV("PHMRF_ENERGY_FULL") != nullptr;
  const bool grid = b->has_grid && b->fwd_w && b->uT && b->uT_valid && b->stamp && b->tick > 0 && b->n >= (1 << 18);
  const bool snapshot = grid && !always_full;
  const bool incremental = snapshot && energy_delta_available(b);
  // (round 6) the round's two energy sums live in the counter bank (slots 120, 121, as doubles -- or 2^-20 fixed-point integers
  // in deterministic mode): the memset that opens the round has zeroed them and the ONE read-back of the bank that closes it
  // carries them -- two fills / copies per round fewer than with the accumulator area
  tic(b, KC_ENERGY);
  double* const at = reinterpret_cast<double*>(b->counters + ROUND_ENERGY_SLOT);
  if (incremental) PHMRF_TRY(launch_energy_delta(b, at));
  else PHMRF_TRY(launch_energy(b, 0.f, at));
  toc(b, KC_ENERGY, 1);
  if (b->tile_top || b->tile_bot)        // (the pin-violation count of a row tile: accum slot 6, tile.hip)
    PHMRF_HIP(hipMemcpyAsync(b->accum_host + 6, b->accum + 6, sizeof(double), hipMemcpyDeviceToHost, b->stream));
  if (snapshot) {       // the snapshot for the next evaluation: every launch from here on carries a later tick
    if (!b->labels_eval) PHMRF_TRY(dev_alloc(&b->labels_eval, (size_t)b->n));
    PHMRF_HIP(hipMemcpyAsync(b->labels_eval, b->labels, (size_t)b->n, hipMemcpyDeviceToDevice, b->stream));
    b->eval_tick = b->tick;
    ++b->tick;
  }
  *incremental_out = incremental;
  *snapshot_out = snapshot;
  return PHMRF_OK;
}

// (the stream has been synchronised)  -> *eu, *ep_raw: unary and pair sum (without beta) after the round
static int energy_round_collect(phmrf_block_t b, double beta, bool incremental, double* eu_carry, double* ep_carry) {
  static const bool check = PHMRF_DEV_ENV("PHMRF_ENERGY_CHECK") != nullptr;
  double du, dp;
  if (b->deterministic) {             // 2^-20 fixed-point integers in the two slots (kernels.hip energy_flush)
    long long q[2];
    std::memcpy(q, b->counters_host + ROUND_ENERGY_SLOT, sizeof(q));
    du = (double)q[0] / 1048576.0;
    dp = (double)q[1] / 1048576.0;
  } else {
    double q[2];
    std::memcpy(q, b->counters_host + ROUND_ENERGY_SLOT, sizeof(q));
    du = q[0];
    dp = q[1];
  }
  if (incremental) {
    *eu_carry += du;
    *ep_carry += dp;
  } else {
    *eu_carry = du;
    *ep_carry = dp;
  }
  if (check && incremental) {
    double fu, fp;
    PHMRF_TRY(energy_now(b, beta, &fu, &fp));
    const double eu = *eu_carry, ep = beta * *ep_carry;
    const double tol = 1e-9 * std::fabs(fu + fp) + 1e-3;
    if (!b->tile_top && !b->tile_bot && (std::fabs(fu - eu) > tol || std::fabs(fp - ep) > tol))
      fprintf(stderr, "[phmrf energy check] incremental %.6f + %.6f, full %.6f + %.6f (diff %.3e, %.3e)\n", eu, ep, fu, fp,
              eu - fu, ep - fp);
  }
  return PHMRF_OK;
}

int phmrf_mrf_energy(phmrf_block_t b, double beta, double* e_total, double* e_unary, double* e_pair) {
  PHMRF_TRY(check_solvable(b));
  double eu, ep;
  PHMRF_TRY(energy_now(b, beta, &eu, &ep));
  if (e_total) *e_total = eu + ep;
  if (e_unary) *e_unary = eu;
  if (e_pair) *e_pair = ep;
  return PHMRF_OK;
}

// The warm start of an E-step.  The reference starts every labelling from labels_local, the labels of the EM iteration with
// the lowest cost so far (phylo_hmrf.py:479, base.py:416-420) -- which can be many iterations old: the start is then far
// from any minimum of the energy under the current parameters and the solve pays for a cold start.  The block's CURRENT
// labels are the previous E-step's result; both candidates are scored under the logprob that is resident now and the lower
// energy wins.  choose = 0: only score (the tiles of a split block decide on their sums).
int phmrf_block_warm_start(phmrf_block_t b, double beta, int slot, int choose, double* e_current, double* e_saved, int* took_saved) {
  PHMRF_TRY(check_solvable(b));
  PHMRF_CHECK(slot >= 0 && slot < 4 && b->saved[slot], PHMRF_ERR_STATE, "label slot is empty");
  if (took_saved) *took_saved = 1;
  const bool report = e_current || e_saved || took_saved;
  if (!b->has_labels || ((b->labels_are_slot >> slot) & 1)) {        // nothing to compare with / the snapshot IS the current labelling
    if (choose && !b->has_labels) PHMRF_TRY(phmrf_block_restore_labels(b, slot));
    if (e_current) *e_current = std::numeric_limits<double>::infinity();
    if (e_saved) *e_saved = 0.0;
    return PHMRF_OK;
  }
  double* const extra = b->accum + (ACCUM_DOUBLES - 4);     // (behind every statistic the accum area can hold)
  if (choose && !report && energy_diff_available(b)) {
    // nobody asked for the two energies: what decides is the SIGN of their difference, which comes from the nodes where the
    // two labellings differ (energy_diff_grid_kernel) -- one light pass instead of two full ones; the choice kernel then
    // compares (difference, 0)
    PHMRF_TRY(zero_accum(b, 4, 2));
    PHMRF_TRY(zero_accum(b, ACCUM_DOUBLES - 4, 2));
    tic(b, KC_ENERGY);
    PHMRF_TRY(launch_energy_diff(b, b->saved[slot]));
    PHMRF_TRY(launch_choose_labels(b, b->saved[slot], b->accum + 4, extra, beta));
    toc(b, KC_ENERGY, 2);
    b->labels_are_slot = 0;
    return PHMRF_OK;
  }
  // the two evaluations and the choice are queued on the block's stream; the host waits only if it wants the numbers
  PHMRF_TRY(zero_accum(b, 4, 2));
  PHMRF_TRY(zero_accum(b, ACCUM_DOUBLES - 4, 2));
  tic(b, KC_ENERGY);
  PHMRF_TRY(launch_energy(b, (float)beta));
  std::swap(b->labels, b->saved[slot]);                     // (score the snapshot in place)
  const int st = launch_energy(b, (float)beta, extra);
  std::swap(b->labels, b->saved[slot]);
  PHMRF_TRY(st);
  if (choose) PHMRF_TRY(launch_choose_labels(b, b->saved[slot], b->accum + 4, extra, beta));
  toc(b, KC_ENERGY, choose ? 3 : 2);
  if (choose) b->labels_are_slot = 0;
  if (!report) return PHMRF_OK;
  PHMRF_HIP(hipMemcpyAsync(b->accum_host + 4, b->accum + 4, 2 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  PHMRF_HIP(hipMemcpyAsync(b->accum_host + 6, extra, 2 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  PHMRF_HIP(hipStreamSynchronize(b->stream));
  double ec[2], es[2];
  for (int which = 0; which < 2; ++which) {
    double* dst = which ? es : ec;
    const double* src = b->accum_host + (which ? 6 : 4);
    if (b->deterministic) {
      long long q[2];
      std::memcpy(q, src, sizeof(q));
      dst[0] = (double)q[0] / 1048576.0;
      dst[1] = (double)q[1] / 1048576.0;
    } else {
      dst[0] = src[0];
      dst[1] = src[1];
    }
  }
  const double cur = ec[0] + beta * ec[1], sav = es[0] + beta * es[1];
  if (e_current) *e_current = cur;
  if (e_saved) *e_saved = sav;
  if (took_saved) *took_saved = !(cur < sav) ? 1 : 0;
  return PHMRF_OK;
}

int phmrf_mrf_icm_sweep(phmrf_block_t b, double beta, int64_t* changed) {
  PHMRF_TRY(check_solvable(b));
  b->labels_are_slot = 0;
  PHMRF_TRY(zero_counter(b));
  PHMRF_TRY(icm_sweep_nocount(b, (float)beta));
  int64_t ch = 0;
  PHMRF_TRY(read_counter(b, &ch));
  if (changed) *changed = ch;
  return PHMRF_OK;
}

int phmrf_mrf_chain_sweep(phmrf_block_t b, double beta, int family, int64_t* changed) {
  PHMRF_TRY(check_solvable(b));
  b->labels_are_slot = 0;
  PHMRF_CHECK(b->has_grid, PHMRF_ERR_STATE, "chain moves need phmrf_block_set_grid");
  PHMRF_CHECK(family >= 0 && family < (int)b->families.size(), PHMRF_ERR_INVALID, "no such chain family");
  PHMRF_TRY(zero_counter(b));
  PHMRF_TRY(chain_sweep_nocount(b, (float)beta, family, 0));
  PHMRF_TRY(chain_sweep_nocount(b, (float)beta, family, 1));
  int64_t ch = 0;
  PHMRF_TRY(read_counter(b, &ch));
  if (changed) *changed = ch;
  return PHMRF_OK;
}

int phmrf_block_prepare_components(phmrf_block_t b) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_CHECK(!b->ss, PHMRF_ERR_STATE, "a solve is in progress");
  if (!b->has_labels || !b->has_graph) return PHMRF_OK;        // (nothing to prepare yet: the pass will do its own)
  return launch_component_prepare(b);
}

int phmrf_mrf_component_pass(phmrf_block_t b, double beta, int64_t* changed) {
  PHMRF_TRY(check_solvable(b));
  b->labels_are_slot = 0;
  PHMRF_TRY(zero_counter(b));
  tic(b, KC_COMPONENT);
  PHMRF_TRY(launch_component_pass(b, (float)beta));
  toc(b, KC_COMPONENT, 1);
  int64_t ch = 0;
  PHMRF_TRY(read_counter(b, &ch));
  if (changed) *changed = ch;
  return PHMRF_OK;
}

int phmrf_mrf_graph_expansion(phmrf_block_t b, double beta, int alpha, int64_t* changed) {
  PHMRF_TRY(check_solvable(b));
  PHMRF_CHECK(b->has_labels, PHMRF_ERR_STATE, "labels not set");
  PHMRF_CHECK(alpha >= 0 && alpha < b->K, PHMRF_ERR_INVALID, "alpha must be in [0, K)");
  PHMRF_CHECK(b->n >= 2, PHMRF_ERR_INVALID, "the graph has fewer than two nodes");
  b->labels_are_slot = 0;
  PHMRF_TRY(zero_counter(b));
  PHMRF_TRY(launch_graph_expansion(b, (float)beta, alpha));
  int64_t ch = 0;
  PHMRF_TRY(read_counter(b, &ch));
  if (changed) *changed = ch;
  return PHMRF_OK;
}

static int strip_pass_nocount(phmrf_block_t b, float beta, int orient, int shift_r, int shift_c, int alpha, int geom = -1,
                              bool timed = true) {
  if (!b->uT_valid) {                       // (before the proposals: on a grid they are formed from the planes)
    tic(b, KC_PROPOSE);
    PHMRF_TRY(launch_unary_planes(b));
    toc(b, KC_PROPOSE, 1);
  }
  if (alpha < 0) {
    tic(b, KC_PROPOSE);
    PHMRF_TRY(launch_propose(b, beta));
    toc(b, KC_PROPOSE, 1);
  }
  if (timed) tic(b, KC_FUSION);
  if (b->tick) ++b->tick;
  PHMRF_TRY(launch_strip_pass(b, beta, orient, shift_r, shift_c, alpha, geom));
  b->work[4] += 1;
  if (timed) toc(b, KC_FUSION, 1);
  return PHMRF_OK;
}

int phmrf_mrf_strip_pass(phmrf_block_t b, double beta, int orient, int shift_r, int shift_c, int alpha, int64_t* changed) {
  PHMRF_TRY(check_solvable(b));
  b->labels_are_slot = 0;
  PHMRF_CHECK(b->has_grid, PHMRF_ERR_STATE, "strip moves need phmrf_block_set_grid");
  PHMRF_CHECK(b->num_neighbor == 8 || b->num_neighbor == 4, PHMRF_ERR_STATE, "bad grid");
  PHMRF_CHECK(orient == 0 || orient == 1, PHMRF_ERR_INVALID, "orient must be 0 or 1");
  PHMRF_CHECK(shift_r >= 0 && shift_r <= 5 && shift_c >= 0 && shift_c <= 63, PHMRF_ERR_INVALID, "shift out of range");
  PHMRF_CHECK(alpha < b->K, PHMRF_ERR_INVALID, "alpha must be < K");
  PHMRF_TRY(zero_counter(b));
  PHMRF_TRY(strip_pass_nocount(b, (float)beta, orient, shift_r, shift_c, alpha));
  PHMRF_TRY(work_fetch_async(b));
  int64_t ch = 0;
  PHMRF_TRY(read_counter(b, &ch));
  work_fold(b);
  if (changed) *changed = ch;
  return PHMRF_OK;
}

int phmrf_mrf_strip_multi_pass(phmrf_block_t b, double beta, int orient, int shift_r, int shift_c, uint64_t label_mask,
                               int64_t* changed) {
  PHMRF_TRY(check_solvable(b));
  b->labels_are_slot = 0;
  PHMRF_CHECK(b->has_grid, PHMRF_ERR_STATE, "strip moves need phmrf_block_set_grid");
  PHMRF_CHECK(b->num_neighbor == 8 || b->num_neighbor == 4, PHMRF_ERR_STATE, "bad grid");
  PHMRF_CHECK(orient == 0 || orient == 1, PHMRF_ERR_INVALID, "orient must be 0 or 1");
  PHMRF_CHECK(shift_r >= 0 && shift_r <= 5 && shift_c >= 0 && shift_c <= 63, PHMRF_ERR_INVALID, "shift out of range");
  PHMRF_CHECK(b->K >= 64 || (label_mask >> b->K) == 0, PHMRF_ERR_INVALID, "label_mask names a label >= K");
  PHMRF_HIP(hipMemsetAsync(b->counters, 0, 128 * sizeof(unsigned long long), b->stream));
  if (!b->uT_valid) PHMRF_TRY(launch_unary_planes(b));
  tic(b, KC_STRIP);
  PHMRF_TRY(launch_strip_multi(b, (float)beta, orient, shift_r, shift_c, label_mask, -1));
  b->work[4] += 1;
  toc(b, KC_STRIP, 1);
  PHMRF_TRY(work_fetch_async(b));
  PHMRF_HIP(hipMemcpyAsync(b->counters_host, b->counters, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
  PHMRF_HIP(hipStreamSynchronize(b->stream));
  work_fold(b);
  int64_t ch = 0;
  for (int a = 0; a < b->K; ++a) ch += (int64_t)b->counters_host[8 + a];
  if (changed) *changed = ch;
  return PHMRF_OK;
}

// ---- coarse alpha-expansions (coarse.hip) ----------------------------------------------------------------------
static const int N_COARSE = 3;
static const int COARSE_SCALE[N_COARSE] = {2, 4, 8};
static const int64_t COARSE_ON_DIV = 8;       // "moved at large": a solve changed >= 1/8 of the labels so far
static const int64_t COARSE_ROUND_DIV = 4;    // "moving at large": a round changed >= 1/4 of the labels (a cold start)

// The twelve child problems of a block (three scales x four labels of a batch) are lean: a child holds what coarsen_kernel
// writes and strip_kernel / coarse_apply_kernel read -- labels, two unary planes, the forward weights, a counter bank -- and
// all twelve come out of ONE device allocation made at the first coarse sweep.  (They used to be full blocks: a stream, three
// pinned host buffers, two events and eleven device buffers each -- some two hundred runtime calls per block, 50 - 60 ms of
// the first cold solve of EVERY block, whatever its size.)
static int coarse_children_create(phmrf_block_t b) {
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t off_lab[12], off_uT[12], off_fwd[12], total = up((size_t)12 * 128 * sizeof(unsigned long long));   // the counter banks first
  int64_t nm[12];
  for (int ls = 0; ls < 12; ++ls) {
    const int s = COARSE_SCALE[ls / 4];
    nm[ls] = coarse_nodes(b, s, s - 1);
    off_lab[ls] = total;
    total += up((size_t)nm[ls]);
    off_uT[ls] = total;
    total += up((size_t)2 * nm[ls] * sizeof(float));
    off_fwd[ls] = total;
    total += up((size_t)nm[ls] * sizeof(float4));
  }
  PHMRF_HIP(hipMalloc(reinterpret_cast<void**>(&b->coarse_arena), total));
  PHMRF_HIP(hipMemsetAsync(b->coarse_arena, 0, (size_t)12 * 128 * sizeof(unsigned long long), b->stream));
  for (int ls = 0; ls < 12; ++ls) {
    phmrf_block* c = new phmrf_block();
    c->n = nm[ls];
    c->S = 1;
    c->K = 2;
    c->device = b->device;
    c->deterministic = b->deterministic;
    c->counters = reinterpret_cast<unsigned long long*>(b->coarse_arena) + (size_t)ls * 128;
    c->labels = reinterpret_cast<uint8_t*>(b->coarse_arena + off_lab[ls]);
    c->uT = reinterpret_cast<float*>(b->coarse_arena + off_uT[ls]);
    c->fwd_w = reinterpret_cast<float4*>(b->coarse_arena + off_fwd[ls]);
    c->uT_valid = true;
    c->has_grid = true;
    c->has_graph = true;
    c->has_logprob = true;
    c->D = 0;
    c->unary_pins = true;
    c->stream = b->stream;
    b->coarse[ls] = c;
  }
  return PHMRF_OK;
}

static void coarse_children_destroy(phmrf_block_t b) {
  for (int ls = 0; ls < 12; ++ls) {
    delete b->coarse[ls];             // (a child owns nothing: its buffers are slices of the arena)
    b->coarse[ls] = nullptr;
  }
  if (b->coarse_arena) (void)hipFree(b->coarse_arena);
  b->coarse_arena = nullptr;
}

static int coarse_child(phmrf_block_t b, int level_slot, phmrf_block** out) {       // level_slot = level * 4 + slot in a batch
  if (!b->coarse_arena) PHMRF_TRY(coarse_children_create(b));
  b->coarse[level_slot]->stream = b->stream;
  b->coarse[level_slot]->num_neighbor = b->num_neighbor;
  *out = b->coarse[level_slot];
  return PHMRF_OK;
}

// every label alpha of `labels_mask` once at one scale / offset: coarsen -> one strip pass per orientation on the
// super-cell grid -> apply.  4 launches per label.
static int coarse_sweep_nocount(phmrf_block_t b, float beta, int level, int off, int shift_r, int shift_c, int alpha_lo,
                                int alpha_hi, unsigned long long label_mask = ~0ull) {
  const int s = COARSE_SCALE[level];
  if (!b->uT_valid) PHMRF_TRY(launch_unary_planes(b));
  // Labels in batches of four: one pass over the block builds the four child problems (the label-independent two thirds of
  // coarsen_kernel's reads once instead of four times), then label by label the child's two strip passes and the apply
  // pass, in order.  A label whose predecessors in the batch moved something gets its problem rebuilt first -- decided on
  // the device (coarse_apply_kernel raises b->coarse_flag, the one-label rebuild returns at once while it is down): the
  // sequence of labellings is the one-label-at-a-time sequence, the host never waits.  PHMRF_COARSE_BATCH=1: one by one.
  static const int batch_env = PHMRF_DEV_ENV("PHMRF_COARSE_BATCH") ? atoi(PHMRF_DEV_ENV("PHMRF_COARSE_BATCH")) : 4;
  const int batch = (batch_env == 1 || batch_env == 2) ? batch_env : 4;
  static const bool no_gate = PHMRF_DEV_ENV("PHMRF_COARSE_NO_GATE") != nullptr;      // development: A/B timing
  static const bool no_stamp_gate = PHMRF_DEV_ENV("PHMRF_COARSE_NO_STAMP_GATE") != nullptr;
  phmrf_block* ch[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int q = 0; q < batch; ++q) PHMRF_TRY(coarse_child(b, level * 4 + q, &ch[q]));
  if (!b->coarse_flag) PHMRF_TRY(dev_alloc(&b->coarse_flag, (size_t)1));
  tic(b, KC_COARSE);
  int n_launch = 0;
  // (the labels of [alpha_lo, alpha_hi) that `label_mask` lists, ascending, in batches)
  std::vector<int> todo;
  for (int a = alpha_lo; a < alpha_hi; ++a)
    if ((label_mask >> a) & 1ull) todo.push_back(a);
  for (size_t t0 = 0; t0 < todo.size(); t0 += (size_t)batch) {
    const int nl = (int)std::min<size_t>((size_t)batch, todo.size() - t0);
    int alphas[4] = {0, 0, 0, 0};
    for (int q = 0; q < nl; ++q) alphas[q] = todo[t0 + q];
    // The apply pass (a thread per fine node) returns at once when the child's two passes switched no super-cell: the
    // child counts its switches in its own counter slot 0, and the apply kernel reads it on the device.  (The batch's pass
    // zeroes those counters and the block's moved-flag itself.)
    for (int q = 0; q < nl; ++q) ch[q]->counter_slot = 0;
    PHMRF_TRY(launch_coarsen_batch(b, ch, alphas, nl == 3 ? 3 : nl, s, off, beta, nullptr, -1, true));
    // (inside a solve the change stamps say WHERE the labels before a label in the batch have moved: its rebuild touches
    //  those wavefronts only -- every apply pass below stamps with a tick later than this one)
    const int since = (b->tick && !no_stamp_gate) ? b->tick : -1;
    ++n_launch;
    for (int q = 0; q < nl; ++q) {
      phmrf_block* c = ch[q];
      if (q > 0) {        // rebuilt only if a label before it in the batch has moved (b->coarse_flag, read on the device)
        phmrf_block* one[1] = {c};
        PHMRF_TRY(launch_coarsen_batch(b, one, &alphas[q], 1, s, off, beta, b->coarse_flag, since, false));
        ++n_launch;
      }
      // (measured: the filtered multi-label kernel is 15-20 % slower than the plain one on these one-label problems)
      PHMRF_TRY(launch_strip_pass(c, beta, 0, shift_r % 6, shift_c % 64, 1, -1));
      PHMRF_TRY(launch_strip_pass(c, beta, 1, (shift_r + 3) % 6, (shift_c + 31) % 64, 1, -1));
      if (b->tick) ++b->tick;
      PHMRF_TRY(launch_coarse_apply(b, c, s, off, alphas[q], no_gate ? nullptr : c->counters, b->coarse_flag,
                                    b->coarse_lab ? b->coarse_lab + level * 64 + alphas[q] : nullptr));
      n_launch += 3;
    }
  }
  toc(b, KC_COARSE, n_launch);
  return PHMRF_OK;
}

int phmrf_mrf_coarse_pass(phmrf_block_t b, double beta, int scale, int offset, int alpha, int shift_r, int shift_c,
                          int64_t* changed) {
  PHMRF_TRY(check_solvable(b));
  b->labels_are_slot = 0;
  PHMRF_CHECK(b->has_grid, PHMRF_ERR_STATE, "coarse moves need phmrf_block_set_grid");
  PHMRF_CHECK(scale == 2 || scale == 4 || scale == 8, PHMRF_ERR_INVALID, "scale must be 2, 4 or 8");
  PHMRF_CHECK(offset >= 0 && offset < scale, PHMRF_ERR_INVALID, "offset must be in [0, scale)");
  PHMRF_CHECK(alpha >= 0 && alpha < b->K, PHMRF_ERR_INVALID, "alpha must be in [0, K)");
  PHMRF_CHECK(shift_r >= 0 && shift_r <= 5 && shift_c >= 0 && shift_c <= 63, PHMRF_ERR_INVALID, "shift out of range");
  PHMRF_TRY(zero_counter(b));
  PHMRF_TRY(coarse_sweep_nocount(b, (float)beta, scale == 2 ? 0 : (scale == 4 ? 1 : 2), offset, shift_r, shift_c, alpha, alpha + 1));
  int64_t ch = 0;
  PHMRF_TRY(read_counter(b, &ch));
  if (changed) *changed = ch;
  return PHMRF_OK;
}

int phmrf_block_coarse_problem(phmrf_block_t b, double beta, int scale, int offset, int alpha, int64_t* nc, float* D_out,
                               float* lam_out) {
  PHMRF_TRY(check_solvable(b));
  PHMRF_CHECK(nc, PHMRF_ERR_INVALID, "nc is NULL");
  PHMRF_CHECK(b->has_grid, PHMRF_ERR_STATE, "coarse moves need phmrf_block_set_grid");
  PHMRF_CHECK(scale == 2 || scale == 4 || scale == 8, PHMRF_ERR_INVALID, "scale must be 2, 4 or 8");
  PHMRF_CHECK(offset >= 0 && offset < scale, PHMRF_ERR_INVALID, "offset must be in [0, scale)");
  PHMRF_CHECK(alpha >= 0 && alpha < b->K, PHMRF_ERR_INVALID, "alpha must be in [0, K)");
  *nc = coarse_nodes(b, scale, offset);
  if (!D_out && !lam_out) return PHMRF_OK;
  phmrf_block* c = nullptr;
  PHMRF_TRY(coarse_child(b, 4 * (scale == 2 ? 0 : (scale == 4 ? 1 : 2)), &c));
  if (!b->uT_valid) PHMRF_TRY(launch_unary_planes(b));
  PHMRF_TRY(launch_coarsen(b, c, scale, offset, alpha, (float)beta));
  if (D_out) PHMRF_TRY(download(D_out, c->uT + *nc, (size_t)*nc * sizeof(float), b->stream));
  if (lam_out) PHMRF_TRY(download(lam_out, c->fwd_w, (size_t)*nc * sizeof(float4), b->stream));
  return PHMRF_OK;
}

// ---- the label solver as a resumable state machine ------------------------------------------------------------------
// phmrf_mrf_solve = begin; { round_launch; round_collect; round_decide } until decided; end.  The pieces are entry points
// of their own so that the tiles of ONE block that live on different GPUs can run their rounds in lockstep: between
// collect and decide the host adds up the tiles' change counters and energies (one small all-gather per round), every tile
// takes the same decision from the sums, and the boundary label rows are exchanged (phylo_hmrf_amd/tiles.py).
}  // extern "C"

namespace {
const int N_COARSE_LV = 3;
const int GEOM_R[3] = {0, 2, 4}, GEOM_C[3] = {0, 21, 42};

void solve_scope_exit(phmrf_block* b) {
  phmrf_solve_state* s = b->ss;
  if (s) b->geom_phase = (s->geom + 1) % 3;
  b->tick = 0;
  b->eval_tick = -1;
  b->counter_slot = 0;
  b->prop_tick = -1;
  b->seed_tick = -1;
  delete s;
  b->ss = nullptr;
}

// first node of grid row i (i == H: n)
int64_t tile_row_first(const phmrf_block* b, int i) {
  return b->diagonal ? (int64_t)i * b->W - ((int64_t)i * (i - 1)) / 2 : (int64_t)i * b->W;
}

// queue the copy of a tile's first and last owned rows into the pinned staging buffer (top row first); whoever
// synchronises the stream next finds them there (phmrf_block_tile_get_boundary then copies without touching the device)
int tile_queue_boundary(phmrf_block* b) {
  int64_t off = 0;
  if (b->tile_top) {
    const int64_t tf = tile_row_first(b, 1), tc = tile_row_first(b, 2) - tf;
    PHMRF_HIP(hipMemcpyAsync(b->xfer_host, b->labels + tf, (size_t)tc, hipMemcpyDeviceToHost, b->stream));
    off = tc;
  }
  if (b->tile_bot) {
    const int64_t bf = tile_row_first(b, b->H - 2), bc = tile_row_first(b, b->H - 1) - bf;
    PHMRF_HIP(hipMemcpyAsync(b->xfer_host + off, b->labels + bf, (size_t)bc, hipMemcpyDeviceToHost, b->stream));
  }
  b->boundary_queued = true;
  return PHMRF_OK;
}

// The coarse-to-fine start of a cold solve (c2f.hip): the labelling problem of the block's 4 x 4 super-cells as a block of
// its own (created at the first cold solve; its graph is built once, the graph of the block being constant over a fit),
// solved from ITS cold start by this same solver -- which recurses while the coarse block is large --, and copied down.
// *started: the block's labels are the prolongated coarse labels (otherwise the caller takes argmax_k logprob).
static int c2f_start(phmrf_block_t b, double beta, const phmrf_solve_opts& o, bool* started) {
  *started = false;
  if (o.coarse_start <= 0 || !b->has_grid || !b->fwd_w || b->num_neighbor != 8 || !b->grid_complete) return PHMRF_OK;
  if (b->tile_top || b->tile_bot || !o.use_strips) return PHMRF_OK;
  if (b->n < 1024 || b->H < 2 * C2F_SCALE || b->W < 2 * C2F_SCALE) return PHMRF_OK;
  const int s = C2F_SCALE;
  if (!b->c2f) {
    const int Hc = (b->H + s - 1) / s, Wc = (b->W + s - 1) / s;
    Geometry gc(Hc, Wc, b->diagonal);
    phmrf_block_t c = nullptr;
    PHMRF_TRY(phmrf_block_create(gc.count(), 1, b->K, &c));
    // the child becomes b->c2f only once it is complete: a failure on the way destroys it, and the next cold solve starts over
    // (a half-built child -- no graph, no grid tables -- must never be solved on)
    auto build = [&]() -> int {
      c->stream = b->stream;                    // (the child's kernels read the parent's logprob and write its labels)
      PHMRF_TRY(dev_alloc(&c->nbr, (size_t)c->n * 8));
      PHMRF_TRY(dev_alloc(&c->wgt, (size_t)c->n * 8));
      c->D = 8;
      PHMRF_TRY(launch_c2f_graph(b, c, Hc, Wc, s));
      if (!c->colour_nodes) PHMRF_TRY(dev_alloc(&c->colour_nodes, (size_t)c->n));
      c->has_graph = true;
      PHMRF_TRY(setup_grid_tables(c, gc, 8));
      c->grid_complete = true;
      return PHMRF_OK;
    };
    const int st = build();
    if (st != PHMRF_OK) {
      (void)hipStreamSynchronize(b->stream);    // (its kernels were queued on the parent's stream)
      c->stream = c->own_stream;
      (void)phmrf_block_destroy(c);
      return st;
    }
    b->c2f = c;
    b->c2f_Hc = Hc;
    b->c2f_Wc = Wc;
  }
  phmrf_block* c = b->c2f;
  c->stream = b->stream;
  PHMRF_TRY(launch_c2f_logprob(b, c, b->c2f_Wc, s));
  c->has_logprob = true;
  c->uT_valid = false;
  phmrf_solve_opts oc = o;
  oc.init_mode = 1;
  PHMRF_TRY(phmrf_mrf_solve(c, beta, &oc, nullptr));
  PHMRF_TRY(launch_c2f_prolong(b, c, b->c2f_Wc, s));
  *started = true;
  return PHMRF_OK;
}

int solve_begin(phmrf_block_t b, double beta, const phmrf_solve_opts* opts, bool want_init_energy) {
  PHMRF_TRY(check_solvable(b));
  if (b->ss) solve_scope_exit(b);                 // (an abandoned solve)
  phmrf_solve_state* s = new phmrf_solve_state();
  b->ss = s;
  phmrf_solve_opts& o = s->o;
  std::memset(&o, 0, sizeof(o));
  o.max_rounds = 64;
  o.use_chains = 1;
  o.use_components = 1;
  o.use_strips = 1;
  o.use_expansion = 1;
  o.use_coarse = 1;
  if (opts) {
    o = *opts;
    if (o.max_rounds <= 0) o.max_rounds = 64;
  }
  s->beta = beta;
  s->bf = (float)beta;
  s->sched_n = b->sched_n > 0 ? b->sched_n : b->n;
  b->labels_are_slot = 0;                        // (whatever the solve does to the labels)
  struct Abort {                                  // a failure below leaves no half-begun solve behind
    phmrf_block* blk;
    bool armed = true;
    ~Abort() { if (armed) solve_scope_exit(blk); }
  } abort_guard{b};
  if (o.init_mode == 1) {
    bool started = false;
    PHMRF_TRY(c2f_start(b, beta, o, &started));        // coarse-to-fine (c2f.hip), where the block qualifies
    if (!started) PHMRF_TRY(launch_argmax_labels(b));
    b->has_labels = true;
  }
  if (want_init_energy) {
    PHMRF_TRY(energy_now(b, beta, &s->eu0, &s->ep0));
    s->have_init_energy = true;
  }
  // a graph without grid geometry gets its path families at its first solve (setup_path_families: once per graph)
  if (o.use_chains && !b->has_grid && b->families.empty() && b->nbr && b->n >= 2) PHMRF_TRY(setup_path_families(b));
  s->chains = o.use_chains && (b->has_grid || !b->families.empty());
  s->strips = o.use_strips && b->has_grid;
  s->tol = o.min_changed > 0 ? o.min_changed : 0;
  const int K = b->K;
  // move types and their change counters (slot in b->counters): chain families 72..75, ICM 76, components 77,
  // strip fusion 78/79, strip expansion of label a: 8 + a.
  s->expansions = s->strips && o.use_expansion;
  // a graph without grid geometry: every label's alpha-expansion over the WHOLE graph by a minimum cut (maxflow.hip), the
  // move gco's expansion() makes; the same change counters 8 + a
  s->graph_expansions = !b->has_grid && o.use_expansion && b->nbr != nullptr && b->n >= 2;
  s->n_fam = s->chains ? (int)b->families.size() : 0;
  for (int f = 0; f < s->n_fam; ++f) s->slots.push_back(72 + f);
  s->slots.push_back(76);
  if (o.use_components) s->slots.push_back(77);
  if (s->strips) {
    s->slots.push_back(78);
    s->slots.push_back(79);
  }
  if (s->expansions || s->graph_expansions)
    for (int a = 0; a < K; ++a) s->slots.push_back(8 + a);
  // coarse alpha-expansions: slots 80 (2 x 2 super-cells), 81 (4 x 4), 82 (8 x 8)
  s->coarse = s->strips && o.use_coarse && b->H >= 4 && b->W >= 4;
  if (s->coarse)
    for (int lv = 0; lv < N_COARSE_LV; ++lv) s->slots.push_back(80 + lv);
  s->active.assign(128, 0);
  for (int sl : s->slots) s->active[sl] = 1;
  s->last_count.assign(128, -1);
  s->ran.assign(128, 0);
  // change stamps + per-strip memo of quiet expansions (exact skip of strips whose inputs did not change)
  if (!b->stamp) PHMRF_TRY(dev_alloc(&b->stamp, (size_t)b->n));
  PHMRF_HIP(hipMemsetAsync(b->stamp, 0, (size_t)b->n * sizeof(uint16_t), b->stream));
  b->tick = 1;
  b->eval_tick = -1;                   // (no energy evaluation in this solve yet: the first one is a full pass)
  b->prop_tick = -1;
  b->seed_tick = -1;
  if (s->chains) {                       // segment memos of all families: one buffer, one memset
    size_t total = 0;
    for (auto& f : b->families)
      for (int p = 0; p < 2; ++p)
        for (int c = 0; c < f.n_colours; ++c) total += (size_t)f.nseg[p][c];
    if (!b->chain_memo || b->chain_memo_count != total) {
      dev_free(b->chain_memo);
      PHMRF_TRY(dev_alloc(&b->chain_memo, total));
      b->chain_memo_count = total;
    }
    size_t off = 0;
    for (auto& f : b->families)
      for (int p = 0; p < 2; ++p)
        for (int c = 0; c < f.n_colours; ++c) {
          f.memo[p][c] = f.nseg[p][c] > 0 ? b->chain_memo + off : nullptr;
          off += (size_t)f.nseg[p][c];
        }
    PHMRF_HIP(hipMemsetAsync(b->chain_memo, 0, (total ? total : 1) * sizeof(uint16_t), b->stream));
  }
  if (s->strips) {                         // (the fusion passes keep a memo, too: slot K)
    int64_t max_strips = 0;
    for (int orient = 0; orient < 2; ++orient) {
      const int Hs = orient ? b->W : b->H, Ws = orient ? b->H : b->W;
      const int64_t ns = (int64_t)((Hs + 5 + 5) / 6) * ((Ws + 63 + 63) / 64);
      max_strips = std::max(max_strips, ns);
    }
    if (!b->memo || b->memo_strips < max_strips) {
      dev_free(b->memo);
      PHMRF_TRY(dev_alloc(&b->memo, (size_t)6 * max_strips * (K + 1)));
      b->memo_strips = max_strips;
    }
    PHMRF_HIP(hipMemsetAsync(b->memo, 0, (size_t)6 * b->memo_strips * (K + 1) * sizeof(uint16_t), b->stream));
    // the two words per strip slot that strip_scan_kernel leaves for the strip launch behind it
    const int64_t slots = strip_scan_slots(b);
    if (!b->scan_out || b->scan_slots < slots) {
      dev_free(b->scan_out);
      PHMRF_TRY(dev_alloc(&b->scan_out, (size_t)2 * slots));
      b->scan_slots = slots;
    }
  }
  // One round runs every ACTIVE move type: chain families, ICM, component moves, strip fusion per orientation, strip
  // alpha-expansion per label.  A type stays active while it still changes labels.  When a round is quiet (at most
  // `min_changed` labels changed, or the energy did not go down) a VERIFICATION round with every type active (and the
  // chain segments cut at their other set of separators) decides: quiet again -> done.  (The energy test also ends the
  // alternation between two labellings of exactly equal energy that different move types prefer; gco stops on the same
  // criterion, GCoptimization.cpp:1298.)
  // the energy before the first round (only needed when the caller asked for it: the first round of a solve that
  // changes labels always improves, and the tolerance refers to the energy after the round).  The tiles of a split block
  // start without it: their schedule runs on sums over the tiles, which begin with the first round's.
  s->e_prev = (s->have_init_energy && !b->tile_top && !b->tile_bot) ? s->eu0 + s->ep0 : std::numeric_limits<double>::infinity();
  // The strip alpha-expansions run on one of three fixed cuts (so that the per-strip memo of quiet runs applies).  The
  // cut ADVANCES after a round that moved the labelling at large (>= 1/64 of the labels: the memo is worth little
  // then), when a verification round begins, and from one solve to the next (b->geom_phase); it STAYS while the solve
  // is mopping up, so those rounds only revisit the strips whose inputs changed -- a warm start pays for one full
  // sweep per solve instead of one per round.
  s->geom = b->geom_phase % 3;
  if (b->tile_top || b->tile_bot) PHMRF_TRY(zero_accum(b, 6, 1));       // the pin-violation counter (tile.hip)
  abort_guard.armed = false;
  return PHMRF_OK;
}

// queue one round: every active move type, then the energy evaluation and the read-back of the change counters
int solve_round_launch(phmrf_block_t b) {
  phmrf_solve_state* s = b->ss;
  PHMRF_CHECK(s, PHMRF_ERR_STATE, "no solve in progress (phmrf_mrf_solve_begin)");
  PHMRF_CHECK(!s->launched, PHMRF_ERR_STATE, "the previous round has not been decided");
  if (s->status != 0) return PHMRF_OK;
  const phmrf_solve_opts& o = s->o;
  const float bf = s->bf;
  const int K = b->K;
  const int r = s->rounds;
  const bool verifying = s->verifying;
  std::vector<char>& active = s->active;
  std::vector<char>& ran = s->ran;
  PHMRF_HIP(hipMemsetAsync(b->counters, 0, 128 * sizeof(unsigned long long), b->stream));
  std::fill(ran.begin(), ran.end(), 0);
  {
    // Chain moves (exact 1-D Viterbi over all K labels): with the strip expansions in place they run in verification
    // rounds only.  Measured (round 2, live-gco parity cases and the whole-genome bench): in ordinary rounds they make
    // 60 % of a warm start's label changes but the strips find the same energy without them -- the 2,001,000-node
    // K=10 cold start even ends 2e-4 LOWER and in 12 rounds instead of 32 (1-D moves leave row / column streaks that
    // the 2-D moves then have to undo) -- and they were 14 % of the device time.  Without strip expansions (general
    // graphs have no chains at all; `use_expansion = 0`) rows and columns run in every round as before.
    const int n_ord_fams = s->expansions ? 0 : (b->has_grid ? 2 : s->n_fam);     // (path families of a general graph: all of them)
    int n_chain = 0;
    for (int f = 0; f < s->n_fam; ++f)
      if (active[72 + f] && (f < n_ord_fams || verifying)) n_chain += b->families[f].n_colours;
    if (n_chain > 0) {
      tic(b, KC_CHAIN);
      for (int f = 0; f < s->n_fam; ++f)
        if (active[72 + f] && (f < n_ord_fams || verifying)) {
          b->counter_slot = 72 + f;
          ran[72 + f] = 1;
          // cut phase 0 in ordinary rounds (so the segment memo applies from the second round on); the other set of
          // separators is used by the verification rounds
          // (the path families of a general graph hold two independent decompositions as their two phases: they take
          //  turns round by round, so the round that verifies a quiet one always looks along the other set of paths)
          PHMRF_TRY(chain_sweep_nocount(b, bf, f, b->has_grid ? (verifying ? 1 : 0) : (s->rounds & 1), false));
        }
      toc(b, KC_CHAIN, n_chain);
    }
  }
  // single-site ICM: every strip cell and every chain node is already optimal given the rest, so ICM only earns its
  // launches on the fixed separator cells; it runs in verification rounds and on graphs without grid moves
  if (active[76] && (verifying || !(s->chains || s->strips) || !b->has_grid)) {
    b->counter_slot = 76;
    ran[76] = 1;
    PHMRF_TRY(icm_sweep_nocount(b, bf));
  }
  // component moves: a full pass over the block (seven kernels) whatever the number of labels that changed.  On grid
  // blocks they run in a solve's first round, after a round that moved the labelling at large, and in verification
  // rounds; the mop-up rounds in between (a few hundred changed labels, of which the pass would take a dozen) skip
  // them.  On general graphs, where they are one of two move types, they run in every round.
  const bool comp_round = s->rounds == 0 || verifying || s->prev_moving || !(s->chains || s->strips) || !b->has_grid;
  // (after a round that moved the labelling at large the pass runs whether or not it was rested: its last count is old)
  if (o.use_components && (active[77] || s->prev_moving) && comp_round) {
    b->counter_slot = 77;
    ran[77] = 1;
    if (b->tick) ++b->tick;
    tic(b, KC_COMPONENT);
    PHMRF_TRY(launch_component_pass(b, bf));
    toc(b, KC_COMPONENT, 1);
  }
  if (s->graph_expansions) {
    // (host-synchronous launches: a few small read-backs per expansion; general graphs are off the hot path)
    for (int a = 0; a < K; ++a)
      if (active[8 + a]) {
        b->counter_slot = 8 + a;
        ran[8 + a] = 1;
        ++b->tick;
        tic(b, KC_STRIP);
        PHMRF_TRY(launch_graph_expansion(b, bf, a));
        toc(b, KC_STRIP, 1);
      }
  }
  if (s->strips) {
    const int geom = s->geom;
    for (int orient = 0; orient < 2; ++orient) {
      unsigned long long lmask = 0ull;
      if (s->expansions)
        for (int a = 0; a < K; ++a)
          if (active[8 + a]) lmask |= 1ull << a;
      if (active[78 + orient]) {
        b->counter_slot = 78 + orient;
        ran[78 + orient] = 1;
        // the fusion pass runs on a cut of its own that moves with the expansions' (so that its memo of quiet strips
        // applies while the cut stays): the expansion cut shifted by half a band / half a segment
        PHMRF_TRY(strip_pass_nocount(b, bf, orient, (GEOM_R[geom] + 3) % 6, (GEOM_C[geom] + 31) % 64, -1, geom));
      } else if (lmask && b->seed && b->has_grid && b->fwd_w && b->D == 8) {
        // (development builds with PHMRF_SEED_MASKS=1 only: b->seed is never allocated otherwise) the proposals' launch also
        // writes the expansions' seed masks (propose_grid_kernel): with the fusion pass at rest it runs for them alone
        if (!b->uT_valid) {
          tic(b, KC_PROPOSE);
          PHMRF_TRY(launch_unary_planes(b));
          toc(b, KC_PROPOSE, 1);
        }
        tic(b, KC_PROPOSE);
        PHMRF_TRY(launch_propose(b, bf));
        toc(b, KC_PROPOSE, 1);
      }
      if (s->expansions) {
        // every active label's expansion of the cut in ONE launch: a wave owns a strip, stages it once and runs the
        // labels back to back behind the exact filter (strip_cols_kernel)
        for (int a = 0; a < K; ++a)
          if (active[8 + a]) ran[8 + a] = 1;
        if (lmask) {
          if (!b->uT_valid) {
            tic(b, KC_PROPOSE);
            PHMRF_TRY(launch_unary_planes(b));
            toc(b, KC_PROPOSE, 1);
          }
          tic(b, KC_STRIP);
          ++b->tick;
          PHMRF_TRY(launch_strip_multi(b, bf, orient, GEOM_R[geom], GEOM_C[geom], lmask, geom));
          b->tick += K;                           // one tick per label inside the launch
          b->work[4] += 1;
          toc(b, KC_STRIP, 1);
        }
      }
    }
  }
  // coarse alpha-expansions.  Coarse scales switch on while the labelling is still moving at large (the previous round
  // changed >= 25 % of the labels: a cold start).  A solve that has moved >= 12.5 % of the labels in all (a far-off warm
  // start of an EM iteration) gets them once at the end, before the tolerance may stop it (force_coarse below); the warm
  // start of a later EM iteration, which moves 1-3 %, does not pay for them at all.
  // A scale that changed labels in its last run stays on (like every move type), with the super-cell grid shifted by
  // one node per round; a verification round tries every shift of both scales.
  // Round 5: inside a scale the LABELS rest like the fine expansions' labels do: a label whose expansion at this scale
  // changed nothing in its last run is left out until the scale is switched on afresh (a round that moved the labelling at
  // large), a verification round or the forced last say runs all labels again.  A coarse round costs what its labels cost
  // -- a share of the coarsen pass, two child strip passes and an apply pass each --, and after the first coarse round of a
  // cold or far-off solve a few labels per scale are still moving (measured: HISTORY.md 3.1 item 6, round 5).  Row tiles keep every label
  // (their schedule runs on sums over the tiles; the per-label counts are local).
  const bool rest_coarse_labels = !(b->tile_top || b->tile_bot) && o.energy_tol_ppb > 0;
  if (s->coarse) {
    bool any_on = false;
    for (int lv = 0; lv < N_COARSE_LV; ++lv) {
      const bool verify_coarse = verifying && (o.energy_tol_ppb == 0 || s->total * COARSE_ON_DIV >= s->sched_n);
      any_on = any_on || verify_coarse || s->force_coarse ||
               (active[80 + lv] && (s->last_changed * COARSE_ROUND_DIV >= s->sched_n || s->coarse_changed[lv] > 0));
    }
    if (any_on && rest_coarse_labels) {
      if (!b->coarse_lab) {
        PHMRF_TRY(dev_alloc(&b->coarse_lab, (size_t)N_COARSE_LV * 64));
        PHMRF_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->coarse_lab_host), N_COARSE_LV * 64 * sizeof(unsigned long long)));
      }
      PHMRF_HIP(hipMemsetAsync(b->coarse_lab, 0, N_COARSE_LV * 64 * sizeof(unsigned long long), b->stream));
    }
    for (int lv = 0; lv < N_COARSE_LV; ++lv) {
      const int sc = COARSE_SCALE[lv];
      // (a verification round tries every shift of every scale -- as long as the solve has moved the labelling at
      //  large or runs to the exact fixed point; the warm start of a later EM iteration under a stopping tolerance,
      //  which moves 1-3 % of the labels, would pay several times its own cost for them)
      const bool verify_coarse = verifying && (o.energy_tol_ppb == 0 || s->total * COARSE_ON_DIV >= s->sched_n);
      const bool on = verify_coarse || s->force_coarse ||
                      (active[80 + lv] && (s->last_changed * COARSE_ROUND_DIV >= s->sched_n || s->coarse_changed[lv] > 0));
      s->coarse_ran[lv] = on;
      if (!on) continue;
      // every label: a verification round, the forced last say, a scale switched on by a round that moved at large
      const bool all_labels = !rest_coarse_labels || verify_coarse || s->force_coarse ||
                              s->last_changed * COARSE_ROUND_DIV >= s->sched_n;
      const unsigned long long kmask = K >= 64 ? ~0ull : ((1ull << K) - 1ull);
      if (all_labels) s->coarse_lab_mask[lv] = kmask;
      s->coarse_all[lv] = all_labels;
      b->counter_slot = 80 + lv;
      ran[80 + lv] = 1;
      for (int off = 0; off < sc; ++off)
        if (verify_coarse || off == r % sc)
          PHMRF_TRY(coarse_sweep_nocount(b, bf, lv, off, (2 * r + lv + off) % 6, (17 * r + 5 * lv + 13 * off) % 64, 0, K,
                                         s->coarse_lab_mask[lv] & kmask));
    }
    if (any_on && rest_coarse_labels)
      PHMRF_HIP(hipMemcpyAsync(b->coarse_lab_host, b->coarse_lab, N_COARSE_LV * 64 * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, b->stream));
  }
  b->counter_slot = 0;
  if (b->timing) PHMRF_TRY(work_fetch_async(b));
  PHMRF_TRY(energy_round_launch(b, &s->incremental, &s->snapshot));
  PHMRF_HIP(hipMemcpyAsync(b->counters_host, b->counters, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                           b->stream));       // (the change counters AND the round's energy sums, slots 120 / 121)
  if (b->tile_top || b->tile_bot) PHMRF_TRY(tile_queue_boundary(b));      // the rows the neighbours need travel with the counters
  s->launched = true;
  s->collected = false;
  return PHMRF_OK;
}

// wait for the round; -> this block's change counters [128] and its (unary, pair without beta) energy after the round
int solve_round_collect(phmrf_block_t b, unsigned long long* counters, double* energy) {
  phmrf_solve_state* s = b->ss;
  PHMRF_CHECK(s, PHMRF_ERR_STATE, "no solve in progress (phmrf_mrf_solve_begin)");
  if (s->status != 0 && !s->launched) {            // decided already: nothing ran
    if (counters) std::memset(counters, 0, 128 * sizeof(unsigned long long));
    if (energy) { energy[0] = s->eu_carry; energy[1] = s->ep_carry; }
    return PHMRF_OK;
  }
  PHMRF_CHECK(s->launched, PHMRF_ERR_STATE, "no round has been launched");
  if (!s->collected) {
    PHMRF_HIP(hipStreamSynchronize(b->stream));
    PHMRF_TRY(energy_round_collect(b, s->beta, s->incremental, &s->eu_carry, &s->ep_carry));
    if (b->timing) work_fold(b, s->rounds == 0);
    s->collected = true;
    if (b->tile_top || b->tile_bot) {
      unsigned long long viol = 0;
      std::memcpy(&viol, b->accum_host + 6, sizeof(viol));
      PHMRF_CHECK(viol == 0, PHMRF_ERR_STATE, "internal: a pinned row of a tile has moved");
    }
  }
  if (counters) {
    std::memcpy(counters, b->counters_host, 128 * sizeof(unsigned long long));
    counters[ROUND_ENERGY_SLOT] = counters[ROUND_ENERGY_SLOT + 1] = 0ull;     // (the energy sums travel in `energy`, not as counters)
  }
  if (energy) { energy[0] = s->eu_carry; energy[1] = s->ep_carry; }
  return PHMRF_OK;
}

// the schedule: what the round's change counters and energy (of this block, or the sums over the tiles of a split block)
// mean for the next round.  *status: 0 another round, 1 converged, 2 stopped by max_rounds / the launch budget
int solve_round_decide(phmrf_block_t b, const unsigned long long* counters, const double* energy, int* status) {
  phmrf_solve_state* s = b->ss;
  PHMRF_CHECK(s, PHMRF_ERR_STATE, "no solve in progress (phmrf_mrf_solve_begin)");
  if (s->status != 0) {
    if (status) *status = s->status;
    return PHMRF_OK;
  }
  PHMRF_CHECK(s->launched && s->collected, PHMRF_ERR_STATE, "round_decide needs a launched and collected round");
  s->launched = false;
  const phmrf_solve_opts& o = s->o;
  std::vector<char>& active = s->active;
  const std::vector<int>& slots = s->slots;
  const int r = s->rounds;
  const double e_now = energy[0] + s->beta * energy[1];
  int64_t ch = 0;
  for (int sl : slots) {
    ch += (int64_t)counters[sl];
    if (s->ran[sl]) s->last_count[sl] = (long long)counters[sl];
  }
  s->total += ch;
  s->last_changed = ch;
  // "this round moved the labelling at large" (the cut advances, the component pass runs again): >= 1/64 of the labels
  // in a solve that has moved at large in all (a cold or far-off start: every new cut finds more), >= 1/16 otherwise --
  // the first round of a warm-started E-step moves 1-3 %, and with the rule at 1/64 its second round was a full sweep on
  // a new cut plus a component pass: 162 instead of 121 ms per E-step on the whole-genome workload for 1e-6 of energy
  const bool moving = ch * (s->total * COARSE_ON_DIV >= s->sched_n ? 64 : 16) >= s->sched_n;
  s->prev_moving = moving;
  if (moving) s->geom = (s->geom + 1) % 3;
  for (int lv = 0; lv < N_COARSE_LV; ++lv)
    if (s->coarse_ran[lv]) {
      s->coarse_changed[lv] = (int64_t)counters[80 + lv];
      // the labels that moved something at this scale stay; the others rest (round_launch decides when all run again)
      if (b->coarse_lab_host && !(b->tile_top || b->tile_bot) && o.energy_tol_ppb > 0) {
        unsigned long long keep = 0ull;
        for (int a = 0; a < b->K; ++a)
          if (b->coarse_lab_host[lv * 64 + a] > 0ull) keep |= 1ull << a;
        s->coarse_lab_mask[lv] = keep;
      }
    }
  ++s->rounds;
  const bool improved = std::isinf(s->e_prev) ? ch > 0 : e_now < s->e_prev - 1e-11 * std::fabs(s->e_prev);
  static const bool trace = getenv("PHMRF_SOLVE_TRACE") != nullptr;   // development aid (one block at a time)
  if (trace) {
    int n_active = 0;
    for (int sl : slots) n_active += active[sl];
    fprintf(stderr, "[phmrf solve] round %d active %d/%d changed %lld energy %.6f delta %.3e\n", r, n_active,
            (int)slots.size(), (long long)ch, e_now, e_now - s->e_prev);
    if (b->timing) {                                // per-round time of each kernel class (ms)
      static double seen[PHMRF_NUM_KERNEL_CLASSES] = {};
      resolve_timing(b);
      fprintf(stderr, "[phmrf solve]   ms:");
      static const char* NM[PHMRF_NUM_KERNEL_CLASSES] = {"emis", "icm", "chain", "comp", "energy", "post", "strip", "prop", "coarse", "fusion"};
      for (int kc = 0; kc < PHMRF_NUM_KERNEL_CLASSES; ++kc) {
        fprintf(stderr, " %s %.2f", NM[kc], b->ms[kc] - seen[kc]);
        seen[kc] = b->ms[kc];
      }
      fprintf(stderr, "  changed by slot:");
      for (int sl : slots)
        if (counters[sl]) fprintf(stderr, " %d:%llu", sl, counters[sl]);
      fprintf(stderr, "\n");
      if (b->counters_host[100])
        fprintf(stderr, "[phmrf solve]   expansion strips: launched %llu, past memo+mask %llu, into DP %llu, DP steps %llu, with a move %llu\n",
                b->counters_host[100], b->counters_host[101], b->counters_host[102], b->counters_host[103],
                b->counters_host[104]);
    }
  }
  const double gain = s->e_prev - e_now;
  if (e_now < s->e_prev) s->e_prev = e_now;
  auto finish = [&](int st) {
    s->status = st;
    if (status) *status = st;
    return PHMRF_OK;
  };
  auto next_round = [&]() {               // (the change stamps are 16-bit launch ticks)
    if (s->rounds >= o.max_rounds || b->tick >= 60000) return finish(2);
    if (status) *status = 0;
    return (int)PHMRF_OK;
  };
  // accepted tolerance: the round (all active types; the rested ones were worth at most a quarter of the tolerance
  // together, see below) changed the energy by less than the tolerance.  A round that RAISED the energy by the
  // tolerance or more (f32 move arithmetic against the f64 energy) is not "converged": it is quiet, and the
  // verification round decides.
  bool coarse_moved = false, coarse_just_ran = s->coarse;
  for (int lv = 0; lv < N_COARSE_LV; ++lv) {
    coarse_moved = coarse_moved || (s->coarse && s->coarse_ran[lv] && s->coarse_changed[lv] > 0);
    coarse_just_ran = coarse_just_ran && s->coarse_ran[lv];
  }
  // (round 5) under a stopping tolerance the coarse scales have their forced "last say" ONCE per solve: it runs every label at
  // every scale, and the rounds after it -- the scales stay on while they move labels, with the labels that moved nothing at
  // rest -- end when a whole round gains less than the tolerance, like any other round.  Until round 5 every coarse move asked
  // for another forced round before the solve might stop (measured on the cold solve of the 12.4 M-node block: two of them, 17
  // of 73 ms, for 1.4 of energy at a tolerance of 59; one more after the rule was tied to the number of labels moved since: 8.5 of
  // 62 ms for 26).  Exact solves (tolerance 0) and row tiles keep the old rule.
  if (coarse_moved && (o.energy_tol_ppb == 0 || b->tile_top || b->tile_bot)) s->coarse_checked = false;
  s->force_coarse = false;
  // (|gain| below the tolerance on either side: a round that moved three labels and changed the f64 energy sum by
  //  2e-13 of itself, up or down, has converged; a rise of the tolerance's size or more has not -- the verification
  //  round decides then)
  if (o.energy_tol_ppb > 0 && std::fabs(gain) < 1e-9 * o.energy_tol_ppb * std::fabs(s->e_prev)) {
    // A solve that has moved the labelling at large (>= 12.5 % of the labels so far: a cold or far-off start, not the
    // warm start of a later EM iteration) does not stop before the coarse scales have run once more and gained less
    // than the tolerance, too: their gains come in few large steps, not in the trickle the tolerance watches.
    if (s->coarse && !s->coarse_checked && !coarse_just_ran && s->total * COARSE_ON_DIV >= s->sched_n) {
      s->coarse_checked = true;
      s->force_coarse = true;
      for (int sl : slots) active[sl] = 1;
      return next_round();
    }
    s->converged = 1;
    return finish(1);
  }
  const bool quiet = ch <= s->tol || !improved;
  if (quiet) {
    if (s->all_active && s->verifying) {
      s->converged = 1;
      return finish(1);
    }
    for (int sl : slots) active[sl] = 1;           // verification round, on the next cut
    s->all_active = true;
    s->verifying = true;
    if (!moving) s->geom = (s->geom + 1) % 3;
    return next_round();
  }
  s->verifying = false;
  // A type stays active while it changes labels.  With an energy tolerance the types whose last run changed the
  // fewest labels are rested until the verification round, as long as ALL rested types together were worth at most a
  // quarter of the stopping tolerance at this round's average gain per changed label (a heuristic about label counts, not
  // a bound on the energy a tolerance stop leaves behind: that is what the parity tests against gco measure).
  int n_act = 0;
  for (int sl : slots) active[sl] = 1;
  if (o.energy_tol_ppb > 0 && ch > 0 && gain > 0) {
    const double budget_labels = 1e-9 * o.energy_tol_ppb * std::fabs(s->e_prev) / 4.0 / (gain / (double)ch);
    // (only types that have run in this solve can be rested, on the count of their last run)
    std::vector<int> by_count;
    for (int sl : slots)
      if (s->last_count[sl] >= 0) by_count.push_back(sl);
    std::sort(by_count.begin(), by_count.end(), [&](int a, int c) { return s->last_count[a] < s->last_count[c]; });
    double used = 0.0;
    for (int sl : by_count) {
      used += (double)s->last_count[sl];
      if (used > budget_labels) break;
      active[sl] = 0;
    }
  } else {
    for (int sl : slots) active[sl] = s->last_count[sl] != 0 ? 1 : 0;
  }
  // (round 5) labels that a coarse scale has just moved are new inputs for the fine moves around them: the strip fusion and
  // the strip expansions run in the next round even if their own last counts had put them to rest -- their memos of quiet
  // strips keep that to the strips near the coarse changes.  (Before, they waited for the forced round: on the cold solve
  // of the 12.4 M-node block five coarse rounds moved 6,245 labels while the fine moves rested, and the forced round after
  // them then found 680 of energy in the fine moves -- twelve tolerances that every earlier stop would have left behind.)
  if (coarse_moved && s->strips && o.energy_tol_ppb > 0 && !(b->tile_top || b->tile_bot)) {
    active[78] = active[79] = 1;
    if (s->expansions)
      for (int a = 0; a < b->K; ++a) active[8 + a] = 1;
  }
  for (int sl : slots) n_act += active[sl];
  s->all_active = n_act == (int)slots.size();
  if (n_act == 0) {
    for (int sl : slots) active[sl] = 1;
    s->all_active = true;
  }
  return next_round();
}

int solve_end(phmrf_block_t b, phmrf_solve_result* res) {
  phmrf_solve_state* s = b->ss;
  PHMRF_CHECK(s, PHMRF_ERR_STATE, "no solve in progress (phmrf_mrf_solve_begin)");
  if (s->launched && !s->collected) (void)hipStreamSynchronize(b->stream);
  b->has_labels = true;
  int st = PHMRF_OK;
  if (res) {
    double eu = 0, ep = 0;
    st = energy_now(b, s->beta, &eu, &ep);
    res->energy = eu + ep;
    res->energy_unary = eu;
    res->energy_pair = ep;
    res->energy_init = s->have_init_energy ? s->eu0 + s->ep0 : std::numeric_limits<double>::quiet_NaN();
    res->rounds = s->rounds;
    res->converged = s->converged;
    res->changed = s->total;
  }
  static const bool child_count = PHMRF_DEV_ENV("PHMRF_CHILD_COUNT") != nullptr;     // development (with PHMRF_SOLVE_TRACE)
  if (child_count) {
    (void)hipStreamSynchronize(b->stream);
    for (int lv = 0; lv < 3; ++lv) {
      unsigned long long tot[5] = {0, 0, 0, 0, 0};
      for (int q = 0; q < 4; ++q) {
        if (!b->coarse[lv * 4 + q]) continue;
        unsigned long long c[5];
        // (a child's passes run with alpha = 1, counter slot 0: strip_kernel's counters 100.. land at 91..)
        if (hipMemcpy(c, b->coarse[lv * 4 + q]->counters + 91, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) continue;
        for (int x = 0; x < 5; ++x) tot[x] += c[x];
      }
      fprintf(stderr, "[phmrf solve] coarse scale %d child strips: seen %llu, past the memo and the pin look %llu, into the DP %llu, DP steps %llu, with a move %llu\n",
              lv == 0 ? 2 : (lv == 1 ? 4 : 8), tot[0], tot[1], tot[2], tot[3], tot[4]);
    }
  }
  solve_scope_exit(b);
  return st;
}
}  // namespace

extern "C" {

int phmrf_mrf_solve_begin(phmrf_block_t b, double beta, const phmrf_solve_opts* opts, int want_init_energy) {
  return solve_begin(b, beta, opts, want_init_energy != 0);
}
int phmrf_mrf_solve_round_launch(phmrf_block_t b) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  return solve_round_launch(b);
}
int phmrf_mrf_solve_round_collect(phmrf_block_t b, uint64_t* counters, double* energy) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  return solve_round_collect(b, reinterpret_cast<unsigned long long*>(counters), energy);
}
int phmrf_mrf_solve_round_decide(phmrf_block_t b, const uint64_t* counters, const double* energy, int* status) {
  PHMRF_CHECK(b && counters && energy, PHMRF_ERR_INVALID, "NULL argument");
  return solve_round_decide(b, reinterpret_cast<const unsigned long long*>(counters), energy, status);
}
int phmrf_mrf_solve_end(phmrf_block_t b, phmrf_solve_result* res) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  return solve_end(b, res);
}

int phmrf_mrf_solve(phmrf_block_t b, double beta, const phmrf_solve_opts* opts, phmrf_solve_result* res) {
  PHMRF_TRY(solve_begin(b, beta, opts, res != nullptr));
  struct Scope {                               // an error inside the loop still ends the solve
    phmrf_block* blk;
    ~Scope() { if (blk->ss) solve_scope_exit(blk); }
  } scope{b};
  int status = 0;
  unsigned long long counters[128];
  double energy[2];
  while (status == 0) {
    if (b->ss->rounds >= b->ss->o.max_rounds || b->tick >= 60000) break;
    PHMRF_TRY(solve_round_launch(b));
    PHMRF_TRY(solve_round_collect(b, counters, energy));
    PHMRF_TRY(solve_round_decide(b, counters, energy, &status));
  }
  return solve_end(b, res);
}


// Several blocks solved in LOCKSTEP ROUNDS from one host thread (round 6): every undecided block's round is queued on its own
// stream, then the rounds are collected and decided in the same order -- the blocks' kernels overlap on the GPU as they do
// when a host thread per block drives them, without the threads.  Each block's state machine is the one phmrf_mrf_solve
// runs: block for block the same labelling (bit-identical under PHMRF_DETERMINISTIC=1).
int phmrf_mrf_solve_group(phmrf_block_t* blocks, int n_blocks, double beta, const phmrf_solve_opts* opts) {
  PHMRF_CHECK(blocks && n_blocks >= 0, PHMRF_ERR_INVALID, "NULL argument");
  for (int i = 0; i < n_blocks; ++i) PHMRF_CHECK(blocks[i], PHMRF_ERR_INVALID, "block is NULL");
  struct Scope {                               // an error inside the loop still ends every solve
    phmrf_block_t* bl;
    int n;
    ~Scope() {
      for (int i = 0; i < n; ++i)
        if (bl[i]->ss) solve_scope_exit(bl[i]);
    }
  } scope{blocks, 0};
  for (int i = 0; i < n_blocks; ++i) {
    PHMRF_TRY(solve_begin(blocks[i], beta, opts, false));
    scope.n = i + 1;
  }
  std::vector<int> status(n_blocks, 0);
  std::vector<char> in_flight(n_blocks, 0);
  unsigned long long counters[128];
  double energy[2];
  // queue a block's next round, unless it is decided or out of rounds
  auto launch = [&](int i) -> int {
    phmrf_block* b = blocks[i];
    if (status[i] != 0) return PHMRF_OK;
    if (b->ss->rounds >= b->ss->o.max_rounds || b->tick >= 60000) {
      status[i] = 2;
      return PHMRF_OK;
    }
    PHMRF_TRY(solve_round_launch(b));
    in_flight[i] = 1;
    return PHMRF_OK;
  };
  int n_in_flight = 0;
  for (int i = 0; i < n_blocks; ++i) {
    PHMRF_TRY(launch(i));
    n_in_flight += in_flight[i];
  }
  // (round 6) the rounds are taken as they END: this thread looks at the streams in turn (hipStreamQuery), and a block whose
  // round has drained is collected, decided and given its next round at once -- no block waits for another's round.  (The
  // first form queued a round of every block, then collected them all in order: 69 - 71 ms per E-step of the whole-genome
  // workload where fourteen threads take 60 - 63.)
  while (n_in_flight > 0) {
    bool progressed = false;
    for (int i = 0; i < n_blocks; ++i) {
      if (!in_flight[i]) continue;
      const hipError_t q = hipStreamQuery(blocks[i]->stream);
      if (q == hipErrorNotReady) continue;
      PHMRF_HIP(q);
      PHMRF_TRY(solve_round_collect(blocks[i], counters, energy));
      PHMRF_TRY(solve_round_decide(blocks[i], counters, energy, &status[i]));
      in_flight[i] = 0;
      --n_in_flight;
      PHMRF_TRY(launch(i));
      n_in_flight += in_flight[i];
      progressed = true;
    }
    if (!progressed) std::this_thread::yield();
  }
  int st = PHMRF_OK;
  for (int i = 0; i < n_blocks; ++i) {
    const int s1 = solve_end(blocks[i], nullptr);
    if (s1 != PHMRF_OK) st = s1;
  }
  scope.n = 0;
  return st;
}

// ---- row tiles (tile.hip) -----------------------------------------------------------------------
static int64_t row_first(const phmrf_block* b, int i) {      // first node of grid row i (i == H: n)
  return b->diagonal ? (int64_t)i * b->W - ((int64_t)i * (i - 1)) / 2 : (int64_t)i * b->W;
}

int phmrf_block_set_tile(phmrf_block_t b, int top, int bottom, int64_t sched_n) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_CHECK(b->has_grid, PHMRF_ERR_STATE, "row tiles need the grid geometry (phmrf_block_set_grid / build_grid_graph)");
  PHMRF_CHECK(!b->ss, PHMRF_ERR_STATE, "a solve is in progress");
  top = top ? 1 : 0;
  bottom = bottom ? 1 : 0;
  PHMRF_CHECK(b->H >= 2 * (top + bottom) + 2, PHMRF_ERR_INVALID, "a tile needs two rows per cut and two more");
  for (int r = 0; r < 2; ++r) {
    dev_free(b->pin_save[r]);
    dev_free(b->pin_label[r]);
    b->pin_first[r] = b->pin_count[r] = b->pin_split[r] = 0;
    b->pin_rows[r] = 0;
  }
  b->pin_saved = false;
  b->tile_top = top;
  b->tile_bot = bottom;
  b->sched_n = sched_n > 0 ? sched_n : 0;
  if (!top && !bottom) {
    b->own0 = 0;
    b->own1 = -1;
    b->own_r0 = 0;
    b->own_r1 = -1;
    return PHMRF_OK;
  }
  b->own_r0 = top ? 1 : 0;
  b->own_r1 = b->H - (bottom ? 1 : 0);
  b->own0 = row_first(b, b->own_r0);
  b->own1 = row_first(b, b->own_r1);
  if (top) {
    b->pin_first[0] = 0;
    b->pin_count[0] = row_first(b, 2);
    b->pin_split[0] = row_first(b, 1);
  }
  if (bottom) {
    b->pin_first[1] = row_first(b, b->H - 2);
    b->pin_count[1] = b->n - b->pin_first[1];
    b->pin_split[1] = row_first(b, b->H - 1) - b->pin_first[1];
  }
  int64_t cap = 0;
  for (int r = 0; r < 2; ++r)
    if (b->pin_count[r] > 0) {
      PHMRF_TRY(dev_alloc(&b->pin_save[r], (size_t)b->pin_count[r] * b->K));
      PHMRF_TRY(dev_alloc(&b->pin_label[r], (size_t)b->pin_count[r]));
      cap += b->pin_count[r];
    }
  if (cap > b->xfer_cap) {
    dev_free(b->xfer);
    if (b->xfer_host) (void)hipHostFree(b->xfer_host);
    b->xfer_host = nullptr;
    PHMRF_TRY(dev_alloc(&b->xfer, (size_t)cap));
    PHMRF_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->xfer_host), (size_t)2 * cap));     // outgoing | incoming rows
    b->xfer_cap = cap;
  }
  return PHMRF_OK;
}

// the node range of the first (from the top) / last (from the bottom) `rows` rows of a pin region
static void pin_range(const phmrf_block* b, int region, int rows, int64_t* first, int64_t* count) {
  *first = b->pin_first[region];
  *count = 0;
  if (rows <= 0 || b->pin_count[region] == 0) return;
  if (rows >= 2) {
    *count = b->pin_count[region];
  } else if (region == 0) {
    *count = b->pin_split[0];
  } else {
    *first = b->pin_first[1] + b->pin_split[1];
    *count = b->pin_count[1] - b->pin_split[1];
  }
}

int phmrf_block_tile_pins(phmrf_block_t b, int n_top, int n_bottom) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_CHECK(b->tile_top || b->tile_bot, PHMRF_ERR_STATE, "the block is not a tile (phmrf_block_set_tile)");
  PHMRF_CHECK(b->has_logprob, PHMRF_ERR_STATE, "logprob not set");
  PHMRF_CHECK(n_top >= 0 && n_top <= 2 && n_bottom >= 0 && n_bottom <= 2, PHMRF_ERR_INVALID, "0, 1 or 2 rows per cut");
  const int want[2] = {b->tile_top ? n_top : 0, b->tile_bot ? n_bottom : 0};
  if (b->tick) ++b->tick;                  // (one tick whatever the tile's cuts: the tiles of a block count alike)
  for (int r = 0; r < 2; ++r) {
    if (b->pin_count[r] == 0) continue;
    int64_t pf, pc, wf, wc;
    pin_range(b, r, want[r], &pf, &pc);
    pin_range(b, r, b->pin_saved ? b->pin_rows[r] : 0, &wf, &wc);
    PHMRF_TRY(launch_tile_pins(b, r, b->pin_first[r], b->pin_count[r], pf, pc, wf, wc, !b->pin_saved));
    b->pin_rows[r] = want[r];
  }
  b->pin_saved = true;
  return PHMRF_OK;
}

// the rows the neighbour tiles need: top_out = the first row this tile owns, bottom_out = the last (either may be NULL)
int phmrf_block_tile_get_boundary(phmrf_block_t b, uint8_t* top_out, uint8_t* bottom_out) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_CHECK(b->tile_top || b->tile_bot, PHMRF_ERR_STATE, "the block is not a tile (phmrf_block_set_tile)");
  // inside a solve the rows were queued behind the round's moves (solve_round_launch) and have arrived with the round's
  // counters (solve_round_collect): no device traffic, no wait
  const bool have = b->boundary_queued && b->ss && b->ss->collected;
  if (!have) {
    PHMRF_TRY(tile_queue_boundary(b));
    PHMRF_HIP(hipStreamSynchronize(b->stream));
  }
  b->boundary_queued = false;
  int64_t off = 0;
  if (b->tile_top) {
    const int64_t tc = row_first(b, 2) - row_first(b, 1);
    if (top_out) std::memcpy(top_out, b->xfer_host, (size_t)tc);
    off = tc;
  }
  if (b->tile_bot && bottom_out) {
    const int64_t bc = row_first(b, b->H - 1) - row_first(b, b->H - 2);
    std::memcpy(bottom_out, b->xfer_host + off, (size_t)bc);
  }
  return PHMRF_OK;
}

// the neighbours' rows: top_in -> this tile's first row (its upper halo), bottom_in -> its last row (either may be NULL)
int phmrf_block_tile_put_halo(phmrf_block_t b, const uint8_t* top_in, const uint8_t* bottom_in) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_CHECK(b->tile_top || b->tile_bot, PHMRF_ERR_STATE, "the block is not a tile (phmrf_block_set_tile)");
  if (b->tick) ++b->tick;
  b->labels_are_slot = 0;
  uint8_t* const in_host = b->xfer_host + b->xfer_cap;      // (the second half of the pinned buffer: the first holds the outgoing rows)
  int64_t off = 0;
  if (top_in && b->tile_top) {
    const int64_t c = row_first(b, 1);
    for (int64_t q = 0; q < c; ++q) PHMRF_CHECK(top_in[q] < b->K, PHMRF_ERR_INVALID, "label out of range [0,K)");
    std::memcpy(in_host, top_in, (size_t)c);
    PHMRF_HIP(hipMemcpyAsync(b->xfer, in_host, (size_t)c, hipMemcpyHostToDevice, b->stream));
    PHMRF_TRY(launch_put_labels(b, 0, c, b->xfer));
    off = c;
  }
  if (bottom_in && b->tile_bot) {
    const int64_t f = row_first(b, b->H - 1), c = b->n - f;
    for (int64_t q = 0; q < c; ++q) PHMRF_CHECK(bottom_in[q] < b->K, PHMRF_ERR_INVALID, "label out of range [0,K)");
    std::memcpy(in_host + off, bottom_in, (size_t)c);
    PHMRF_HIP(hipMemcpyAsync(b->xfer + off, in_host + off, (size_t)c, hipMemcpyHostToDevice, b->stream));
    PHMRF_TRY(launch_put_labels(b, f, c, b->xfer + off));
  }
  // (no wait: inside a solve the next call comes after the next round has been collected -- a stream synchronisation --,
  //  so the copies have left the staging buffers by then; outside a solve the caller's next synchronising call does it)
  if (!b->ss) PHMRF_HIP(hipStreamSynchronize(b->stream));
  return PHMRF_OK;
}

// ---- b3 posterior / stats -----------------------------------------------------------------------
static int posterior_launch(phmrf_block_t b, double beta, int estimate_type, bool write_post) {
  PHMRF_TRY(check_solvable(b));
  PHMRF_CHECK(b->has_X, PHMRF_ERR_STATE, "observations not set");
  const int ns = n_stats(b);
  PHMRF_CHECK(8 + ns <= ACCUM_DOUBLES, PHMRF_ERR_UNSUPPORTED, "K*(1+S+S*S) too large");
  if (write_post && !b->posteriors) PHMRF_TRY(dev_alloc(&b->posteriors, (size_t)b->n * b->K));
  PHMRF_TRY(zero_accum(b, 0, 4));
  PHMRF_TRY(zero_accum(b, 8, ns));
  tic(b, KC_POSTERIOR);
  PHMRF_TRY(launch_posterior_stats(b, (float)beta, estimate_type, write_post));
  toc(b, KC_POSTERIOR, 1);
  return PHMRF_OK;
}

int phmrf_posterior_stats(phmrf_block_t b, double beta, int estimate_type, double* stats_out, double* costs_out,
                          double* posteriors_out) {
  PHMRF_CHECK(b && stats_out && costs_out, PHMRF_ERR_INVALID, "NULL argument");
  PHMRF_TRY(posterior_launch(b, beta, estimate_type, posteriors_out != nullptr));
  const int ns = n_stats(b);
  PHMRF_HIP(hipMemcpyAsync(b->accum_host, b->accum, (8 + ns) * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  PHMRF_HIP(hipStreamSynchronize(b->stream));
  std::memcpy(costs_out, b->accum_host, 4 * sizeof(double));
  std::memcpy(stats_out, b->accum_host + 8, ns * sizeof(double));
  if (posteriors_out) {
    const size_t cnt = (size_t)b->n * b->K;
    std::vector<float> tmp(cnt);
    PHMRF_TRY(download(tmp.data(), b->posteriors, cnt * sizeof(float), b->stream));
    for (size_t i = 0; i < cnt; ++i) posteriors_out[i] = tmp[i];
  }
  return PHMRF_OK;
}

int phmrf_posterior_stats_dev(phmrf_block_t b, double beta, int estimate_type, double* out_dev) {
  PHMRF_CHECK(b && out_dev, PHMRF_ERR_INVALID, "NULL argument");
  PHMRF_TRY(posterior_launch(b, beta, estimate_type, false));
  const int ns = n_stats(b);
  PHMRF_HIP(hipMemcpyAsync(out_dev, b->accum + 8, ns * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
  PHMRF_HIP(hipMemcpyAsync(out_dev + ns, b->accum, 4 * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
  return PHMRF_OK;
}

// ---- initialisation -----------------------------------------------------------------------------
int phmrf_kmeans_step(phmrf_block_t b, const double* centers, int write_labels, double* out) {
  PHMRF_CHECK(b && centers && out, PHMRF_ERR_INVALID, "NULL argument");
  PHMRF_CHECK(b->has_X, PHMRF_ERR_STATE, "observations not set");
  const int K = b->K, S = b->S, NP = K * S + K + 1;
  PHMRF_CHECK(8 + NP <= ACCUM_DOUBLES, PHMRF_ERR_UNSUPPORTED, "K*S too large");
  std::vector<float> c((size_t)K * S);
  for (int i = 0; i < K * S; ++i) {
    PHMRF_CHECK(std::isfinite(centers[i]), PHMRF_ERR_INVALID, "centres must be finite");
    c[i] = (float)centers[i];
  }
  PHMRF_TRY(upload(b->emis_params, c.data(), c.size() * sizeof(float), b->stream));   // K*S floats fit the emission pack
  PHMRF_TRY(zero_accum(b, 8, NP));
  PHMRF_TRY(launch_kmeans_step(b, b->emis_params, write_labels != 0, b->accum + 8));
  PHMRF_HIP(hipMemcpyAsync(b->accum_host + 8, b->accum + 8, NP * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  PHMRF_HIP(hipStreamSynchronize(b->stream));
  std::memcpy(out, b->accum_host + 8, NP * sizeof(double));
  if (write_labels) {
    b->has_labels = true;
    b->labels_are_slot = 0;
  }
  return PHMRF_OK;
}

int phmrf_kmeans_moments(phmrf_block_t b, const double* centers, int write_labels, double* out) {
  PHMRF_CHECK(b && centers && out, PHMRF_ERR_INVALID, "NULL argument");
  PHMRF_CHECK(b->has_X, PHMRF_ERR_STATE, "observations not set");
  const int K = b->K, S = b->S, NP = K * S + K + 1 + K * S * S;
  PHMRF_CHECK(S <= 8, PHMRF_ERR_UNSUPPORTED, "second moments: S must be in [1,8]");
  PHMRF_CHECK(8 + NP <= ACCUM_DOUBLES, PHMRF_ERR_UNSUPPORTED, "K*S*S too large");
  std::vector<float> c((size_t)K * S);
  for (int i = 0; i < K * S; ++i) {
    PHMRF_CHECK(std::isfinite(centers[i]), PHMRF_ERR_INVALID, "centres must be finite");
    c[i] = (float)centers[i];
  }
  PHMRF_TRY(upload(b->emis_params, c.data(), c.size() * sizeof(float), b->stream));
  PHMRF_TRY(zero_accum(b, 8, NP));
  PHMRF_TRY(launch_kmeans_step(b, b->emis_params, write_labels != 0, b->accum + 8, true));
  PHMRF_HIP(hipMemcpyAsync(b->accum_host + 8, b->accum + 8, NP * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  PHMRF_HIP(hipStreamSynchronize(b->stream));
  std::memcpy(out, b->accum_host + 8, NP * sizeof(double));
  if (write_labels) {
    b->has_labels = true;
    b->labels_are_slot = 0;
  }
  return PHMRF_OK;
}

// ---- measurement --------------------------------------------------------------------------------
int phmrf_block_enable_timing(phmrf_block_t b, int enable) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  b->timing = enable != 0;
  return PHMRF_OK;
}

int phmrf_block_set_timing_classes(phmrf_block_t b, uint32_t class_mask) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  b->timing_mask = class_mask;
  return PHMRF_OK;
}

int phmrf_block_get_timing(phmrf_block_t b, int capacity, double* ms, int64_t* launches) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_CHECK(capacity >= 0, PHMRF_ERR_INVALID, "capacity < 0");
  resolve_timing(b);
  for (int i = 0; i < PHMRF_NUM_KERNEL_CLASSES && i < capacity; ++i) {
    if (ms) ms[i] = b->ms[i];
    if (launches) launches[i] = b->launches[i];
  }
  return PHMRF_OK;
}

int phmrf_block_get_timing_first(phmrf_block_t b, int capacity, double* ms, int64_t* launches) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  PHMRF_CHECK(capacity >= 0, PHMRF_ERR_INVALID, "capacity < 0");
  resolve_timing(b);
  for (int i = 0; i < PHMRF_NUM_KERNEL_CLASSES && i < capacity; ++i) {
    if (ms) ms[i] = b->ms_first[i];
    if (launches) launches[i] = b->launches_first[i];
  }
  return PHMRF_OK;
}

int phmrf_block_reset_timing(phmrf_block_t b) {
  PHMRF_CHECK(b, PHMRF_ERR_INVALID, "block is NULL");
  resolve_timing(b);
  for (int i = 0; i < PHMRF_NUM_KERNEL_CLASSES; ++i) {
    b->ms[i] = b->ms_first[i] = 0;
    b->launches[i] = b->launches_first[i] = 0;
  }
  for (int q = 0; q < 10; ++q) b->work[q] = b->work_first[q] = 0;
  PHMRF_HIP(hipMemsetAsync(b->work_acc, 0, WORK_BANKS * WORK_SLOTS * sizeof(unsigned long long), b->stream));
  b->intervals.clear();
  return PHMRF_OK;
}

int phmrf_time_base_reset(void) {
  int dev = 0;
  PHMRF_HIP(hipGetDevice(&dev));
  PHMRF_CHECK(dev >= 0 && dev < 64, PHMRF_ERR_UNSUPPORTED, "device index >= 64");
  if (!g_time_base[dev]) PHMRF_HIP(hipEventCreate(&g_time_base[dev]));
  PHMRF_HIP(hipDeviceSynchronize());
  PHMRF_HIP(hipEventRecord(g_time_base[dev], nullptr));
  PHMRF_HIP(hipEventSynchronize(g_time_base[dev]));
  return PHMRF_OK;
}

int phmrf_block_get_work(phmrf_block_t b, int64_t* out) {
  PHMRF_CHECK(b && out, PHMRF_ERR_INVALID, "NULL argument");
  for (int q = 0; q < 8; ++q) out[q] = b->work[q];
  return PHMRF_OK;
}

int phmrf_block_get_work_first(phmrf_block_t b, int64_t* out) {
  PHMRF_CHECK(b && out, PHMRF_ERR_INVALID, "NULL argument");
  for (int q = 0; q < 8; ++q) out[q] = b->work_first[q];
  return PHMRF_OK;
}

int phmrf_block_get_work_ex(phmrf_block_t b, int first_round_only, int capacity, int64_t* out) {
  PHMRF_CHECK(b && out, PHMRF_ERR_INVALID, "NULL argument");
  for (int q = 0; q < 10 && q < capacity; ++q) out[q] = first_round_only ? b->work_first[q] : b->work[q];
  return PHMRF_OK;
}

int phmrf_block_get_intervals(phmrf_block_t b, int kclass, double* out, int64_t capacity, int64_t* count) {
  PHMRF_CHECK(b && count, PHMRF_ERR_INVALID, "NULL argument");
  PHMRF_CHECK(kclass >= 0 && kclass < PHMRF_NUM_KERNEL_CLASSES, PHMRF_ERR_INVALID, "no such kernel class");
  resolve_timing(b);
  int64_t c = 0;
  for (const auto& iv : b->intervals)
    if (iv.kclass == kclass) {
      if (out && c < capacity) {
        out[2 * c] = iv.t0;
        out[2 * c + 1] = iv.t1;
      }
      ++c;
    }
  *count = c;
  return PHMRF_OK;
}

}  // extern "C"
