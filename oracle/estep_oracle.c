/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's CPU E-step, never linked by the product.
 *
 * What it restates (file:line in /root/reference):
 *   oracle_emission        phylo_hmrf.py:266-268 -> sklearn-0.18 log_multivariate_normal_density(...,'full')
 *                          (Cholesky, +1e-7*I retry, triangular solve, log-det; published algorithm, the module is
 *                          not vendored: pinned against scipy's logpdf and the golden fixtures)
 *   oracle_swap            pygco.cut_general_graph(..., algorithm='swap') (phylo_hmrf.py:496-498) -> gco-v3.0
 *                          GCoptimization::swap / oneSwapIteration / alpha_beta_swap (GCoptimization.cpp:1282-1394),
 *                          binary energy construction of energy.h:204-243 (add_term1 / add_term2), integer energies
 *                          (int32 terms, int64 totals, GCoptimization.h:164-171).  The max-flow inside is Dinic's
 *                          algorithm instead of gco's Boykov-Kolmogorov (any exact max-flow yields the same cut
 *                          VALUE; when the minimum cut is not unique the label choice follows BK's convention:
 *                          a node goes to the sink label only if it can still reach the sink in the residual graph).
 *   oracle_posterior_stats phylo_hmrf.py:334-355, :374-396, :398-468, :311-314 (per-node loops, float64)
 *   oracle_energy          SURVEY.md 8(a-E)
 *
 * Pinned by tests/test_oracle_c.py against the golden fixtures and against the compiled reference gco
 * (oracle/_ref) on the same integer problems.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int phmrf_oracle_version(void) { return 1; }

/* ------------------------------------------------------------------------------------------------ emission */
/* X [n,S], means [K,S], covars [K,S,S] -> logprob [n,K]; returns 0, or 1 if a covariance is not PD. */
int oracle_emission(const double* X, int64_t n, int S, int K, const double* means, const double* covars, double* out) {
  double* L = (double*)malloc(sizeof(double) * S * S);
  double* y = (double*)malloc(sizeof(double) * S);
  const double log2pi = log(2.0 * M_PI);
  for (int k = 0; k < K; ++k) {
    const double* cv = covars + (size_t)k * S * S;
    int ok = 0;
    for (int attempt = 0; attempt < 2 && !ok; ++attempt) {
      const double jit = attempt ? 1e-7 : 0.0;
      ok = 1;
      memset(L, 0, sizeof(double) * S * S);
      for (int i = 0; i < S && ok; ++i)
        for (int j = 0; j <= i; ++j) {
          double s = cv[i * S + j] + (i == j ? jit : 0.0);
          for (int t = 0; t < j; ++t) s -= L[i * S + t] * L[j * S + t];
          if (i == j) {
            if (!(s > 0.0)) { ok = 0; break; }
            L[i * S + i] = sqrt(s);
          } else {
            L[i * S + j] = s / L[j * S + j];
          }
        }
    }
    if (!ok) { free(L); free(y); return 1; }
    double logdet = 0.0;
    for (int i = 0; i < S; ++i) logdet += 2.0 * log(L[i * S + i]);
    for (int64_t r = 0; r < n; ++r) {
      double q = 0.0;
      for (int i = 0; i < S; ++i) {      /* forward substitution: L y = x - mu */
        double s = X[r * S + i] - means[(size_t)k * S + i];
        for (int t = 0; t < i; ++t) s -= L[i * S + t] * y[t];
        y[i] = s / L[i * S + i];
        q += y[i] * y[i];
      }
      out[r * K + k] = -0.5 * (q + S * log2pi + logdet);
    }
  }
  free(L);
  free(y);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ max-flow */
typedef struct {
  int nv, ne, cap_e;
  int* head;   /* per vertex: first arc */
  int* nxt;    /* per arc */
  int* to;
  int64_t* cap;
  int* level;
  int* it;
  int* queue;
} Flow;

static void flow_init(Flow* f, int nv, int max_arcs) {
  f->nv = nv; f->ne = 0; f->cap_e = max_arcs;
  f->head = (int*)malloc(sizeof(int) * nv);
  for (int i = 0; i < nv; ++i) f->head[i] = -1;
  f->nxt = (int*)malloc(sizeof(int) * max_arcs);
  f->to = (int*)malloc(sizeof(int) * max_arcs);
  f->cap = (int64_t*)malloc(sizeof(int64_t) * max_arcs);
  f->level = (int*)malloc(sizeof(int) * nv);
  f->it = (int*)malloc(sizeof(int) * nv);
  f->queue = (int*)malloc(sizeof(int) * nv);
}
static void flow_free(Flow* f) {
  free(f->head); free(f->nxt); free(f->to); free(f->cap); free(f->level); free(f->it); free(f->queue);
}
static void flow_arc(Flow* f, int a, int b, int64_t cab, int64_t cba) {
  f->to[f->ne] = b; f->cap[f->ne] = cab; f->nxt[f->ne] = f->head[a]; f->head[a] = f->ne++;
  f->to[f->ne] = a; f->cap[f->ne] = cba; f->nxt[f->ne] = f->head[b]; f->head[b] = f->ne++;
}
static int flow_bfs(Flow* f, int s, int t) {
  for (int i = 0; i < f->nv; ++i) f->level[i] = -1;
  int qh = 0, qt = 0;
  f->queue[qt++] = s; f->level[s] = 0;
  while (qh < qt) {
    const int v = f->queue[qh++];
    for (int e = f->head[v]; e >= 0; e = f->nxt[e])
      if (f->cap[e] > 0 && f->level[f->to[e]] < 0) { f->level[f->to[e]] = f->level[v] + 1; f->queue[qt++] = f->to[e]; }
  }
  return f->level[t] >= 0;
}
/* iterative blocking-flow DFS (explicit stack: the graphs are large) */
static int64_t flow_dfs(Flow* f, int s, int t, int* stack_v, int* stack_e) {
  int64_t total = 0;
  for (;;) {
    int top = 0, v = s;
    stack_v[0] = s;
    while (v != t) {
      int e = f->it[v];
      while (e >= 0 && !(f->cap[e] > 0 && f->level[f->to[e]] == f->level[v] + 1)) e = f->nxt[e];
      f->it[v] = e;
      if (e < 0) {              /* dead end: retreat */
        if (top == 0) return total;
        f->level[v] = -1;
        --top;
        v = stack_v[top];
        f->it[v] = f->nxt[f->it[v]];
        continue;
      }
      stack_e[top] = e;
      v = f->to[e];
      stack_v[++top] = v;
    }
    int64_t push = INT64_MAX;
    for (int i = 0; i < top; ++i) if (f->cap[stack_e[i]] < push) push = f->cap[stack_e[i]];
    for (int i = 0; i < top; ++i) { f->cap[stack_e[i]] -= push; f->cap[stack_e[i] ^ 1] += push; }
    total += push;
  }
}
static int64_t flow_max(Flow* f, int s, int t) {
  int* sv = (int*)malloc(sizeof(int) * (f->nv + 1));
  int* se = (int*)malloc(sizeof(int) * (f->nv + 1));
  int64_t flow = 0;
  while (flow_bfs(f, s, t)) {
    for (int i = 0; i < f->nv; ++i) f->it[i] = f->head[i];
    flow += flow_dfs(f, s, t, sv, se);
  }
  free(sv); free(se);
  return flow;
}
/* reach[v] = 1 if v can still reach t in the residual graph */
static void flow_sink_side(Flow* f, int t, char* reach) {
  memset(reach, 0, f->nv);
  int qh = 0, qt = 0;
  f->queue[qt++] = t; reach[t] = 1;
  while (qh < qt) {
    const int v = f->queue[qh++];
    for (int e = f->head[v]; e >= 0; e = f->nxt[e]) {
      const int u = f->to[e];          /* arc u->v is e^1 */
      if (!reach[u] && f->cap[e ^ 1] > 0) { reach[u] = 1; f->queue[qt++] = u; }
    }
  }
}

/* ------------------------------------------------------------------------------------------------ energy */
static int64_t int_energy(int64_t n, int K, const int* unary, const int* smooth, int64_t E, const int* s1,
                          const int* s2, const int* w, const int* lab) {
  int64_t e = 0;
  for (int64_t i = 0; i < n; ++i) e += unary[i * K + lab[i]];
  for (int64_t k = 0; k < E; ++k) e += (int64_t)w[k] * smooth[lab[s1[k]] * K + lab[s2[k]]];
  return e;
}

/* ------------------------------------------------------------------------------------------------ swap */
/* gco's swap(max_cycles) on integer energies.  labels: in = initial, out = result.  Returns the final energy;
 * *cycles_out = number of cycles executed.  Returns -1 if a pair term is not submodular (gco raises there). */
int64_t oracle_swap(int64_t n, int K, const int* unary, const int* smooth, int64_t E, const int* s1, const int* s2,
                    const int* w, int* labels, int max_cycles, int* cycles_out) {
  /* CSR adjacency with edge ids */
  int64_t* ptr = (int64_t*)calloc(n + 1, sizeof(int64_t));
  for (int64_t k = 0; k < E; ++k) { ptr[s1[k] + 1]++; ptr[s2[k] + 1]++; }
  for (int64_t i = 0; i < n; ++i) ptr[i + 1] += ptr[i];
  int* adj = (int*)malloc(sizeof(int) * 2 * (E > 0 ? E : 1));
  int* adjw = (int*)malloc(sizeof(int) * 2 * (E > 0 ? E : 1));
  int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
  memcpy(fill, ptr, sizeof(int64_t) * (n + 1));
  for (int64_t k = 0; k < E; ++k) {
    adj[fill[s1[k]]] = s2[k]; adjw[fill[s1[k]]++] = w[k];
    adj[fill[s2[k]]] = s1[k]; adjw[fill[s2[k]]++] = w[k];
  }
  free(fill);
  int* var = (int*)malloc(sizeof(int) * n);
  int* active = (int*)malloc(sizeof(int) * n);
  if (max_cycles < 0) max_cycles = 10000000;
  int64_t new_e = int_energy(n, K, unary, smooth, E, s1, s2, w, labels), old_e = new_e + 1;
  int cycle = 0, bad = 0;
  while (old_e > new_e && cycle < max_cycles && !bad) {                 /* GCoptimization.cpp:1298-1305 */
    old_e = new_e;
    for (int a = 0; a < K && !bad; ++a)                                 /* :1325-1331: alpha up, beta down, alpha<beta */
      for (int b = K - 1; b >= 0 && !bad; --b) {
        if (!(a < b)) continue;
        int m = 0;
        for (int64_t i = 0; i < n; ++i) {
          var[i] = -1;
          if (labels[i] == a || labels[i] == b) { var[i] = m; active[m++] = (int)i; }
        }
        if (m == 0) continue;
        int64_t narc = 0;
        for (int v = 0; v < m; ++v) narc += ptr[active[v] + 1] - ptr[active[v]];
        Flow f;
        flow_init(&f, m + 2, (int)(2 * (narc / 2 + 2 * (int64_t)m) + 16));
        const int S_ = m, T_ = m + 1;
        int64_t* e0 = (int64_t*)calloc(m, sizeof(int64_t));             /* cost of x=0 (alpha) */
        int64_t* e1 = (int64_t*)calloc(m, sizeof(int64_t));             /* cost of x=1 (beta)  */
        for (int v = 0; v < m; ++v) {
          const int i = active[v];
          e0[v] += unary[(int64_t)i * K + a];                           /* :367-378 */
          e1[v] += unary[(int64_t)i * K + b];
          for (int64_t q = ptr[i]; q < ptr[i + 1]; ++q) {
            const int j = adj[q];
            const int64_t wq = adjw[q];
            if (var[j] < 0) {                                           /* inactive neighbour: unary terms (:396-398) */
              e0[v] += wq * smooth[a * K + labels[j]];
              e1[v] += wq * smooth[b * K + labels[j]];
            } else if (j < i) {                                         /* active pair once (nSite < site, :399-406) */
              const int u = var[j];
              const int64_t A = wq * smooth[a * K + a], B = wq * smooth[a * K + b];
              const int64_t C = wq * smooth[b * K + a], D = wq * smooth[b * K + b];
              if (A + D > B + C) { bad = 1; break; }                    /* :306-307 */
              /* energy.h add_term2(x=v, y=u, A=E00, B=E01, C=E10, D=E11):
               * E(x,y) = A + (C-A) x + (D-C) y + (B+C-A-D) (1-x) y ; the constant A is dropped, the last term is an
               * arc x -> y cut when x is on the source side (0) and y on the sink side (1) */
              e1[v] += C - A;
              e1[u] += D - C;
              flow_arc(&f, v, u, B + C - A - D, 0);
            }
          }
          if (bad) break;
        }
        if (!bad) {
          for (int v = 0; v < m; ++v) {
            /* add_term1(x, E0, E1) -> tweights(source cap = E1, sink cap = E0); subtract the common part */
            const int64_t lo = e0[v] < e1[v] ? e0[v] : e1[v];
            if (e1[v] - lo > 0) flow_arc(&f, S_, v, e1[v] - lo, 0);
            if (e0[v] - lo > 0) flow_arc(&f, v, T_, e0[v] - lo, 0);
          }
          flow_max(&f, S_, T_);
          char* reach = (char*)malloc(m + 2);
          flow_sink_side(&f, T_, reach);
          for (int v = 0; v < m; ++v) labels[active[v]] = reach[v] ? b : a;   /* :1379-1383 */
          free(reach);
        }
        free(e0); free(e1);
        flow_free(&f);
      }
    new_e = int_energy(n, K, unary, smooth, E, s1, s2, w, labels);
    ++cycle;
  }
  if (cycles_out) *cycles_out = cycle;
  free(ptr); free(adj); free(adjw); free(var); free(active);
  return bad ? -1 : new_e;
}

/* ------------------------------------------------------------------------------------------------ posteriors */
/* stats_out = post[K] | obs[K,S] | obsobsT[K,S,S]; costs_out = pairwise_cost, pairwise_cost_normalize, unary_cost,
 * cost1 (NORMALISED by n like the reference); post_out [n,K] optional.  No max shift, like the reference. */
int oracle_posterior_stats(const double* X, const double* logprob, int64_t n, int S, int K, int64_t E, const int64_t* edges,
                           const double* w, const int* labels, double beta, int estimate_type, double* stats_out,
                           double* costs_out, double* post_out) {
  double* pp = (double*)calloc((size_t)n * K, sizeof(double));
  int* deg = (int*)calloc(n, sizeof(int));
  double pair = 0.0;
  for (int64_t e = 0; e < E; ++e) {
    const int64_t a = edges[2 * e], b = edges[2 * e + 1];
    const double ww = estimate_type == 3 ? w[e] : 1.0;
    for (int k = 0; k < K; ++k) {                 /* V[l_other, k] * w  (phylo_hmrf.py:426-434) */
      pp[a * K + k] += (labels[b] != k ? beta : 0.0) * ww;
      pp[b * K + k] += (labels[a] != k ? beta : 0.0) * ww;
    }
    deg[a]++; deg[b]++;
    if (labels[a] != labels[b]) pair += 2.0 * beta * ww;      /* counted from both endpoints (:438-447) */
  }
  for (int64_t i = 0; i < n; ++i)
    if (!deg[i])
      for (int k = 0; k < K; ++k) pp[i * K + k] = labels[i] != k ? beta : 0.0;   /* :421-423 */
  const int ns = K * (1 + S + S * S);
  memset(stats_out, 0, sizeof(double) * ns);
  double un = 0.0, pcn = 0.0;
  double* g = (double*)malloc(sizeof(double) * K);
  for (int64_t i = 0; i < n; ++i) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < K; ++k) { g[k] = exp(logprob[i * K + k] - pp[i * K + k]); s1 += g[k]; s2 += exp(-pp[i * K + k]); }
    const int li = labels[i];
    un -= logprob[i * K + li];
    pcn -= log(exp(-pp[i * K + li]) / s2 + 1e-16);
    for (int k = 0; k < K; ++k) {
      const double gk = g[k] / s1;
      if (post_out) post_out[i * K + k] = gk;
      stats_out[k] += gk;
      for (int a = 0; a < S; ++a) {
        stats_out[K + k * S + a] += gk * X[i * S + a];
        for (int b = 0; b < S; ++b) stats_out[K + K * S + (k * S + a) * S + b] += gk * X[i * S + a] * X[i * S + b];
      }
    }
  }
  costs_out[0] = pair / n;
  costs_out[1] = pcn / n;
  costs_out[2] = un / n;
  costs_out[3] = costs_out[1] + costs_out[2];
  free(pp); free(deg); free(g);
  return 0;
}

double oracle_energy(const double* logprob, int64_t n, int K, int64_t E, const int64_t* edges, const double* w,
                     const int* labels, double beta) {
  double e = 0.0;
  for (int64_t i = 0; i < n; ++i) e -= logprob[i * K + labels[i]];
  for (int64_t k = 0; k < E; ++k)
    if (labels[edges[2 * k]] != labels[edges[2 * k + 1]]) e += beta * w[k];
  return e;
}
