// libphmrf_host.so: the M-step objective and its analytic gradient (see include/phmrf_host.h).
// Plain C++ (g++), no HIP.  Mirrors phylo_hmrf_amd/mstep.py OUObjective.value_and_grad term by term; that NumPy version
// is the oracle of the unit test (tests/test_mstep.py) and itself pinned on the reference's objective values
// (tests/golden/mstep_objective.npz).
#include "../../include/phmrf_host.h"

#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace {

constexpr int MAXS = 32;
constexpr double SMALL_EPS = 1e-16;   // phylo_hmrf.py:49

// eigenvalues of a symmetric S x S matrix by cyclic Jacobi rotations (S is the number of species: tiny)
void sym_eigenvalues(const double* A, int S, double* ev) {
  double a[MAXS * MAXS];
  std::memcpy(a, A, sizeof(double) * S * S);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < S; ++i)
      for (int j = 0; j < S; ++j) (i == j ? diag : off) += a[i * S + j] * a[i * S + j];
    if (off <= 1e-30 * (diag > 0 ? diag : 1.0)) break;
    for (int p = 0; p < S; ++p)
      for (int q = p + 1; q < S; ++q) {
        const double apq = a[p * S + q];
        if (apq == 0.0) continue;
        const double theta = (a[q * S + q] - a[p * S + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < S; ++k) {   // columns p, q
          const double akp = a[k * S + p], akq = a[k * S + q];
          a[k * S + p] = c * akp - s * akq;
          a[k * S + q] = s * akp + c * akq;
        }
        for (int k = 0; k < S; ++k) {   // rows p, q
          const double apk = a[p * S + k], aqk = a[q * S + k];
          a[p * S + k] = c * apk - s * aqk;
          a[q * S + k] = s * apk + c * aqk;
        }
      }
  }
  for (int i = 0; i < S; ++i) ev[i] = a[i * S + i];
}

// np.linalg.cond(V) < 1/eps for the symmetric V (phylo_hmrf.py:1109)
bool well_conditioned(const double* V, int S) {
  double ev[MAXS];
  sym_eigenvalues(V, S, ev);
  double lo = std::numeric_limits<double>::infinity(), hi = 0.0;
  for (int i = 0; i < S; ++i) {
    const double v = std::fabs(ev[i]);
    if (!std::isfinite(v)) return false;
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
  return lo > 0 && hi / lo < 1.0 / std::numeric_limits<double>::epsilon();
}

// inverse and determinant by LU with partial pivoting (what LAPACK getrf/getri do for numpy)
bool invert(const double* V, int S, double* Vi, double* det) {
  double a[MAXS * MAXS];
  std::memcpy(a, V, sizeof(double) * S * S);
  for (int i = 0; i < S; ++i)
    for (int j = 0; j < S; ++j) Vi[i * S + j] = i == j ? 1.0 : 0.0;
  double d = 1.0;
  for (int c = 0; c < S; ++c) {
    int piv = c;
    for (int r = c + 1; r < S; ++r)
      if (std::fabs(a[r * S + c]) > std::fabs(a[piv * S + c])) piv = r;
    if (a[piv * S + c] == 0.0) return false;
    if (piv != c) {
      for (int k = 0; k < S; ++k) {
        std::swap(a[piv * S + k], a[c * S + k]);
        std::swap(Vi[piv * S + k], Vi[c * S + k]);
      }
      d = -d;
    }
    const double pv = a[c * S + c];
    d *= pv;
    for (int k = 0; k < S; ++k) {
      a[c * S + k] /= pv;
      Vi[c * S + k] /= pv;
    }
    for (int r = 0; r < S; ++r) {
      if (r == c) continue;
      const double f = a[r * S + c];
      if (f == 0.0) continue;
      for (int k = 0; k < S; ++k) {
        a[r * S + k] -= f * a[c * S + k];
        Vi[r * S + k] -= f * Vi[c * S + k];
      }
    }
  }
  *det = d;
  return true;
}

}  // namespace

extern "C" int phmrf_host_version(void) { return 1; }

extern "C" int phmrf_ou_objective(const phmrf_tree_tables* t, const double* p, double post, const double* obs,
                                  const double* oo, double n_samples, double reg, double min_covar, double* f_out,
                                  double* grad, double* V_out, double* mu_out) {
  if (!t || !p || !obs || !oo || !f_out) return PHMRF_HOST_ERR_INVALID;
  const int N = t->N, S = t->S, P = t->n_pairs, B = N - 1;
  if (N < 2 || S < 1 || S > MAXS || N > 4 * MAXS) return PHMRF_HOST_ERR_INVALID;
  const int NP = 3 * B + 2;
  const double* beta = p + 1;
  const double* lam = p + 1 + B;
  const double* theta = p + 1 + 2 * B;   // theta_0 .. theta_B (N entries)
  double e[4 * MAXS], ratio[4 * MAXS], mean[4 * MAXS], var[4 * MAXS];
  // ---- node moments (phylo_hmrf.py:999-1015) ----------------------------------------------------------------------
  e[0] = 0.0;
  ratio[0] = 0.0;
  for (int i = 1; i < N; ++i) {
    e[i] = std::exp(-beta[i - 1]);
    ratio[i] = beta[i - 1] > 1e-07 ? lam[i - 1] / (2.0 * beta[i - 1]) : 0.0;
  }
  for (int i = 0; i < N; ++i) {
    mean[i] = 0.0;
    var[i] = 0.0;
    if (t->parent[i] < 0) {
      mean[i] = theta[i];
      var[i] = p[0];
    }
  }
  for (int q = 0; q < t->n_order; ++q) {
    const int i = t->order[q], pa = t->parent[i];
    mean[i] = mean[pa] * e[i] + theta[i] * (1.0 - e[i]);
    var[i] = ratio[i] * (1.0 - e[i] * e[i]) + var[pa] * e[i] * e[i];
  }
  // ---- covariance of the leaves (:1018-1031) ----------------------------------------------------------------------
  std::vector<double> ex(P), s2(P);
  double V[MAXS * MAXS];
  for (int i = 0; i < S * S; ++i) V[i] = 0.0;
  for (int k = 0; k < P; ++k) {
    double s1 = 0.0;
    for (int j = 1; j < N; ++j) s1 += t->A2[(size_t)k * N + j] * beta[j - 1];
    ex[k] = std::exp(-s1);
    s2[k] = var[t->pair_anc[k]] * ex[k];
    V[t->pair_a[k] * S + t->pair_b[k]] = s2[k];
    V[t->pair_b[k] * S + t->pair_a[k]] = s2[k];
  }
  double mu[MAXS];
  for (int s = 0; s < S; ++s) {
    V[s * S + s] = var[t->leaf_vec[s]] + min_covar;   // :1090
    mu[s] = mean[t->leaf_vec[s]];
  }
  // ---- ill-conditioned V: add min_covar * I up to 10 times (:1108-1133) -------------------------------------------
  bool well = well_conditioned(V, S);
  for (int cnt = 0; !well && cnt < 10; ++cnt) {
    for (int s = 0; s < S; ++s) V[s * S + s] += min_covar;
    well = well_conditioned(V, S);
  }
  if (!well) return PHMRF_HOST_ILL_CONDITIONED;
  double Vi[MAXS * MAXS], detV = 0.0;
  if (!invert(V, S, Vi, &detV)) return PHMRF_HOST_ILL_CONDITIONED;
  // ---- S_w and the value (:1093-1113) -----------------------------------------------------------------------------
  double Sw[MAXS * MAXS];
  for (int a = 0; a < S; ++a)
    for (int b = 0; b < S; ++b) Sw[a * S + b] = oo[a * S + b] - obs[a] * mu[b] - obs[b] * mu[a] + mu[a] * mu[b] * post;
  double tr = 0.0, pp = 0.0;
  for (int i = 0; i < S * S; ++i) tr += Vi[i] * Sw[i];
  for (int i = 0; i < NP; ++i) pp += p[i] * p[i];
  *f_out = post * std::log(detV + SMALL_EPS) / n_samples + tr / n_samples + reg * pp;
  if (V_out) std::memcpy(V_out, V, sizeof(double) * S * S);
  if (mu_out) std::memcpy(mu_out, mu, sizeof(double) * S);
  if (!grad) return PHMRF_HOST_OK;
  // ---- reverse sweep ----------------------------------------------------------------------------------------------
  double G[MAXS * MAXS], T1[MAXS * MAXS];
  for (int a = 0; a < S; ++a)       // T1 = Vi Sw
    for (int b = 0; b < S; ++b) {
      double acc = 0.0;
      for (int k = 0; k < S; ++k) acc += Vi[a * S + k] * Sw[k * S + b];
      T1[a * S + b] = acc;
    }
  const double dscale = post * (detV / (detV + SMALL_EPS));
  for (int a = 0; a < S; ++a)       // G = (dscale Vi - Vi Sw Vi) / n
    for (int b = 0; b < S; ++b) {
      double acc = 0.0;
      for (int k = 0; k < S; ++k) acc += T1[a * S + k] * Vi[k * S + b];
      G[a * S + b] = (dscale * Vi[a * S + b] - acc) / n_samples;
    }
  double gmu[MAXS];
  for (int a = 0; a < S; ++a) {
    double acc = 0.0;
    for (int k = 0; k < S; ++k) acc += Vi[a * S + k] * (post * mu[k] - obs[k]);
    gmu[a] = 2.0 * acc / n_samples;
  }
  double gvar[4 * MAXS], gmean[4 * MAXS], gbeta_full[4 * MAXS], ge[4 * MAXS], gratio[4 * MAXS], gtheta[4 * MAXS];
  for (int i = 0; i < N; ++i) gvar[i] = gmean[i] = gbeta_full[i] = ge[i] = gratio[i] = gtheta[i] = 0.0;
  for (int s = 0; s < S; ++s) {
    gvar[t->leaf_vec[s]] += G[s * S + s];
    gmean[t->leaf_vec[s]] += gmu[s];
  }
  for (int k = 0; k < P; ++k) {
    const double gpair = 2.0 * G[t->pair_a[k] * S + t->pair_b[k]];   // V_ab appears twice
    gvar[t->pair_anc[k]] += gpair * ex[k];
    const double c = -gpair * s2[k];
    for (int j = 0; j < N; ++j) gbeta_full[j] += t->A2[(size_t)k * N + j] * c;
  }
  for (int q = t->n_order - 1; q >= 0; --q) {
    const int i = t->order[q], pa = t->parent[i];
    gvar[pa] += gvar[i] * e[i] * e[i];
    gratio[i] += gvar[i] * (1.0 - e[i] * e[i]);
    ge[i] += gvar[i] * (2.0 * e[i] * (var[pa] - ratio[i]));
    gmean[pa] += gmean[i] * e[i];
    gtheta[i] += gmean[i] * (1.0 - e[i]);
    ge[i] += gmean[i] * (mean[pa] - theta[i]);
  }
  double groot = 0.0;
  for (int i = 0; i < N; ++i)
    if (t->parent[i] < 0) {
      gtheta[i] += gmean[i];
      groot += gvar[i];
    }
  grad[0] = groot + 2.0 * reg * p[0];
  for (int i = 1; i < N; ++i) {
    const double b = beta[i - 1];
    double gb = gbeta_full[i] - ge[i] * e[i];
    double gl = 0.0;
    if (b > 1e-07) {
      gl = gratio[i] / (2.0 * b);
      gb += -gratio[i] * lam[i - 1] / (2.0 * b * b);
    }
    grad[i] = gb + 2.0 * reg * p[i];
    grad[B + i] = gl + 2.0 * reg * p[B + i];
  }
  for (int i = 0; i < N; ++i) grad[1 + 2 * B + i] = gtheta[i] + 2.0 * reg * p[1 + 2 * B + i];
  return PHMRF_HOST_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// One state's SLSQP run, natively: the loop of scipy 1.15 `_minimize_slsqp` (bounds only, no constraints) around
// SciPy's own Fortran core -- whose entry point the caller takes from the f2py object
// (scipy.optimize._slsqp.slsqp._cpointer) -- with the objective above evaluated in place.  Same routine, same calling
// sequence, same iterates as the Python loop (phylo_hmrf_amd/mstep.py `_slsqp_lean`); what goes away is ~20 us of
// interpreter and f2py argument handling per objective evaluation (~300 evaluations per state and EM iteration).
typedef void (*slsqp_fn)(int* m, int* meq, int* la, int* n, double* x, double* xl, double* xu, double* f, double* c,
                         double* g, double* a, double* acc, int* iter, int* mode, double* w, int* l_w, int* jw, int* l_jw,
                         double* alpha, double* f0, double* gs, double* h1, double* h2, double* h3, double* h4, double* t,
                         double* t0, double* tol, int* iexact, int* incons, int* ireset, int* itermx, int* line, int* n1,
                         int* n2, int* n3);

extern "C" int phmrf_ou_slsqp(const phmrf_tree_tables* tree, void* slsqp_entry, double post, const double* obs,
                              const double* oo, double n_samples, double reg, double min_covar, const double* x0,
                              double lower, double upper, double acc, int maxiter, double* x_out, int* mode_out,
                              int* n_eval_out) {
  if (!tree || !slsqp_entry || !obs || !oo || !x0 || !x_out || !mode_out) return PHMRF_HOST_ERR_INVALID;
  const int N = tree->N;
  if (N < 2 || N > 4 * MAXS) return PHMRF_HOST_ERR_INVALID;
  int n = 3 * (N - 1) + 2;
  constexpr int MAXN = 3 * (4 * MAXS) + 2;
  const int n1 = n + 1;
  int m = 0, meq = 0, la = 1;
  const int mineq = m - meq + n1 + n1;
  int l_w = (3 * n1 + m) * (n1 + 1) + (n1 - meq + 1) * (mineq + 2) + 2 * mineq + (n1 + mineq) * (n1 - meq) + 2 * meq + n1 +
            ((n + 1) * n) / 2 + 2 * m + 3 * n + 3 * n1 + 1;
  int l_jw = mineq;
  std::vector<double> w((size_t)l_w, 0.0);
  std::vector<int> jw((size_t)l_jw, 0);
  double x[MAXN], xl[MAXN], xu[MAXN], xc[MAXN], g[MAXN + 1], gr[MAXN], a[MAXN + 1], c[1] = {0.0};
  for (int i = 0; i < n; ++i) {
    xl[i] = lower;
    xu[i] = upper;
    x[i] = x0[i] < lower ? lower : (x0[i] > upper ? upper : x0[i]);
  }
  for (int i = 0; i <= n; ++i) a[i] = 0.0;
  int mode = 0, iter = maxiter;
  double accv = acc;
  double sf[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // alpha f0 gs h1 h2 h3 h4 t t0 tol
  int si[8] = {0, 0, 0, 0, 0, 0, 0, 0};                // iexact incons ireset itermx line n1 n2 n3
  int n_eval = 0;
  auto evaluate = [&](bool want_grad, double* f_out) -> int {     // (gh11403: SLSQP may leave the box by an ulp or two)
    for (int i = 0; i < n; ++i) xc[i] = x[i] < lower ? lower : (x[i] > upper ? upper : x[i]);
    ++n_eval;
    return phmrf_ou_objective(tree, xc, post, obs, oo, n_samples, reg, min_covar, f_out, want_grad ? gr : nullptr, nullptr, nullptr);
  };
  double fx = 0.0;
  int st = evaluate(true, &fx);
  if (st != PHMRF_HOST_OK) return st;
  for (int i = 0; i < n; ++i) g[i] = gr[i];
  g[n] = 0.0;
  slsqp_fn core = reinterpret_cast<slsqp_fn>(slsqp_entry);
  for (;;) {
    core(&m, &meq, &la, &n, x, xl, xu, &fx, c, g, a, &accv, &iter, &mode, w.data(), &l_w, jw.data(), &l_jw, &sf[0], &sf[1],
         &sf[2], &sf[3], &sf[4], &sf[5], &sf[6], &sf[7], &sf[8], &sf[9], &si[0], &si[1], &si[2], &si[3], &si[4], &si[5],
         &si[6], &si[7]);
    if (mode == 1) {                       // function evaluation
      st = evaluate(false, &fx);
      if (st != PHMRF_HOST_OK) return st;
    } else if (mode == -1) {               // gradient evaluation
      double dummy;
      st = evaluate(true, &dummy);
      if (st != PHMRF_HOST_OK) return st;
      for (int i = 0; i < n; ++i) g[i] = gr[i];
      g[n] = 0.0;
    }
    if (mode != 1 && mode != -1) break;
  }
  for (int i = 0; i < n; ++i) x_out[i] = x[i];
  *mode_out = mode;
  if (n_eval_out) *n_eval_out = n_eval;
  return PHMRF_HOST_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// The whole M-step of one EM iteration: `_do_mstep` (phylo_hmrf.py:1500-1528) = K independent `_ou_optimize2` runs
// (:1327-1403), one host thread per state.  Per state, exactly what phylo_hmrf_amd/mstep.py `_solve_state` does around
// phmrf_ou_slsqp: the start points in order (the first is used, the others are the reference's retries, :1332-1342), the
// box test of the result (`_check_params`, :1405-1425), the fall-back to the state's initial parameters (:1346-1349), and
// the objective at the chosen point with its mean and covariance.  A state that meets an ill-conditioned covariance
// anywhere is reported with status PHMRF_HOST_ILL_CONDITIONED and left to the caller (its Python loop takes the
// reference's pseudo-inverse path).  The states share nothing, so the result does not depend on the number of threads.
#include <thread>

namespace {

// `_check_params`: 1 inside the box, <= -1 outside (beta, lambda in [0, 100], theta in [-100, 100]; NaN counts as outside)
int check_box(const double* p, int N) {
  const int B = N - 1;
  for (int i = 0; i < 2 * B; ++i)
    if (!(p[1 + i] >= 0.0 && p[1 + i] <= 100.0)) return -1;
  for (int i = 0; i < N; ++i)
    if (!(p[1 + 2 * B + i] >= -100.0 && p[1 + 2 * B + i] <= 100.0)) return -1;
  return 1;
}

}  // namespace

extern "C" int phmrf_ou_mstep(const phmrf_tree_tables* tree, void* slsqp_entry, int K, const double* post, const double* obs,
                              const double* oo, double n_samples, double reg, double min_covar, const double* guesses,
                              int n_guesses, const double* init_params, double lower, double upper, double acc, int maxiter,
                              int n_threads, double* params_out, double* lik_out, double* mean_out, double* V_out,
                              int* status_out) {
  if (!tree || !slsqp_entry || K <= 0 || !post || !obs || !oo || !guesses || n_guesses <= 0 || !init_params || !params_out ||
      !lik_out || !mean_out || !V_out || !status_out)
    return PHMRF_HOST_ERR_INVALID;
  const int N = tree->N, S = tree->S;
  if (N < 2 || N > 4 * MAXS || S < 1 || S > MAXS) return PHMRF_HOST_ERR_INVALID;
  const int P = 3 * (N - 1) + 2;
  auto one_state = [&](int c) {
    const double* ob = obs + (size_t)c * S;
    const double* o2 = oo + (size_t)c * S * S;
    double* x = params_out + (size_t)c * P;
    bool ok = false;
    status_out[c] = PHMRF_HOST_OK;
    for (int r = 0; r < n_guesses && !ok; ++r) {
      int mode = 0, n_eval = 0;
      const int st = phmrf_ou_slsqp(tree, slsqp_entry, post[c], ob, o2, n_samples, reg, min_covar,
                                    guesses + ((size_t)c * n_guesses + r) * P, lower, upper, acc, maxiter, x, &mode, &n_eval);
      if (st != PHMRF_HOST_OK) {
        status_out[c] = st;
        return;
      }
      ok = check_box(x, N) > 0;
    }
    if (!ok)
      for (int i = 0; i < P; ++i) x[i] = init_params[(size_t)c * P + i];
    const int st = phmrf_ou_objective(tree, x, post[c], ob, o2, n_samples, reg, min_covar, lik_out + c, nullptr,
                                      V_out + (size_t)c * S * S, mean_out + (size_t)c * S);
    if (st != PHMRF_HOST_OK) status_out[c] = st;
  };
  int nt = n_threads < 1 ? 1 : (n_threads > K ? K : n_threads);
  if (nt == 1) {
    for (int c = 0; c < K; ++c) one_state(c);
    return PHMRF_HOST_OK;
  }
  std::vector<std::thread> th;
  th.reserve((size_t)nt);
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t]() {
      for (int c = t; c < K; c += nt) one_state(c);
    });
  for (auto& t : th) t.join();
  return PHMRF_HOST_OK;
}
