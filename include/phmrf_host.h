/* phmrf_host.h -- host-only helper of the EM loop (libphmrf_host.so, plain C ABI, no HIP dependency).
 *
 * The M-step of Phylo-HMRF fits, per hidden state, the 3B+2 Ornstein-Uhlenbeck parameters of the species tree to the
 * sufficient statistics the GPU E-step produced (reference: phylo_hmrf.py:1500-1528 `_do_mstep`, :1327-1403
 * `_ou_optimize2`, objective :1038-1138 `_ou_lik_varied_constraint`).  It is O(K) work independent of the number of
 * nodes, so it stays on the host (SciPy SLSQP, as in the reference); but once the E-step takes milliseconds the K x ~200
 * objective evaluations in NumPy cap the EM rate.  This entry point evaluates the objective and its analytic gradient
 * natively; phylo_hmrf_amd/mstep.py binds it with ctypes and keeps the NumPy version for the ill-conditioned fall-back
 * and as the test oracle of this function.
 */
#ifndef PHMRF_HOST_H
#define PHMRF_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHMRF_HOST_API __attribute__((visibility("default")))

#define PHMRF_HOST_OK 0
#define PHMRF_HOST_ERR_INVALID 1   /* bad argument                                                            */
#define PHMRF_HOST_ILL_CONDITIONED 2 /* V stayed ill-conditioned after 10 x min_covar (phylo_hmrf.py:1108-1133):
                                        the caller takes the pseudo-inverse path                               */

/* Species-tree tables (phylo_hmrf.py:715-919), all indices 0-based:
 *   N nodes, parent[N] (-1 for a root), order[n_order] = the non-root nodes root-first,
 *   leaf_vec[S] = leaf node of feature column s, n_pairs = S(S-1)/2 leaf pairs (pair_a < pair_b, indices into leaf_vec)
 *   with their most recent common ancestor pair_anc and the 0/1 branch-path matrix A2[n_pairs, N].              */
typedef struct phmrf_tree_tables {
  int32_t N, S, n_pairs, n_order;
  const int32_t* parent;
  const int32_t* order;
  const int32_t* leaf_vec;
  const int32_t* pair_a;
  const int32_t* pair_b;
  const int32_t* pair_anc;
  const double* A2;
} phmrf_tree_tables;

/* f(p) = post * log(det V + 1e-16) / n + tr(V^-1 S_w) / n + reg * |p|^2        (phylo_hmrf.py:1093-1113)
 *   V = cov_OU(p) + min_covar * I (+ min_covar * I while ill-conditioned, at most 10 times),
 *   S_w = oo - obs mu^T - mu obs^T + post * mu mu^T,  mu = OU leaf means,
 *   p = [ root variance | beta_1..B | lambda_1..B | theta_0..B ],  B = N - 1.
 * grad (3B+2 doubles) may be NULL; V_out (S*S) and mu_out (S) may be NULL.                                     */
PHMRF_HOST_API int phmrf_ou_objective(const phmrf_tree_tables* tree, const double* p, double post, const double* obs,
                                      const double* oo, double n_samples, double reg, double min_covar, double* f_out,
                                      double* grad, double* V_out, double* mu_out);

/* One state's SLSQP run over the objective above (phylo_hmrf.py:1383-1384: scipy.optimize.minimize(method='SLSQP',
 * bounds lower <= p <= upper)), driven natively: `slsqp_entry` is the address of SciPy's own Fortran SLSQP core, taken by
 * the caller from the f2py object (scipy.optimize._slsqp.slsqp._cpointer; SciPy 1.15 calling sequence, 32-bit INTEGERs).
 * Same routine and calling sequence as scipy's `_minimize_slsqp`, hence the same iterates; x_out = the final point,
 * *mode_out = SLSQP's exit mode (0 = converged), *n_eval_out = objective evaluations.  Returns
 * PHMRF_HOST_ILL_CONDITIONED as soon as an evaluation meets an ill-conditioned V: the caller repeats the state with its
 * Python loop, which takes the reference's pseudo-inverse path there.                                            */
PHMRF_HOST_API int phmrf_ou_slsqp(const phmrf_tree_tables* tree, void* slsqp_entry, double post, const double* obs,
                                  const double* oo, double n_samples, double reg, double min_covar, const double* x0,
                                  double lower, double upper, double acc, int maxiter, double* x_out, int* mode_out,
                                  int* n_eval_out);

/* The M-step of one EM iteration, `_do_mstep` (phylo_hmrf.py:1500-1528): K independent `_ou_optimize2` runs
 * (:1327-1403) on `n_threads` host threads (the states share nothing: the result does not depend on the thread count).
 * Per state c: phmrf_ou_slsqp from guesses[c, r, :] for r = 0 .. n_guesses-1 until the result passes `_check_params`
 * (:1405-1425: beta, lambda in [0, 100], theta in [-100, 100]); if none does, the state keeps init_params[c, :]
 * (:1346-1349); params_out[c, :] = the chosen point, lik_out[c] = the objective there, mean_out[c, :] / V_out[c, :, :] =
 * its OU leaf means and covariance (without the EM-time + min_covar I of :1524).  status_out[c] = PHMRF_HOST_OK, or
 * PHMRF_HOST_ILL_CONDITIONED when an evaluation of that state met an ill-conditioned covariance: the caller then repeats
 * THAT state with its Python loop (pseudo-inverse path); its outputs are undefined.
 * Arrays are C-order float64: post[K], obs[K,S], oo[K,S,S], guesses[K,n_guesses,3B+2], init_params[K,3B+2].           */
PHMRF_HOST_API int phmrf_ou_mstep(const phmrf_tree_tables* tree, void* slsqp_entry, int K, const double* post,
                                  const double* obs, const double* oo, double n_samples, double reg, double min_covar,
                                  const double* guesses, int n_guesses, const double* init_params, double lower,
                                  double upper, double acc, int maxiter, int n_threads, double* params_out,
                                  double* lik_out, double* mean_out, double* V_out, int* status_out);

/* Pre-processing: the reference's median fill of empty contact-map cells, `near_interpolation1`
 * (symmetric != 0: square matrix, upper triangle scanned, value mirrored; utility.py:603-631) and
 * `near_interpolation1a` (symmetric == 0: general matrix; utility.py:633-660).  mtx: C-order float64 [n1,n2], updated
 * IN PLACE in raster order exactly like the reference's loops: a cell below `threshold` (THRESH1 = 1e-5, :47) takes the
 * median of its 8 neighbours when that median is above the threshold.  n_low / n_filled (may be NULL) return the
 * reference's cnt1 / cnt2.                                                                                       */
PHMRF_HOST_API int phmrf_median_fill(double* mtx, int64_t n1, int64_t n2, int symmetric, double threshold,
                                     int64_t* n_low, int64_t* n_filled);

/* Pre-processing, filter_mode 1: the reference's `denoise_bilateral(img, sigma_color, sigma_spatial, multichannel=False)`
 * of one channel (utility.py:1575-1582; scikit-image's `_denoise_cy._denoise_bilateral`, restated from the published
 * algorithm -- scikit-image is not available here: PARITY UNPINNED).  img, out: C-order float64 [rows, cols], distinct
 * buffers; win_size <= 0: max(5, 2 ceil(3 sigma_spatial) + 1); bins <= 0: 10000; mode 'constant' with cval 0.  A
 * negative pixel (skimage raises ValueError) or a non-positive sigma returns PHMRF_HOST_ERR_INVALID.              */
PHMRF_HOST_API int phmrf_bilateral(const double* img, int64_t rows, int64_t cols, double sigma_color,
                                   double sigma_spatial, int win_size, int bins, double* out);

PHMRF_HOST_API int phmrf_host_version(void);

#ifdef __cplusplus
}
#endif
#endif
