// gfx950 kernel of the EM initialisation: one Lloyd step of k-means on the device-resident observations.
//
// The reference initialises the hidden states with sklearn's MiniBatchKMeans (phylo_hmrf.py:234-238, batch 2000,
// n_init 10) -- on a whole-genome run (88.8 M bin-pairs) that is minutes of single-core work in front of an EM
// iteration that now takes half a second.  This step keeps X where it already is:
//   label_i = argmin_k |x_i - c_k|^2   (lowest k wins ties),   sums[k] += x_i,  counts[k] += 1,  inertia += min_k |.|^2
// One thread per node, centres via wave-uniform scalar loads (K*S floats), per-workgroup partial sums in LDS (f32 within
// a tile of 256 nodes), f64 atomics across workgroups -- the accumulation scheme of the posterior kernel.
// HBM-bound: 4S bytes per node in, 1 byte out.

#include "common.h"

namespace phmrf {
namespace {

// OUTER: also the per-cluster second moments sum_i x_i x_i^T (K*S*S values behind the others): with the sums and the
// counts they are all the per-cluster OU fit of the initialisation (phylo_hmrf.py:1246-1325 works on a cluster's mean
// and X^T X / n only) and the global covariance (:258) need -- no host pass over the observations.
template <int S, bool OUTER>
__global__ __launch_bounds__(256) void kmeans_step_kernel(const float* __restrict__ X, int64_t n, int K,
                                                          const float* __restrict__ centers, uint8_t* __restrict__ labels,
                                                          double* __restrict__ acc /* [K*S sums | K counts | inertia | K*S*S] */,
                                                          int64_t own0, int64_t own1) {
  // (every node is labelled; only the nodes [own0, own1) -- all of them unless the block is a row tile -- enter the sums)
  extern __shared__ float part[];                 // [K*S sums | K counts | 1 inertia | K*S*S second moments]
  const int NP = K * S + K + 1 + (OUTER ? K * S * S : 0);
  for (int q = threadIdx.x; q < NP; q += blockDim.x) part[q] = 0.f;
  __syncthreads();
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += (int64_t)gridDim.x * blockDim.x) {
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
      float best = 3.0e38f;
      int bk = 0;
      for (int k = 0; k < K; ++k) {
        const float* c = centers + k * S;          // wave-uniform: scalar loads
        float d = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float t = x[s] - c[s];
          d = fmaf(t, t, d);
        }
        if (d < best) { best = d; bk = k; }
      }
      if (labels) labels[i] = (uint8_t)bk;
      if (i >= own0 && i < own1) {
#pragma unroll
      for (int s = 0; s < S; ++s) atomicAdd(part + bk * S + s, x[s]);     // LDS float atomics
      atomicAdd(part + K * S + bk, 1.f);
      atomicAdd(part + K * S + K, best);
      if (OUTER) {
        float* oo = part + K * S + K + 1 + bk * S * S;
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int t = 0; t < S; ++t) atomicAdd(oo + s * S + t, x[s] * x[t]);
      }
      }
    }
    __syncthreads();
    // flush the tile's partial sums (f32 over at most 256 nodes) into the f64 accumulators
    for (int q = threadIdx.x; q < NP; q += blockDim.x) {
      const float v = part[q];
      if (v != 0.f) atomicAdd(acc + q, (double)v);
      part[q] = 0.f;
    }
    __syncthreads();
  }
}

}  // namespace

int launch_kmeans_step(const phmrf_block* b, const float* centers_dev, bool write_labels, double* acc_dev, bool outer) {
  const int K = b->K;
  if (outer && b->S > 8) return fail(PHMRF_ERR_UNSUPPORTED, "second moments: S must be in [1,8]");
  const size_t lds = (size_t)(K * b->S + K + 1 + (outer ? K * b->S * b->S : 0)) * sizeof(float);
  int64_t g64 = (b->n + 255) / 256;
  const int grid = (int)(g64 > 2048 ? 2048 : g64);
  const int64_t own0 = b->own1 >= 0 ? b->own0 : 0, own1 = b->own1 >= 0 ? b->own1 : b->n;
  switch (b->S) {
#define PHMRF_CASE(S_)                                                                                               \
  case S_:                                                                                                           \
    if (outer && S_ <= 8)                                                                                            \
      hipLaunchKernelGGL((kmeans_step_kernel<(S_ <= 8 ? S_ : 1), true>), dim3(grid), dim3(256), lds, b->stream, b->X, b->n, \
                         K, centers_dev, write_labels ? b->labels : nullptr, acc_dev, own0, own1);                   \
    else                                                                                                             \
      hipLaunchKernelGGL((kmeans_step_kernel<S_, false>), dim3(grid), dim3(256), lds, b->stream, b->X, b->n, K,      \
                         centers_dev, write_labels ? b->labels : nullptr, acc_dev, own0, own1);                      \
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
