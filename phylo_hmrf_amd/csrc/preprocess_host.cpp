// libphmrf_host.so, pre-processing part (see include/phmrf_host.h): the reference's "nearest neighbour interpolation"
// of empty contact-map cells (utility.py:603-660) -- a SEQUENTIAL in-place raster scan (a filled cell feeds the
// windows of the cells after it), so it stays on the host; the Python double loop of the reference takes minutes per
// chromosome, this takes milliseconds.
#include "../../include/phmrf_host.h"

#include <algorithm>

extern "C" int phmrf_median_fill(double* mtx, int64_t n1, int64_t n2, int symmetric, double threshold,
                                 int64_t* n_low, int64_t* n_filled) {
  if (!mtx || n1 < 0 || n2 < 0) return PHMRF_HOST_ERR_INVALID;
  if (symmetric && n1 != n2) return PHMRF_HOST_ERR_INVALID;
  int64_t cnt1 = 0, cnt2 = 0;
  // rows 2 .. n1-2; columns i .. n2-2 (symmetric: upper triangle, mirrored write, utility.py:612-626) or
  // 2 .. n2-2 (general matrix, :642-656).  The scan starts at index 2, not 1: kept as in the reference.
  for (int64_t i = 2; i < n1 - 1; ++i) {
    for (int64_t j = symmetric ? i : 2; j < n2 - 1; ++j) {
      if (!(mtx[i * n2 + j] < threshold)) continue;
      ++cnt1;
      double w[8];
      int k = 0;
      for (int di = -1; di <= 1; ++di)
        for (int dj = -1; dj <= 1; ++dj)
          if (di || dj) w[k++] = mtx[(i + di) * n2 + (j + dj)];
      std::sort(w, w + 8);
      const double m1 = (w[3] + w[4]) / 2.0;          // np.median of 8 values: mean of the two middle ones
      if (m1 > threshold) {
        ++cnt2;
        mtx[i * n2 + j] = m1;
        if (symmetric) mtx[j * n2 + i] = m1;
      }
    }
  }
  if (n_low) *n_low = cnt1;
  if (n_filled) *n_filled = cnt2;
  return PHMRF_HOST_OK;
}
