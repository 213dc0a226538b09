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

// ---------------------------------------------------------------------------------------------------------------
// filter_mode 1: the reference calls skimage.restoration.denoise_bilateral(img, sigma_color, sigma_spatial,
// multichannel=False) channel by channel (utility.py:1575-1582).  scikit-image is neither in this image nor under
// /root/reference, so this restates the algorithm of its `_denoise_cy._denoise_bilateral` as published (0.13 / 0.14, the
// versions of the reference's Python-2 era) for ONE channel -- PARITY UNPINNED:
//   window  win = max(5, 2 ceil(3 sigma_spatial) + 1), centred; pixels outside the image count as 0 (mode 'constant',
//           cval 0) and DO take part with their weight;
//   spatial weight  exp(-0.5 (d / sigma_spatial)^2) of the Euclidean pixel distance d;
//   colour weight   a table of `bins` = 10000 entries, entry b = exp(-0.5 (b max / bins / sigma_color)^2), looked up at
//           bin min(floor(|centre - value| bins / max), bins - 1), max = the image maximum;
//   out = sum(value * w) / sum(w).  An image with min == max is returned unchanged; a negative value is an error there
//   (ValueError) and here.  Rows are independent: a few host threads share them.
#include <cmath>
#include <thread>
#include <vector>

extern "C" int phmrf_bilateral(const double* img, int64_t rows, int64_t cols, double sigma_color, double sigma_spatial,
                               int win_size, int bins, double* out) {
  if (!img || !out || rows <= 0 || cols <= 0 || !(sigma_color > 0.0) || !(sigma_spatial > 0.0)) return PHMRF_HOST_ERR_INVALID;
  if (bins <= 0) bins = 10000;
  if (win_size <= 0) win_size = std::max(5, 2 * (int)std::ceil(3.0 * sigma_spatial) + 1);
  if (win_size % 2 == 0) return PHMRF_HOST_ERR_INVALID;
  double mn = img[0], mx = img[0];
  for (int64_t q = 1; q < rows * cols; ++q) {
    mn = std::min(mn, img[q]);
    mx = std::max(mx, img[q]);
  }
  if (mn == mx) {
    std::copy(img, img + rows * cols, out);
    return PHMRF_HOST_OK;
  }
  if (mn < 0.0 || mx == 0.0) return PHMRF_HOST_ERR_INVALID;
  const int ext = (win_size - 1) / 2;
  std::vector<double> color_lut((size_t)bins), range_lut((size_t)win_size * win_size);
  for (int b = 0; b < bins; ++b) {
    const double v = (double)b * mx / (double)bins / sigma_color;
    color_lut[(size_t)b] = std::exp(-0.5 * v * v);
  }
  for (int kr = 0; kr < win_size; ++kr)
    for (int kc = 0; kc < win_size; ++kc) {
      const double d = std::sqrt((double)((kr - ext) * (kr - ext) + (kc - ext) * (kc - ext))) / sigma_spatial;
      range_lut[(size_t)kr * win_size + kc] = std::exp(-0.5 * d * d);
    }
  const double dist_scale = (double)bins / mx;
  auto run_rows = [&](int64_t r0, int64_t r1) {
    for (int64_t r = r0; r < r1; ++r)
      for (int64_t c = 0; c < cols; ++c) {
        const double centre = img[r * cols + c];
        double total = 0.0, weight_sum = 0.0;
        for (int wr = -ext; wr <= ext; ++wr) {
          const int64_t rr = r + wr;
          const bool row_in = rr >= 0 && rr < rows;
          const double* rl = range_lut.data() + (size_t)(wr + ext) * win_size;
          for (int wc = -ext; wc <= ext; ++wc) {
            const int64_t cc = c + wc;
            const double value = (row_in && cc >= 0 && cc < cols) ? img[rr * cols + cc] : 0.0;
            const double dist = std::fabs(centre - value);          // sqrt(t * t) of the one channel
            int64_t bin = (int64_t)(dist * dist_scale);
            if (bin > bins - 1) bin = bins - 1;
            const double w = rl[wc + ext] * color_lut[(size_t)bin];
            total += value * w;
            weight_sum += w;
          }
        }
        out[r * cols + c] = total / weight_sum;
      }
  };
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt == 0 ? 1 : (nt > 16 ? 16 : nt);
  if ((int64_t)nt > rows) nt = (unsigned)rows;
  if (nt <= 1) {
    run_rows(0, rows);
  } else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(run_rows, rows * t / nt, rows * (t + 1) / nt);
    for (auto& t : th) t.join();
  }
  return PHMRF_HOST_OK;
}
