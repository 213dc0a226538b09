"""Raw Hi-C text -> (samples, len_vec, edge_list_vec): the reference's loader (utility.py) restated for the host, so
the CLI runs from `example_input`-style directories without medpy / skimage and in seconds instead of minutes.

Kept names and outputs (file:line in /root/reference):
  quantile_contact_vec(chrom_vec, resolution, ref_filename, filename_list, species)        utility.py:2463-2505
  load_data_chromosome2(chrom_vec, x_max, x_min, resolution, num_neighbor, filter_mode, sigma, diagonal_typeId,
                        ref_filename, filename_list, species, data_path, annotation)       utility.py:267-534
      -> samples float64 [n,S], len_vec rows [n, start, stop, H, W, start_bin1, start_bin2, region_id, type, chrom]
         (utility.py:455-456, :528), edge_list_vec: per block float64 [E,3] = (id1, id2, d) sorted by (id1, id2)
Steps restated: merge of the species' contact files on the union of their loci (multi_contact_matrix3A, :2507-2570,
:2631-2662), per-species min/max normalisation and log(1+x) (:867-897, :367), synteny regions with the hg38 chr3 / chr6
centromere split (subregion1, :2111-2190), selection of a region's bin pairs (:1331-1345), the contact-map image
(:2192-2229, :2332-2366), the median fill of empty cells (:603-660; native, libphmrf_host.so), the smoothing filter
(:1566-1588), the image -> node array (:2295-2330, :2368-2401) and the edge list with feature distances (:1871-2053).

Python-2 semantics kept where they matter (the reference is Python 2): `N = ceil(chrom_size / resolution)` and
`x1 / resolution` are INTEGER divisions there (:2516, :2542).

Filters: filter_mode 0 is medpy's `anisotropic_diffusion(img, niter=5, kappa=50, gamma=0.1, option=1)` (:412, :1573).
medpy is not in this image and not under /root/reference, so `anisotropic_diffusion` below restates its published
algorithm (Perona-Malik, exponential conductance, float32 working array) -- PARITY UNPINNED for that one function; the
rest of the pipeline is pinned on fixtures recorded from the reference's own loader (tests/golden/).  filter_mode 1 is
skimage's `denoise_bilateral(img, sigma_color=0.5, sigma_spatial=5)` (:1575-1582): scikit-image is absent as well, so
`denoise_bilateral` below restates its published algorithm (native, libphmrf_host.so) -- PARITY UNPINNED likewise.  Any
other value with sigma > 0 is scipy's Gaussian filter, as in the reference.
"""
from __future__ import print_function

import ctypes
import math
import os

import numpy as np

THRESH1 = 1e-05                                                   # utility.py:47
REGION_POINTS_HG38 = np.asarray([[3, 90279522, 93797661], [6, 57542947, 61520508]])   # utility.py:385


def _host():
    from . import mstep
    L = mstep.host_lib()
    if not getattr(L, "_pre_ready", False):
        L.phmrf_median_fill.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                        ctypes.c_double, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        L.phmrf_median_fill.restype = ctypes.c_int
        L.phmrf_bilateral.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        L.phmrf_bilateral.restype = ctypes.c_int
        L._pre_ready = True
    return L


def median_fill(mtx, symmetric):
    """near_interpolation1 (symmetric) / near_interpolation1a (general): in place, returns (cnt1, cnt2)."""
    if not (mtx.dtype == np.float64 and mtx.flags.c_contiguous and mtx.ndim == 2):
        raise ValueError("median_fill needs a C-contiguous float64 matrix")
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    st = _host().phmrf_median_fill(mtx.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), mtx.shape[0], mtx.shape[1],
                                   int(bool(symmetric)), THRESH1, ctypes.byref(a), ctypes.byref(b))
    if st != 0:
        raise ValueError("phmrf_median_fill: invalid argument (a symmetric fill needs a square matrix)")
    return a.value, b.value


def anisotropic_diffusion(img, niter=1, kappa=50, gamma=0.1, option=1):
    """Perona-Malik diffusion as medpy.filter.smoothing.anisotropic_diffusion computes it for a 2-D image with unit
    voxel spacing (published algorithm; medpy is absent here -> parity unpinned): float32 working copy; per iteration
    forward differences along each axis (0 at the far border), flux = c(delta) * delta with c = exp(-(delta/kappa)^2)
    (option 1) or 1 / (1 + (delta/kappa)^2) (option 2), divergence by backward difference of the flux, explicit step
    `out += gamma * div`."""
    out = np.array(img, dtype=np.float32, copy=True)
    if option not in (1, 2):
        raise ValueError("option must be 1 or 2")
    deltas = [np.zeros_like(out) for _ in range(out.ndim)]
    for _ in range(int(niter)):
        for ax in range(out.ndim):
            sl = tuple(slice(None, -1) if j == ax else slice(None) for j in range(out.ndim))
            deltas[ax][sl] = np.diff(out, axis=ax)
        if option == 1:
            flux = [np.exp(-(d / kappa) ** 2.0) * d for d in deltas]
        else:
            flux = [1.0 / (1.0 + (d / kappa) ** 2.0) * d for d in deltas]
        for ax in range(out.ndim):
            sl = tuple(slice(1, None) if j == ax else slice(None) for j in range(out.ndim))
            flux[ax][sl] = np.diff(flux[ax], axis=ax)
        out += gamma * np.sum(flux, axis=0)
    return out


def denoise_bilateral(img, sigma_color=None, sigma_spatial=1, win_size=None, bins=10000):
    """skimage.restoration.denoise_bilateral(img, sigma_color, sigma_spatial, multichannel=False) of a 2-D float image,
    as the reference calls it for filter_mode 1 (utility.py:1575-1582): restated from the published algorithm of
    scikit-image 0.13 / 0.14 (`_denoise_cy._denoise_bilateral`, mode 'constant', cval 0) -- scikit-image is absent here,
    PARITY UNPINNED.  Native (libphmrf_host.so `phmrf_bilateral`); oracle/ref_numpy.py holds the NumPy restatement the
    tests compare it with."""
    a = np.ascontiguousarray(img, dtype=np.float64)
    if a.ndim != 2:
        raise ValueError("denoise_bilateral: a 2-D image (one channel) is expected")
    if sigma_color is None:
        sigma_color = float(a.std())                              # skimage: `sigma_color or image.std()`
    out = np.empty_like(a)
    dp = ctypes.POINTER(ctypes.c_double)
    st = _host().phmrf_bilateral(a.ctypes.data_as(dp), a.shape[0], a.shape[1], float(sigma_color), float(sigma_spatial),
                                 0 if win_size is None else int(win_size), int(bins), out.ctypes.data_as(dp))
    if st != 0:
        raise ValueError("denoise_bilateral: image must contain only positive values, sigmas must be positive, the "
                         "window odd")
    return out


def _apply_filter(mtx1, filter_mode, filter_param1, filter_param2, sigma):
    """utility.py:1566-1588 / :1752-1774, channel by channel, in place."""
    dim1 = mtx1.shape[-1]
    if filter_mode == 0:
        for i in range(dim1):
            if filter_param1 < 0:
                mtx1[:, :, i] = anisotropic_diffusion(mtx1[:, :, i], niter=10, kappa=50, gamma=0.1, option=1)
            else:
                mtx1[:, :, i] = anisotropic_diffusion(mtx1[:, :, i], niter=filter_param1, kappa=filter_param2, gamma=0.1,
                                                      option=1)
    elif filter_mode == 1:
        for i in range(dim1):
            if filter_param1 < 0:
                mtx1[:, :, i] = denoise_bilateral(mtx1[:, :, i], sigma_color=0.5, sigma_spatial=5)
            else:
                mtx1[:, :, i] = denoise_bilateral(mtx1[:, :, i], sigma_color=filter_param1, sigma_spatial=filter_param2)
    elif sigma > 0:
        import scipy.ndimage
        for i in range(dim1):
            mtx1[:, :, i] = scipy.ndimage.gaussian_filter(mtx1[:, :, i], sigma)


# ---- merge of the species' contact files -----------------------------------------------------------------------
def _chrom_bins(ref_chromsize, chrom, resolution):
    key = "chr%s" % chrom
    with open(ref_chromsize) as f:
        for line in f:
            t = line.split("\t")
            if t[0] == key:
                return int(int(t[1]) // int(resolution))      # Python 2: math.ceil(int / int) floors first (:2516)
    raise ValueError("chrom size error: %s not in %s" % (key, ref_chromsize))


def _read_contacts(path, chrom, resolution):
    import pandas as pd
    fn = "%s/chr%s.%dK.txt" % (path, chrom, int(resolution / 1000))
    if not os.path.exists(fn):
        raise IOError("File %s does not exist. Please check." % fn)
    d = pd.read_csv(fn, header=None, sep="\t")
    x1 = np.asarray(d[0]).astype(np.int64) // int(resolution)  # Python 2 integer division (:2542)
    x2 = np.asarray(d[1]).astype(np.int64) // int(resolution)
    value = np.asarray(d[2], dtype=np.float64).copy()
    value[np.isnan(value)] = -1                                 # :2546-2547
    return x1, x2, value


def multi_contact_matrix3A(chrom, resolution, ref_chromsize, filename_list, species):
    """-> position int64 [m,3] = (bin1, bin2, serial) and x float64 [m,S] on the union of the species' loci, rows in
    ascending serial order; 0 where a species has no entry (utility.py:2507-2570, :2631-2662)."""
    N = _chrom_bins(ref_chromsize, chrom, resolution)
    per = []
    serial1 = np.zeros(0, dtype=np.int64)
    for path in filename_list[:len(species)]:
        x1, x2, value = _read_contacts(path, chrom, resolution)
        ser = np.int64(N * x1 + x2)
        per.append((ser, x1, x2, value))
        serial1 = np.union1d(serial1, ser)
    m = serial1.shape[0]
    x = np.zeros((m, len(species)))
    pos = np.zeros((m, 3), dtype=np.int64)
    for i, (ser, x1, x2, value) in enumerate(per):
        idx = np.searchsorted(serial1, ser)
        x[idx, i] = value
        pos[idx, 0], pos[idx, 1] = x1, x2
    pos[:, 2] = serial1
    return pos, x


def normalize_feature(x1, x_min, x_max):
    """utility.py:867-897 (x1 is modified in place like there)."""
    n2 = x1.shape[1]
    vec1 = np.zeros((n2, 2))
    for i in range(n2):
        x = x1[:, i]
        x[x < 0] = 0
        vec1[i] = [np.min(x), np.max(x)]
    if x_min < 0:
        x_min = np.median(vec1[:, 0])
    if x_max < 0:
        x_max = np.median(vec1[:, 1])
    for i in range(n2):
        x = x1[:, i]
        m1, m2 = vec1[i]
        x1[:, i] = x_min + (x - m1) * 1.0 * (x_max - x_min) / (m2 - m1)
    return x1, vec1, x_min, x_max


def quantile_contact_vec(chrom_vec, resolution, ref_filename, filename_list, species):
    """Per chromosome and species: [np.percentile(values >= 0, q) for q in 0.05, 0.25, 0.50, 0.75, 0.95 | min of the
    values > 0 | max | max / (column 4 + 1e-16) | #values > 0 | #values >= 0]   (utility.py:2463-2505).  NaN entries
    count as -1.  The reference hands the FRACTIONS 0.05 .. 0.95 to np.percentile (:2484, :2496), i.e. the 0.05th .. 0.95th
    percentiles -- kept; only column 6 (the maximum) is used downstream (phylo_hmrf.py:1654-1655)."""
    rows = []
    for chrom in chrom_vec:
        _chrom_bins(ref_filename, chrom, resolution)
        for path in filename_list[:len(species)]:
            _, _, values = _read_contacts(path, chrom, resolution)
            b1, b2 = values > 0, values >= 0
            r = np.zeros(10)
            r[0:5] = np.percentile(values[b2], [0.05, 0.25, 0.50, 0.75, 0.95])
            r[5] = np.min(values[b1])
            r[6] = np.max(values)
            r[7] = np.max(values) / (r[4] + 1e-16)
            r[8], r[9] = b1.sum(), b2.sum()
            rows.append(r)
    return np.asarray(rows)


# ---- synteny regions ---------------------------------------------------------------------------------------------
def subregion1(filename, chrom_id, resolution, region_points):
    """-> list of [position1, position2, position1a, position2a, length, length_a, region_id, region_id1, chrom_id]
    (utility.py:2111-2190): one diagonal region per synteny block; a block that spans a centromere gap listed in
    `region_points` becomes two diagonal regions and the off-diagonal region between them."""
    t = np.loadtxt(filename, dtype="int", delimiter="\t")
    t = np.atleast_2d(t)
    region_list = [[int(t[i, 0]), int(t[i, 1]), int(t[i, 2]), i] for i in range(t.shape[0])]
    threshold = resolution * 2
    for point1, point2 in region_points:
        vec1 = np.asarray(region_list)
        b = np.where((vec1[:, 0] < point1 - threshold) & (vec1[:, 1] > point2 + threshold))[0]
        if len(b) > 0:
            id1 = int(b[0])
            region_id = int(vec1[id1, 3])
            start1, stop1 = int(vec1[id1, 0]), int(point1)
            start2, stop2 = int(point2), int(vec1[id1, 1])
            region_list[id1] = [start2, stop2, stop2 - start2, region_id]
            region_list.insert(id1, [start1, stop1, stop1 - start1, region_id])
    ids = np.asarray([r[3] for r in region_list])
    out = []
    region_id1 = 0
    for region_id in np.sort(np.unique(ids)):
        b = np.where(ids == region_id)[0]
        if len(b) == 1:
            p1, p2, ln = region_list[b[0]][0:3]
            out.append([p1, p2, p1, p2, ln, ln, int(region_id), region_id1, chrom_id])
            region_id1 += 1
        else:
            for i in range(len(b)):
                for j in range(i, len(b)):
                    p1, p2, ln = region_list[b[i]][0:3]
                    p1a, p2a, lna = region_list[b[j]][0:3]
                    out.append([p1, p2, p1a, p2a, ln, lna, int(region_id), region_id1, chrom_id])
                    region_id1 += 1
    return out


def grid_edges(X, H, W, diagonal, num_neighbor=8):
    """edge_weightlist_grid3_undirected_unsym (diagonal block: nodes = upper triangle row-major; utility.py:1871-1973)
    / edge_weightlist_grid3_undirected (full H x W block; :1975-2053): rows (id1, id2, d) sorted by (id1, id2) with
    d = |x1 - x2|^2 / (|x1| |x2| + 1e-16), halved between two diagonal nodes of a diagonal block."""
    from .graph_host import grid_edges as _ge
    return _ge(X, H, W, diagonal, num_neighbor)


def _region_block(region, x, position, resolution, num_neighbor, filter_mode, filter_param1, filter_param2, sigma):
    """load_data_chromosome_sub3 (utility.py:470-534) for one region -> (samples, t_lenvec (8 fields), edge_list)."""
    position1, position2, position1a, position2a = region[0:4]
    region_id1 = region[6]
    type_id1 = 1 if (position1 == position1a and position2 == position2a) else 0
    xp1, xp2 = position[:, 0] * resolution, (position[:, 1] + 1) * resolution            # border_type 0 (:1333-1335)
    sel = np.where((xp1 >= position1) & (xp1 <= position2) & (xp2 >= position1a) & (xp2 <= position2a))[0]
    value, pos = x[sel, :], position[sel, :]
    S = value.shape[1]
    if type_id1 == 1:
        start = int(min(pos[:, 0].min(), pos[:, 1].min()))                                # :2206-2212
        stop = int(max(pos[:, 0].max(), pos[:, 1].max()))
        win = stop - start + 1
        mtx1 = np.zeros((win, win, S))
        i1, i2 = pos[:, 0] - start, pos[:, 1] - start
        mtx1[i1, i2] = value
        mtx1[i2, i1] = value
        for s in range(S):
            plane = np.ascontiguousarray(mtx1[:, :, s])
            median_fill(plane, True)
            mtx1[:, :, s] = plane
        _apply_filter(mtx1, filter_mode, filter_param1, filter_param2, sigma)
        ii, jj = np.triu_indices(win)                                                      # :2310-2317
        data1 = mtx1[ii, jj, :]
        edges = grid_edges(data1, win, win, True, num_neighbor)
        H = W = win
        start1 = start2 = start                                                            # np.min(t_position)
    else:
        start1, start2 = int(pos[:, 0].min()), int(pos[:, 1].min())                       # :2346-2350
        H, W = int(pos[:, 0].max()) - start1 + 1, int(pos[:, 1].max()) - start2 + 1
        mtx1 = np.zeros((H, W, S))
        mtx1[pos[:, 0] - start1, pos[:, 1] - start2] = value
        for s in range(S):
            plane = np.ascontiguousarray(mtx1[:, :, s])
            median_fill(plane, False)
            mtx1[:, :, s] = plane
        _apply_filter(mtx1, filter_mode, filter_param1, filter_param2, sigma)
        data1 = mtx1.reshape(H * W, S).copy()
        edges = grid_edges(data1, H, W, False, num_neighbor)
    return data1, [data1.shape[0], H, W, start1, start2, region_id1, type_id1, region[8]], edges


def load_data_chromosome2(chrom_vec, x_max, x_min, resolution, num_neighbor, filter_mode, sigma, diagonal_typeId,
                          ref_filename, filename_list, species, data_path, annotation=""):
    samples, len_vec, edge_list_vec = [], [], []
    offset = 0
    filter_param1, filter_param2 = (5, 50) if filter_mode == 0 else (-1, -1)               # utility.py:411-412
    for chrom_id in sorted(int(c) for c in chrom_vec):                                    # np.argsort(chrom ids), :313
        position, x1 = multi_contact_matrix3A(str(chrom_id), resolution, ref_filename, filename_list, species)
        x1, _, _, _ = normalize_feature(x1, x_min, x_max)
        x = np.log(1 + x1)                                                                 # :367
        pts = [REGION_POINTS_HG38[i, 1:] for i in np.where(REGION_POINTS_HG38[:, 0] == chrom_id)[0]]
        regions = subregion1("%s/chr%s.synteny.txt" % (data_path, chrom_id), chrom_id, resolution, pts)
        if diagonal_typeId == 1:
            regions = [r for r in regions if r[0] == r[2] and r[1] == r[3]]               # :396-400
        for region in regions:
            data1, lv, edges = _region_block(region, x, position, resolution, num_neighbor, filter_mode, filter_param1,
                                             filter_param2, sigma)
            n = data1.shape[0]
            lv[1:1] = [offset, offset + n]                                                 # :453-454, :322-324
            offset += n
            samples.append(data1)
            len_vec.append(lv)
            edge_list_vec.append(edges)
    return np.concatenate(samples, axis=0), len_vec, edge_list_vec
