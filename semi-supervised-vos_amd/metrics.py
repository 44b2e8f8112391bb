"""J (region similarity) and F (boundary measure), the two numbers `main.py evaluation` reports.

What is computed is what the reference's evaluation computes (src/utils/metrics.py:15-45 region Jaccard, :67-121 boundary
F-measure with a disk tolerance of ceil(0.008 * image diagonal), :124-181 the half-pixel-offset boundary map) and the public names
and arguments are the same - evaluate_segmentation, eval_j, eval_f, f_measure - so callers and result files are interchangeable.
HOW it is computed is this project's own:

  * regions: every object pair of a frame is scored from ONE joint histogram of the two index maps (`pair_jaccard`), instead of a
    pair of boolean images per object;
  * boundaries: a mask's boundary is "differs from its east, south or south-east neighbour" evaluated on an edge-replicated pad
    (`boundary_map`; one expression, no border fix-ups), for a whole stack of objects at once;
  * tolerance matching: "lies within a disk of radius r of the other boundary" is decided by the exact Euclidean distance transform
    (integer nearest-pixel offsets from scipy, compared as dy^2 + dx^2 <= r^2) instead of dilating both boundary maps with a disk
    footprint - the same set (a disk footprint IS the set of offsets with dy^2 + dx^2 <= r^2, and only real pixels dilate), in
    O(pixels) instead of O(pixels * r^2).

Pinned by: the reference's own eval_j / boundary maps (goldens G8, produced by running the reference) and, for F - which the
reference computes with scikit-image, absent from this image - a committed fixture made by brute-force disk dilation
(tests/golden/make_f_fixture.py) plus closed-form cases.
"""
import numpy as np


# ---------------------------------------------------------------------------------------------------------------------------
# regions
def _as_bool(x):
    x = np.asarray(x)
    return x if x.dtype == np.bool_ else x.astype(bool)


def eval_j(annotation, segmentation, void_pixels=None):
    """Jaccard index of two binary masks (or of two stacks of them: the last two axes are the image), ignoring void pixels;
    1 where both are empty.  Scalar for a single pair of maps."""
    a, s = _as_bool(annotation), _as_bool(segmentation)
    if a.shape != s.shape:
        raise AssertionError(f'annotation {a.shape} and segmentation {s.shape} differ in shape')
    if void_pixels is not None:
        v = _as_bool(void_pixels)
        if v.shape != a.shape:
            raise AssertionError(f'annotation {a.shape} and void pixels {v.shape} differ in shape')
        valid = ~v
        a, s = a & valid, s & valid
    both = np.count_nonzero(a & s, axis=(-2, -1))
    either = np.count_nonzero(a | s, axis=(-2, -1))
    if np.ndim(either) == 0:
        return 1 if either == 0 else both / either
    out = np.ones(either.shape, dtype=np.float64)
    np.divide(both, either, out=out, where=either != 0)
    return out


def pair_jaccard(gt_index, seg_index, pairs):
    """Jaccard of (gt_index == g) against (seg_index == s) for every (g, s) in `pairs`, from one joint histogram of the two
    uint8 index maps: |A & S| = H[g, s], |A| = row sum, |S| = column sum."""
    g = np.asarray(gt_index, dtype=np.int64).ravel()
    s = np.asarray(seg_index, dtype=np.int64).ravel()
    hist = np.bincount(g * 256 + s, minlength=65536).reshape(256, 256)
    rows, cols = hist.sum(1), hist.sum(0)
    out = []
    for gi, si in pairs:
        inter = int(hist[gi, si])
        union = int(rows[gi] + cols[si]) - inter
        out.append(1.0 if union == 0 else inter / union)
    return np.asarray(out, dtype=np.float64)


# ---------------------------------------------------------------------------------------------------------------------------
# boundaries
def boundary_map(mask):
    """One-pixel-wide boundary of a binary mask (or of a stack (..., H, W) of masks): a pixel is on the boundary when it differs
    from its east, south or south-east neighbour; the neighbours of the last row / column are the row / column itself (edge
    replication), which reproduces the reference's special-cased borders and its always-clear bottom-right corner."""
    m = _as_bool(mask)
    pad = [(0, 0)] * (m.ndim - 2) + [(0, 1), (0, 1)]
    q = np.pad(m, pad, mode='edge')
    here = q[..., :-1, :-1]
    return (here != q[..., :-1, 1:]) | (here != q[..., 1:, :-1]) | (here != q[..., 1:, 1:])


def _seg2bmap(seg, width=None, height=None):
    """The reference's name for boundary_map (src/utils/metrics.py:124).  Only the same-size case exists here (the reference's
    rescaling branch has no caller)."""
    seg = np.asarray(seg)
    if np.atleast_3d(seg).shape[2] != 1:
        raise AssertionError('a single-channel mask is expected')
    h, w = seg.shape[:2]
    if (w if width is None else width, h if height is None else height) != (w, h):
        raise NotImplementedError('boundary maps are only built at the size of the segmentation')
    return boundary_map(seg)


def within_radius(boundary, radius):
    """Pixels whose Euclidean distance to the nearest True pixel of `boundary` is <= radius (exact integer arithmetic on the
    offsets of the distance transform).  Equals the dilation of `boundary` by a disk footprint of that radius."""
    from scipy.ndimage import distance_transform_edt
    b = _as_bool(boundary)
    if not b.any():
        return np.zeros(b.shape, dtype=bool)
    nearest = distance_transform_edt(~b, return_distances=False, return_indices=True)
    yy, xx = np.indices(b.shape, sparse=True)
    dy, dx = nearest[0] - yy, nearest[1] - xx
    return dy * dy + dx * dx <= radius * radius


def disk(radius):
    """Footprint of the disk tolerance, for reference and tests: integer offsets with dy^2 + dx^2 <= radius^2."""
    r = int(radius)
    off = np.arange(-r, r + 1)
    return off[:, None] ** 2 + off[None, :] ** 2 <= radius * radius


def boundary_tolerance(shape, bound_th=0.008):
    """Radius of the tolerance disk: bound_th itself when >= 1 (pixels), else ceil(bound_th * |image diagonal|)."""
    return bound_th if bound_th >= 1 else np.ceil(bound_th * np.linalg.norm(shape))


def _harmonic(n_fg, n_gt, fg_hits, gt_hits):
    """F from boundary sizes and matched counts; an empty boundary on one side is precision 1 / recall 0 (or the reverse), two
    empty boundaries agree perfectly."""
    if n_fg == 0 or n_gt == 0:
        if n_fg == n_gt:
            return 1.0
        return 0.0          # one of precision / recall is 0: the harmonic mean is 0
    precision, recall = fg_hits / float(n_fg), gt_hits / float(n_gt)
    total = precision + recall
    return 0 if total == 0 else 2 * precision * recall / total


def f_measure(foreground_mask, gt_mask, void_pixels=None, bound_th=0.008):
    """Boundary F-measure of a predicted mask against the ground truth."""
    fg, gt = _as_bool(foreground_mask), _as_bool(gt_mask)
    if np.atleast_3d(fg).shape[2] != 1:
        raise AssertionError('a single-channel mask is expected')
    if void_pixels is not None:
        valid = ~_as_bool(void_pixels)
        fg, gt = fg & valid, gt & valid
    radius = boundary_tolerance(fg.shape, bound_th)
    fg_b, gt_b = boundary_map(fg), boundary_map(gt)
    fg_hits = np.count_nonzero(fg_b & within_radius(gt_b, radius))
    gt_hits = np.count_nonzero(gt_b & within_radius(fg_b, radius))
    return _harmonic(np.count_nonzero(fg_b), np.count_nonzero(gt_b), fg_hits, gt_hits)


def eval_f(annotation, segmentation, void_pixels=None, bound_th=0.008):
    """f_measure of one pair of maps, or frame by frame over a (T, H, W) stack."""
    annotation, segmentation = np.asarray(annotation), np.asarray(segmentation)
    if annotation.shape != segmentation.shape or (void_pixels is not None and np.shape(void_pixels) != annotation.shape):
        raise AssertionError('annotation, segmentation and void pixels must have one shape')
    if annotation.ndim == 2:
        return f_measure(segmentation, annotation, void_pixels, bound_th=bound_th)
    if annotation.ndim != 3:
        raise ValueError(f'boundary evaluation takes (H, W) maps or (T, H, W) stacks, not {annotation.ndim} dimensions')
    voids = [None] * len(annotation) if void_pixels is None else void_pixels
    return np.fromiter((f_measure(s, a, v, bound_th=bound_th) for a, s, v in zip(annotation, segmentation, voids)),
                       dtype=np.float64, count=len(annotation))


def evaluate_segmentation(annotation, segmentation, void_pixels=None, threshold=0.008):
    """(J, F) of one pair of masks."""
    return eval_j(annotation, segmentation, void_pixels), eval_f(annotation, segmentation, void_pixels, threshold)


def frame_scores(gt_index, seg_index, pairs, bound_th=0.008):
    """(J, F) per object pair of one frame, J from the joint histogram, F per pair from the boundary stacks: (len(pairs), 2)."""
    gt_index, seg_index = np.asarray(gt_index), np.asarray(seg_index)
    j = pair_jaccard(gt_index, seg_index, pairs)
    radius = boundary_tolerance(gt_index.shape, bound_th)
    gt_b = boundary_map(np.stack([gt_index == g for g, _ in pairs]))
    seg_b = boundary_map(np.stack([seg_index == s for _, s in pairs]))
    f = [_harmonic(np.count_nonzero(sb), np.count_nonzero(gb), np.count_nonzero(sb & within_radius(gb, radius)),
                   np.count_nonzero(gb & within_radius(sb, radius))) for gb, sb in zip(gt_b, seg_b)]
    return np.stack([j, np.asarray(f, dtype=np.float64)], axis=1)
