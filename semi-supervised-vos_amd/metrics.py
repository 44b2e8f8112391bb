"""J (region similarity) and F (boundary measure) of the reference's evaluation (src/utils/metrics.py:11-162), numpy + scipy.

SURVEY.md section 8f rank 4: the tool that states "mask IoU delta" in the reference's own terms.  Same names and arguments:
evaluate_segmentation, eval_j, eval_f, f_measure.  The reference dilates the boundary maps with scikit-image
(`skimage.morphology.dilation(b.astype(uint8), disk(r))`, metrics.py:89-92), which is not in this image; for 0/1 images a flat grey
dilation IS the binary dilation with the same footprint and zero padding, so `scipy.ndimage.binary_dilation(b, structure=disk(r))`
restates it (parity for f_measure is pinned by known answers only - the reference's own F cannot be run here; eval_j and the
boundary map are pinned by goldens made from the reference)."""
import numpy as np


def evaluate_segmentation(annotation, segmentation, void_pixels=None, threshold=0.008):
    """reference metrics.py:11-12"""
    return eval_j(annotation, segmentation, void_pixels), eval_f(annotation, segmentation, void_pixels, threshold)


def eval_j(annotation, segmentation, void_pixels=None):
    """Jaccard index |A & S| / |A | S| outside the void pixels, 1 where the union is empty (reference metrics.py:15-45).
    Works on single maps (returns a scalar) and on stacks (reduces the last two axes)."""
    assert annotation.shape == segmentation.shape, \
        f'Annotation({annotation.shape}) and segmentation:{segmentation.shape} dimensions do not match.'
    annotation = annotation.astype(bool)
    segmentation = segmentation.astype(bool)
    if void_pixels is not None:
        assert annotation.shape == void_pixels.shape, \
            f'Annotation({annotation.shape}) and void pixels:{void_pixels.shape} dimensions do not match.'
        void_pixels = void_pixels.astype(bool)
    else:
        void_pixels = np.zeros_like(segmentation)
    keep = np.logical_not(void_pixels)
    inters = np.sum((segmentation & annotation) & keep, axis=(-2, -1))
    union = np.sum((segmentation | annotation) & keep, axis=(-2, -1))
    with np.errstate(divide='ignore', invalid='ignore'):
        j = inters / union
    if j.ndim == 0:
        j = 1 if np.isclose(union, 0) else j
    else:
        j[np.isclose(union, 0)] = 1
    return j


def eval_f(annotation, segmentation, void_pixels=None, bound_th=0.008):
    """reference metrics.py:48-64: per-frame f_measure of a stack, or of a single map"""
    assert annotation.shape == segmentation.shape
    if void_pixels is not None:
        assert annotation.shape == void_pixels.shape
    if annotation.ndim == 3:
        f_res = np.zeros(annotation.shape[0])
        for frame_id in range(annotation.shape[0]):
            vp = None if void_pixels is None else void_pixels[frame_id]
            f_res[frame_id] = f_measure(segmentation[frame_id], annotation[frame_id], vp, bound_th=bound_th)
        return f_res
    if annotation.ndim == 2:
        return f_measure(segmentation, annotation, void_pixels, bound_th=bound_th)
    raise ValueError(f'db_eval_boundary does not support tensors with {annotation.ndim} dimensions')


def disk(radius):
    """skimage.morphology.disk: (2r+1)x(2r+1) footprint of the points with x^2 + y^2 <= r^2"""
    L = np.arange(-radius, radius + 1)
    X, Y = np.meshgrid(L, L)
    return (X ** 2 + Y ** 2) <= radius ** 2


def f_measure(foreground_mask, gt_mask, void_pixels=None, bound_th=0.008):
    """Boundary F-measure (reference metrics.py:67-121): precision / recall of the 1-pixel boundary maps of the two masks,
    each matched against the other one dilated by a disk of radius ceil(bound_th * |image diagonal|)."""
    from scipy.ndimage import binary_dilation
    assert np.atleast_3d(foreground_mask).shape[2] == 1
    void_pixels = np.zeros_like(foreground_mask).astype(bool) if void_pixels is None else void_pixels.astype(bool)
    bound_pix = bound_th if bound_th >= 1 else np.ceil(bound_th * np.linalg.norm(foreground_mask.shape))
    fg_boundary = _seg2bmap(foreground_mask * np.logical_not(void_pixels))
    gt_boundary = _seg2bmap(gt_mask * np.logical_not(void_pixels))
    fp = disk(bound_pix)
    fg_dil = binary_dilation(fg_boundary, structure=fp)
    gt_dil = binary_dilation(gt_boundary, structure=fp)
    gt_match = gt_boundary * fg_dil
    fg_match = fg_boundary * gt_dil
    n_fg = np.sum(fg_boundary)
    n_gt = np.sum(gt_boundary)
    if n_fg == 0 and n_gt > 0:
        precision, recall = 1, 0
    elif n_fg > 0 and n_gt == 0:
        precision, recall = 0, 1
    elif n_fg == 0 and n_gt == 0:
        precision, recall = 1, 1
    else:
        precision = np.sum(fg_match) / float(n_fg)
        recall = np.sum(gt_match) / float(n_gt)
    return 0 if precision + recall == 0 else 2 * precision * recall / (precision + recall)


def _seg2bmap(seg, width=None, height=None):
    """Binary boundary map, one pixel wide, offset half a pixel towards the origin (reference metrics.py:124-181; only the
    same-size case is used by f_measure and built here)."""
    seg = np.asarray(seg).astype(bool)
    assert np.atleast_3d(seg).shape[2] == 1
    h, w = seg.shape[:2]
    width = w if width is None else width
    height = h if height is None else height
    if (width, height) != (w, h):
        raise NotImplementedError('boundary maps are only built at the size of the segmentation')
    e = np.zeros_like(seg)
    s = np.zeros_like(seg)
    se = np.zeros_like(seg)
    e[:, :-1] = seg[:, 1:]
    s[:-1, :] = seg[1:, :]
    se[:-1, :-1] = seg[1:, 1:]
    b = seg ^ e | seg ^ s | seg ^ se
    b[-1, :] = seg[-1, :] ^ e[-1, :]
    b[:, -1] = seg[:, -1] ^ s[:, -1]
    b[-1, -1] = 0
    return b
