"""Operator interface of the reference's propagation step (src/model/predict.py), backed by the HIP engine.

Same names, argument order and meaning as the reference:
    predict(ref, target, ref_label, weight_dense, weight_sparse, frame_idx, range, ref_num, temperature,
            probability_propagation) -> (d, H*W) f32                      (reference predict.py:19-71)
    sample_frames(frame_idx, take_range, num_refs) -> LongTensor          (predict.py:74-89)
    get_labels(label, d, H, W, H_d, W_d) -> (d, 1, H_d*W_d) int32        (predict.py:92-96)
    prepare_first_frame(curr_video, save_prediction, annotation, ...)     (predict.py:99-155)
    get_spatial_weight(shape, sigma) -> SpatialPrior                      (predict.py:158-175)

The one deliberate difference: `get_spatial_weight` does not materialise the (HW, HW) matrix (165 MB at 480p, with a
330 MB transient) - it returns a `SpatialPrior` that just remembers sigma, and the kernel evaluates the prior from pixel
coordinates.  `predict` also accepts real (HW, HW) tensors (e.g. produced by the reference) and recovers sigma from
them, so it is a drop-in either way.  There is no CPU path: tensors must live on the HIP device.
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import engine as _engine
from .config import Config


class SpatialPrior:
    """Stand-in for the reference's dense spatial-weight matrix: w[i,j] = exp(-((i/W - j/W)^2 + (i%W - j%W)^2)/sigma^2)
    with the reference's fractional-row quirk (predict.py:168).  `dense()` materialises it (tests only)."""

    def __init__(self, shape, sigma):
        self.shape = (int(shape[0]), int(shape[1]))
        self.sigma = float(sigma)

    def dense(self, device=None):
        H, W = self.shape
        idx = torch.arange(H * W, dtype=torch.long, device=device).reshape(H * W, 1)
        coords = torch.cat((idx.div(float(W)), idx % W), -1)
        d2 = (coords - coords.unsqueeze(1)).float().pow(2).sum(-1)
        return (-d2 / self.sigma ** 2).exp()


def get_spatial_weight(shape, sigma, t_loc=None):
    if t_loc is not None:
        raise NotImplementedError('t_loc is never used by the reference (predict.py:170-171 has no caller)')
    return SpatialPrior(shape, sigma)


def _sigma_of(weight, W_d):
    """sigma from either a SpatialPrior or a dense (HW,HW) weight tensor: w[0,1] = exp(-(1/W^2 + 1)/sigma^2)."""
    if weight is None:
        return None
    if isinstance(weight, SpatialPrior):
        return weight.sigma
    if weight.shape[0] < 2:
        return 8.0
    w01 = float(weight[0, 1])
    return math.sqrt(-(1.0 + 1.0 / (W_d * W_d)) / math.log(w01))


_ENGINES = {}


def _engine_for(device, H_d, W_d, ref_num):
    key = (device.index, H_d, W_d)
    eng = _ENGINES.get(key)
    if eng is None or eng.cfg.ref_num < ref_num:
        if eng is not None:
            eng.close()
        eng = _engine.PropagationEngine(H_d, W_d, device=device.index, ref_num=max(ref_num, 9))
        _ENGINES[key] = eng
    return eng


def sample_frames(frame_idx, take_range, num_refs):
    idx = _engine.sample_frames_list(frame_idx, take_range, num_refs)
    return torch.tensor(idx, dtype=torch.long, device=Config.DEVICE)


def predict(ref, target, ref_label, weight_dense, weight_sparse, frame_idx, range, ref_num, temperature,
            probability_propagation):
    """
    :param ref: (N, feature_dim, H, W)   history features (all previous frames)
    :param target: (feature_dim, H, W)
    :param ref_label: (d, N, H*W)
    :param weight_dense / weight_sparse: SpatialPrior or (H*W, H*W) tensors (None in probability mode)
    :return: (d, H*W) f32
    """
    if not ref.is_cuda:
        raise _engine.VospropError('predict(): tensors must be on the HIP device - the engine has no CPU path')
    T, C, H_d, W_d = ref.shape
    eng = _engine_for(ref.device, H_d, W_d, ref_num)
    s1 = _sigma_of(weight_dense, W_d) or 8.0
    s2 = _sigma_of(weight_sparse, W_d) or 21.0
    return eng.predict(ref, target, ref_label, frame_idx, range, ref_num, temperature, s1, s2, probability_propagation)


def index_to_onehot(idx, d):
    """reference src/utils/utils.py:59-68"""
    n = idx.shape[0]
    return torch.zeros(d, n, device=idx.device).scatter_(0, idx.view(1, -1), 1)


def get_labels(label, d, H, W, H_d, W_d):
    label_1hot = index_to_onehot(label.view(-1), d).reshape(1, d, H, W)
    label_1hot = F.interpolate(label_1hot, size=(H_d, W_d), mode='nearest')
    return label_1hot.reshape(d, -1).unsqueeze(1).type(torch.int32)


def prepare_first_frame(curr_video, save_prediction, annotation, sigma1=8, sigma2=21, inference_strategy='single',
                        probability_propagation=False, scale=None):
    from PIL import Image
    first_annotation = Image.open(annotation)
    label_np = np.asarray(first_annotation)
    (H, W) = label_np.shape
    H_d = int(np.ceil(H * Config.SCALE))
    W_d = int(np.ceil(W * Config.SCALE))
    palette = first_annotation.getpalette()
    d = int(np.max(label_np)) + 1
    label = torch.from_numpy(label_np.astype(np.int64)).to(Config.DEVICE)
    label_1hot = get_labels(label, d, H, W, H_d, W_d)
    weight_dense = get_spatial_weight((H_d, W_d), sigma1) if not probability_propagation else None
    weight_sparse = get_spatial_weight((H_d, W_d), sigma2) if not probability_propagation else None
    if save_prediction is not None:
        save_path = os.path.join(save_prediction, curr_video)
        os.makedirs(save_path, exist_ok=True)
        first_annotation.save(os.path.join(save_path, '00000.png'))
    if inference_strategy == 'hor-flip':
        return label_1hot, get_labels(torch.fliplr(label), d, H, W, H_d, W_d), d, palette, weight_dense, weight_sparse
    if inference_strategy == 'ver-flip':        # the name the reference's loop passes (inference_utils.py:219)
        return label_1hot, get_labels(torch.flipud(label), d, H, W, H_d, W_d), d, palette, weight_dense, weight_sparse
    if inference_strategy in ('2-scale', 'hor-2-scale', '3-scale'):
        H_d_2 = int(np.ceil(H * Config.SCALE * scale))
        W_d_2 = int(np.ceil(W * Config.SCALE * scale))
        weight_dense_2 = get_spatial_weight((H_d_2, W_d_2), sigma1) if not probability_propagation else None
        weight_sparse_2 = get_spatial_weight((H_d_2, W_d_2), sigma2) if not probability_propagation else None
        label_1hot_2 = get_labels(label, d, H, W, H_d_2, W_d_2)
        if inference_strategy == '3-scale':     # predict.py:145-153: the scaled set replaces the plain one
            return label_1hot_2, d, palette, weight_dense_2, weight_sparse_2
        return (label_1hot, label_1hot_2), d, palette, (weight_dense, weight_dense_2), (weight_sparse, weight_sparse_2)
    return label_1hot, d, palette, weight_dense, weight_sparse      # 'single', 'multimodel' (:142-144) and anything else
