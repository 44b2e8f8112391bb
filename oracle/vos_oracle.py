"""CPU ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

A CPU restatement (torch-CPU / numpy, fp32, same op ORDER as the reference so results are
bit-comparable) of the reference's label-propagation hot path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the
product (`semi-supervised-vos_amd/`) must fail loudly when its HIP library is missing and
never falls back to anything in here.

Parity status: PINNED.  Every function below is checked in `tests/test_oracle_golden.py`
against `tests/golden/reference_goldens.npz`, which was produced by importing and running the
reference's own code in the build container (`tests/golden/make_goldens.py`).

All `file:line` citations are into the reference tree (hynekdav/semi-supervised-VOS).
"""
import numpy as np
import torch

CONTINUOUS_FRAME = 4      # src/config.py:13
SCALE = 0.125             # src/config.py:12


def sample_frames(frame_idx, take_range, num_refs):
    """src/model/predict.py:74-89.  Returns a python list of history indices.

    frame_idx <= num_refs : every previous frame.  Otherwise (num_refs-3) indices
    linspace(max(frame_idx-4-take_range,0), frame_idx-4) truncated toward zero (float64
    arithmetic, exactly numpy's), followed by frame_idx-3, -2, -1.
    """
    if frame_idx <= num_refs:
        return list(range(frame_idx))
    dense_num = CONTINUOUS_FRAME - 1
    sparse_num = num_refs - dense_num
    ref_end = frame_idx - dense_num - 1
    ref_start = max(ref_end - take_range, 0)
    idx = np.linspace(ref_start, ref_end, sparse_num).astype(int).tolist()
    idx += [frame_idx - dense_num + j for j in range(dense_num)]
    return idx


def get_spatial_weight(shape, sigma):
    """src/model/predict.py:158-175 (t_loc branch never taken).

    NOTE the quirk kept on purpose: the 'row' coordinate is idx / float(W) - a TRUE division of
    the flat index (predict.py:168) - so dist^2 = (di/W)^2-ish fractional rows, not grid rows.
    """
    H, W = shape
    index = torch.arange(H * W, dtype=torch.long).reshape(H * W, 1)
    coords = torch.cat((index.div(float(W)), index % W), -1)        # (HW,2) f32
    diff = coords - coords.unsqueeze(1)                              # (HW,HW,2)
    d2 = diff.float().pow(2).sum(-1)
    return (-d2 / sigma ** 2).exp()


def index_to_onehot(idx, d):
    """src/utils/utils.py:59-68: zeros(d,n).scatter_(0, idx, 1) -> f32 (d,n)."""
    idx = torch.as_tensor(idx).long().view(-1)
    return torch.zeros(d, idx.shape[0]).scatter_(0, idx.view(1, -1), 1)


def nearest_src_index(out_size, in_size):
    """Index map of F.interpolate(mode='nearest') as used at predict.py:94 and
    inference_utils.py:74: src = floor(dst * in/out) computed in f32 like ATen does."""
    scale = np.float32(in_size) / np.float32(out_size)
    src = np.floor(np.arange(out_size, dtype=np.float32) * scale).astype(np.int64)
    return np.minimum(src, in_size - 1)


def get_labels(label, d, H, W, H_d, W_d):
    """src/model/predict.py:92-96: one-hot -> nearest resize -> (d,1,H_d*W_d) int32."""
    label = torch.as_tensor(label).long()
    oh = index_to_onehot(label.view(-1), d).reshape(1, d, H, W)
    oh = torch.nn.functional.interpolate(oh, size=(H_d, W_d), mode='nearest')
    return oh.reshape(d, -1).unsqueeze(1).type(torch.int32)


def feature_map_size(H, W):
    """src/model/predict.py:109-110."""
    return int(np.ceil(H * SCALE)), int(np.ceil(W * SCALE))


def predict(ref, target, ref_label, weight_dense, weight_sparse, frame_idx, take_range, ref_num,
            temperature, probability_propagation, topk=0, affinity_bf16=False):
    """src/model/predict.py:19-71, op for op.

    ref (T,C,H,W) f32; target (C,H,W); ref_label (d,T,HW); weights (HW,HW) or None.
    Softmax is over ALL N*HW reference rows jointly (dim=0, :55); the spatial prior multiplies
    AFTER normalisation (:59-66): sigma2 for all but the last 4 sampled frames when
    frame_idx > 15, else sigma1 everywhere; the output is NOT renormalised (:70).
    """
    ref = torch.as_tensor(ref)
    target = torch.as_tensor(target)
    ref_label = torch.as_tensor(ref_label)
    d = ref_label.shape[0]
    sample_idx = torch.tensor(sample_frames(frame_idx, take_range, ref_num), dtype=torch.long)
    ref_sel = ref.index_select(0, sample_idx)
    lab_sel = ref_label.index_select(1, sample_idx).reshape(d, -1)
    num_ref, C, H, W = ref_sel.shape
    R = ref_sel.permute(0, 2, 3, 1).reshape(-1, C)
    T = target.reshape(C, -1)
    S = R.mm(T)
    if affinity_bf16:
        # NOT the reference's CPU path: the affinity as the engine's materialised variant keeps it in HBM (bf16; the reference's
        # CUDA autocast path keeps it in fp16) - checker of tests/test_gpu_configs.py::test_config5_materialised_affinity
        S = S.to(torch.bfloat16).to(torch.float32)
    S *= temperature
    S = S.softmax(dim=0)
    S = S.contiguous().view(num_ref, H * W, H * W)
    if not probability_propagation:
        if frame_idx > 15:
            S[:-CONTINUOUS_FRAME] *= weight_sparse
            S[-CONTINUOUS_FRAME:] *= weight_dense
        else:
            S = S.mul(weight_dense)
    S = S.view(-1, H * W)
    if topk and topk < S.shape[0]:
        # NOT in the reference (parity unpinned by the reference, pinned by k >= N*HW == dense):
        # per target pixel keep the k largest entries of the weighted affinity, zero the rest, no renormalisation.
        kth = S.topk(topk, dim=0).values[-1:]
        S = torch.where(S >= kth, S, torch.zeros_like(S))
    return lab_sel.float().mm(S.float())


def spatial_weight_columns(shape, sigma, cols):
    """Columns `cols` of get_spatial_weight(shape, sigma) without forming the (HW, HW) matrix (829 MB at 720p): the same
    f32 operations on the same operands (predict.py:167-173), so the entries are bit-identical to slicing the full matrix."""
    H, W = shape
    index = torch.arange(H * W, dtype=torch.long).reshape(H * W, 1)
    coords = torch.cat((index.div(float(W)), index % W), -1)        # (HW,2) f32
    cols = torch.as_tensor(cols, dtype=torch.long)
    diff = coords[cols].unsqueeze(0) - coords.unsqueeze(1)           # [p, j, :] = coords[cols[j]] - coords[p]  (as full[p, t])
    d2 = diff.float().pow(2).sum(-1)
    return (-d2 / sigma ** 2).exp()


def predict_columns(ref, target, ref_label, sigma1, sigma2, frame_idx, take_range, ref_num, temperature,
                    probability_propagation, cols, topk=0, affinity_bf16=False):
    """`predict` restricted to the target pixels `cols`: returns predict(...)[:, cols].  Every target pixel is an independent
    column of the reference's computation (mm column, softmax over dim 0, weight column, label mm column; predict.py:49-70), so
    the full-size BASELINE configs (720p: a 7.5 GB f32 affinity, three times) can be checked on a few hundred columns in
    seconds.  Same op order as `predict`; pinned to it by tests/test_oracle_golden.py::test_predict_columns_is_a_slice_of_predict."""
    ref = torch.as_tensor(ref)
    target = torch.as_tensor(target)
    ref_label = torch.as_tensor(ref_label)
    cols = torch.as_tensor(cols, dtype=torch.long)
    d = ref_label.shape[0]
    sample_idx = torch.tensor(sample_frames(frame_idx, take_range, ref_num), dtype=torch.long)
    ref_sel = ref.index_select(0, sample_idx)
    lab_sel = ref_label.index_select(1, sample_idx).reshape(d, -1)
    num_ref, C, H, W = ref_sel.shape
    R = ref_sel.permute(0, 2, 3, 1).reshape(-1, C)
    T = target.reshape(C, -1)[:, cols].contiguous()
    S = R.mm(T)
    if affinity_bf16:
        S = S.to(torch.bfloat16).to(torch.float32)
    S *= temperature
    S = S.softmax(dim=0)
    S = S.contiguous().view(num_ref, H * W, cols.numel())
    if not probability_propagation:
        w_dense = spatial_weight_columns((H, W), sigma1, cols)
        if frame_idx > 15:
            S[:-CONTINUOUS_FRAME] *= spatial_weight_columns((H, W), sigma2, cols)
            S[-CONTINUOUS_FRAME:] *= w_dense
        else:
            S = S.mul(w_dense)
    S = S.view(-1, cols.numel())
    if topk and topk < S.shape[0]:
        kth = S.topk(topk, dim=0).values[-1:]
        S = torch.where(S >= kth, S, torch.zeros_like(S))
    return lab_sel.float().mm(S.float())


class VideoState:
    """The per-video state `inference_single` keeps in module globals
    (src/utils/inference_utils.py:25,33-48)."""

    def __init__(self, first_label, sigma1=8.0, sigma2=21.0, probability_propagation=False, map_scale=None,
                 label_transform=None, out_hw=None):
        """map_scale / label_transform / out_hw: the multi-branch strategies' variants of prepare_first_frame
        (predict.py:130-153) - scaled label map, flipped first label, fixed output size (3-scale)."""
        label = np.asarray(first_label)
        self.H, self.W = label.shape
        k = SCALE if map_scale is None else SCALE * map_scale       # predict.py:138-139,148-149
        self.H_d, self.W_d = int(np.ceil(self.H * k)), int(np.ceil(self.W * k))
        self.d = int(label.max()) + 1                                # predict.py:113
        self.prob = bool(probability_propagation)
        lab0 = label if label_transform is None else np.ascontiguousarray(label_transform(label))
        self.label_history = get_labels(lab0.astype(np.int64), self.d, self.H, self.W, self.H_d, self.W_d)
        self.out_hw = (self.H, self.W) if out_hw is None else tuple(out_hw)
        if self.prob:
            self.w_dense = self.w_sparse = None                      # predict.py:117-118
        else:
            self.w_dense = get_spatial_weight((self.H_d, self.W_d), sigma1)
            self.w_sparse = get_spatial_weight((self.H_d, self.W_d), sigma2)
        self.feats_history = None
        self.frame_idx = 0


def rollout_step(state, features, frame_range, ref_num, temperature, topk=0):
    """One iteration of the loop body, src/utils/inference_utils.py:33-75.

    features (1,C,H_d,W_d) f32.  Frame 0 only seeds the history.  Returns (prediction (d,HW) f32,
    mask (H,W) int64) or (None, None) for frame 0.
    """
    features = torch.as_tensor(features).float()
    if state.frame_idx == 0:
        state.feats_history = features
        state.frame_idx = 1
        return None, None
    pred = predict(state.feats_history, features[0], state.label_history, state.w_dense, state.w_sparse,
                   state.frame_idx, frame_range, ref_num, temperature, state.prob, topk)
    if state.prob:
        new_label = pred.unsqueeze(1)                                # :68
    else:
        new_label = index_to_onehot(torch.argmax(pred, 0), state.d).unsqueeze(1)   # :70
    state.label_history = torch.cat((state.label_history, new_label), 1)           # :71
    state.feats_history = torch.cat((state.feats_history, features), 0)            # :72
    up = torch.nn.functional.interpolate(pred.view(1, state.d, state.H_d, state.W_d),
                                         size=state.out_hw, mode='nearest')         # :74
    mask = torch.argmax(up, 1)[0]                                                   # :75
    state.frame_idx += 1
    state.last_upsampled = up
    return pred, mask


def rollout(first_label, feats, frame_range=40, ref_num=9, temperature=1.0, sigma1=8.0, sigma2=21.0,
            probability_propagation=False, topk=0):
    """`inference_single` for one video with the encoder outputs supplied (T,C,H_d,W_d).
    Returns (preds (T-1,d,HW) f32, masks (T-1,H,W) u8)."""
    st = VideoState(first_label, sigma1, sigma2, probability_propagation)
    preds, masks = [], []
    for t in range(feats.shape[0]):
        p, m = rollout_step(st, torch.as_tensor(feats[t:t + 1]), frame_range, ref_num, temperature, topk)
        if p is not None:
            preds.append(p.numpy())
            masks.append(m.numpy().astype(np.uint8))
    return np.stack(preds), np.stack(masks)


# ---- multi-branch strategies (src/utils/inference_utils.py:90-595) ------------------------------------------------
REDUCTIONS = {'maximum': torch.maximum, 'minimum': torch.minimum, 'mean': lambda x, y: (x + y) / 2.0}   # :18-20

# strategy -> (first-label transform of branch 2, branch 2 uses the scaled map, un-flip applied to branch 2's output)
TWO_BRANCH = {
    'hor-flip': (np.fliplr, False, 'fliplr'),       # :90-187; prepare_first_frame 'hor-flip' predict.py:130-132
    'vert-flip': (np.flipud, False, 'fliplr'),      # :196-298; un-flipped with fliplr all the same (:282)
    '2-scale': (None, True, None),                  # :300-413
    'hor-2-scale': (None, True, 'hflip'),           # flip_pred=True (:389-390); labels are NOT mirrored (:326)
    'multimodel': (None, False, None),              # :416-511
}


def fuse_two(up_a, up_b, prob, reduction, unflip):
    """Per-frame fusion.  up_* are the up-sampled predictions (1,d,H,W).  Label mode (:158-178): argmax each, un-flip
    the second CLASS MAP, element-wise maximum of the class indices.  Probability mode: torch.fliplr acts on the
    (1,d,H,W) tensor, i.e. reverses the class axis (:166); hflip reverses W (:390); reduce, cast to half, argmax."""
    if not prob:
        a, b = torch.argmax(up_a, 1)[0], torch.argmax(up_b, 1)[0]
        if unflip == 'fliplr':
            b = torch.fliplr(b)
        elif unflip == 'hflip':
            b = b.flip(-1)
        return torch.maximum(a, b)
    b = up_b
    if unflip == 'fliplr':
        b = torch.fliplr(b)
    elif unflip == 'hflip':
        b = b.flip(-1)
    return torch.argmax(REDUCTIONS[reduction](up_a, b).half(), 1)[0]


def rollout_two_branch(strategy, first_label, feats_a, feats_b, scale=None, reduction='mean', frame_range=40, ref_num=9,
                       temperature=1.0, sigma1=8.0, sigma2=21.0, probability_propagation=False):
    """The two-chain strategies for one video with the encoder outputs supplied.  -> masks (T-1,H,W) u8."""
    tf, scaled, unflip = TWO_BRANCH[strategy]
    sa = VideoState(first_label, sigma1, sigma2, probability_propagation)
    sb = VideoState(first_label, sigma1, sigma2, probability_propagation, map_scale=scale if scaled else None,
                    label_transform=tf)
    masks = []
    for t in range(feats_a.shape[0]):
        pa, _ = rollout_step(sa, torch.as_tensor(feats_a[t:t + 1]), frame_range, ref_num, temperature)
        pb, _ = rollout_step(sb, torch.as_tensor(feats_b[t:t + 1]), frame_range, ref_num, temperature)
        if pa is not None:
            m = fuse_two(sa.last_upsampled, sb.last_upsampled, probability_propagation, reduction, unflip)
            masks.append(m.numpy().astype(np.uint8))
    return np.stack(masks)


def rollout_3_scale(first_label, feats_per_scale, scales, output_size=(480, 910), frame_range=40, ref_num=9,
                    temperature=1.0, sigma1=8.0, sigma2=21.0, probability_propagation=False):
    """:514-595: one chain per scale (label map ceil(H*0.125*s)), class maps at the fixed output size, maximum of the
    three class maps (:594)."""
    outs = []
    for s, feats in zip(scales, feats_per_scale):
        st = VideoState(first_label, sigma1, sigma2, probability_propagation, map_scale=s, out_hw=output_size)
        masks = []
        for t in range(feats.shape[0]):
            p, m = rollout_step(st, torch.as_tensor(feats[t:t + 1]), frame_range, ref_num, temperature)
            if p is not None:
                masks.append(m.numpy().astype(np.int8))
        outs.append(np.stack(masks))
    return np.maximum(np.maximum(outs[0], outs[1]), outs[2]).astype(np.uint8)


def eval_j(annotation, segmentation):
    """Jaccard index of two binary maps, src/utils/metrics.py:15-45 (no void pixels):
    |A & S| / |A | S|, defined as 1 when the union is empty."""
    a = np.asarray(annotation).astype(bool)
    s = np.asarray(segmentation).astype(bool)
    inter = np.sum(a & s, axis=(-2, -1))
    union = np.sum(a | s, axis=(-2, -1))
    with np.errstate(divide='ignore', invalid='ignore'):
        j = inter / union
    return np.where(union == 0, 1.0, j)


def mask_iou_per_object(ref_masks, test_masks, d):
    """Per-object Jaccard (eval_j) of two index-mask stacks, averaged over frames; the
    'mask IoU delta vs CPU ref' of BASELINE.json is 1 - this."""
    out = []
    for k in range(1, d):
        out.append(float(np.mean(eval_j(ref_masks == k, test_masks == k))))
    return out
