"""Deterministic input builders shared by `make_goldens.py` (which feeds them to the reference)
and by the parity tests (which feed them to the oracle and to the HIP engine).

Everything comes from `numpy.random.RandomState(seed)` - its stream is frozen across numpy
versions - so only seeds/shapes and the reference's OUTPUTS need to be committed.
"""
from pathlib import Path

import numpy as np

G1_CASES = [(40, 9), (40, 5), (10, 9)]
G2_SHAPES = [(30, 54), (60, 107)]


def g2_sample_pairs(h, w, n=64, seed=7):
    rs = np.random.RandomState(seed + h * 1000 + w)
    hw = h * w
    ii = rs.randint(0, hw, size=n)
    # half of the pairs are near neighbours so the weights are not all ~0
    jj = rs.randint(0, hw, size=n)
    near = (ii + rs.randint(-3 * w, 3 * w, size=n)) % hw
    jj[: n // 2] = near[: n // 2]
    return ii.astype(np.int64), jj.astype(np.int64)


def g3_mask(seed=11, H=240, W=427):
    """3 objects + background on a 240p canvas (uint8 class indices)."""
    rs = np.random.RandomState(seed)
    m = np.zeros((H, W), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in (1, 2, 3):
        cy, cx = rs.randint(40, H - 40), rs.randint(60, W - 60)
        ry, rx = rs.randint(15, 50), rs.randint(20, 70)
        if k == 2:
            m[max(cy - ry, 0):cy + ry, max(cx - rx, 0):cx + rx] = k
        else:
            m[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = k
    return m


# predict() cases (G4 tiny, G5 config-1 shape)
PREDICT_CASES = [
    dict(name='g4_tiny', seed=101, C=32, hw=(6, 9), T=45, d=3, scale=0.5,
         frame_idx=[1, 5, 9, 10, 12, 15, 16, 17, 40, 44], range=40, ref_num=9,
         sigma1=8.0, sigma2=21.0, temperature=1.0),
    dict(name='g4_tiny_n5_t07', seed=102, C=32, hw=(5, 7), T=30, d=4, scale=0.6,
         frame_idx=[3, 5, 6, 16, 29], range=10, ref_num=5,
         sigma1=3.0, sigma2=6.0, temperature=0.7),
    dict(name='g5_cfg1', seed=103, C=256, hw=(30, 54), T=2, d=4, scale=0.25,
         frame_idx=[1], range=40, ref_num=9,
         sigma1=8.0, sigma2=21.0, temperature=1.0),
]


def predict_inputs(case):
    """-> ref (T,C,Hd,Wd) f32 (frame f is the target when frame_idx == f), tgt alias, raw label idx (T,HW)."""
    rs = np.random.RandomState(case['seed'])
    Hd, Wd = case['hw']
    feats = (rs.standard_normal((case['T'], case['C'], Hd, Wd)) * case['scale']).astype(np.float32)
    lab = rs.randint(0, case['d'], size=(case['T'], Hd * Wd)).astype(np.int64)
    return feats, feats, lab


def predict_labels(case, prob):
    """Label history (d, T, HW) f32: one-hot in label mode, column-normalised random in probability mode."""
    _, _, lab = predict_inputs(case)
    d = case['d']
    T, HW = lab.shape
    if not prob:
        oh = np.zeros((d, T, HW), dtype=np.float32)
        tt, pp = np.meshgrid(np.arange(T), np.arange(HW), indexing='ij')
        oh[lab, tt, pp] = 1.0
        return oh
    rs = np.random.RandomState(case['seed'] + 5000)
    p = rs.uniform(0.0, 1.0, size=(d, T, HW)).astype(np.float32)
    p /= p.sum(axis=0, keepdims=True)
    return p.astype(np.float32)


# inference_single roll-outs (G6): encoder replaced by seeded features
ROLLOUT_CASES = [
    dict(name='g6_roll', seed=201, video='clipA', C=256, image_hw=(96, 160), T=22, n_obj=3, scale=0.25,
         drift=0.35, range=40, ref_num=9, sigma1=8.0, sigma2=21.0, temperature=1.0),
    dict(name='g6_roll_n5', seed=202, video='clipB', C=64, image_hw=(72, 100), T=20, n_obj=2, scale=0.45,
         drift=0.3, range=6, ref_num=5, sigma1=4.0, sigma2=9.0, temperature=1.0),
]

DAVIS_PALETTE = [0, 0, 0, 128, 0, 0, 0, 128, 0, 128, 128, 0, 0, 0, 128, 128, 0, 128, 0, 128, 128, 128, 128, 128]


def rollout_annotation(case):
    H, W = case['image_hw']
    rs = np.random.RandomState(case['seed'] + 1)
    m = np.zeros((H, W), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(1, case['n_obj'] + 1):
        cy, cx = rs.randint(H // 5, 4 * H // 5), rs.randint(W // 5, 4 * W // 5)
        ry, rx = rs.randint(H // 10, H // 4), rs.randint(W // 10, W // 4)
        m[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = k
    return m


def write_rollout_annotation(case, ann_dir):
    from PIL import Image
    m = rollout_annotation(case)
    p = Path(ann_dir) / case['video']
    p.mkdir(parents=True, exist_ok=True)
    im = Image.fromarray(m, mode='P')
    pal = DAVIS_PALETTE + [0] * (768 - len(DAVIS_PALETTE))
    im.putpalette(pal)
    im.save(p / '00000.png')
    return p / '00000.png'


def rollout_features(case):
    """Temporally correlated features: a per-class prototype field plus drifting noise, so that the
    propagated masks are non-trivial but stable (argmax margins far above float noise)."""
    H, W = case['image_hw']
    Hd, Wd = int(np.ceil(H * 0.125)), int(np.ceil(W * 0.125))
    rs = np.random.RandomState(case['seed'])
    C, T = case['C'], case['T']
    m = rollout_annotation(case)
    src_r = (np.arange(Hd) * H) // Hd
    src_c = (np.arange(Wd) * W) // Wd
    md = m[src_r][:, src_c]                                  # nearest down-sample (Hd, Wd)
    protos = rs.standard_normal((case['n_obj'] + 1, C)).astype(np.float32)
    noise = rs.standard_normal((C, Hd, Wd)).astype(np.float32)
    feats = np.empty((T, C, Hd, Wd), dtype=np.float32)
    for t in range(T):
        # the scene drifts one feature-map column every 2 frames (and one row every 5)
        mdt = np.roll(np.roll(md, t // 2, axis=1), t // 5, axis=0)
        base = protos[mdt].transpose(2, 0, 1)                # (C,Hd,Wd)
        noise = np.sqrt(1 - case['drift'] ** 2) * noise + case['drift'] * rs.standard_normal((C, Hd, Wd))
        feats[t] = (0.6 * base + 1.0 * noise) * case['scale']
    return feats.astype(np.float32)


# ---- multi-branch strategies (G7): one clip, the encoder replaced by seeded features per (branch, map size) ----
STRATEGY_CASE = dict(name='g7', seed=401, video='clipS', C=256, image_hw=(96, 160), T=19, n_obj=3, scale=0.25, drift=0.3,
                     range=40, ref_num=9, sigma1=8.0, sigma2=21.0, temperature=1.0, scale2=1.15, out3=(480, 910))
STRATEGIES_2 = ('hor-flip', 'vert-flip', '2-scale', 'hor-2-scale', 'multimodel')
# (strategy, probability, fusion) combinations stored as goldens
STRATEGY_RUNS = [(s, False, 'mean') for s in STRATEGIES_2] + [(s, True, 'mean') for s in STRATEGIES_2] + \
                [('hor-flip', True, 'maximum'), ('2-scale', True, 'minimum')]


def strategy_map_hw(case, scale=None):
    H, W = case['image_hw']
    k = 0.125 if scale is None else 0.125 * scale
    return int(np.ceil(H * k)), int(np.ceil(W * k))


def strategy_input_hw(case, strategy, branch):
    """Size of the image the dataset hands to the encoder for a branch (reference datasets.py:156-162)."""
    H, W = case['image_hw']
    if strategy in ('2-scale', 'hor-2-scale') and branch == 1:
        return int(np.ceil(H * case['scale2'])), int(np.ceil(W * case['scale2']))
    return H, W


def strategy_features(case, branch_seed, map_hw, flip=None):
    """(T,C,Hd,Wd) f32 features of one chain: the clip's class layout (nearest down-sample of the annotation onto the
    chain's map, mirrored / flipped for the flipped branches) as prototypes + temporally correlated noise."""
    H, W = case['image_hw']
    Hd, Wd = map_hw
    rs = np.random.RandomState(case['seed'] + 17 * branch_seed)
    C, T = case['C'], case['T']
    m = rollout_annotation(case)
    if flip == 'w':
        m = m[:, ::-1]
    elif flip == 'h':
        m = m[::-1, :]
    md = m[(np.arange(Hd) * H) // Hd][:, (np.arange(Wd) * W) // Wd]
    protos = np.random.RandomState(case['seed']).standard_normal((case['n_obj'] + 1, C)).astype(np.float32)
    noise = rs.standard_normal((C, Hd, Wd)).astype(np.float32)
    feats = np.empty((T, C, Hd, Wd), dtype=np.float32)
    for t in range(T):
        mdt = np.roll(np.roll(md, (t // 2) * (-1 if flip == 'w' else 1), axis=1), (t // 5) * (-1 if flip == 'h' else 1),
                      axis=0)
        base = protos[mdt].transpose(2, 0, 1)
        noise = np.sqrt(1 - case['drift'] ** 2) * noise + case['drift'] * rs.standard_normal((C, Hd, Wd))
        feats[t] = (0.6 * base + 1.0 * noise) * case['scale']
    return feats.astype(np.float32)


def strategy_branch_features(case, strategy):
    """-> [features of branch 0, features of branch 1] for a two-branch strategy."""
    flip = {'hor-flip': 'w', 'vert-flip': 'h', 'hor-2-scale': 'w'}.get(strategy)
    scaled = strategy in ('2-scale', 'hor-2-scale')
    a = strategy_features(case, 0, strategy_map_hw(case))
    b = strategy_features(case, 1, strategy_map_hw(case, case['scale2'] if scaled else None), flip)
    return [a, b]


def three_scale_features(case):
    scales = (0.9, 1.0, case['scale2'])
    return scales, [strategy_features(case, 10 + i, strategy_map_hw(case, s)) for i, s in enumerate(scales)]


# ---- J / boundary-map fixtures (G8) ----
def metric_masks(seed=501, n=6, H=48, W=64):
    """(annotation, segmentation, void) boolean stacks: blobs, a shifted copy of them with noise, a void band; frame 4 has an
    empty annotation, frame 5 is empty in both (union == 0 -> J = 1)."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    ann = np.zeros((n, H, W), dtype=bool)
    seg = np.zeros((n, H, W), dtype=bool)
    for i in range(n):
        cy, cx, ry, rx = rs.randint(12, H - 12), rs.randint(16, W - 16), rs.randint(5, 14), rs.randint(6, 18)
        ann[i] = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        dy, dx = rs.randint(-3, 4), rs.randint(-4, 5)
        seg[i] = np.roll(np.roll(ann[i], dy, 0), dx, 1) ^ (rs.uniform(size=(H, W)) < 0.01)
    ann[4] = False
    ann[5] = False
    seg[5] = False
    void = np.zeros((n, H, W), dtype=bool)
    void[:, :, W // 2 - 2: W // 2 + 2] = True
    return ann, seg, void


def onehot_indices(seed=5, n=40, d=5):
    return np.random.RandomState(seed).randint(0, d, size=n).astype(np.int64)


# ---- encoder fixtures: deterministic weights keyed by parameter NAME (order independent) ----
def fill_state_dict(sd, seed=301):
    """Return {key: numpy array} for every entry of a VOSNet state dict: the same values whatever module
    registration order the implementation uses."""
    import zlib
    out = {}
    for key, val in sd.items():
        shape = tuple(val.shape)
        rs = np.random.RandomState((zlib.crc32(key.encode()) + seed) % (2 ** 31))
        if key.endswith('num_batches_tracked'):
            out[key] = np.zeros(shape, dtype=np.int64)
        elif key.endswith('running_var'):
            out[key] = rs.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif key.endswith('running_mean'):
            out[key] = (rs.standard_normal(shape) * 0.1).astype(np.float32)
        elif len(shape) == 1 and key.endswith('weight'):
            out[key] = rs.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif len(shape) == 1:
            out[key] = (rs.standard_normal(shape) * 0.1).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            out[key] = (rs.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
    return out


def encoder_input(seed=302, H=64, W=96):
    return np.random.RandomState(seed).standard_normal((1, 3, H, W)).astype(np.float32)
