"""The oracle (torch-CPU restatement + plain-C restatement) against the golden vectors that were
produced by running the reference itself (tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest
import torch

import inputs as gin
from oracle import c_oracle
from oracle import vos_oracle as vo


def relerr(a, b):
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-30)))


@pytest.mark.parametrize('rng,nref', gin.G1_CASES)
def test_sample_frames(goldens, rng, nref):
    g = goldens[f'g1_sample_r{rng}_n{nref}']
    for fi in range(1, 121):
        want = [int(x) for x in g[fi - 1] if x >= 0]
        assert vo.sample_frames(fi, rng, nref) == want
        assert c_oracle.sample_frames(fi, rng, nref) == want
    # the ring-capacity claim of DESIGN.md: nothing older than frame_idx-4-range is ever sampled
    for fi in range(1, 121):
        assert min(vo.sample_frames(fi, rng, nref)) >= max(fi - 4 - rng, 0) or fi <= nref


def test_spatial_weight_full(goldens):
    g = goldens['g2_w_4x6_s8']
    assert np.array_equal(vo.get_spatial_weight((4, 6), 8.0).numpy(), g)
    c = c_oracle.spatial_weight(4, 6, 8.0)
    assert relerr(c, g) < 1e-6


@pytest.mark.parametrize('h,w', gin.G2_SHAPES)
@pytest.mark.parametrize('sigma', [8.0, 21.0])
def test_spatial_weight_samples(goldens, h, w, sigma):
    ii, jj = gin.g2_sample_pairs(h, w)
    g = goldens[f'g2_w_{h}x{w}_s{int(sigma)}']
    full = c_oracle.spatial_weight(h, w, sigma)
    assert np.allclose(full[ii, jj], g, rtol=2e-6, atol=0)
    if h * w <= 2000:
        assert np.array_equal(vo.get_spatial_weight((h, w), sigma).numpy()[ii, jj], g)


def test_get_labels(goldens):
    mask = gin.g3_mask()
    H, W = mask.shape
    Hd, Wd = vo.feature_map_size(H, W)
    d = int(mask.max()) + 1
    g = goldens['g3_labels']
    assert (Hd, Wd) == (30, 54) and g.shape == (d, 1, Hd * Wd)
    assert np.array_equal(vo.get_labels(mask.astype(np.int64), d, H, W, Hd, Wd).numpy(), g)
    assert np.array_equal(c_oracle.get_labels(mask, Hd, Wd, d), g[:, 0].astype(np.float32))


def test_onehot(goldens):
    assert np.array_equal(vo.index_to_onehot(gin.onehot_indices(), 5).numpy(), goldens['onehot'])


@pytest.mark.parametrize('case', gin.PREDICT_CASES, ids=lambda c: c['name'])
@pytest.mark.parametrize('prob', [False, True])
def test_predict(goldens, case, prob):
    ref, tgt, _ = gin.predict_inputs(case)
    lh = gin.predict_labels(case, prob)
    Hd, Wd = case['hw']
    wd = vo.get_spatial_weight((Hd, Wd), case['sigma1'])
    ws = vo.get_spatial_weight((Hd, Wd), case['sigma2'])
    for fi in case['frame_idx']:
        g = goldens[f"{case['name']}_{'prob' if prob else 'label'}_f{fi}"]
        p = vo.predict(ref[:fi], tgt[fi], lh[:, :fi], None if prob else wd, None if prob else ws, fi,
                       case['range'], case['ref_num'], case['temperature'], prob).numpy()
        # same op order as the reference; BLAS blocking depends on the thread count, hence not bit-equal
        assert relerr(p, g) < 5e-6, f'torch oracle differs at frame_idx={fi}: {relerr(p, g)}'
        c = c_oracle.predict(ref[:fi], tgt[fi], lh[:, :fi], fi, case['range'], case['ref_num'],
                             case['temperature'], case['sigma1'], case['sigma2'], prob)
        assert relerr(c, g) < 2e-5, f'C oracle differs at frame_idx={fi}: {relerr(c, g)}'


@pytest.mark.parametrize('case', gin.ROLLOUT_CASES, ids=lambda c: c['name'])
@pytest.mark.parametrize('prob', [False, True])
def test_rollout(goldens, case, prob):
    tag = f"{case['name']}_{'prob' if prob else 'label'}"
    ann = gin.rollout_annotation(case)
    feats = gin.rollout_features(case)
    preds, masks = vo.rollout(ann, feats, case['range'], case['ref_num'], case['temperature'], case['sigma1'],
                              case['sigma2'], prob)
    assert np.array_equal(masks, goldens[f'{tag}_masks'])
    assert relerr(preds, goldens[f'{tag}_preds']) < 2e-5
    # C glue: argmax + nearest up-sample of the golden predictions reproduces the golden masks
    H, W = case['image_hw']
    Hd, Wd = vo.feature_map_size(H, W)
    for i in range(preds.shape[0]):
        _, m = c_oracle.argmax_upsample(goldens[f'{tag}_preds'][i], Hd, Wd, H, W)
        assert np.array_equal(m, goldens[f'{tag}_masks'][i])


def test_eval_j():
    a = np.zeros((4, 5), bool); a[1:3, 1:4] = True
    b = np.zeros((4, 5), bool); b[2:4, 2:5] = True
    assert vo.eval_j(a, b) == pytest.approx(2 / 10)
    assert vo.eval_j(np.zeros((3, 3)), np.zeros((3, 3))) == 1.0


@pytest.mark.parametrize('strategy,prob,fusion', gin.STRATEGY_RUNS)
def test_two_branch_strategies(goldens, strategy, prob, fusion):
    """oracle.rollout_two_branch against the reference's inference_hor_flip / _ver_flip / _2_scale / _multimodel."""
    case = gin.STRATEGY_CASE
    fa, fb = gin.strategy_branch_features(case, strategy)
    masks = vo.rollout_two_branch(strategy, gin.rollout_annotation(case), fa, fb, scale=case['scale2'], reduction=fusion,
                                  frame_range=case['range'], ref_num=case['ref_num'], temperature=case['temperature'],
                                  sigma1=case['sigma1'], sigma2=case['sigma2'], probability_propagation=prob)
    g = goldens[f"g7_{strategy}_{'prob_' + fusion if prob else 'label'}_masks"]
    assert masks.shape == g.shape
    assert np.mean(masks == g) > 0.9995          # thread-count dependent f32 sums may flip a near-tie pixel
    assert len(np.unique(g)) >= 3                # the fixture is not degenerate


@pytest.mark.parametrize('prob', [False, True])
def test_three_scale_strategy(goldens, prob):
    case = gin.STRATEGY_CASE
    scales, feats = gin.three_scale_features(case)
    masks = vo.rollout_3_scale(gin.rollout_annotation(case), feats, scales, output_size=case['out3'],
                               frame_range=case['range'], ref_num=case['ref_num'], temperature=case['temperature'],
                               sigma1=case['sigma1'], sigma2=case['sigma2'], probability_propagation=prob)
    g = goldens[f"g7_3-scale_{'prob' if prob else 'label'}_masks"]
    assert masks.shape == g.shape == (case['T'] - 1,) + tuple(case['out3'])
    assert np.mean(masks == g) > 0.9995


@pytest.mark.parametrize('prob', [False, True])
@pytest.mark.parametrize('fi,topk', [(5, 0), (20, 0), (20, 7)])
def test_predict_columns_is_a_slice_of_predict(prob, fi, topk):
    """The column-subset form the full-size GPU tests use (tests/test_gpu_configs.py) equals the pinned oracle on those columns -
    same ops, same order, so bit for bit up to the mm kernel's blocking (<= 1e-6 relative)."""
    if prob and topk:
        pytest.skip('top-k is label-propagation only')
    rs = np.random.RandomState(fi * 31 + topk)
    Hd, Wd, T, d = 9, 13, fi + 1, 4
    feats = (rs.standard_normal((T, 64, Hd, Wd)) * 0.4).astype(np.float32)
    lab = rs.randint(0, d, size=(T, Hd * Wd))
    oh = np.zeros((d, T, Hd * Wd), np.float32)
    tt, pp = np.meshgrid(np.arange(T), np.arange(Hd * Wd), indexing='ij')
    oh[lab, tt, pp] = 1.0
    wd, ws = vo.get_spatial_weight((Hd, Wd), 8.0), vo.get_spatial_weight((Hd, Wd), 21.0)
    cols = np.array([0, 1, 12, 13, 57, 58, 103, Hd * Wd - 1])
    assert torch.equal(vo.spatial_weight_columns((Hd, Wd), 8.0, cols), wd[:, cols])
    full = vo.predict(feats[:fi], feats[fi], oh[:, :fi], None if prob else wd, None if prob else ws, fi, 40, 9, 1.0, prob,
                      topk=topk).numpy()
    part = vo.predict_columns(feats[:fi], feats[fi], oh[:, :fi], 8.0, 21.0, fi, 40, 9, 1.0, prob, cols, topk=topk).numpy()
    assert np.allclose(part, full[:, cols], rtol=1e-6, atol=1e-12)
