"""The multi-branch inference strategies (hor-flip, vert-flip, 2-scale, hor-2-scale, multimodel, 3-scale) of the package's
inference_utils - the engine's chains + the fusion - against the masks the REFERENCE's own strategy functions wrote for the
same seeded encoder outputs (tests/golden/make_goldens.py, G7).  GPU only.

Tolerance: the golden run is f32 end to end, the engine contracts in bf16 -> at least 99 % of the pixels of every clip
must carry the same class, and the per-object IoU with the reference's masks must be >= 0.97 (the fused masks are
maxima of class indices, so a one-pixel boundary shift of either branch shows up in the result)."""
import numpy as np
import pytest
import torch

import inputs as gin
from oracle import vos_oracle as vo

pytestmark = pytest.mark.gpu


class FakeEncoder:
    """Stands in for VOSNet: hands back the seeded features of (frame, branch) in call order, batched like the input."""

    def __init__(self, feats, dev):
        self.feats = torch.from_numpy(np.ascontiguousarray(feats)).to(dev)
        self.i = 0

    def __call__(self, x):
        n = x.shape[0]
        f = self.feats[self.i:self.i + n]
        self.i += n
        assert (int(np.ceil(x.shape[2] / 8)), int(np.ceil(x.shape[3] / 8))) == tuple(f.shape[-2:])
        return f


def _read_masks(save_dir, case):
    from PIL import Image
    return np.stack([np.asarray(Image.open(save_dir / case['video'] / f'{i:05d}.png')).astype(np.uint8)
                     for i in range(1, case['T'])])


def _check(masks, g, d):
    assert masks.shape == g.shape
    same = float(np.mean(masks == g))
    assert same >= 0.99, f'{(1 - same) * 100:.2f} % of the pixels differ'
    present = [k for k in range(1, d) if (g == k).any()]
    iou = [float(np.mean(vo.eval_j(g == k, masks == k))) for k in present]
    assert min(iou) >= 0.97, iou


@pytest.fixture()
def on_gpu(vos):
    assert torch.cuda.is_available()
    vos.Config.DEVICE = torch.device('cuda', 0)
    return vos


@pytest.mark.parametrize('strategy,prob,fusion', gin.STRATEGY_RUNS)
@pytest.mark.parametrize('batch', [1, 4])
def test_two_branch_strategy_vs_reference(on_gpu, goldens, tmp_path, strategy, prob, fusion, batch):
    vos = on_gpu
    iu = vos.inference_utils
    case = gin.STRATEGY_CASE
    dev = vos.Config.DEVICE
    H, W = case['image_hw']
    T = case['T']
    fa, fb = gin.strategy_branch_features(case, strategy)
    gin.write_rollout_annotation(case, tmp_path / 'ann')
    sizes = [gin.strategy_input_hw(case, strategy, b) for b in (0, 1)]
    if strategy == 'multimodel':
        loader = [(torch.zeros(1, 3, H, W), (case['video'],)) for _ in range(T)]
    else:
        loader = [([torch.zeros(1, 3, *sizes[0]), torch.zeros(1, 3, *sizes[1])], (case['video'],)) for _ in range(T)]
    m0, m1 = FakeEncoder(fa, dev), FakeEncoder(fb, dev)
    head = (loader, T, tmp_path / 'ann', case['video'], str(tmp_path / 'save'), case['sigma1'], case['sigma2'],
            case['range'], case['ref_num'], case['temperature'], prob)
    stats = {}
    opts = dict(stats=stats, encoder_batch=batch)
    # the package's two-branch loop calls models[b] once per branch and batch: give each branch its own fake encoder
    iu._inference_two_branch(strategy, [m0, m1], *head, case['scale2'], fusion, True, **opts)
    assert stats['frames'] == T and stats['videos'] == 1
    masks = _read_masks(tmp_path / 'save', case)
    _check(masks, goldens[f"g7_{strategy}_{'prob_' + fusion if prob else 'label'}_masks"], case['n_obj'] + 1)
    ann0 = np.asarray(__import__('PIL.Image', fromlist=['Image']).open(tmp_path / 'save' / case['video'] / '00000.png'))
    assert np.array_equal(ann0, gin.rollout_annotation(case))


@pytest.mark.parametrize('prob', [False, True])
def test_three_scale_vs_reference(on_gpu, goldens, tmp_path, prob):
    vos = on_gpu
    case = gin.STRATEGY_CASE
    H, W = case['image_hw']
    T = case['T']
    scales, f3 = gin.three_scale_features(case)
    gin.write_rollout_annotation(case, tmp_path / 'ann')

    class ThreePass:
        """one fake encoder per pass; the pass is recognised by the size of the (pre-scaled) input"""

        def __init__(self):
            self.enc = {tuple(f.shape[-2:]): FakeEncoder(f, vos.Config.DEVICE) for f in f3}

        def __call__(self, x):
            return self.enc[(int(np.ceil(x.shape[2] / 8)), int(np.ceil(x.shape[3] / 8)))](x)

    loader = [(torch.zeros(1, 3, H, W), (case['video'],)) for _ in range(T)]
    vos.inference_utils.inference_3_scale(ThreePass(), loader, T, tmp_path / 'ann', case['video'], str(tmp_path / 'save'),
                                          case['sigma1'], case['sigma2'], case['range'], case['ref_num'],
                                          case['temperature'], prob, case['scale2'], True, encoder_batch=8)
    masks = _read_masks(tmp_path / 'save', case)
    _check(masks, goldens[f"g7_3-scale_{'prob' if prob else 'label'}_masks"], case['n_obj'] + 1)


def test_map_size_mismatch_is_reported(on_gpu, tmp_path):
    """A second-branch feature map that is not ceil(H*0.125*scale) is an error (the reference dies in mm())."""
    vos = on_gpu
    case = gin.STRATEGY_CASE
    H, W = case['image_hw']
    fa, _ = gin.strategy_branch_features(case, '2-scale')
    gin.write_rollout_annotation(case, tmp_path / 'ann')
    loader = [([torch.zeros(1, 3, H, W), torch.zeros(1, 3, H, W)], (case['video'],)) for _ in range(2)]
    m = FakeEncoder(fa, vos.Config.DEVICE)
    m2 = FakeEncoder(fa, vos.Config.DEVICE)
    with pytest.raises(vos.VospropError, match='label map'):
        vos.inference_utils._inference_two_branch('2-scale', [m, m2], loader, 2, tmp_path / 'ann', case['video'], None, 8.0,
                                                  21.0, 40, 9, 1.0, False, 1.15, 'mean', True)
