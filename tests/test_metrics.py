"""J / F evaluation: semi-supervised-vos_amd/metrics.py and evaluation.py.

  * eval_j and the boundary map: bit for bit against the reference's own functions (goldens G8, made by running the reference);
  * f_measure: the reference's needs scikit-image (absent here), so F is pinned by tests/golden/f_measure_fixture.npz - values
    computed by an independent brute-force method (explicit disk stamping, tests/golden/make_f_fixture.py) - and by closed forms;
  * the vectorised per-frame scorer and the chunked evaluation command against the per-object functions.
CPU only."""
import importlib
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import inputs as gin

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / 'tests' / 'golden'


@pytest.fixture(scope='module')
def M():
    return importlib.import_module('semi-supervised-vos_amd.metrics')


def test_eval_j_and_boundary_map_match_the_reference(goldens, M):
    """metrics.eval_j / _seg2bmap against the reference's own functions (goldens G8)."""
    ann, seg, void = gin.metric_masks()
    assert np.array_equal(np.asarray(M.eval_j(ann, seg), dtype=np.float64), goldens['g8_j_stack'])
    assert np.array_equal(np.asarray(M.eval_j(ann, seg, void), dtype=np.float64), goldens['g8_j_stack_void'])
    single = np.asarray([float(M.eval_j(ann[i], seg[i])) for i in range(ann.shape[0])])
    assert np.array_equal(single, goldens['g8_j_single'])
    assert goldens['g8_j_stack'][5] == 1.0 and goldens['g8_j_stack'][4] == 0.0        # empty union / empty annotation
    bm = np.stack([M._seg2bmap(seg[i]) for i in range(seg.shape[0])]).astype(np.uint8)
    assert np.array_equal(bm, goldens['g8_bmap'])


def test_f_measure_known_answers(M):
    """The reference's F needs scikit-image (absent): pinned by known answers - identical masks, empty masks, a shift inside /
    outside the matching radius, and the flat-dilation identity the restatement rests on."""
    from scipy.ndimage import binary_dilation, grey_dilation
    H, W = 120, 160                                   # diagonal 200 -> bound = ceil(0.008 * 200) = 2 pixels
    a = np.zeros((H, W), bool)
    a[30:80, 40:110] = True
    assert M.f_measure(a, a) == 1.0
    z = np.zeros_like(a)
    assert M.f_measure(z, z) == 1.0 and M.f_measure(z, a) == 0.0 and M.f_measure(a, z) == 0.0
    assert M.f_measure(np.roll(a, 2, 1), a) == 1.0        # every boundary pixel within 2 pixels of the other boundary
    far = M.f_measure(np.roll(a, 12, 1), a)
    assert 0.0 < far < 0.8                                 # only the horizontal edges still match
    assert np.allclose(M.eval_f(np.stack([a, a]), np.stack([a, np.roll(a, 12, 1)])), [1.0, far])
    j, f = M.evaluate_segmentation(a, np.roll(a, 12, 1))
    assert abs(j - (50 * 58) / (50 * 82)) < 1e-12 and f == far
    d = M.disk(3.0)
    assert d.shape == (7, 7) and d.sum() == 29 and d[0, 3] and not d[0, 2]
    b = M._seg2bmap(a)
    assert np.array_equal(binary_dilation(b, structure=d), grey_dilation(b.astype(np.uint8), footprint=d, mode='constant') > 0)


def test_evaluation_command(tmp_path):
    """`main.py evaluation -g ... -c ...` over a tiny tree: perfect results score 1, and the J of a known shift comes out."""
    from PIL import Image
    E = importlib.import_module('semi-supervised-vos_amd.evaluation')
    m = gin.rollout_annotation(gin.STRATEGY_CASE)
    for root, shift in (('gt', 0), ('same', 0), ('moved', 5)):
        for i in range(3):
            d = tmp_path / root / 'v'
            d.mkdir(parents=True, exist_ok=True)
            im = Image.fromarray(np.roll(m, shift, 1), mode='P')
            im.putpalette(gin.DAVIS_PALETTE + [0] * (768 - len(gin.DAVIS_PALETTE)))
            im.save(d / f'{i:05d}.png')
    j, f, jf = E.evaluation_command_impl(tmp_path / 'gt', tmp_path / 'same', disable=True, processes=2)
    assert (j, f, jf) == (1.0, 1.0, 1.0)
    j2, f2, jf2 = E.evaluation_command_impl(tmp_path / 'gt', tmp_path / 'moved', disable=True, processes=2)
    assert 0.3 < j2 < 1.0 and 0.0 < f2 < 1.0 and abs(jf2 - (j2 + f2) / 2) < 1e-12
    out = subprocess.run([sys.executable, str(ROOT / 'main.py'), 'evaluation', '-g', str(tmp_path / 'gt'), '-c',
                          str(tmp_path / 'same')], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'j_mean=1.0' in out.stdout, out.stdout + out.stderr


def test_f_measure_matches_the_bruteforce_fixture(M):
    """Distance-transform matching == explicit disk dilation, on random blobs with shifts, holes, void pixels, relative and absolute
    thresholds (fixture values were NOT computed with this package or scipy)."""
    fx = np.load(GOLD / 'f_measure_fixture.npz')
    for i in range(int(fx['n'])):
        void = fx[f'void{i}'] if f'void{i}' in fx.files else None
        got = M.f_measure(fx[f'fg{i}'], fx[f'gt{i}'], void, bound_th=float(fx[f'th{i}']))
        assert abs(got - float(fx[f'f{i}'])) < 1e-12, (i, got, float(fx[f'f{i}']))
        assert np.array_equal(M.boundary_map(fx[f'fg{i}']), fx[f'bmap{i}'])


def test_within_radius_is_the_disk_dilation(M):
    from scipy.ndimage import binary_dilation
    rs = np.random.RandomState(3)
    for shape, r in [((40, 57), 2), ((31, 31), 5), ((64, 20), 1), ((17, 90), 7.5)]:
        b = rs.rand(*shape) < 0.02
        assert np.array_equal(M.within_radius(b, r), binary_dilation(b, structure=M.disk(r)))
    assert not M.within_radius(np.zeros((5, 5), bool), 3).any()


def test_frame_scores_equal_the_per_object_functions(M):
    """One joint histogram + boundary stacks for all objects of a frame == evaluate_segmentation object by object, including the
    reference's positional pairing when an object is missing on one side."""
    rs = np.random.RandomState(11)
    gt = np.zeros((72, 100), np.uint8)
    gt[10:40, 10:50] = 1
    gt[35:65, 45:90] = 2
    gt[5:20, 70:95] = 3
    seg = np.roll(gt, 3, 1)
    seg[seg == 3] = 0                                    # object 3 missing from the result: pairs are (0,0), (1,1), (2,2)
    pairs = list(zip(np.unique(gt).tolist(), np.unique(seg).tolist()))
    assert pairs == [(0, 0), (1, 1), (2, 2)]
    got = M.frame_scores(gt, seg, pairs)
    want = np.array([M.evaluate_segmentation(gt == g, seg == s) for g, s in pairs], dtype=np.float64)
    assert np.allclose(got, want, rtol=0, atol=1e-15)
    noise = rs.randint(0, 4, size=gt.shape).astype(np.uint8)
    pairs = [(0, 0), (1, 1), (2, 2), (3, 3), (1, 3)]
    assert np.allclose(M.frame_scores(gt, noise, pairs), [M.evaluate_segmentation(gt == g, noise == s) for g, s in pairs], atol=1e-15)
