"""The f32 parity path (VOSPROP_PREC_F32: f32 features in the ring, v_mfma_f32_32x32x2_f32, f32 prior, csrc/prop_f32.h) against the
reference's UN-ROUNDED golden vectors (tests/golden/reference_goldens.npz: outputs of the reference's own fp32 CPU code,
src/model/predict.py:19-71, src/utils/inference_utils.py:23-87) - the tolerance SURVEY.md section 8(c) states for it:
    |out - golden| <= 1e-4 relative on well-conditioned columns (column total >= 1e-6), masks IoU >= 0.999.
The bf16 path needs 3e-2 absolute against the same goldens (tests/test_gpu_parity.py); this one does not round the features."""
import importlib
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import inputs as gin
from oracle import vos_oracle as vo

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
F32 = 1      # VOSPROP_PREC_F32


def pad_c(x, C=256, axis=1):
    pad = [(0, 0)] * x.ndim
    pad[axis] = (0, C - x.shape[axis])
    return np.pad(x, pad)


def check_rel(got, want, rel=1e-4, well=1e-6):
    colsum = want.sum(0, keepdims=True)
    wellc = np.broadcast_to(colsum >= well, want.shape)
    err = np.abs(got - want)
    scale = np.maximum(np.abs(want), colsum * 1e-2)
    worst = float(np.max(err[wellc] / (scale[wellc] + 1e-30))) if wellc.any() else 0.0
    assert worst <= rel, f'max rel err {worst:.3e}'
    assert np.all(err[~wellc] <= 1e-8 + rel * 50 * np.abs(want[~wellc]))
    return worst


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch.device('cuda', 0)


@pytest.mark.parametrize('case', gin.PREDICT_CASES, ids=lambda c: c['name'])
@pytest.mark.parametrize('prob', [False, True])
def test_predict_f32_vs_unrounded_reference_goldens(vos, goldens, dev, case, prob):
    """G4 / G5: predict() at the tiny and the config-1 shape, both modes, both sides of the frame_idx > 15 switch."""
    ref, tgt, _ = gin.predict_inputs(case)
    lh = gin.predict_labels(case, prob)
    Hd, Wd = case['hw']
    eng = vos.PropagationEngine(Hd, Wd, device=0, ref_num=case['ref_num'], frame_range=case['range'], precision=F32)
    ref_dev = torch.from_numpy(pad_c(ref)).to(dev)
    lab_dev = torch.from_numpy(lh).to(dev)
    for fi in case['frame_idx']:
        got = eng.predict(ref_dev[:fi], ref_dev[fi], lab_dev[:, :fi], fi, case['range'], case['ref_num'],
                          case['temperature'], case['sigma1'], case['sigma2'], prob).cpu().numpy()
        g = goldens[f"{case['name']}_{'prob' if prob else 'label'}_f{fi}"]
        check_rel(got, g)
    eng.close()


@pytest.mark.parametrize('case', gin.ROLLOUT_CASES, ids=lambda c: c['name'])
@pytest.mark.parametrize('prob', [False, True])
def test_rollout_f32_vs_unrounded_reference_goldens(vos, goldens, dev, case, prob):
    """G6: begin_video + step over a whole clip == the reference's inference_single: identical masks, predictions to 1e-4."""
    tag = f"{case['name']}_{'prob' if prob else 'label'}"
    ann = gin.rollout_annotation(case)
    feats = gin.rollout_features(case)
    H, W = case['image_hw']
    Hd, Wd = vos.feature_map_size(H, W)
    eng = vos.PropagationEngine(Hd, Wd, device=0, ref_num=case['ref_num'], frame_range=case['range'], sigma1=case['sigma1'],
                                sigma2=case['sigma2'], temperature=case['temperature'], probability=prob, precision=F32)
    d = eng.begin_video(ann)
    fd = torch.from_numpy(pad_c(feats)).to(dev)
    preds, masks = [], []
    for t in range(feats.shape[0]):
        p, m = eng.step(fd[t])
        if t:
            preds.append(p.cpu().numpy())
            masks.append(m.cpu().numpy())
    preds, masks = np.stack(preds), np.stack(masks)
    gm, gp = goldens[f'{tag}_masks'], goldens[f'{tag}_preds']
    iou = vo.mask_iou_per_object(gm, masks, d)
    assert min(iou) >= 0.999, iou
    assert np.mean(masks != gm) <= 1e-3
    # predictions: frame by frame while the label histories still agree exactly (label mode feeds arg-maxes back)
    for t in range(preds.shape[0]):
        check_rel(preds[t], gp[t], rel=2e-4 if prob else 1e-4)
        if not prob and not np.array_equal(preds[t].argmax(0), gp[t].argmax(0)):
            break
    eng.close()


def test_f32_full_480p_against_the_oracle_columns(vos, dev):
    """BASELINE config 2 shape with UN-ROUNDED f32 features, N = 9, frame_idx = 20, both modes, on a column subset."""
    Hd, Wd, T, d, fi = 60, 107, 21, 4, 20
    rs = np.random.RandomState(8)
    feats = (rs.standard_normal((T, 256, Hd, Wd)) * 0.25).astype(np.float32)
    lab = rs.randint(0, d, size=(T, Hd * Wd))
    oh = np.zeros((d, T, Hd * Wd), dtype=np.float32)
    tt, pp = np.meshgrid(np.arange(T), np.arange(Hd * Wd), indexing='ij')
    oh[lab, tt, pp] = 1.0
    cols = np.unique(np.concatenate([np.arange(32), np.arange(6400, Hd * Wd), rs.choice(Hd * Wd, 256, replace=False)]))
    eng = vos.PropagationEngine(Hd, Wd, device=0, precision=F32)
    fd, ld = torch.from_numpy(feats).to(dev), torch.from_numpy(oh).to(dev)
    for prob in (False, True):
        got = eng.predict(fd[:fi], fd[fi], ld[:, :fi], fi, 40, 9, 1.0, 8.0, 21.0, prob).cpu().numpy()
        want = vo.predict_columns(feats[:fi], feats[fi], oh[:, :fi], 8.0, 21.0, fi, 40, 9, 1.0, prob, cols).numpy()
        check_rel(got[:, cols], want)
    eng.close()


def test_f32_edge_shapes_and_sources(vos, dev):
    """Ragged maps, d = 32, N = 1, peaky logits (the running max moves late), and every feature source type / layout."""
    for (Hd, Wd, T, d, fi, scale) in [(5, 7, 12, 32, 11, 0.25), (1, 1, 3, 1, 2, 0.25), (17, 19, 6, 3, 5, 1.0), (8, 8, 2, 2, 1, 0.25)]:
        rs = np.random.RandomState(Hd * 100 + Wd)
        feats = (rs.standard_normal((T, 256, Hd, Wd)) * scale).astype(np.float32)
        lab = rs.randint(0, d, size=(T, Hd * Wd))
        oh = np.zeros((d, T, Hd * Wd), dtype=np.float32)
        tt, pp = np.meshgrid(np.arange(T), np.arange(Hd * Wd), indexing='ij')
        oh[lab, tt, pp] = 1.0
        wd, ws = vo.get_spatial_weight((Hd, Wd), 8.0), vo.get_spatial_weight((Hd, Wd), 21.0)
        eng = vos.PropagationEngine(Hd, Wd, device=0, precision=F32)
        fd, ld = torch.from_numpy(feats).to(dev), torch.from_numpy(oh).to(dev)
        for prob in (False, True):
            got = eng.predict(fd[:fi], fd[fi], ld[:, :fi], fi, 40, 9, 1.0, 8.0, 21.0, prob).cpu().numpy()
            want = vo.predict(feats[:fi], feats[fi], oh[:, :fi], None if prob else wd, None if prob else ws, fi, 40, 9, 1.0, prob).numpy()
            check_rel(got, want)
        eng.close()
    # channels-last and half-precision sources enter the f32 ring unrounded beyond their own type
    Hd, Wd, T = 12, 20, 6
    g = torch.Generator().manual_seed(3)
    ann = np.zeros((Hd * 8, Wd * 8), np.uint8)
    ann[: Hd * 4] = 1
    ann[:, : Wd * 3] = 2
    for dtype in (torch.float32, torch.float16, torch.bfloat16):
        batch = (torch.randn(T, 256, Hd, Wd, generator=g) * 0.25).to(dtype).to(dev)
        outs = []
        for src in (batch, batch.contiguous(memory_format=torch.channels_last)):
            eng = vos.PropagationEngine(Hd, Wd, device=0, precision=F32)
            eng.begin_video(ann)
            outs.append([eng.step(src[t])[0] for t in range(T)][1:])
            eng.close()
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        want, _ = vo.rollout(ann, batch.float().cpu().numpy(), 40, 9, 1.0, 8.0, 21.0, False)
        check_rel(outs[0][0].cpu().numpy(), want[0])
    with pytest.raises(vos.VospropError):
        vos.PropagationEngine(Hd, Wd, device=0, precision=F32, topk=5)          # top-k: bf16 path only


def test_inference_cli_f32_path_matches_the_unrounded_oracle(tmp_path):
    """`main.py inference --propagation-precision f32 --encoder-dtype f32`: the whole pipeline in the reference's CPU arithmetic;
    the oracle gets the UN-ROUNDED encoder features and the masks must agree on >= 99.9 % of the pixels."""
    from PIL import Image
    sys.path.insert(0, str(ROOT / 'tests'))
    from test_gpu_cli import _make_dataset
    vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
    ds = importlib.import_module('semi-supervised-vos_amd.datasets')
    ann, frames = _make_dataset(tmp_path / 'data', n_frames=8)
    torch.manual_seed(0)
    net = vn.VOSNet('resnet18')
    ckpt = tmp_path / 'ckpt.pth.tar'
    torch.save({'state_dict': net.state_dict()}, ckpt)
    out = subprocess.run([sys.executable, 'main.py', 'inference', '-d', str(tmp_path / 'data'), '-r', str(ckpt), '-m', 'resnet18',
                          '-s', str(tmp_path / 'out'), '--encoder-dtype', 'f32', '--propagation-precision', 'f32', '--ref_num', '5',
                          '--frame_range', '6', '--encoder-batch', '1', '--no-encoder-graph'],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    net.eval().cuda()
    for vid, imgs in frames.items():
        with torch.no_grad():
            feats = torch.cat([net(ds.normalize_image(Image.fromarray(im))[None].cuda()) for im in imgs]).cpu().numpy()
        _, want = vo.rollout(ann, feats, 6, 5, 1.0, 8.0, 21.0, False)
        got = np.stack([np.asarray(Image.open(tmp_path / 'out' / vid / f'{i:05d}.png')) for i in range(1, len(imgs))])
        assert np.mean(got != want) <= 1e-3, f'{vid}: {np.mean(got != want) * 100:.3f} % of pixels differ'
