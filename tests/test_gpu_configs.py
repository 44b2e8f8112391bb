"""One GPU test per BASELINE.json config, each at the config's FULL size, HIP engine (through the C ABI) against the oracle.

The torch oracle materialises the (N*HW) x HW affinity three times (reference src/model/predict.py:49-66): 1.5 GB per copy at 480p,
7.5 GB at 720p.  Target pixels are independent columns of that computation, so the big shapes are checked on a SUBSET of target
columns with `vo.predict_columns` (same ops in the same order, pinned to `vo.predict` by tests/test_oracle_golden.py): 512 random
columns + the whole last (ragged) target tile + the first tile, which the oracle finishes in seconds.  Tolerances are those of
tests/test_gpu_parity.py (bf16 MFMA path, inputs pre-rounded to bf16: 4e-3 relative on well-conditioned columns).
"""
import importlib
import json
import os
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


def bf16_round(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).to(torch.float32).numpy()


def random_case(seed, Hd, Wd, T, d, scale=0.25):
    rs = np.random.RandomState(seed)
    feats = bf16_round((rs.standard_normal((T, 256, Hd, Wd)) * scale).astype(np.float32))
    lab = rs.randint(0, d, size=(T, Hd * Wd))
    oh = np.zeros((d, T, Hd * Wd), dtype=np.float32)
    tt, pp = np.meshgrid(np.arange(T), np.arange(Hd * Wd), indexing='ij')
    oh[lab, tt, pp] = 1.0
    return feats, oh


def some_columns(HW, seed, n_random=512):
    rs = np.random.RandomState(seed)
    last_tile = np.arange((HW - 1) // 256 * 256, HW)          # the ragged last target tile of the kernel's 256-pixel tiling
    cols = np.unique(np.concatenate([np.arange(0, 32), last_tile, rs.choice(HW, size=min(n_random, HW), replace=False)]))
    return cols


def check_close(got, want, rel, well=1e-6, abs_small=1e-6):
    colsum = want.sum(0, keepdims=True)
    wellc = np.broadcast_to(colsum >= well, want.shape)
    err = np.abs(got - want)
    scale = np.maximum(np.abs(want), colsum * 1e-2)
    assert np.all(err[wellc] <= rel * scale[wellc] + abs_small), f'max rel err {np.max(err[wellc] / (scale[wellc] + 1e-30)):.3e}'
    assert np.all(err[~wellc] <= np.maximum(abs_small, rel * np.abs(want[~wellc]) * 50))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch.device('cuda', 0)


def test_config1_240p_pair_resnet18_map(vos, dev):
    """configs[0]: a 2-frame 240p pair (427x240 -> 30x54 map, N = 1 reference frame), both modes, full oracle."""
    Hd, Wd = vos.feature_map_size(240, 427)
    assert (Hd, Wd) == (30, 54)
    feats, oh = random_case(11, Hd, Wd, 2, 3)
    wd, ws = vo.get_spatial_weight((Hd, Wd), 8.0), vo.get_spatial_weight((Hd, Wd), 21.0)
    eng = vos.PropagationEngine(Hd, Wd, device=0)
    fd, ld = torch.from_numpy(feats).to(dev), torch.from_numpy(oh).to(dev)
    for prob in (False, True):
        got = eng.predict(fd[:1], fd[1], ld[:, :1], 1, 40, 9, 1.0, 8.0, 21.0, prob).cpu().numpy()
        want = vo.predict(feats[:1], feats[1], oh[:, :1], None if prob else wd, None if prob else ws, 1, 40, 9, 1.0, prob).numpy()
        check_close(got, want, rel=4e-3)
    eng.close()


@pytest.mark.parametrize('prob', [False, True])
def test_config2_480p_dense_full_size(vos, dev, prob):
    """configs[1]: 854x480 -> 60x107, HW = 6 420, N = 9, frame_idx = 20 (both sigma branches), dense, 4 classes."""
    Hd, Wd, T, d, fi = 60, 107, 21, 4, 20
    feats, oh = random_case(22 + prob, Hd, Wd, T, d)
    cols = some_columns(Hd * Wd, 2)
    eng = vos.PropagationEngine(Hd, Wd, device=0)
    fd, ld = torch.from_numpy(feats).to(dev), torch.from_numpy(oh).to(dev)
    got = eng.predict(fd[:fi], fd[fi], ld[:, :fi], fi, 40, 9, 1.0, 8.0, 21.0, prob).cpu().numpy()
    want = vo.predict_columns(feats[:fi], feats[fi], oh[:, :fi], 8.0, 21.0, fi, 40, 9, 1.0, prob, cols).numpy()
    check_close(got[:, cols], want, rel=4e-3)
    if prob:
        assert np.allclose(got.sum(0), 1.0, atol=2e-3)
    eng.close()


def test_config3_480p_top20_ref5_full_size(vos, dev):
    """configs[2]: 60x107, --ref_num 5, k = 20 (top-k is NOT in the reference: oracle restatement, label mode)."""
    Hd, Wd, T, d, fi, k = 60, 107, 21, 4, 20, 20
    feats, oh = random_case(33, Hd, Wd, T, d)
    cols = some_columns(Hd * Wd, 3)
    eng = vos.PropagationEngine(Hd, Wd, device=0, ref_num=5, topk=k)
    fd, ld = torch.from_numpy(feats).to(dev), torch.from_numpy(oh).to(dev)
    got = eng.predict(fd[:fi], fd[fi], ld[:, :fi], fi, 40, 5, 1.0, 8.0, 21.0, False).cpu().numpy()
    want = vo.predict_columns(feats[:fi], feats[fi], oh[:, :fi], 8.0, 21.0, fi, 40, 5, 1.0, False, cols, topk=k).numpy()
    dense = vo.predict_columns(feats[:fi], feats[fi], oh[:, :fi], 8.0, 21.0, fi, 40, 5, 1.0, False, cols).numpy()
    assert np.any(np.abs(want - dense) > 1e-4), 'test would not see a missing top-k'
    assert np.max(np.abs(got[:, cols] - want)) <= 2e-4 * max(1.0, want.max()), np.max(np.abs(got[:, cols] - want))
    eng.close()


@pytest.mark.parametrize('prob', [False, True])
def test_config5_720p_dense_full_size(vos, dev, prob):
    """configs[4]: 1280x720 -> 90x160, HW = 14 400, N = 9, frame_idx = 20, dense (the reference would materialise 7.46 GB)."""
    Hd, Wd = vos.feature_map_size(720, 1280)
    assert (Hd, Wd) == (90, 160)
    T, d, fi = 21, 4, 20
    feats, oh = random_case(55 + prob, Hd, Wd, T, d)
    cols = some_columns(Hd * Wd, 5)
    eng = vos.PropagationEngine(Hd, Wd, device=0)
    fd, ld = torch.from_numpy(feats).to(dev), torch.from_numpy(oh).to(dev)
    got = eng.predict(fd[:fi], fd[fi], ld[:, :fi], fi, 40, 9, 1.0, 8.0, 21.0, prob).cpu().numpy()
    want = vo.predict_columns(feats[:fi], feats[fi], oh[:, :fi], 8.0, 21.0, fi, 40, 9, 1.0, prob, cols).numpy()
    check_close(got[:, cols], want, rel=4e-3)
    if prob:
        assert np.allclose(got.sum(0), 1.0, atol=2e-3)
    else:
        assert np.all(got >= 0) and np.all(got.sum(0) <= 1.0 + 2e-3)
    eng.close()


def test_config5_720p_rollout_through_step(vos, dev):
    """configs[4] through the stateful path: begin_video + 19 steps at 90x160 (ring slots, label packing, combine, fused mask
    up-sampling at 720p), the last step (frame_idx = 19 > 15: N = 9, both sigmas) checked against the oracle fed with the
    engine's own label history on a column subset."""
    H, W = 720, 1280
    Hd, Wd = vos.feature_map_size(H, W)
    T, d = 20, 4
    rs = np.random.RandomState(7)
    feats = bf16_round((rs.standard_normal((T, 256, Hd, Wd)) * 0.25).astype(np.float32))
    ann = np.zeros((H, W), np.uint8)
    ann[100:400, 200:600] = 1
    ann[300:650, 700:1100] = 2
    ann[50:200, 900:1200] = 3
    eng = vos.PropagationEngine(Hd, Wd, device=0)
    assert eng.begin_video(ann) == d
    fd = torch.from_numpy(feats).to(dev)
    st = vo.VideoState(ann, 8.0, 21.0, False)
    labels = [st.label_history[:, 0].float()]                     # (d, HW) one-hot of the down-sampled first annotation
    pred = mask = None
    for t in range(T):
        pred, mask = eng.step(fd[t])
        if pred is not None and t < T - 1:
            labels.append(vo.index_to_onehot(pred.argmax(0).cpu(), d))
    hist = torch.stack(labels, 1).numpy()                         # (d, T-1, HW): the label history the engine propagated from
    fi = T - 1
    cols = some_columns(Hd * Wd, 9, n_random=256)
    want = vo.predict_columns(feats[:fi], feats[fi], hist, 8.0, 21.0, fi, 40, 9, 1.0, False, cols).numpy()
    check_close(pred.cpu().numpy()[:, cols], want, rel=4e-3)
    low = pred.view(-1, Hd, Wd).cpu().argmax(0).to(torch.float32)
    up = torch.nn.functional.interpolate(low[None, None], size=(H, W), mode='nearest')[0, 0].to(torch.uint8)
    assert torch.equal(mask.cpu(), up)
    eng.close()


@pytest.mark.parametrize('Hd,Wd,prob', [(12, 20, False), (17, 19, True), (90, 160, False)])
def test_config5_materialised_affinity(vos, dev, Hd, Wd, prob):
    """configs[4], the HBM-stress variant: the engine writes the (N HW) x HW affinity to HBM as bf16 and reads it back (the shape of
    the reference's own algorithm, src/model/predict.py:49-55; 7.46 GB per step at 90x160).  Checked against the oracle with its
    affinity rounded to bf16 at the same point, and against the fused engine: same masks wherever the margin is clear."""
    T, d, fi = 21, 4, 20
    feats, oh = random_case(77 + Hd, Hd, Wd, T, d)
    HW = Hd * Wd
    cols = some_columns(HW, 6, n_random=256) if HW > 2000 else np.arange(HW)
    fd, ld = torch.from_numpy(feats).to(dev), torch.from_numpy(oh).to(dev)
    em = vos.PropagationEngine(Hd, Wd, device=0, materialise=True)
    ef = vos.PropagationEngine(Hd, Wd, device=0)
    got = em.predict(fd[:fi], fd[fi], ld[:, :fi], fi, 40, 9, 1.0, 8.0, 21.0, prob).cpu().numpy()
    fused = ef.predict(fd[:fi], fd[fi], ld[:, :fi], fi, 40, 9, 1.0, 8.0, 21.0, prob).cpu().numpy()
    want = vo.predict_columns(feats[:fi], feats[fi], oh[:, :fi], 8.0, 21.0, fi, 40, 9, 1.0, prob, cols, affinity_bf16=True).numpy()
    # a score that sits on a bf16 rounding boundary can round the other way in the engine (f32 sums in MFMA order): one such
    # element moves its probability by 2^(2^-8 |s| c) - 1; the tolerance is per column total, not per element
    colsum = want.sum(0, keepdims=True)
    assert np.all(np.abs(got[:, cols] - want) <= 1e-2 * np.maximum(colsum, 1e-6) + 1e-6), float(np.max(np.abs(got[:, cols] - want) / np.maximum(colsum, 1e-6)))
    srt = np.sort(fused, axis=0)
    clear = (srt[-1] - srt[-2]) > 5e-2 * srt[-1]
    assert clear.sum() > 50 and np.all(got.argmax(0)[clear] == fused.argmax(0)[clear])
    st = em.last_stats()
    assert st['bytes'] > 2.0 * 9 * HW * HW * 2.0            # the stats price the written + re-read affinity
    em.close(); ef.close()
    with pytest.raises(vos.VospropError):
        vos.PropagationEngine(Hd, Wd, device=0, materialise=True, topk=5)


def _davis_like_dataset(root, lengths, H=64, W=96):
    from PIL import Image
    case = dict(gin.ROLLOUT_CASES[0], image_hw=(H, W))
    ann = gin.rollout_annotation(case)
    for i, n in enumerate(lengths):
        vid = f'v{i:02d}'
        (root / 'JPEGImages' / '480p' / vid).mkdir(parents=True)
        (root / 'Annotations' / '480p' / vid).mkdir(parents=True)
        rs = np.random.RandomState(100 + i)
        base = rs.randint(0, 255, (H // 8, W // 8, 3)).astype(np.float32)
        for t in range(n):
            base = np.clip(base + rs.randn(*base.shape) * 6, 0, 255)
            img = Image.fromarray(base.astype(np.uint8)).resize((W, H), Image.BILINEAR)
            img.save(root / 'JPEGImages' / '480p' / vid / f'{t:05d}.png')
        im = Image.fromarray(ann, mode='P')
        im.putpalette(gin.DAVIS_PALETTE + [0] * (768 - 24))
        im.save(root / 'Annotations' / '480p' / vid / '00000.png')
    return ann


def test_config4_thirty_videos_over_eight_shards(tmp_path):
    """configs[3]: the full DAVIS-2017-val layout - 30 videos whose lengths vary 3x - dealt to EIGHT shards by LPT (sharding.py),
    every shard run as its own `main.py inference --shard r 8` process on this box's one GPU (at most four at a time: the box
    allows few processes on its card).  Checks what the 8-GPU run relies on: every video is produced by exactly one shard,
    complete; no shard's load exceeds the LPT bound; every PNG is byte-identical to a single-process run's over all 30 videos
    (`--deterministic`, default f16 encoder; reference: videos are independent, src/utils/inference_utils.py:28-48)."""
    from PIL import Image
    sharding = importlib.import_module('semi-supervised-vos_amd.sharding')
    vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
    rs = np.random.RandomState(4)
    lengths = [int(v) for v in rs.randint(4, 13, size=30)]         # 4 .. 12 frames: DAVIS val's 34 .. 104, scaled down
    _davis_like_dataset(tmp_path / 'data', lengths)
    names = {f'v{i:02d}': n for i, n in enumerate(lengths)}
    shards, load = sharding.lpt_assign(names, 8)
    assert sorted(v for s in shards for v in s) == sorted(names) and max(load) <= sum(lengths) / 8 + max(lengths)
    torch.manual_seed(0)
    torch.save({'state_dict': vn.VOSNet('resnet18').state_dict()}, tmp_path / 'ckpt.pth.tar')
    base = [sys.executable, 'main.py', 'inference', '-d', str(tmp_path / 'data'), '-r', str(tmp_path / 'ckpt.pth.tar'), '-m', 'resnet18',
            '--ref_num', '5', '--frame_range', '6', '--io-workers', '1', '--deterministic']
    one = subprocess.run(base + ['-s', str(tmp_path / 'one')], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    stats = []
    for wave in (range(0, 4), range(4, 8)):
        procs = [subprocess.Popen(base + ['-s', str(tmp_path / 'eight'), '--shard', str(r), '8'], cwd=ROOT, stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True) for r in wave]
        for r, pr in zip(wave, procs):
            out, err = pr.communicate(timeout=900)
            assert pr.returncode == 0, err[-2000:]
            st = json.loads([l for l in out.splitlines() if l.startswith('{"vosprop_stats"')][0])['vosprop_stats']
            assert st['shard'] == [r, 8]
            stats.append(st)
    assert sum(st['frames'] for st in stats) == sum(lengths) and sum(st['videos'] for st in stats) == 30
    assert [st['frames'] for st in stats] == load
    n_classes = set()
    for vid, n in names.items():
        for i in range(n):
            a = np.asarray(Image.open(tmp_path / 'one' / vid / f'{i:05d}.png'))
            b = np.asarray(Image.open(tmp_path / 'eight' / vid / f'{i:05d}.png'))
            assert a.shape == b.shape and np.array_equal(a, b), f'{vid}/{i:05d}.png: {np.mean(a != b) * 100:.3f} % of pixels differ'
        n_classes.add(len(np.unique(b)))
    assert max(n_classes) >= 2          # not all collapsed to background
