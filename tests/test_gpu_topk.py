"""Top-k variant (NOT in the reference: "parity unpinned by the reference"), round-3 build: one
scoring pass + a re-score of the marked tiles (prop_dense.h TK 1 / 2, aux_kernels.h topk_select2 / topk_combine2).  What pins it:
the oracle's restatement (vo.predict(..., topk=k)) and the identity `every non-zero entry kept == dense`, here at RAGGED MULTI-TILE
shapes (round 2 only had it at 4x4).  More cases (vs oracle at four shapes, a 20-frame roll-out, full 480p N = 5 k = 20) are in
test_gpu_parity.py and test_gpu_configs.py."""
import numpy as np
import pytest
import torch

from oracle import vos_oracle as vo

pytestmark = pytest.mark.gpu


def bf16_round(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).to(torch.float32).numpy()


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch.device('cuda', 0)


def _sparse_support_case(seed, Hd, Wd, T, d, hot_per_frame):
    """Features for which every target pixel has at most `hot_per_frame` reference pixels per frame with a NON-ZERO probability:
    `hot` reference pixels carry a large component along a direction u every target pixel shares (score ~ +250 above the rest), so
    the softmax of every other reference pixel underflows to exactly 0 in f32 - in the reference's arithmetic and in the
    engine's.  The dense result then has at most N * hot_per_frame non-zero terms per column: keeping the k >= that many largest
    changes nothing."""
    rs = np.random.RandomState(seed)
    HW = Hd * Wd
    u = np.zeros(256, np.float32)
    u[:8] = 1.0
    feats = (rs.standard_normal((T, 256, Hd, Wd)) * 0.05).astype(np.float32)
    feats[:, :8] = 0.0
    for t in range(T):
        hot = rs.choice(HW, hot_per_frame, replace=False)
        f = feats[t].reshape(256, HW)
        f[:8, hot] = 4.0 + rs.rand(8, hot_per_frame).astype(np.float32)      # distinct scores: no ties among the kept terms
        f[:8, :] += 0.0
    feats[T - 1, :8] = 8.0          # every TARGET pixel looks along u: hot scores ~ 8 * 8 * 4.5 = 290, the others ~ 0
    feats = bf16_round(feats)
    lab = rs.randint(0, d, size=(T, HW))
    oh = np.zeros((d, T, HW), np.float32)
    tt, pp = np.meshgrid(np.arange(T), np.arange(HW), indexing='ij')
    oh[lab, tt, pp] = 1.0
    return feats, oh


@pytest.mark.parametrize('Hd,Wd,T,fi,hot,k', [
    (17, 19, 2, 1, 32, 32),      # HW = 323: 11 reference tiles (ragged tail), 2 target tiles, N = 1, exactly k non-zero terms
    (17, 19, 5, 4, 8, 32),       # N = 4 frames x 8 = 32 non-zero terms spread over frames
    (23, 31, 21, 20, 3, 32),     # HW = 713: 3 target tiles, frame_idx = 20: N = 9 frames x 3 = 27 <= k, both sigma branches
])
def test_topk_keeping_every_nonzero_term_equals_dense(vos, dev, Hd, Wd, T, fi, hot, k):
    feats, oh = _sparse_support_case(7 + Hd * Wd + T, Hd, Wd, T, 3, hot)
    fd, ld = torch.from_numpy(feats).to(dev), torch.from_numpy(oh).to(dev)
    e_k = vos.PropagationEngine(Hd, Wd, device=0, topk=k)
    e_d = vos.PropagationEngine(Hd, Wd, device=0)
    a = e_k.predict(fd[:fi], fd[fi], ld[:, :fi], fi, 40, 9, 1.0, 8.0, 21.0, False).cpu().numpy()
    b = e_d.predict(fd[:fi], fd[fi], ld[:, :fi], fi, 40, 9, 1.0, 8.0, 21.0, False).cpu().numpy()
    wd, ws = vo.get_spatial_weight((Hd, Wd), 8.0), vo.get_spatial_weight((Hd, Wd), 21.0)
    want = vo.predict(feats[:fi], feats[fi], oh[:, :fi], wd, ws, fi, 40, 9, 1.0, False).numpy()
    want_k = vo.predict(feats[:fi], feats[fi], oh[:, :fi], wd, ws, fi, 40, 9, 1.0, False, topk=k).numpy()
    assert np.array_equal(want, want_k), 'the case does not have <= k non-zero terms per column'
    assert want.sum(0).max() > 1e-3          # the prior leaves something
    # top-k weights are exact f32 (no bf16 packing on that path), the dense engine rounds them to bf16: 4e-3 relative between them
    assert np.max(np.abs(a - want)) <= 2e-4 * max(1.0, want.max()), np.max(np.abs(a - want))
    assert np.max(np.abs(a - b)) <= 4e-3 * max(b.max(), 1e-6), np.max(np.abs(a - b))
    e_k.close(); e_d.close()


@pytest.mark.parametrize('scale', [0.25, 1.0], ids=['flat_logits', 'peaky_logits'])
def test_topk_mask_only_steps_equal_steps_with_prediction(vos, dev, scale):
    """A top-k step that is asked for the mask only skips the dense launch that supplies the softmax denominators (they scale every
    class of a column alike): a 22-frame roll-out (N = 5, k = 20, frame_idx > 15 at the end) gives the masks of the roll-out that
    returns predictions, and the predictions match the oracle's top-k at the last frame (fed with the engine's own label history)."""
    H, W = 144, 232
    Hd, Wd = vos.feature_map_size(H, W)
    HW = Hd * Wd
    rs = np.random.RandomState(31)
    ann = np.zeros((H, W), np.uint8)
    ann[20:90, 30:120] = 1
    ann[70:130, 100:210] = 2
    T, d = 22, 3
    base = rs.randn(256, Hd, Wd).astype(np.float32)
    feats = []
    for t in range(T):
        base = 0.9 * base + 0.45 * rs.randn(256, Hd, Wd).astype(np.float32)
        feats.append(bf16_round(base * scale))
    feats = np.stack(feats)
    out = {}
    for want_pred in (True, False):
        eng = vos.PropagationEngine(Hd, Wd, device=0, ref_num=5, topk=20)
        eng.begin_video(ann)
        masks, preds = [], []
        for t in range(T):
            p, m = eng.step(torch.from_numpy(feats[t]).to(dev), want_pred=want_pred, want_mask=True)
            if m is not None:
                masks.append(m.cpu().numpy())
                preds.append(None if p is None else p.cpu().numpy())
        eng.close()
        out[want_pred] = (masks, preds)
    worst = max(float(np.mean(a != b)) for a, b in zip(out[True][0], out[False][0]))
    assert worst <= 2e-3, worst
    # the last step of the roll-out that returns predictions against the oracle, on the engine's own label history
    masks, preds = out[True]
    ys = [next(y for y in range(H) if min(int(np.floor(np.float32(y) * (np.float32(Hd) / np.float32(H)))), Hd - 1) == i) for i in range(Hd)]
    xs = [next(x for x in range(W) if min(int(np.floor(np.float32(x) * (np.float32(Wd) / np.float32(W)))), Wd - 1) == j) for j in range(Wd)]
    cls0 = np.asarray(vo.get_labels(ann.astype(np.int64), d, H, W, Hd, Wd)).reshape(d, HW).argmax(0)
    cls_hist = [cls0] + [m[np.ix_(ys, xs)].reshape(-1) for m in masks]
    fi = T - 1
    oh = np.zeros((d, fi, HW), np.float32)
    for t in range(fi):
        oh[cls_hist[t], t, np.arange(HW)] = 1.0
    cols = np.sort(rs.choice(HW, 400, replace=False))
    want = vo.predict_columns(feats[:fi], feats[fi], oh, 8.0, 21.0, fi, 40, 5, 1.0, False, cols, topk=20).numpy()
    got = preds[-1][:, cols]
    assert np.max(np.abs(got - want)) <= 2e-4 * max(1.0, want.max()), np.max(np.abs(got - want))


def test_topk_mass_ties_take_the_fallback_and_stay_finite(vos, dev):
    """Constant features: every score of a column is the same number, so every group maximum ties (the packed maxima differ only
    by their index bits, inside the selection margin) - far more than the 512 candidates the select kernel carries through LDS.
    That takes its streamed-radix fallback and the "more groups than slots" clamps of pass 2 / combine.  Which of the tied
    elements are kept is arbitrary (the oracle's `S >= kth` keeps ALL of them, i.e. the dense result): the engine must terminate,
    return finite, non-negative numbers that sum to at most the dense total, and a mask of valid classes."""
    Hd, Wd, T, d, fi, k = 30, 54, 8, 3, 7, 20
    HW = Hd * Wd
    feats = np.full((T, 256, Hd, Wd), 0.125, np.float32)
    rs = np.random.RandomState(3)
    lab = rs.randint(0, d, size=(T, HW))
    oh = np.zeros((d, T, HW), np.float32)
    tt, pp = np.meshgrid(np.arange(T), np.arange(HW), indexing='ij')
    oh[lab, tt, pp] = 1.0
    fd, ld = torch.from_numpy(feats).to(dev), torch.from_numpy(oh).to(dev)
    e_k = vos.PropagationEngine(Hd, Wd, device=0, ref_num=5, topk=k)
    e_d = vos.PropagationEngine(Hd, Wd, device=0, ref_num=5)
    a = e_k.predict(fd[:fi], fd[fi], ld[:, :fi], fi, 40, 5, 1.0, 8.0, 21.0, False).cpu().numpy()
    b = e_d.predict(fd[:fi], fd[fi], ld[:, :fi], fi, 40, 5, 1.0, 8.0, 21.0, False).cpu().numpy()
    assert np.all(np.isfinite(a)) and np.all(a >= 0)
    assert np.all(a.sum(0) <= b.sum(0) * (1 + 1e-3) + 1e-9)
    assert a.sum() > 0
    # the capacity clamps are REPORTED (vosprop_topk_overflows), never silent: here none fires (the spatial prior breaks the ties of the
    # scores - only the frames of one sigma class tie, five ways), so the kept set is exact and the counters read zero
    over = e_k.topk_overflows()
    assert len(over) == 3 and over == (0, 0, 0), over
    e_k.close(); e_d.close()


@pytest.mark.parametrize('k', [20, 32])
def test_topk_capacity_clamps_do_not_fire_on_flat_logits(vos, dev, k):
    """The three capacity limits of the top-k kernels (dump slots per lane, 40 groups per pixel in the combine kernel, 512
    candidates per pixel in the select kernel) drop candidates when they fire.  On flat, slowly varying logits at 480p - many near-equal
    group maxima, the hard case short of exact ties - with the LARGEST k (32: room for only 8 extra groups) they must not fire
    (vosprop_topk_overflows == 0), and the prediction must be the oracle's top-k restatement on a column sample."""
    Hd, Wd, T, d, fi = 60, 107, 7, 4, 6
    HW = Hd * Wd
    rs = np.random.RandomState(5 + k)
    low = rs.randn(T, 256, 8, 14).astype(np.float32)
    feats = torch.nn.functional.interpolate(torch.from_numpy(low), size=(Hd, Wd), mode='bilinear', align_corners=False).numpy()
    feats = bf16_round(feats * 0.2 + rs.randn(T, 256, Hd, Wd).astype(np.float32) * 0.02)       # smooth + a little texture
    lab = rs.randint(0, d, size=(T, HW))
    oh = np.zeros((d, T, HW), np.float32)
    tt, pp = np.meshgrid(np.arange(T), np.arange(HW), indexing='ij')
    oh[lab, tt, pp] = 1.0
    fd, ld = torch.from_numpy(feats).to(dev), torch.from_numpy(oh).to(dev)
    eng = vos.PropagationEngine(Hd, Wd, device=0, ref_num=5, topk=k)
    got = eng.predict(fd[:fi], fd[fi], ld[:, :fi], fi, 40, 5, 1.0, 8.0, 21.0, False).cpu().numpy()
    over = eng.topk_overflows()
    eng.close()
    assert over == (0, 0, 0), over
    cols = np.sort(rs.choice(HW, 400, replace=False))
    want = vo.predict_columns(feats[:fi], feats[fi], oh[:, :fi], 8.0, 21.0, fi, 40, 5, 1.0, False, cols, topk=k).numpy()
    tot = np.maximum(want.sum(0), 1e-30)
    assert np.max(np.abs(got[:, cols] - want) / tot) <= 2e-2, np.max(np.abs(got[:, cols] - want) / tot)


_FORCED_RADIX_CHILD = r'''
import importlib, json, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from oracle import vos_oracle as vo
vos = importlib.import_module('semi-supervised-vos_amd')
Hd, Wd, T, d, fi, k = 23, 31, 7, 4, 6, 20
HW = Hd * Wd
rs = np.random.RandomState(77)
feats = torch.from_numpy((rs.randn(T, 256, Hd, Wd) * 0.25).astype(np.float32)).to(torch.bfloat16).to(torch.float32).numpy()
lab = rs.randint(0, d, size=(T, HW))
oh = np.zeros((d, T, HW), np.float32)
tt, pp = np.meshgrid(np.arange(T), np.arange(HW), indexing='ij')
oh[lab, tt, pp] = 1.0
dev = torch.device('cuda', 0)
eng = vos.PropagationEngine(Hd, Wd, device=0, ref_num=5, topk=k)
got = eng.predict(torch.from_numpy(feats[:fi]).to(dev), torch.from_numpy(feats[fi]).to(dev), torch.from_numpy(oh[:, :fi]).to(dev),
                  fi, 40, 5, 1.0, 8.0, 21.0, False).cpu().numpy()
over = eng.topk_overflows()
eng.close()
want = vo.predict_columns(feats[:fi], feats[fi], oh[:, :fi], 8.0, 21.0, fi, 40, 5, 1.0, False, np.arange(HW), topk=k).numpy()
tot = np.maximum(want.sum(0), 1e-30)
print(json.dumps({'err': float(np.max(np.abs(got.reshape(d, HW) - want) / tot)), 'over': list(over)}))
'''


def test_topk_select_radix_fallback_matches_the_oracle(tmp_path):
    """topk_combine2_kernel finds a column's k-th largest key among the candidates it compacted into LDS; a column with more
    candidates than that buffer holds falls back to a radix selection over ALL its keys.  Natural inputs never get there (the
    overflow counters of the tests above read zero), so the fallback is FORCED here for every column (VOSPROP_TK_FORCE_RADIX=1, read
    once per process: hence the child process) and checked against the oracle's top-k restatement like the default path."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    script = tmp_path / 'child.py'
    script.write_text(_FORCED_RADIX_CHILD)
    env = dict(os.environ, VOSPROP_TK_FORCE_RADIX='1')
    r = subprocess.run([sys.executable, str(script), str(root)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['err'] <= 2e-2, out
