#!/usr/bin/env python3
"""Generate golden vectors for the label-propagation hot path by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_goldens.py

It imports the reference's own `src.model.predict` / `src.utils.inference_utils` / `src.utils.metrics` with four
in-process shims (nothing installed, nothing fetched):
  * `loguru`  -> stub module with a no-op `logger`      (imported at src/utils/utils.py:9)
  * `np.int`  -> `int`                                   (removed alias used at src/model/predict.py:85)
  * `torchvision.transforms.ColorJitter` -> dummy class  (import chain inference_utils.py:15 -> transforms.py:9,50)
  * `skimage` -> empty module, import only           (src/utils/metrics.py:6-7; no function that uses it is run)
and stores ONLY data (inputs' seeds/shapes and the reference's outputs) in `tests/golden/*.npz`.
Every input is regenerated from `numpy.random.RandomState(seed)` (frozen stream) by
`tests/golden/inputs.py`, which the tests share with this script.
"""
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import inputs as gin  # noqa: E402  (shared deterministic input builders)

REF = Path('/root/reference')


def _install_shims():
    np.int = int  # noqa: NPY001 - alias removed in numpy>=1.24, used by the reference
    loguru = types.ModuleType('loguru')

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    loguru.logger = _Logger()
    sys.modules['loguru'] = loguru
    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')

    class ColorJitter:  # only has to exist as a base class
        def __init__(self, *a, **k):
            pass

    tvt.ColorJitter = ColorJitter
    tv.transforms = tvt
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.transforms'] = tvt
    sys.path.insert(0, str(REF))


def main():
    _install_shims()
    torch.set_num_threads(4)
    from src.config import Config
    Config.DEVICE = torch.device('cpu')
    from src.model import predict as rp
    from src.utils import inference_utils as riu
    from src.utils.utils import index_to_onehot

    out = {}

    # ---- G1: sample_frames (src/model/predict.py:74-89) -------------------------------------
    for (rng, nref) in gin.G1_CASES:
        rows = []
        for fi in range(1, 121):
            idx = rp.sample_frames(fi, rng, nref).tolist()
            rows.append(idx + [-1] * (nref - len(idx)))
        out[f'g1_sample_r{rng}_n{nref}'] = np.asarray(rows, dtype=np.int32)

    # ---- G2: get_spatial_weight (src/model/predict.py:158-175) -----------------------------
    out['g2_w_4x6_s8'] = rp.get_spatial_weight((4, 6), 8.0).numpy()
    for (h, w) in gin.G2_SHAPES:
        ii, jj = gin.g2_sample_pairs(h, w)
        for sigma in (8.0, 21.0):
            full = rp.get_spatial_weight((h, w), sigma)
            out[f'g2_w_{h}x{w}_s{int(sigma)}'] = full[ii, jj].numpy()
            del full

    # ---- G3: get_labels (src/model/predict.py:92-96) ---------------------------------------
    mask = gin.g3_mask()
    H, W = mask.shape
    H_d, W_d = int(np.ceil(H * 0.125)), int(np.ceil(W * 0.125))
    d = int(mask.max()) + 1
    lab = rp.get_labels(torch.from_numpy(mask).long(), d, H, W, H_d, W_d)
    out['g3_labels'] = lab.numpy()

    # ---- G4/G5: predict (src/model/predict.py:19-71) ---------------------------------------
    for case in gin.PREDICT_CASES:
        ref, tgt, lab_hist = gin.predict_inputs(case)
        Hd, Wd = case['hw']
        wd = rp.get_spatial_weight((Hd, Wd), case['sigma1'])
        ws = rp.get_spatial_weight((Hd, Wd), case['sigma2'])
        for prob in (False, True):
            lh = gin.predict_labels(case, prob)
            for fi in case['frame_idx']:
                pred = rp.predict(torch.from_numpy(ref[:fi].copy()), torch.from_numpy(tgt[fi].copy()),
                                  torch.from_numpy(lh[:, :fi].copy()),
                                  None if prob else wd, None if prob else ws,
                                  fi, case['range'], case['ref_num'], case['temperature'], prob)
                out[f"{case['name']}_{'prob' if prob else 'label'}_f{fi}"] = pred.numpy().astype(np.float32)

    # ---- G6: inference_single roll-out (src/utils/inference_utils.py:23-87) ----------------
    for case in gin.ROLLOUT_CASES:
        for prob in (False, True):
            with tempfile.TemporaryDirectory() as td:
                td = Path(td)
                ann_dir = td / 'ann'
                save_dir = td / 'save'
                gin.write_rollout_annotation(case, ann_dir)
                feats = gin.rollout_features(case)          # (T, C, Hd, Wd) f32
                T = feats.shape[0]
                H, W = case['image_hw']
                loader = [(torch.zeros(1, 3, H, W), (case['video'],)) for _ in range(T)]
                counter = {'i': 0}

                def model(x, _f=feats, _c=counter):
                    t = torch.from_numpy(_f[_c['i']:_c['i'] + 1].copy())
                    _c['i'] += 1
                    return t

                preds = []
                orig_predict = riu.predict

                def spy(*a, **k):
                    p = orig_predict(*a, **k)
                    preds.append(p.numpy().astype(np.float32).copy())
                    return p

                riu.predict = spy
                try:
                    riu.inference_single(model, loader, T, ann_dir, case['video'], str(save_dir),
                                         case['sigma1'], case['sigma2'], case['range'], case['ref_num'],
                                         case['temperature'], prob, True)
                finally:
                    riu.predict = orig_predict
                from PIL import Image
                masks = []
                for i in range(1, T):
                    im = Image.open(save_dir / case['video'] / f'{i:05d}.png')
                    assert im.mode == 'P'
                    masks.append(np.asarray(im).astype(np.uint8))
                    if i == 1:
                        out[f"{case['name']}_palette"] = np.asarray(im.getpalette(), dtype=np.uint8)
                tag = f"{case['name']}_{'prob' if prob else 'label'}"
                out[f'{tag}_preds'] = np.stack(preds)           # (T-1, d, HW)
                out[f'{tag}_masks'] = np.stack(masks)           # (T-1, H, W) u8

    # ---- G7: the multi-branch strategies (src/utils/inference_utils.py:90-595) with seeded encoder outputs ----------
    from PIL import Image
    case = gin.STRATEGY_CASE
    H, W = case['image_hw']
    T = case['T']

    class FakeEncoder:
        """Returns the pre-generated features of (frame, branch); checks that the map size a stride-8 encoder would
        produce for the input it is handed equals the size of those features."""

        def __init__(self, feats_by_call):
            self.feats, self.i = feats_by_call, 0

        def __call__(self, x):
            f = self.feats[self.i]
            self.i += 1
            assert (int(np.ceil(x.shape[2] / 8)), int(np.ceil(x.shape[3] / 8))) == f.shape[-2:], (x.shape, f.shape)
            return torch.from_numpy(f[None].copy())

    def read_masks(save_dir):
        return np.stack([np.asarray(Image.open(save_dir / case['video'] / f'{i:05d}.png')).astype(np.uint8)
                         for i in range(1, T)])

    for (strategy, prob, fusion) in gin.STRATEGY_RUNS:
        fa, fb = gin.strategy_branch_features(case, strategy)
        with tempfile.TemporaryDirectory() as td:
            td = Path(td)
            gin.write_rollout_annotation(case, td / 'ann')
            sizes = [gin.strategy_input_hw(case, strategy, b) for b in (0, 1)]
            if strategy == 'multimodel':
                loader = [(torch.zeros(1, 3, H, W), (case['video'],)) for _ in range(T)]
                m0, m1 = FakeEncoder(list(fa)), FakeEncoder(list(fb))
            else:
                loader = [([torch.zeros(1, 3, *sizes[0]), torch.zeros(1, 3, *sizes[1])], (case['video'],)) for _ in range(T)]
                m0 = FakeEncoder([f for t in range(T) for f in (fa[t], fb[t])])      # model(input_l); model(input_r)
            head = (loader, T, td / 'ann', case['video'], str(td / 'save'), case['sigma1'], case['sigma2'], case['range'],
                    case['ref_num'], case['temperature'], prob)
            if strategy == 'hor-flip':
                riu.inference_hor_flip(m0, *head, fusion, True)
            elif strategy == 'vert-flip':
                riu.inference_ver_flip(m0, *head, fusion, True)
            elif strategy == '2-scale':
                riu.inference_2_scale(m0, *head, case['scale2'], fusion, False, True)
            elif strategy == 'hor-2-scale':
                riu.inference_2_scale(m0, *head, case['scale2'], fusion, True, True)
            else:
                riu.inference_multimodel(m0, m1, *head, fusion, True)
            out[f"g7_{strategy}_{'prob_' + fusion if prob else 'label'}_masks"] = read_masks(td / 'save')

    scales, f3 = gin.three_scale_features(case)
    for prob in (False, True):
        with tempfile.TemporaryDirectory() as td:
            td = Path(td)
            gin.write_rollout_annotation(case, td / 'ann')
            loader = [(torch.zeros(1, 3, H, W), (case['video'],)) for _ in range(T)]
            enc = FakeEncoder([f for fs in f3 for f in fs])                           # three sequential passes
            riu.inference_3_scale(enc, loader, T, td / 'ann', case['video'], str(td / 'save'), case['sigma1'],
                                  case['sigma2'], case['range'], case['ref_num'], case['temperature'], prob,
                                  case['scale2'], True)
            out[f"g7_3-scale_{'prob' if prob else 'label'}_masks"] = read_masks(td / 'save')

    # ---- G8: eval_j and the boundary map (src/utils/metrics.py:15-45,124-181).  metrics.py imports scikit-image at module
    # level (absent here); an empty stand-in module lets the import succeed and NOTHING that uses it is run: eval_j and
    # _seg2bmap do not touch it, f_measure (which does) is not executed - its parity stays unpinned ----
    sk = types.ModuleType('skimage')
    skm = types.ModuleType('skimage.morphology')
    skm.disk = None
    sk.morphology = skm
    sys.modules['skimage'] = sk
    sys.modules['skimage.morphology'] = skm
    from src.utils import metrics as rmet
    ann, seg, void = gin.metric_masks()
    out['g8_j_stack'] = np.asarray(rmet.eval_j(ann, seg), dtype=np.float64)
    out['g8_j_stack_void'] = np.asarray(rmet.eval_j(ann, seg, void), dtype=np.float64)
    out['g8_j_single'] = np.asarray([float(rmet.eval_j(ann[i], seg[i])) for i in range(ann.shape[0])], dtype=np.float64)
    out['g8_bmap'] = np.stack([rmet._seg2bmap(seg[i].copy()) for i in range(seg.shape[0])]).astype(np.uint8)

    # ---- index_to_onehot (src/utils/utils.py:59-68) -----------------------------------------
    idx = torch.from_numpy(gin.onehot_indices())
    out['onehot'] = index_to_onehot(idx, 5).numpy()

    # ---- encoder (src/model/vos_net.py:9-51 over src/model/backbone/resnet.py), built WITHOUT the network fetch:
    # resnetXX(pretrained=False) + the same wrapping VOSNet.__init__ does (vos_net.py:18-23) ----
    import json
    import torch.nn as nn
    from src.model.backbone import resnet as rresnet
    from src.model.vos_net import VOSNet as RefVOSNet
    enc_keys = {}
    for name in ('resnet18', 'resnet50', 'resnet101'):
        net = RefVOSNet.__new__(RefVOSNet)
        nn.Module.__init__(net)
        net.model = name
        net.backbone = nn.Sequential(*list(getattr(rresnet, name)(pretrained=False).children())[0:8])
        if name != 'resnet18':
            net.adjust_dim = nn.Conv2d(1024, 256, kernel_size=1, stride=1, padding=0, bias=False)
            net.bn256 = nn.BatchNorm2d(256)
        sd = net.state_dict()
        enc_keys[name] = {k: list(v.shape) for k, v in sd.items()}
        if name != 'resnet101':
            net.load_state_dict({k: torch.from_numpy(v) for k, v in gin.fill_state_dict(sd).items()})
            net.eval()
            with torch.no_grad():
                y = net(torch.from_numpy(gin.encoder_input()))
            out[f'enc_{name}_out'] = y.numpy().astype(np.float32)
    (HERE / 'encoder_keys.json').write_text(json.dumps(enc_keys, indent=0, sort_keys=True))

    np.savez_compressed(HERE / 'reference_goldens.npz', **out)
    tot = sum(v.nbytes for v in out.values())
    print(f'wrote {len(out)} arrays, {tot / 1e6:.2f} MB raw ->', HERE / 'reference_goldens.npz',
          os.path.getsize(HERE / 'reference_goldens.npz') / 1e6, 'MB')


if __name__ == '__main__':
    main()
