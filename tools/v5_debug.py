import importlib, sys, numpy as np, torch
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / 'tests' / 'golden'))
import inputs as gin
vos = importlib.import_module('semi-supervised-vos_amd')
g = np.load(R / 'tests/golden/reference_goldens.npz')
case = gin.ROLLOUT_CASES[0]
ann = gin.rollout_annotation(case); feats = gin.rollout_features(case)
H, W = case['image_hw']; Hd, Wd = vos.feature_map_size(H, W)
eng = vos.PropagationEngine(Hd, Wd, device=0, ref_num=case['ref_num'], frame_range=case['range'])
eng.begin_video(ann)
fd = torch.from_numpy(np.pad(feats, ((0,0),(0,256-feats.shape[1]),(0,0),(0,0)))).cuda()
gp = g['g6_roll_label_preds']
for t in range(feats.shape[0]):
    p, m = eng.step(fd[t])
    if p is None: continue
    p = p.cpu().numpy(); err = np.abs(p - gp[t-1])
    k, c = np.unravel_index(err.argmax(), err.shape)
    print(f'frame {t}: max err {err.max():.4f} at class {k} col {c} (got {p[k,c]:.4f} want {gp[t-1][k,c]:.4f}); colsum got {p[:,c].sum():.4f} want {gp[t-1][:,c].sum():.4f}; cols with err>0.01: {np.unique(np.where(err>0.01)[1])[:24].tolist()}')
