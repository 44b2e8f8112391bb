import sys, ctypes, importlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import numpy as np, torch
import test_gpu_parity as tp
from oracle import vos_oracle as vo
vos = importlib.import_module('semi-supervised-vos_amd')
Hd, Wd, T, d, fi, k = 12, 20, 10, 4, 9, 20
feats, oh = tp._random_case(4321 + Hd * Wd + k, Hd, Wd, T, d)
dev = torch.device('cuda', 0)
eng = vos.PropagationEngine(Hd, Wd, device=0, topk=k)
wd, ws = vo.get_spatial_weight((Hd, Wd), 8.0), vo.get_spatial_weight((Hd, Wd), 21.0)
fd, ld = torch.from_numpy(feats).to(dev), torch.from_numpy(oh).to(dev)
got = eng.predict(fd[:fi], fd[fi], ld[:, :fi], fi, 40, 9, 1.0, 8.0, 21.0, False).cpu().numpy()
want = vo.predict(feats[:fi], feats[fi], oh[:, :fi], wd, ws, fi, 40, 9, 1.0, False, topk=k).numpy()
HW = Hd * Wd
L = vos._native.lib()
tg = np.zeros(HW, np.float32); te = np.zeros(HW, np.float32); gr = np.zeros(HW, np.int32)
L.vosprop_debug_topk.argtypes = [ctypes.c_void_p] * 4
print('dbg rc', L.vosprop_debug_topk(eng._ctx, tg.ctypes.data_as(ctypes.c_void_p), te.ctypes.data_as(ctypes.c_void_p), gr.ctypes.data_as(ctypes.c_void_p)))
# oracle E = S c + log2 w per column
idx = vo.sample_frames(fi, 40, 9)
R = torch.from_numpy(feats[idx]).permute(0, 2, 3, 1).reshape(-1, 256)
Tt = torch.from_numpy(feats[fi]).reshape(256, -1)
S = (R @ Tt).numpy() * 1.4426950408889634
W = np.concatenate([wd.numpy() if isinstance(wd, torch.Tensor) else wd] * len(idx), 0)
E = S + np.log2(np.maximum(W, 1e-300))
kth = np.sort(E, axis=0)[-k]
err = np.abs(got - want).max(0)
bad = np.where(err > 2e-4)[0]
print('bad columns', bad[:20], 'n', len(bad))
print('groups dumped: min/mean/max', gr.min(), gr.mean(), gr.max())
print('thr_elem <= kth everywhere:', np.all(te <= kth + 1e-3), 'worst', (te - kth).max())
for t in bad[:6]:
    print(t, 'err', err[t], 'te', te[t], 'tg', tg[t], 'kth', kth[t], 'groups', gr[t], 'got', got[:, t], 'want', want[:, t])
