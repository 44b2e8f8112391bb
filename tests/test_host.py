"""Host-side logic and the C-ABI surface, on CPU (no GPU, no compute calls through the ABI)."""
import collections
import ctypes
import importlib
import json
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import inputs as gin

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol(vos):
    hdr = (ROOT / 'include' / 'vosprop.h').read_text()
    declared = sorted(set(re.findall(r'\b(vosprop_[a-z_]+)\s*\(', hdr)))
    assert len(declared) >= 12
    lib = vos._native.lib()
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/vosprop.h but not exported'
    assert sorted(vos._native.SYMBOLS) == declared
    assert b'gfx950' in lib.vosprop_version()


def test_config_struct_matches_header(vos):
    lib = vos._native.lib()
    cfg = vos._native.Config()
    lib.vosprop_default_config(ctypes.byref(cfg), 60, 107)
    # reference CLI defaults, src/inference.py:19-31
    assert (cfg.ref_num, cfg.frame_range, cfg.sigma1, cfg.sigma2, cfg.temperature) == (9, 40, 8.0, 21.0, 1.0)
    assert (cfg.feat_h, cfg.feat_w, cfg.channels, cfg.probability, cfg.topk) == (60, 107, 256, 0, 0)


@pytest.mark.parametrize('rng,nref', gin.G1_CASES)
def test_sample_frames_through_the_abi(vos, goldens, rng, nref):
    g = goldens[f'g1_sample_r{rng}_n{nref}']
    for fi in range(1, 121):
        assert vos.sample_frames_list(fi, rng, nref) == [int(v) for v in g[fi - 1] if v >= 0]


def test_sample_frames_with_fewer_than_three_references(vos):
    """ref_num < CONTINUOUS_FRAME - 1: past frame ref_num the reference asks np.linspace for a negative number of samples and
    raises ValueError (src/model/predict.py:80-85) - the ABI reports VOSPROP_E_INVALID there instead of returning 3 indices for
    a 2-slot ring (round-1 advisor finding); ref_num == 3 gives the three continuous frames alone, as numpy's empty linspace does."""
    from oracle import vos_oracle as vo
    for nref in (1, 2):
        for fi in range(1, nref + 1):
            assert vos.sample_frames_list(fi, 40, nref) == vo.sample_frames(fi, 40, nref) == list(range(fi))
        for fi in (nref + 1, nref + 5, 30):
            with pytest.raises(ValueError):
                vo.sample_frames(fi, 40, nref)
            with pytest.raises(ValueError):
                vos.sample_frames_list(fi, 40, nref)
    for fi in (4, 9, 57):
        assert vos.sample_frames_list(fi, 40, 3) == vo.sample_frames(fi, 40, 3) == [fi - 3, fi - 2, fi - 1]


def test_create_without_gpu_fails_loudly(vos):
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(vos.VospropError):
        vos.PropagationEngine(8, 8)
    pred = importlib.import_module('semi-supervised-vos_amd.predict')
    with pytest.raises(vos.VospropError):       # no silent CPU fallback behind the reference's predict() signature
        pred.predict(torch.zeros(1, 256, 4, 4), torch.zeros(256, 4, 4), torch.zeros(2, 1, 16), None, None, 1, 40, 9, 1.0, True)


def test_missing_library_is_an_error(vos, tmp_path):
    code = ("import importlib, os; os.environ['VOSPROP_LIB'] = r'%s';"
            "v = importlib.import_module('semi-supervised-vos_amd');\n"
            "try:\n    v._native.lib(); print('LOADED')\nexcept v.VospropError as e:\n    print('ERR', 'no CPU fallback' in str(e))")
    out = subprocess.run([sys.executable, '-c', code % (tmp_path / 'nope.so')], cwd=ROOT, capture_output=True, text=True)
    assert 'ERR True' in out.stdout, out.stdout + out.stderr


def test_spatial_prior_matches_reference_weights(goldens):
    pred = importlib.import_module('semi-supervised-vos_amd.predict')
    w = pred.get_spatial_weight((4, 6), 8.0)
    assert np.array_equal(w.dense().numpy(), goldens['g2_w_4x6_s8'])
    # sigma is recovered from a dense matrix the reference (or anyone) hands to predict()
    assert pred._sigma_of(torch.from_numpy(goldens['g2_w_4x6_s8']), 6) == pytest.approx(8.0, rel=1e-5)
    assert pred._sigma_of(pred.get_spatial_weight((30, 54), 21.0).dense(), 54) == pytest.approx(21.0, rel=1e-5)


def test_get_labels_matches_reference(goldens):
    pred = importlib.import_module('semi-supervised-vos_amd.predict')
    pred.Config.DEVICE = torch.device('cpu')
    mask = gin.g3_mask()
    lab = pred.get_labels(torch.from_numpy(mask.astype(np.int64)), 4, 240, 427, 30, 54)
    assert np.array_equal(lab.numpy(), goldens['g3_labels'])


def test_lpt_sharding_is_a_partition():
    sh = importlib.import_module('semi-supervised-vos_amd.sharding')
    rs = np.random.RandomState(0)
    lengths = {f'v{i:02d}': int(rs.randint(34, 105)) for i in range(30)}     # DAVIS-2017-val-like
    for world in (1, 2, 4, 8):
        shards, load = sh.lpt_assign(lengths, world)
        flat = sorted(v for s in shards for v in s)
        assert flat == sorted(lengths) and len(set(flat)) == 30
        assert max(load) <= sum(lengths.values()) / world + max(lengths.values())
        assert [sh.shard_for_rank(lengths, r, world) for r in range(world)] == [sorted(s) for s in shards]


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, str(ROOT))
    sh = importlib.import_module('semi-supervised-vos_amd.sharding')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lengths = {f'clip{i}': 30 + 7 * i for i in range(9)}
    mine = sh.shard_for_rank(lengths, rank, world)
    frames = torch.tensor([sum(lengths[v] for v in mine)], dtype=torch.int64)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)          # host-side bookkeeping only: the data path has no collective
    dist.all_reduce(frames)
    q.put((rank, gathered, int(frames.item())))
    dist.destroy_process_group()


def test_two_rank_sharding_over_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    for _, gathered, total in res:
        assert sorted(gathered[0] + gathered[1]) == sorted(f'clip{i}' for i in range(9))
        assert not set(gathered[0]) & set(gathered[1])
        assert total == sum(30 + 7 * i for i in range(9))


def test_dataset_order_and_normalisation(tmp_path):
    from PIL import Image
    ds_mod = importlib.import_module('semi-supervised-vos_amd.datasets')
    rs = np.random.RandomState(3)
    for vid in ('zebra', 'ant'):
        d = tmp_path / 'JPEGImages' / '480p' / vid
        d.mkdir(parents=True)
        for i in (2, 0, 1):
            Image.fromarray(rs.randint(0, 255, (16, 24, 3), dtype=np.uint8)).save(d / f'{i:05d}.png')
    ds = ds_mod.InferenceDataset(tmp_path / 'JPEGImages' / '480p')
    assert [n for _, n in ds.imgs] == ['ant'] * 3 + ['zebra'] * 3
    assert [Path(p).name for p, _ in ds.imgs[:3]] == ['00000.png', '00001.png', '00002.png']
    x, name = ds[0]
    raw = np.asarray(Image.open(ds.imgs[0][0]).convert('RGB'), np.float32) / 255
    want = (raw - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
    assert name == 'ant' and x.shape == (3, 16, 24) and np.allclose(x.numpy(), want.transpose(2, 0, 1), atol=1e-6)
    assert len(ds_mod.InferenceDataset(tmp_path / 'JPEGImages' / '480p', videos=['zebra'])) == 3


def test_save_predictions_png_mode_p(tmp_path, goldens):
    from PIL import Image
    utils = importlib.import_module('semi-supervised-vos_amd.utils')
    masks = goldens['g6_roll_label_masks'][:3]
    pal = gin.DAVIS_PALETTE + [0] * (768 - len(gin.DAVIS_PALETTE))
    utils.save_predictions(masks, pal, str(tmp_path), 'clipA')
    for i in range(3):
        im = Image.open(tmp_path / 'clipA' / f'{i + 1:05d}.png')
        assert im.mode == 'P' and np.array_equal(np.asarray(im), masks[i]) and im.getpalette()[:24] == gin.DAVIS_PALETTE


def test_load_model_accepts_reference_checkpoint_forms(tmp_path):
    utils = importlib.import_module('semi-supervised-vos_amd.utils')
    vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
    utils.Config.DEVICE = torch.device('cpu')
    net = vn.VOSNet('resnet18')
    sd = net.state_dict()
    torch.save({'state_dict': sd, 'epoch': 3}, tmp_path / 'a.pth.tar')
    torch.save({('module.' + k): v for k, v in sd.items()}, tmp_path / 'b.pth.tar')
    for f in ('a.pth.tar', 'b.pth.tar'):
        m = utils.load_model(vn.VOSNet('resnet18'), str(tmp_path / f))
        assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), sd.values()))
    with pytest.raises(FileNotFoundError):
        utils.load_model(net, str(tmp_path / 'missing.pth'))


def test_cli_flags_match_the_reference():
    inf = importlib.import_module('semi-supervised-vos_amd.inference')
    opts = {o.name: o for o in inf.inference_command.params}
    want = {'ref_num': 9, 'model': 'resnet50', 'temperature': 1.0, 'frame_range': 40, 'sigma_1': 8.0, 'sigma_2': 21.0,
            'device': 'cuda', 'inference_strategy': 'single', 'additional_model_type': 'resnet50', 'probability': False,
            'scale': 1.15, 'fusion': 'mean'}                     # reference src/inference.py:19-47
    for k, v in want.items():
        assert opts[k].default == v, k
    for k in ('data', 'resume', 'save'):
        assert opts[k].required
    assert '-n' in opts['ref_num'].opts and '-d' in opts['data'].opts and '-r' in opts['resume'].opts
    assert '-m' in opts['model'].opts and '-t' in opts['temperature'].opts and '-s' in opts['save'].opts
    assert list(opts['inference_strategy'].type.choices) == ['single', 'hor-flip', 'vert-flip', '2-scale', 'multimodel',
                                                             'hor-2-scale', '3-scale']


def test_wrappers_keep_the_reference_signatures(vos):
    """inference_hor_flip(model, loader, total_len, annotation_dir, last_video, save, sigma_1, sigma_2, frame_range, ref_num,
    temperature, probability_propagation, reduction_str, disable) etc. - positional order of the reference."""
    import inspect
    iu = vos.inference_utils
    base = ['inference_loader', 'total_len', 'annotation_dir', 'last_video', 'save', 'sigma_1', 'sigma_2', 'frame_range',
            'ref_num', 'temperature', 'probability_propagation']
    want = {
        'inference_hor_flip': ['model'] + base + ['reduction_str', 'disable'],
        'inference_ver_flip': ['model'] + base + ['reduction_str', 'disable'],
        'inference_2_scale': ['model'] + base + ['scale', 'reduction_str', 'flip_pred', 'disable'],
        'inference_multimodel': ['model', 'additional_model'] + base + ['reduction_str', 'disable'],
        'inference_3_scale': ['model'] + base + ['scale', 'disable'],
    }
    for name, params in want.items():
        got = list(inspect.signature(getattr(iu, name)).parameters)[:len(params)]
        assert got == params, name


@pytest.mark.parametrize('strategy', ['hor-flip', 'vert-flip', '2-scale', 'hor-2-scale', 'multimodel'])
@pytest.mark.parametrize('prob,fusion', [(False, 'mean'), (True, 'mean'), (True, 'maximum'), (True, 'minimum')])
def test_fusion_matches_the_oracle(vos, strategy, prob, fusion):
    """fuse_two (device-agnostic torch) == the oracle's restatement of the reference's per-frame fusion, including the
    class-axis fliplr of probability mode."""
    from oracle import vos_oracle as vo
    rs = np.random.RandomState(3)
    d, H, W = 4, 6, 6            # square on purpose: fliplr of (1,d,H,W) with d != W would not even be an image flip
    ua = torch.from_numpy(rs.uniform(size=(1, d, H, W)).astype(np.float32))
    ub = torch.from_numpy(rs.uniform(size=(1, d, H, W)).astype(np.float32))
    unflip = vo.TWO_BRANCH[strategy][2]
    want = vo.fuse_two(ua, ub, prob, fusion, unflip).numpy()
    spec = vos.inference_utils._TWO_BRANCH[strategy]
    assert spec['unflip'] == unflip and spec['scaled'] == vo.TWO_BRANCH[strategy][1]
    if prob:
        got = vos.inference_utils.fuse_two(ua, ub, True, fusion, spec['unflip'])
    else:
        got = vos.inference_utils.fuse_two(ua.argmax(1)[0].to(torch.uint8), ub.argmax(1)[0].to(torch.uint8), False, fusion,
                                           spec['unflip'])
    assert np.array_equal(got.numpy(), want)


def test_lowres_class_map_equals_get_labels(goldens):
    from oracle import vos_oracle as vo
    import inputs as gin
    m = gin.g3_mask()
    H, W = m.shape
    for (hd, wd) in [(30, 54), (35, 62), (27, 49)]:
        cls = importlib.import_module('semi-supervised-vos_amd').inference_utils.lowres_class_map(m, hd, wd)
        oh = vo.get_labels(m.astype(np.int64), int(m.max()) + 1, H, W, hd, wd)[:, 0].numpy()
        assert np.array_equal(oh.argmax(0).reshape(hd, wd), cls) and np.all(oh.sum(0) == 1)


def test_dataset_strategy_pairs(tmp_path, vos):
    """hor-flip / vert-flip / 2-scale / hor-2-scale items are (frame, transformed frame) pairs (reference datasets.py:148-162)."""
    from PIL import Image
    ds_mod = importlib.import_module('semi-supervised-vos_amd.datasets')
    rs = np.random.RandomState(0)
    (tmp_path / 'v').mkdir()
    img = rs.randint(0, 255, size=(24, 40, 3)).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / 'v' / '00000.png')
    base, _ = ds_mod.InferenceDataset(tmp_path)[0]
    (a, b), name = ds_mod.InferenceDataset(tmp_path, inference_strategy='hor-flip')[0]
    assert name == 'v' and torch.equal(a, base) and torch.equal(b, base.flip(-1))
    (a, b), _ = ds_mod.InferenceDataset(tmp_path, inference_strategy='vert-flip')[0]
    assert torch.equal(b, base.flip(-2))
    (a, b), _ = ds_mod.InferenceDataset(tmp_path, inference_strategy='2-scale', scale=1.15)[0]
    assert torch.equal(a, base) and tuple(b.shape) == (3, 28, 46)
    (a, b2), _ = ds_mod.InferenceDataset(tmp_path, inference_strategy='hor-2-scale', scale=1.15)[0]
    assert tuple(b2.shape) == (3, 28, 46) and not torch.equal(b, b2)
    x, _ = ds_mod.InferenceDataset(tmp_path, inference_strategy='3-scale', scale=1.15)[0]
    assert torch.equal(x, base)
    with pytest.raises(ValueError):
        ds_mod.InferenceDataset(tmp_path, inference_strategy='diagonal')


def test_prepare_first_frame_strategy_forms(tmp_path, vos):
    """Return arities / shapes of prepare_first_frame for every strategy (reference predict.py:128-155)."""
    import inputs as gin
    P = importlib.import_module('semi-supervised-vos_amd.predict')
    vos.Config.DEVICE = torch.device('cpu')
    case = gin.STRATEGY_CASE
    ann = gin.write_rollout_annotation(case, tmp_path)
    H, W = case['image_hw']
    lab, d, pal, wd, ws = P.prepare_first_frame('v', None, ann)
    assert tuple(lab.shape) == (d, 1, 12 * 20) and wd.sigma == 8 and ws.sigma == 21
    l, r, d2, pal, wd, ws = P.prepare_first_frame('v', None, ann, inference_strategy='hor-flip')
    assert tuple(r.shape) == tuple(l.shape)
    l, r, *_ = P.prepare_first_frame('v', None, ann, inference_strategy='ver-flip')
    assert tuple(r.shape) == tuple(l.shape)
    (l, l2), d, pal, (wd, wd2), (ws, ws2) = P.prepare_first_frame('v', None, ann, inference_strategy='2-scale', scale=1.15)
    assert tuple(l2.shape) == (d, 1, 14 * 23) and wd2.shape == (14, 23) and wd.shape == (12, 20)
    l3, d, pal, wd3, ws3 = P.prepare_first_frame('v', None, ann, inference_strategy='3-scale', scale=0.9,
                                                  probability_propagation=True)
    assert tuple(l3.shape) == (d, 1, 11 * 18) and wd3 is None and ws3 is None
    assert len(P.prepare_first_frame('v', None, ann, inference_strategy='multimodel')) == 5


def test_async_mask_writer(tmp_path, vos, goldens):
    """AsyncMaskWriter writes exactly what save_predictions writes (names, mode P, palette, pixels), from tensors or arrays,
    several videos in flight, and close() re-raises a failed job."""
    from PIL import Image
    io = importlib.import_module('semi-supervised-vos_amd.io_pipeline')
    U = importlib.import_module('semi-supervised-vos_amd.utils')
    pal = [int(v) for v in goldens['g6_roll_palette']]
    rs = np.random.RandomState(1)
    vids = {f'v{i}': rs.randint(0, 4, size=(5, 24, 32)).astype(np.uint8) for i in range(4)}
    w = io.AsyncMaskWriter(str(tmp_path / 'a'), workers=3)
    for k, (name, m) in enumerate(vids.items()):
        w.submit(name, pal, [torch.from_numpy(x) for x in m] if k % 2 else m)
    assert w.close() == 20
    for name, m in vids.items():
        U.save_predictions(m, pal, str(tmp_path / 'b'), name)
        for i in range(1, 6):
            a = Image.open(tmp_path / 'a' / name / f'{i:05d}.png')
            b = Image.open(tmp_path / 'b' / name / f'{i:05d}.png')
            assert a.mode == b.mode == 'P' and a.getpalette() == b.getpalette()
            assert np.array_equal(np.asarray(a), np.asarray(b)) and np.array_equal(np.asarray(a), m[i - 1])
    assert io.AsyncMaskWriter(None).close() == 0
    bad = io.AsyncMaskWriter(str(tmp_path / 'file_in_the_way'), workers=1)
    (tmp_path / 'file_in_the_way').write_text('x')
    bad.submit('v', pal, vids['v0'])
    with pytest.raises(Exception):
        bad.close()


def test_loader_keeps_the_reference_order_with_many_workers(tmp_path, vos):
    from PIL import Image
    io = importlib.import_module('semi-supervised-vos_amd.io_pipeline')
    ds_mod = importlib.import_module('semi-supervised-vos_amd.datasets')
    for v in ('b', 'a'):
        (tmp_path / v).mkdir()
        for i in range(5):
            Image.fromarray(np.full((8, 8, 3), 10 * i + (100 if v == 'b' else 0), np.uint8)).save(tmp_path / v / f'{i:05d}.png')
    ds = ds_mod.InferenceDataset(tmp_path)
    ref = [(x, (n,)) for x, n in (ds[i] for i in range(len(ds)))]
    got = list(io.make_loader(ds, io_workers=3, pin=False))
    assert [n for _, (n,) in got] == ['a'] * 5 + ['b'] * 5
    for (x, _), (y, _) in zip(ref, got):
        assert torch.equal(x[None], y)
    assert 1 <= io.default_io_workers() <= 8


def test_raw_uint8_dataset_and_device_normalisation(tmp_path, vos):
    """The fast path (uint8 frames through the loader, ToTensor + Normalize after the H2D copy) gives the tensors the reference's
    dataset gives (here on the CPU device: the same torch f32 operations in the same order)."""
    from PIL import Image
    ds_mod = importlib.import_module('semi-supervised-vos_amd.datasets')
    rs = np.random.RandomState(3)
    (tmp_path / 'v').mkdir()
    Image.fromarray(rs.randint(0, 256, size=(20, 28, 3)).astype(np.uint8)).save(tmp_path / 'v' / '00000.png')
    ref, _ = ds_mod.InferenceDataset(tmp_path)[0]
    raw, _ = ds_mod.InferenceDataset(tmp_path, raw_uint8=True)[0]
    assert raw.dtype == torch.uint8 and tuple(raw.shape) == (20, 28, 3)
    assert torch.equal(ds_mod.normalize_on_device(raw[None])[0], ref)
    (a, b), _ = ds_mod.InferenceDataset(tmp_path, raw_uint8=True, inference_strategy='hor-flip')[0]
    assert torch.equal(b, a.flip(1))


def test_shm_frame_loader(tmp_path, vos):
    """ShmFrameLoader: in-order, bit-identical frames out of the shared ring with far fewer slots than frames (recycling), an
    oversized frame through the fallback path, and a decode failure surfaced in the consumer."""
    from PIL import Image
    io = importlib.import_module('semi-supervised-vos_amd.io_pipeline')
    ds_mod = importlib.import_module('semi-supervised-vos_amd.datasets')
    rs = np.random.RandomState(0)
    for v, (h, w) in (('a', (16, 24)), ('b', (16, 24)), ('c', (20, 24))):      # video c does not fit a slot
        (tmp_path / v).mkdir()
        for i in range(9):
            Image.fromarray(rs.randint(0, 256, (h, w, 3)).astype(np.uint8)).save(tmp_path / v / f'{i:05d}.png')
    ds = ds_mod.InferenceDataset(tmp_path, raw_uint8=True)
    L = io.ShmFrameLoader(ds, workers=3, slots=6, register=False)
    assert len(L) == 27
    got, batch = [], []
    for x, (name,) in L:
        got.append((x.clone(), name))
        batch.append(x)
        if len(batch) == 4:                  # the consumer holds a few frames, then hands them back
            L.recycle(batch, None)
            batch = []
    L.recycle(batch, None)
    L.close()
    assert [n for _, n in got] == ['a'] * 9 + ['b'] * 9 + ['c'] * 9
    for i, (x, _) in enumerate(got):
        assert x.dtype == torch.uint8 and torch.equal(x[0], ds[i][0])

    class Broken(ds_mod.InferenceDataset):
        def __getitem__(self, i):
            if i == 5:
                raise OSError('truncated file')
            return super().__getitem__(i)
    bad = io.ShmFrameLoader(Broken(tmp_path, raw_uint8=True), workers=2, slots=6, register=False)
    with pytest.raises(RuntimeError, match='truncated file'):
        for x, _ in bad:
            bad.recycle([x], None)
    bad.close()
    with pytest.raises(ValueError):
        io.ShmFrameLoader(ds_mod.InferenceDataset(tmp_path), workers=1)


@pytest.mark.parametrize('TT,NT,la', [(26, 1809, 0), (26, 1809, 1), (26, 201, 0), (7, 57, 0), (7, 459, 0), (7, 459, 3), (57, 4050, 0),
                                      (57, 4050, 2), (1, 8, 0), (1, 8, 5), (1, 1, 0), (33, 999, 0), (33, 999, 7), (64, 640, 0), (64, 640, 4)])
def test_work_plan_covers_every_unit_once(vos, TT, NT, la):
    """The segment table the kernels walk (engine.hip build_segments, through the vosprop_debug_plan test hook): every
    (target tile, reference tile) unit exactly once, a workgroup only touches its own XCD's eighth of the reference stream,
    loads are balanced, and the lockstep map keeps the workgroups of an XCD on the same reference tiles.  la = how many of the
    last target tile's 8 waves hold a column of the map (0: not told): prop_mask_kernel's plan gives that tile's faster steps
    (its other waves only stage) a longer head."""
    L = vos._native.lib()
    L.vosprop_debug_plan.restype = ctypes.c_int
    L.vosprop_debug_plan.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int]
    n = L.vosprop_debug_plan(TT, NT, None, 0, la)
    buf = (ctypes.c_int * (4 * n))()
    assert L.vosprop_debug_plan(TT, NT, buf, n, la) == n
    rows = np.ctypeslib.as_array(buf).reshape(n, 4)
    cover = np.zeros((TT, NT), np.int32)
    load = collections.Counter()
    for b, tt, r_lo, ns in rows:
        x = b % 8
        assert ns > 0 and x * NT // 8 <= r_lo and r_lo + ns <= (x + 1) * NT // 8
        cover[tt, r_lo:r_lo + ns] += 1
        w = 0.66 + 0.34 * la / 8 if (la and tt == TT - 1) else 1.0      # (step time on the last tile, build_segments' model)
        load[b] += ns * w + 16      # the lockstep map prices a segment start at 16 tile steps (VOSPROP_SEGCOST; measured)
    assert cover.min() == 1 and cover.max() == 1
    if TT * NT >= 8 * 32 * 8:
        per_xcd = collections.defaultdict(list)
        # (the last tile's primary cannot walk more than its whole part: with la it may finish early - it is left out)
        capped = {b for b, tt, r_lo, ns in rows if la and tt == TT - 1 and ns == (b % 8 + 1) * NT // 8 - (b % 8) * NT // 8}
        for b, v in load.items():
            if b not in capped:
                per_xcd[b % 8].append(v)
        for x, v in per_xcd.items():
            assert max(v) <= min(v) * 1.1 + 8, (x, sorted(v))
    if TT >= 32:
        # first round of the lockstep map: the 32 workgroups of XCD 0 start at the same reference tile with the same length
        first = {}
        for b, tt, r_lo, ns in rows:
            if b % 8 == 0 and b not in first:
                first[b] = (r_lo, ns)
        assert len(set(first.values())) == 1
    if (TT, NT, la) == (7, 459, 0):
        # 240p: too few steps to amortise a second segment start - every workgroup gets ONE segment (whole workgroups per tile)
        assert collections.Counter(rows[:, 0].tolist()).most_common(1)[0][1] == 1 and len(set(rows[:, 0].tolist())) == 256


@pytest.fixture(scope='module')
def engine_isa(tmp_path_factory):
    """csrc/engine.hip compiled to gfx950 ISA text with the build's own flags (__graft_entry__.build)."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not Path(hipcc).exists():
        pytest.skip('no hipcc')
    src = ROOT / 'semi-supervised-vos_amd' / 'csrc'
    out = tmp_path_factory.mktemp('isa') / 'engine.s'
    subprocess.run([hipcc, '-O3', '--offload-arch=gfx950', '-std=c++17', '-fno-slp-vectorize', '-mllvm', '-amdgpu-mfma-vgpr-form=1', '-S',
                    '--cuda-device-only', '-o', str(out), 'engine.hip'], cwd=src, check=True, capture_output=True, timeout=600)
    return out.read_text()


def test_mask_kernel_keeps_its_registers_out_of_scratch(engine_isa):
    """prop_mask_kernel's tile loop is one asm statement with ~200 hard-bound registers; hipcc parks values it cannot place in
    scratch.  The kernel runs WITHOUT scratch now; round 4 twice found out late that it did not: 84 bytes per lane when a C++ branch
    right behind the statement made hipcc spill 16 output registers per segment (+12 MB of HBM writes per 480p launch that only
    the WRITE_SIZE counter showed), and 20 bytes for a hoisted zero vector whose reload's vmcnt(0) held wave 0's first LDS-DMA
    pieces back until the target fragments had landed.  This keeps either from coming back unseen."""
    import re
    m = re.search(r'\.amdhsa_kernel _ZN7vosprop16prop_mask_kernel.*?\.amdhsa_private_segment_fixed_size (\d+)', engine_isa, re.S)
    assert m, 'prop_mask_kernel not found in the ISA'
    assert int(m.group(1)) == 0, f'prop_mask_kernel uses {m.group(1)} bytes of scratch per lane'


def test_no_compiler_generated_m0_reader_in_the_propagation_kernels(engine_isa):
    """glds16s / glds16s2 (csrc/prop_bf16.h) set M0 inside an asm statement and do not restore it (M0 is a reserved register: it
    cannot be put on a clobber list).  That is only sound while hipcc itself emits nothing that READS M0 in those kernels
    (v_movrel / s_movrel indirect indexing, s_sendmsg, GDS ops, its own `... lds` loads): this test compiles the engine to ISA and
    fails if any line of a propagation kernel mentions m0 outside the three forms the asm statements themselves produce."""
    import re
    text = engine_isa
    ours = (re.compile(r'^\s*s_add_u32 m0, \S+, \S+\s*$'), re.compile(r'^\s*s_mov_b32 m0, \S+\s*$'),
            re.compile(r'^\s*s_mov_b32 \S+, m0\s*$'))
    kernel, bad, seen = None, [], 0
    for ln in text.splitlines():
        m = re.match(r'^(_ZN7vosprop\w+):', ln)
        if m:
            kernel = m.group(1)
        if kernel and ('prop_dense_kernel' in kernel or 'prop_mask_kernel' in kernel):
            code = ln.split(';')[0]
            if re.search(r'\bm0\b', code):
                seen += 1
                if not any(p.match(code) for p in ours):
                    bad.append((kernel[:60], code.strip()))
    assert seen > 0, 'no LDS-DMA piece found: the check is not looking at the right kernels'
    assert not bad, bad[:5]


def test_mask_loop_is_generated_and_passes_its_hazard_check():
    """csrc/prop_mask_loop.inc - the hand-ordered tile loop of prop_mask_kernel - is the output of tools/gen_mask_loop.py: the
    generator derives every s_waitcnt lgkmcnt from a model of the LDS return queue and FAILS on any dependent pair closer than the
    wait states the hardware does not interlock (MFMA result -> VALU / other MFMA, VALU -> MFMA operand, transcendental -> VALU,
    M0 write -> LDS-DMA).  Regenerating must reproduce the committed file byte for byte (so the checks ran on what ships)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, str(ROOT / 'tools' / 'gen_mask_loop.py'), '--check'], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


def test_mask_loop_hazard_checker_catches_a_short_distance():
    """The checker is not vacuous: the same stream with the first softmax row pulled next to the chain's last MFMA must be refused."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('gen_mask_loop', ROOT / 'tools' / 'gen_mask_loop.py')
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    ins = [g.Ins('v_mfma_f32_32x32x16_bf16 v[128:143], v[192:195], v[64:67], v[128:143]', 'mfma',
                 reads=set(range(192, 196)) | set(range(64, 68)) | set(range(128, 144)), writes=range(128, 144))]
    ins += [g.Ins('s_nop 0', 'nop') for _ in range(3)]
    ins += [g.Ins('v_exp_f32 v52, v128', 'trans', reads=[128], writes=[52])]
    with pytest.raises(SystemExit):
        g.check_hazards(ins, 0)
    ins[1:4] = [g.Ins('s_nop 7', 'nop', nops=8), g.Ins('s_nop 3', 'nop', nops=4)]
    g.check_hazards(ins, 0)


def test_set_deterministic_switches_the_library_and_the_solver_family(monkeypatch):
    """inference.set_deterministic: small problems switch MIOpen's atomic split-K solver family off through its environment switch
    (read by the library before its first convolution), large ones keep the library's defaults; the pointwise-GEMM tuner's flag
    (vosprop_set_deterministic, process-wide) follows and reports its previous state.  No GPU needed: nothing is launched."""
    import os
    inf = importlib.import_module('semi-supervised-vos_amd.inference')
    native = importlib.import_module('semi-supervised-vos_amd._native')
    name = inf._MIOPEN_NONDETERMINISTIC_SOLVERS[0]
    monkeypatch.delenv(name, raising=False)
    native.lib().vosprop_set_deterministic(0)
    try:
        inf.set_deterministic(True, pixels_per_batch=32 * 12 * 20)          # the failing round-2 test's size
        assert os.environ.get(name) == '0'
        assert native.lib().vosprop_set_deterministic(1) == 1
        inf.set_deterministic(True, pixels_per_batch=32 * 60 * 107)         # 480p at the CLI's batch: defaults stay
        assert name not in os.environ
        inf.set_deterministic(True)                                         # unknown size: the safe choice
        assert os.environ.get(name) == '0'
        inf.set_deterministic(False)
        assert name not in os.environ and native.lib().vosprop_set_deterministic(0) == 0
    finally:
        os.environ.pop(name, None)
        native.lib().vosprop_set_deterministic(0)
        import torch
        torch.backends.cudnn.benchmark = False
