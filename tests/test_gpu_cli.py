"""End to end on the GPU: `main.py inference` over a tiny synthetic DAVIS-layout dataset, checked against the oracle."""
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


def _make_dataset(root, n_frames=12, H=96, W=160):
    from PIL import Image
    case = dict(gin.ROLLOUT_CASES[0], image_hw=(H, W))
    ann = gin.rollout_annotation(case)
    frames = {}
    for vid, seed in (('bear', 1), ('camel', 2)):
        (root / 'JPEGImages' / '480p' / vid).mkdir(parents=True)
        (root / 'Annotations' / '480p' / vid).mkdir(parents=True)
        rs = np.random.RandomState(seed)
        base = rs.randint(0, 255, (H // 8, W // 8, 3)).astype(np.float32)
        imgs = []
        for i in range(n_frames):
            base = np.clip(base + rs.randn(*base.shape) * 6, 0, 255)
            img = np.asarray(Image.fromarray(base.astype(np.uint8)).resize((W, H), Image.BILINEAR))
            Image.fromarray(img).save(root / 'JPEGImages' / '480p' / vid / f'{i:05d}.png')   # lossless, so the oracle sees the same pixels
            imgs.append(img)
        frames[vid] = imgs
        im = Image.fromarray(ann, mode='P')
        im.putpalette(gin.DAVIS_PALETTE + [0] * (768 - 24))
        im.save(root / 'Annotations' / '480p' / vid / '00000.png')
    return ann, frames


def test_inference_cli_matches_oracle(tmp_path):
    from PIL import Image
    vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
    ds = importlib.import_module('semi-supervised-vos_amd.datasets')
    ann, frames = _make_dataset(tmp_path / 'data')
    torch.manual_seed(0)
    net = vn.VOSNet('resnet18')
    ckpt = tmp_path / 'ckpt.pth.tar'
    torch.save({'state_dict': net.state_dict()}, ckpt)
    out = subprocess.run([sys.executable, 'main.py', 'inference', '-d', str(tmp_path / 'data'), '-r', str(ckpt), '-m',
                          'resnet18', '-s', str(tmp_path / 'out'), '--encoder-dtype', 'f32', '--ref_num', '5',
                          '--frame_range', '6'], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    stats = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"vosprop_stats"')][0])['vosprop_stats']
    assert stats['frames'] == 24 and stats['videos'] == 2
    net.eval().cuda()
    for vid, imgs in frames.items():
        with torch.no_grad():
            feats = torch.cat([net(ds.normalize_image(Image.fromarray(im))[None].cuda()) for im in imgs])
        # the engine stores features as bf16: the oracle gets the same rounded numbers (a random-init encoder on noise
        # frames gives near-tied logits, so feeding it un-rounded features would test the rounding, not the engine)
        feats = feats.to(torch.bfloat16).float().cpu().numpy()
        _, want = vo.rollout(ann, feats, 6, 5, 1.0, 8.0, 21.0, False)
        first = Image.open(tmp_path / 'out' / vid / '00000.png')
        assert first.mode == 'P' and np.array_equal(np.asarray(first), ann)
        got = np.stack([np.asarray(Image.open(tmp_path / 'out' / vid / f'{i:05d}.png')) for i in range(1, len(imgs))])
        assert Image.open(tmp_path / 'out' / vid / '00001.png').mode == 'P'
        assert np.mean(got != want) <= 0.01, f'{vid}: {np.mean(got != want) * 100:.2f} % of pixels differ'
        assert min(vo.mask_iou_per_object(want, got, int(ann.max()) + 1)) >= 0.97


def test_device_normalisation_equals_host_normalisation():
    """uint8 -> f32 ToTensor + Normalize on the GPU (the loader fast path) vs the reference's host transform: IEEE f32 division
    and subtraction on both sides, so the tensors are identical."""
    from PIL import Image
    ds = importlib.import_module('semi-supervised-vos_amd.datasets')
    rs = np.random.RandomState(0)
    img = Image.fromarray(rs.randint(0, 256, size=(64, 96, 3)).astype(np.uint8))
    host = ds.normalize_image(img)
    dev = ds.normalize_on_device(ds.raw_image(img)[None].cuda())[0].cpu()
    assert torch.equal(host, dev)


def test_c_abi_standalone_driver(tmp_path):
    """examples/cbench.cpp: a C++ program that only includes include/vosprop.h and links libvosprop.so (no Python, no torch in the
    process) runs a 480p clip through begin_video / step / time_last_propagation."""
    exe = tmp_path / 'cbench'
    lib_dir = ROOT / 'semi-supervised-vos_amd'
    build = subprocess.run(['hipcc', '-O2', '--offload-arch=gfx950', f'-I{ROOT / "include"}', str(ROOT / 'examples' / 'cbench.cpp'),
                            f'-L{lib_dir}', '-lvosprop', '-o', str(exe)], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    import os
    env = dict(os.environ, LD_LIBRARY_PATH=f'{lib_dir}:' + os.environ.get('LD_LIBRARY_PATH', ''))
    run = subprocess.run([str(exe), '24'], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    assert 'kernel' in run.stdout and 'N=9' in run.stdout


@pytest.mark.parametrize('strategy', ['hor-flip', '2-scale'])
def test_inference_cli_strategy_matches_oracle(tmp_path, strategy):
    """`main.py inference --inference-strategy ...` end to end (paired dataset items, two chains, GPU fusion, PNG output) against
    the oracle's restatement of the same strategy fed with the same encoder's features."""
    from PIL import Image, ImageOps
    vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
    ds = importlib.import_module('semi-supervised-vos_amd.datasets')
    ann, frames = _make_dataset(tmp_path / 'data', n_frames=10)
    torch.manual_seed(0)
    net = vn.VOSNet('resnet18')
    ckpt = tmp_path / 'ckpt.pth.tar'
    torch.save({'state_dict': net.state_dict()}, ckpt)
    scale = 1.25
    out = subprocess.run([sys.executable, 'main.py', 'inference', '-d', str(tmp_path / 'data'), '-r', str(ckpt), '-m',
                          'resnet18', '-s', str(tmp_path / 'out'), '--encoder-dtype', 'f32', '--ref_num', '5',
                          '--frame_range', '6', '--inference-strategy', strategy, '--scale', str(scale)],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    net.eval().cuda()

    def feats_of(imgs):
        with torch.no_grad():
            f = torch.cat([net(ds.normalize_image(im)[None].cuda()) for im in imgs])
        return f.to(torch.bfloat16).float().cpu().numpy()        # the engine keeps features as bf16

    for vid, arrs in frames.items():
        imgs = [Image.fromarray(a) for a in arrs]
        if strategy == 'hor-flip':
            second = [ImageOps.mirror(im) for im in imgs]
        else:
            size2 = tuple(int(v) for v in np.ceil(np.array(imgs[0].size) * scale))
            second = [im.resize(size2, Image.LANCZOS) for im in imgs]
        want = vo.rollout_two_branch(strategy, ann, feats_of(imgs), feats_of(second), scale=scale, frame_range=6, ref_num=5)
        got = np.stack([np.asarray(Image.open(tmp_path / 'out' / vid / f'{i:05d}.png')) for i in range(1, len(imgs))])
        assert got.shape == want.shape
        assert np.mean(got != want) <= 0.015, f'{vid}: {np.mean(got != want) * 100:.2f} % of pixels differ'


def test_inference_cli_multimodel_and_three_scale(tmp_path):
    """The two remaining dispatch paths of the CLI: `multimodel` (second checkpoint, element-wise maximum of the two chains' class
    maps) and `3-scale` (three passes, class maps at the reference's fixed 480x910, maximum of the three)."""
    from PIL import Image
    vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
    ds = importlib.import_module('semi-supervised-vos_amd.datasets')
    ann, frames = _make_dataset(tmp_path / 'data', n_frames=8)
    nets = []
    for i, seed in enumerate((0, 1)):
        torch.manual_seed(seed)
        nets.append(vn.VOSNet('resnet18'))
        torch.save({'state_dict': nets[-1].state_dict()}, tmp_path / f'ckpt{i}.pth.tar')
    common = [sys.executable, 'main.py', 'inference', '-d', str(tmp_path / 'data'), '-r', str(tmp_path / 'ckpt0.pth.tar'), '-m',
              'resnet18', '--encoder-dtype', 'f32', '--ref_num', '5', '--frame_range', '6']
    out = subprocess.run(common + ['-s', str(tmp_path / 'mm'), '--inference-strategy', 'multimodel', '--additional-model',
                                   str(tmp_path / 'ckpt1.pth.tar'), '--additional-model-type', 'resnet18'],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    out3 = subprocess.run(common + ['-s', str(tmp_path / 's3'), '--inference-strategy', '3-scale', '--scale', '1.25'],
                          cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out3.returncode == 0, out3.stderr[-2000:]
    for n in nets:
        n.eval().cuda()

    def feats_of(net, imgs, size=None):
        with torch.no_grad():
            xs = [ds.normalize_image(im)[None].cuda() for im in imgs]
            if size is not None:
                xs = [torch.nn.functional.interpolate(x, size=size, mode='nearest') for x in xs]
            f = torch.cat([net(x) for x in xs])
        return f.to(torch.bfloat16).float().cpu().numpy()

    for vid, arrs in frames.items():
        imgs = [Image.fromarray(a) for a in arrs]
        want = vo.rollout_two_branch('multimodel', ann, feats_of(nets[0], imgs), feats_of(nets[1], imgs), frame_range=6, ref_num=5)
        got = np.stack([np.asarray(Image.open(tmp_path / 'mm' / vid / f'{i:05d}.png')) for i in range(1, len(imgs))])
        assert np.mean(got != want) <= 0.015, f'multimodel {vid}: {np.mean(got != want) * 100:.2f} % of pixels differ'
        H, W = ann.shape
        scales = (0.9, 1.0, 1.25)
        f3 = [feats_of(nets[0], imgs, (int(np.ceil(H * s)), int(np.ceil(W * s)))) for s in scales]
        want3 = vo.rollout_3_scale(ann, f3, scales, output_size=(480, 910), frame_range=6, ref_num=5)
        got3 = np.stack([np.asarray(Image.open(tmp_path / 's3' / vid / f'{i:05d}.png')) for i in range(1, len(imgs))])
        assert got3.shape == want3.shape == (len(imgs) - 1, 480, 910)
        assert np.mean(got3 != want3) <= 0.02, f'3-scale {vid}: {np.mean(got3 != want3) * 100:.2f} % of pixels differ'


def test_inference_cli_sharded_over_two_processes(tmp_path):
    """`--gpus 2 --deterministic`: one child process per shard (here both on GPU 0 via VOSPROP_SHARD_DEVICES), whole videos dealt out
    by LPT, the parent sums the per-shard statistics; every PNG of the sharded run is BYTE-IDENTICAL to the one-process run's
    (SURVEY.md section 4 tier iii: videos are independent in the reference, src/utils/inference_utils.py:28-48, so sharding
    correctness is equality, not a tolerance).  Default encoder precision (f16, the reference's autocast).  Without
    --deterministic the two runs differ in ~1 % of the pixels of this noise dataset: MIOpen's f16 3x3 convolutions of small maps
    are not reproducible from launch to launch (tools/determinism_probe.py)."""
    import os
    from PIL import Image
    vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
    _make_dataset(tmp_path / 'data', n_frames=9)
    torch.manual_seed(0)
    torch.save({'state_dict': vn.VOSNet('resnet18').state_dict()}, tmp_path / 'ckpt.pth.tar')
    base = [sys.executable, 'main.py', 'inference', '-d', str(tmp_path / 'data'), '-r', str(tmp_path / 'ckpt.pth.tar'), '-m',
            'resnet18', '--ref_num', '5', '--frame_range', '6', '--deterministic']
    one = subprocess.run(base + ['-s', str(tmp_path / 'one')], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run(base + ['-s', str(tmp_path / 'two'), '--gpus', '2'], cwd=ROOT, capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, VOSPROP_SHARD_DEVICES='0,0'))
    assert two.returncode == 0, two.stderr[-2000:]
    summary = json.loads(two.stdout.strip().splitlines()[-1])
    assert summary['gpus'] == 2 and summary['frames'] == 18
    for vid in ('bear', 'camel'):
        for i in range(9):
            a = np.asarray(Image.open(tmp_path / 'one' / vid / f'{i:05d}.png'))
            b = np.asarray(Image.open(tmp_path / 'two' / vid / f'{i:05d}.png'))
            assert np.array_equal(a, b), f'{vid}/{i:05d}.png: {np.mean(a != b) * 100:.3f} % of pixels differ'
        # the masks are not degenerate: more than one class survives to the last frame
        assert len(np.unique(np.asarray(Image.open(tmp_path / 'two' / vid / '00008.png')))) >= 2


def test_deterministic_runs_repeat_bit_for_bit(tmp_path):
    """Two one-process `--deterministic` runs of the same job write identical PNGs, and `vosprop_set_deterministic` reports its
    previous state (the switch is process-wide)."""
    from PIL import Image
    vn = importlib.import_module('semi-supervised-vos_amd.vos_net')
    native = importlib.import_module('semi-supervised-vos_amd._native')
    assert native.lib().vosprop_set_deterministic(1) in (0, 1)
    assert native.lib().vosprop_set_deterministic(0) == 1
    _make_dataset(tmp_path / 'data', n_frames=7)
    torch.manual_seed(0)
    torch.save({'state_dict': vn.VOSNet('resnet18').state_dict()}, tmp_path / 'ckpt.pth.tar')
    base = [sys.executable, 'main.py', 'inference', '-d', str(tmp_path / 'data'), '-r', str(tmp_path / 'ckpt.pth.tar'), '-m',
            'resnet18', '--ref_num', '5', '--frame_range', '6', '--deterministic', '--encoder-batch', '4']
    for tag in ('a', 'b'):
        out = subprocess.run(base + ['-s', str(tmp_path / tag)], cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
    for vid in ('bear', 'camel'):
        for i in range(7):
            assert (tmp_path / 'a' / vid / f'{i:05d}.png').read_bytes() == (tmp_path / 'b' / vid / f'{i:05d}.png').read_bytes()
