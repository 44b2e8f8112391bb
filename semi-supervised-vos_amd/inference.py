"""`main.py inference` with the reference's flags (src/inference.py:18-113) plus --gpus / --encoder-dtype.

--gpus N > 1 starts one process per GPU; whole videos are dealt longest-first to the GPUs (sharding.py); there is no
collective on the data path, only a host-side sum of {frames, seconds} for the fps line."""
import json
import os
import subprocess
import sys
from pathlib import Path

import click
import torch
import torch.utils.data

from .config import Config
from .datasets import InferenceDataset, list_videos
from .io_pipeline import ShmFrameLoader, default_io_workers, make_loader
from .inference_utils import (EncoderNotReproducible, inference_2_scale, inference_3_scale, inference_hor_flip, inference_multimodel,
                              inference_single, inference_ver_flip)
from .sharding import shard_for_rank
from .utils import load_model
from .vos_net import GraphedEncoder, VOSNet

_DTYPES = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': None}


@click.command(name='inference')
@click.option('--ref_num', '-n', type=int, default=9, help='Number of reference frames for inference.')
@click.option('--data', '-d', type=click.Path(file_okay=False, dir_okay=True), required=True,
              help='Path to inference dataset folder.')
@click.option('--resume', '-r', type=click.Path(file_okay=True, dir_okay=False), required=True,
              help='Path to the trained checkpoint.')
@click.option('--model', '-m', type=click.Choice(['resnet18', 'resnet50', 'resnet101', 'facebook']), default='resnet50',
              help='Network architecture, resnet18, resnet50, resnet101 or facebook.')
@click.option('--temperature', '-t', type=float, default=1.0, help='Temperature parameter.')
@click.option('--frame_range', type=int, default=40, help='Range of frames for inference.')
@click.option('--sigma_1', type=float, default=8.0, help='Smaller sigma in the motion model for dense spatial weight')
@click.option('--sigma_2', type=float, default=21.0, help='Larger sigma in the motion model for dense spatial weight.')
@click.option('--save', '-s', type=click.Path(file_okay=False, dir_okay=True), required=True,
              help='Path to save predictions.')
@click.option('--device', type=click.Choice(['cpu', 'cuda']), default='cuda', help='Device to run computing on.')
@click.option('--inference-strategy',
              type=click.Choice(['single', 'hor-flip', 'vert-flip', '2-scale', 'multimodel', 'hor-2-scale', '3-scale']),
              default='single', help='Inference strategy.')
@click.option('--additional-model', type=click.Path(file_okay=True, dir_okay=False), required=False,
              help='Path to the additional checkpoint.')
@click.option('--additional-model-type', type=click.STRING, required=False, default='resnet50',
              help='Type of additional model type.')
@click.option('--probability/--no-probability', default=False, required=False,
              help='Should probability or labels be propagated.')
@click.option('--scale', default=1.15, required=False, type=click.FLOAT, help='Scale for 2nd image in 2-scale strategy.')
@click.option('--fusion', default='mean', type=click.Choice(['maximum', 'minimum', 'mean']),
              help='Fusion operation for probability propagation.')
@click.option('--gpus', type=int, default=1, help='[engine] GPUs of this node to shard the videos over.')
@click.option('--encoder-dtype', type=click.Choice(sorted(_DTYPES)), default='f16',
              help='[engine] encoder precision (default f16: the reference runs it under fp16 autocast on GPU).')
@click.option('--propagation-precision', type=click.Choice(['bf16', 'f32']), default='bf16',
              help='[engine] arithmetic of the propagation step: bf16 MFMA (fast path) or f32 MFMA with f32 features (the parity '
                   'path: the reference\'s CPU arithmetic, ~1/16 of the matrix rate).')
@click.option('--encoder-batch', type=int, default=32, help='[engine] frames per encoder call (look-ahead).')
@click.option('--io-workers', type=int, default=None,
              help='[engine] JPEG decode processes (default: min(8, cores - 1); the reference uses 1).')
@click.option('--png-workers', type=int, default=2, help='[engine] PNG encoder threads.')
@click.option('--miopen-find/--no-miopen-find', default=False,
              help='[engine] let MIOpen time its convolution solvers per batch shape (+16 % steady-state throughput, 10-20 s of '
                   'search per shape at start-up: for long jobs).')
@click.option('--encoder-graph/--no-encoder-graph', default=True,
              help='[engine] replay the encoder forward of full batches as one captured HIP graph.')
@click.option('--deterministic/--no-deterministic', default=False,
              help='[engine] reproducible masks: MIOpen restricted to its deterministic solvers (no atomic split-K), the pointwise-GEMM '
                   'algorithm of a layer a pure function of the problem (no timing race, no cache), encoder batches cut at video '
                   'boundaries. A --gpus N run then writes byte-identical PNGs to a one-process run (videos are independent, '
                   'reference inference_utils.py:28-48).')
@click.option('--shard', type=(int, int), default=(0, 1), hidden=True, help='[engine] internal: rank, world')
def inference_command(ref_num, data, resume, model, temperature, frame_range, sigma_1, sigma_2, save, device,
                      inference_strategy, additional_model, additional_model_type, probability, scale, fusion, gpus,
                      encoder_dtype, propagation_precision, encoder_batch, io_workers, png_workers, miopen_find, encoder_graph,
                      deterministic, shard):
    if gpus > 1 and shard == (0, 1):
        return _launch_shards(gpus)
    inference_command_impl(ref_num, data, resume, model, temperature, frame_range, sigma_1, sigma_2, save, device,
                           inference_strategy, additional_model, additional_model_type, probability, scale, fusion,
                           encoder_dtype=encoder_dtype, shard=shard, encoder_batch=encoder_batch, io_workers=io_workers,
                           png_workers=png_workers, encoder_graph=encoder_graph, miopen_find=miopen_find,
                           propagation_precision=propagation_precision, deterministic=deterministic)


# MIOpen solver families that are switched OFF in reproducible mode (environment variables read by MIOpen when it looks for a
# solver): the assembly implicit-GEMM family for NHWC forward convolutions is the one whose small-map kernels (`igemm_fwd_gtcx35_nhwc_
# ..._gkgs`: "gemm-k global split") split the reduction over workgroups and accumulate with atomics (tools/determinism_probe.py:
# 11-24 % of a 3x3 convolution's f16 outputs change from one launch to the next).
_MIOPEN_NONDETERMINISTIC_SOLVERS = ('MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC',)


# The split-K kernels are MIOpen's choice for SMALL problems only (too few output tiles to fill 256 CUs: 12x20 maps at batch 32 =
# 60 tiles); at >= this many stride-8 output pixels per encoder batch (480p at batch 32: 205 440) the family's kernels are the
# plain ones and reproducible (probe), and they are 7x faster than what the library falls back to without the family (a slow CK
# instance in immediate mode: 2.5 ms instead of 0.36 ms per 480p frame).  So the family is only switched off below the threshold -
# and whatever was decided, the CLI CHECKS the encoder for reproducibility before it trusts it (encoder_is_reproducible).
_SPLIT_K_PIXELS = 131072


_SET_BY_US = set()      # MIOpen environment switches THIS module set (and may therefore remove again)


def set_deterministic(on=True, pixels_per_batch=None):
    """Process-wide reproducible mode of the encoder (measured with tools/determinism_probe.py).  Call it
    before the first convolution of the process (the CLI does, first thing).  pixels_per_batch: encoder batch x ceil(H/8) x
    ceil(W/8) of the job (None = unknown: the safe, slow choice).
      * MIOpen: for small problems the solver family with atomic split-K kernels is disabled through its environment switch (read
        once by the library, hence "before the first convolution"), so the library picks its next choice - which is reproducible
        on this stack (the probe checks every convolution call bitwise).  NOT `torch.backends.cudnn.deterministic`: that sets
        MIOPEN_CONVOLUTION_ATTRIB_DETERMINISTIC, and this MIOpen then falls back to its naive reference kernel
        (`naive_conv_ab_nonpacked_fwd_nhwc`, 6.6 ms per call at 12x20 maps: `main.py inference` at 480p ran at 6 frames/s
        instead of 820 - correct, useless);
      * a timed solver search (--miopen-find) is a race between solvers: off;
      * vosprop_set_deterministic: the pointwise-GEMM algorithm of a layer is the first gated candidate in the library's rank
        order - no timing race between candidates, no cache file - so every process picks the same kernel.
    The propagation kernels need nothing: fixed work map, partials merged in a fixed order, no atomics on values."""
    from . import _native
    small = pixels_per_batch is None or pixels_per_batch < _SPLIT_K_PIXELS
    for name in _MIOPEN_NONDETERMINISTIC_SOLVERS:
        if name in os.environ and name not in _SET_BY_US:
            continue                        # the user's own setting of a MIOpen switch is never touched
        if on and small:
            os.environ[name] = '0'
            _SET_BY_US.add(name)
        elif name in _SET_BY_US:
            os.environ.pop(name, None)
            _SET_BY_US.discard(name)
    if on:
        torch.backends.cudnn.benchmark = False
    _native.lib().vosprop_set_deterministic(1 if on else 0)


def encoder_is_reproducible(net, shape, dtype, device, repeats=2):
    """Encode the same random batch `repeats` + 1 times and compare the features bitwise (the deterministic mode's self-check: a
    library update or an unusual frame size must not silently bring an atomic-accumulate kernel back)."""
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.randn(shape, generator=g).to(device)
    if dtype is not None:
        x = x.to(dtype)
    x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        net(x)                                   # warm: solver look-ups, GEMM plans
        ref = net(x).clone()
        same = all(torch.equal(net(x), ref) for _ in range(repeats))
    torch.cuda.synchronize()
    return same


def visible_devices(n):
    """Device ordinals (as strings for HIP_VISIBLE_DEVICES) of the first n GPUs this process is allowed to use: the entries of
    an inherited HIP_VISIBLE_DEVICES / ROCR-style list when one is set, else 0..n-1."""
    inherited = [d.strip() for d in os.environ.get('HIP_VISIBLE_DEVICES', '').split(',') if d.strip() != '']
    if inherited:
        if len(inherited) < n:
            raise SystemExit(f'--gpus {n} but HIP_VISIBLE_DEVICES={os.environ["HIP_VISIBLE_DEVICES"]} lists {len(inherited)}')
        return inherited[:n]
    return [str(i) for i in range(n)]


def _launch_shards(gpus):
    """One child process per GPU (HIP_VISIBLE_DEVICES pins it); children never exec after touching the GPU."""
    argv = [a for a in sys.argv[1:]]
    procs = []
    # VOSPROP_SHARD_DEVICES="0,0": device ordinal per shard (testing the sharded path on a box with fewer GPUs than shards)
    devs = [d for d in os.environ.get('VOSPROP_SHARD_DEVICES', '').split(',') if d != '']
    if not devs:      # shard r -> the r-th device THIS process may see: an inherited HIP_VISIBLE_DEVICES is a list to index into
        devs = visible_devices(gpus)
    for r in range(gpus):
        env = dict(os.environ, HIP_VISIBLE_DEVICES=devs[r % len(devs)], HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, sys.argv[0]] + argv + ['--shard', str(r), str(gpus)], env=env,
                                      stdout=subprocess.PIPE, text=True))
    frames, secs, rc = 0, 0.0, 0
    for p in procs:
        out, _ = p.communicate()
        rc |= p.returncode
        for line in out.splitlines():
            if line.startswith('{"vosprop_stats"'):
                st = json.loads(line)['vosprop_stats']
                frames += st['frames']
                secs = max(secs, st['seconds'])
    print(json.dumps({'gpus': gpus, 'frames': frames, 'seconds': secs, 'frames_per_s': frames / secs if secs else 0.0}))
    if rc:
        raise SystemExit(rc)


def inference_command_impl(ref_num, data, resume, model, temperature, frame_range, sigma_1, sigma_2, save, device,
                           inference_strategy, additional_resume, additional_model_type, probability_propagation, scale,
                           reduction, disable=False, encoder_dtype='f16', shard=(0, 1), encoder_batch=32, io_workers=None,
                           png_workers=2, encoder_graph=True, miopen_find=False, propagation_precision='bf16',
                           deterministic=False):
    if deterministic:
        # before the first convolution of the process: the job's frame size decides which MIOpen solvers are allowed
        first = next(iter(list_videos(str(Path(data) / 'JPEGImages/480p')).values()), None)
        det_shape = None
        if first:
            from PIL import Image
            with Image.open(first[0]) as im0:
                w0, h0 = im0.size
            det_shape = (max(1, encoder_batch), 3, h0, w0)
        # strategies with scaled branches (2-scale, hor-2-scale, 3-scale) encode other sizes as well: unknown = the safe choice
        # (solver family off); every distinct input shape is verified bitwise on its first batch anyway (encoded_branches)
        same_size = inference_strategy in ('single', 'multimodel', 'hor-flip', 'vert-flip')
        set_deterministic(True, None if (det_shape is None or not same_size) else det_shape[0] * -(-h0 // 8) * -(-w0 // 8))
        miopen_find = False        # a timed solver search is a race between solvers: its winner can differ from process to process
    if Config.DEVICE.type != device:
        Config.DEVICE = torch.device(device)
    if Config.DEVICE.type == 'cuda':
        Config.DEVICE = torch.device('cuda', torch.cuda.current_device())
    net = VOSNet(model=model)
    net = load_model(net, resume)
    dtype = _DTYPES[encoder_dtype] if Config.DEVICE.type == 'cuda' else None
    # an f16 / bf16 encoder hands its channels-last features over as they are: the propagation kernel reads the target frame in
    # place (f16 is converted to bf16 as it is loaded) and combine_kernel carries the ring copy - no push launch per frame
    fdt = None
    net.prepare_for_inference(Config.DEVICE, dtype, miopen_find=miopen_find, feature_dtype=fdt)
    additional = None
    if inference_strategy == 'multimodel':      # reference src/inference.py:65-71
        if not additional_resume:
            raise click.UsageError("--inference-strategy multimodel needs --additional-model")
        additional = load_model(VOSNet(model=additional_model_type), additional_resume)
        additional.prepare_for_inference(Config.DEVICE, dtype, miopen_find=miopen_find, feature_dtype=fdt)

    if deterministic and Config.DEVICE.type == 'cuda' and det_shape is not None and inference_strategy in ('single', 'multimodel'):
        for enc in [net] + ([additional] if additional is not None else []):
            if not encoder_is_reproducible(enc, det_shape, dtype, Config.DEVICE):
                raise click.ClickException('--deterministic: the encoder is NOT bit-reproducible for input '
                                           f'{det_shape} on this software stack (a library kernel accumulates with atomics); '
                                           'see DESIGN.md section 7.1 / tools/determinism_probe.py')
    if Config.DEVICE.type == 'cuda' and encoder_graph:
        # full batches of one resolution replay a captured HIP graph; everything else (last batch of a video, another
        # resolution) runs the eager module
        net = GraphedEncoder(net)
        if additional is not None:
            additional = GraphedEncoder(additional)

    data_dir = str(Path(data) / 'JPEGImages/480p')
    videos = None
    if shard[1] > 1:
        lengths = {k: len(v) for k, v in list_videos(data_dir).items()}
        videos = shard_for_rank(lengths, shard[0], shard[1])
    dataset = InferenceDataset(data_dir, disable=disable, inference_strategy=inference_strategy, scale=scale, videos=videos,
                               raw_uint8=Config.DEVICE.type == 'cuda')
    n_io = default_io_workers() if io_workers is None else io_workers
    single_tensor = inference_strategy in ('single', 'multimodel', '3-scale')
    if Config.DEVICE.type == 'cuda' and single_tensor and n_io > 0 and len(dataset) > 0:
        loader = ShmFrameLoader(dataset, workers=n_io)       # zero-copy decode ring (io_pipeline.py)
    else:
        loader = make_loader(dataset, n_io)
    annotation_dir = Path(data) / 'Annotations/480p'
    if len(dataset) == 0:
        print(json.dumps({'vosprop_stats': {'frames': 0, 'videos': 0, 'seconds': 0.0, 'shard': list(shard)}}))
        return
    last_video = dataset.imgs[0][1]
    stats = {}
    with torch.no_grad():
        head = (loader, len(dataset), annotation_dir, last_video, save, sigma_1, sigma_2, frame_range, ref_num,
                temperature, probability_propagation)
        opts = dict(encoder_dtype=dtype, stats=stats, encoder_batch=encoder_batch, png_workers=png_workers,
                    precision={'bf16': 0, 'f32': 1}[propagation_precision], align_videos=deterministic)
        try:
            _run_strategy(inference_strategy, net, additional, head, opts, disable, reduction, scale)
        except EncoderNotReproducible as e:      # --deterministic: a batch of some input shape did not repeat bit for bit
            raise click.ClickException(str(e))
    if hasattr(loader, 'close'):
        loader.close()
    stats['shard'] = list(shard)
    print(json.dumps({'vosprop_stats': stats}))


def _run_strategy(inference_strategy, net, additional, head, opts, disable, reduction, scale):
    """The reference's dispatch over its strategies (src/inference.py:84-103)."""
    if True:
        if inference_strategy == 'single':
            inference_single(net, *head, disable, **opts)
        elif inference_strategy == 'hor-flip':
            inference_hor_flip(net, *head, reduction, disable, **opts)
        elif inference_strategy == 'vert-flip':
            inference_ver_flip(net, *head, reduction, disable, **opts)
        elif inference_strategy == '2-scale':
            inference_2_scale(net, *head, scale, reduction, False, disable, **opts)
        elif inference_strategy == 'multimodel':
            inference_multimodel(net, additional, *head, reduction, disable, **opts)
        elif inference_strategy == 'hor-2-scale':
            inference_2_scale(net, *head, scale, reduction, True, disable, **opts)
        elif inference_strategy == '3-scale':
            inference_3_scale(net, *head, scale, disable, **opts)
