#!/usr/bin/env python3
"""Headline benchmark: 480p frames/sec of the `main.py inference` hot loop (encoder -> label propagation -> mask).

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or plainly as `python bench.py --gpus N` - then this process starts
the N rank processes itself BEFORE it touches any GPU (a process that has initialised HIP must never exec) and passes rank 0's JSON
line through.  Videos shard across ranks with no data-path collective - every rank runs its own synthetic clip, "weak" scaling;
RCCL carries only the barrier and the max-reduce of the wall time.  A step = one frame of BASELINE.json
configs[1] (DAVIS-2017-shaped 480p clip, ResNet-50 encoder, dense affinity, ref_num 9): encoder forward on
PyTorch-ROCm + the hand-written HIP propagation (push, fused affinity/softmax/prior/label kernel, combine,
label pack, mask up-sample).  Frames are resident in HBM before the timed region; masks stay in HBM.
The clip is primed to frame_idx >= 20 first, so every timed step propagates from N = 9 reference frames with
both sigma branches live (reference src/model/predict.py:59-64).

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the definition of every field.
"""
import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"

WORKLOADS = {
    # name: (H, W, ref_num, topk, encoder)
    'davis480p_r50_dense': dict(H=480, W=854, ref_num=9, topk=0, model='resnet50'),
    'davis480p_r50_top20_ref5': dict(H=480, W=854, ref_num=5, topk=20, model='resnet50'),
    'ytvos720p_r50_dense': dict(H=720, W=1280, ref_num=9, topk=0, model='resnet50'),
    'pair240p_r18': dict(H=240, W=427, ref_num=9, topk=0, model='resnet18'),
    # BASELINE.json configs[4], second reading ("HBM-bandwidth stress"): the (N HW) x HW affinity written to HBM as bf16 and read
    # back, as the reference's own algorithm does (src/model/predict.py:49-55); its roofline is HBM, not MFMA
    'ytvos720p_r50_dense_materialised': dict(H=720, W=1280, ref_num=9, topk=0, model='resnet50', materialise=True),
}
HBM_PEAK_GBS = 8000.0                # /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW" (spec; ~6.3 TB/s achievable)


def synthetic_clip(H, W, n_frames, seed, device, tint_scale=1.0):
    """Low-frequency noise field drifting ~0.15 sigma per frame, ImageNet-normalised scale.
    Returns (n_frames,3,H,W) f32 on `device` and the first-frame annotation (H,W) u8 with 3 objects (d = 4)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    lo = torch.randn(3, 16, 28, generator=g)
    frames = []
    for _ in range(n_frames):
        lo = (1 - 0.15 ** 2) ** 0.5 * lo + 0.15 * torch.randn(3, 16, 28, generator=g)
        frames.append(torch.nn.functional.interpolate(lo[None], size=(H, W), mode='bilinear', align_corners=False)[0])
    clip = torch.stack(frames)
    yy, xx = np.mgrid[0:H, 0:W]
    ann = np.zeros((H, W), np.uint8)
    ann[(yy - 0.35 * H) ** 2 / (0.18 * H) ** 2 + (xx - 0.3 * W) ** 2 / (0.12 * W) ** 2 <= 1] = 1
    ann[int(0.55 * H):int(0.85 * H), int(0.5 * W):int(0.7 * W)] = 2
    ann[(yy - 0.3 * H) ** 2 / (0.12 * H) ** 2 + (xx - 0.78 * W) ** 2 / (0.1 * W) ** 2 <= 1] = 3
    # the three objects have a colour of their own (a constant offset inside their region, every frame): on the bare noise field
    # random-init features say nothing about the objects and every label history collapses to background within a few frames -
    # masks that are all zero check nothing (mask_parity, end_to_end.mask_class_histogram)
    tint = torch.tensor([[0.0, 0.0, 0.0], [2.0, -2.0, 0.5], [-2.0, 2.0, 2.0], [0.5, -2.0, 2.0]])
    clip = clip + tint_scale * tint[torch.from_numpy(ann).long()].permute(2, 0, 1)[None]
    return clip.to(device), ann


def kernel_source_hash():
    """sha1 (12 hex digits) over the propagation-kernel sources: ties a committed PMC profile to the code it measured."""
    import hashlib
    h = hashlib.sha1()
    src = ROOT / 'semi-supervised-vos_amd' / 'csrc'
    for name in sorted(p.name for p in src.glob('prop_*')) + ['common.h', 'aux_kernels.h', 'engine.hip']:      # (prop_*: .h and the generated .inc)
        h.update(name.encode())
        h.update((src / name).read_bytes())
    return h.hexdigest()[:12]


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(wl, cfg, model_state, feats_hist, labels_hist_cls, ann, frames_cpu, n_time=5, n_time_1t=2):
    """the oracle (torch-CPU restatement of the reference, oracle/vos_oracle.py) + the same encoder in
    fp32 on the host cores, on a bounded sample of the bench clip: `n_time` warm frames at frame_idx >= 20 (N = ref_num, both
    sigma branches live) with all cores of this job's share, MEDIAN per-frame time, encoder and propagation timed separately;
    then `n_time_1t` frames on ONE thread.  CPU model and core count are part of the record."""
    from oracle import vos_oracle as vo
    vos_net = importlib.import_module('semi-supervised-vos_amd.vos_net')
    # the GPU box gives one GPU's job a 16-core share (os.cpu_count() reports the whole host)
    threads = min(16, len(os.sched_getaffinity(0)))
    net = vos_net.VOSNet(wl['model'])
    net.load_state_dict(model_state)
    net.eval()
    st = vo.VideoState(ann, cfg['sigma1'], cfg['sigma2'], False)
    T0 = feats_hist.shape[0]
    st.feats_history = feats_hist.float()
    oh = torch.zeros(st.d, T0, st.H_d * st.W_d)
    oh.scatter_(0, labels_hist_cls.long().unsqueeze(0), 1.0)
    st.label_history = oh
    st.frame_idx = T0

    def run(n_frames, first):
        enc, prop = [], []
        with torch.no_grad():
            for i in range(first, first + n_frames):
                t0 = time.perf_counter()
                f = net(frames_cpu[i % frames_cpu.shape[0]][None])
                t1 = time.perf_counter()
                vo.rollout_step(st, f, cfg['frame_range'], wl['ref_num'], cfg['temperature'])
                t2 = time.perf_counter()
                enc.append(t1 - t0)
                prop.append(t2 - t1)
        return np.asarray(enc), np.asarray(prop)

    torch.set_num_threads(threads)
    run(1, 0)                                     # untimed: allocator, thread pool, oneDNN primitives
    enc, prop = run(n_time, 1)
    dt = float(np.median(enc + prop))
    torch.set_num_threads(1)
    enc1, prop1 = run(n_time_1t, 1 + n_time)
    dt1 = float(np.median(enc1 + prop1))
    torch.set_num_threads(threads)
    return {'value': 1.0 / dt, 'unit': 'frames/s', 'cores': threads, 'kind': 'port', 'cpu_model': cpu_model_name(),
            'host_cores_visible': len(os.sched_getaffinity(0)),
            'propagation_only_frames_per_s': 1.0 / float(np.median(prop)), 'encoder_ms': float(np.median(enc)) * 1e3,
            'one_thread': {'value': 1.0 / dt1, 'unit': 'frames/s', 'cores': 1, 'frames': n_time_1t,
                           'propagation_only_frames_per_s': 1.0 / float(np.median(prop1))},
            'sample': f'median of {n_time} warm frames (after 1 untimed) of the fp32 {wl["model"]} encoder + oracle predict + argmax + '
                      f'up-sample at frame_idx {T0 + 1}..{T0 + n_time}, N={wl["ref_num"]}, torch {threads} threads; then '
                      f'{n_time_1t} frames on 1 thread'}


def mask_parity(wl, cfg, ann, feats, gpu_masks, n_frames=7):
    """'mask IoU delta vs CPU ref' (BASELINE.json metric): the first frames of the bench clip propagated by the oracle on the
    host from the SAME encoder features the GPU used, compared with the masks the engine produced for those frames."""
    from oracle import vos_oracle as vo
    n = min(n_frames, feats.shape[0])
    topk = wl['topk']
    _, want = vo.rollout(ann, feats[:n].numpy(), cfg['frame_range'], wl['ref_num'], cfg['temperature'], cfg['sigma1'],
                         cfg['sigma2'], False, topk=topk)
    got = np.stack(gpu_masks[1:n])
    d = int(ann.max()) + 1
    iou = vo.mask_iou_per_object(want, got, d)
    return {'frames': n - 1, 'pixels_differing': float(np.mean(got != want)), 'per_object_iou': [round(v, 5) for v in iou],
            'class_histogram_last_frame': [int(v) for v in np.bincount(got[-1].reshape(-1), minlength=d)],
            'iou_delta': round(1.0 - min(iou), 5),
            'what': 'engine masks (mask-only steps, the timed kernel form) vs oracle (torch-CPU restatement of the reference) on the same encoder features'}


def end_to_end_leg(args, wl, dev, clip, pool, net, eng, enc_dtype, fence, world, on_gloo, ann, n_prime):
    """K frames from host memory to host memory: uint8 HWC frames (what a JPEG decoder hands over) sit in PINNED host memory, go
    over PCIe batch by batch on a copy stream one batch ahead of the compute stream, are normalised on the device with the
    reference's ToTensor + Normalize (datasets.normalize_on_device: bit-identical table look-up), encoded, propagated, and every
    mask is copied back into pinned host memory.  The clock stops when the last mask is on the host."""
    ds = importlib.import_module('semi-supervised-vos_amd.datasets')
    H, W, B, K = wl['H'], wl['W'], max(1, args.encoder_batch), args.steps
    mean = torch.tensor(ds.IMAGENET_MEAN, device=dev).view(1, 3, 1, 1)
    std = torch.tensor(ds.IMAGENET_STD, device=dev).view(1, 3, 1, 1)
    u8 = ((clip.float() * std + mean) * 255.0).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    frames_host = torch.empty(u8.shape, dtype=torch.uint8).pin_memory()
    frames_host.copy_(u8)
    del u8
    masks_host = torch.empty((K, H, W), dtype=torch.uint8).pin_memory()
    masks_host.fill_(255)          # no class is 255: a mask that never landed shows
    bufs = [torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    copied = [torch.cuda.Event(), torch.cuda.Event()]
    consumed = [torch.cuda.Event(), torch.cuda.Event()]
    copy_stream = torch.cuda.Stream(dev)
    main = torch.cuda.current_stream(dev)

    mask_dev = [torch.empty((B, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
    masks_done = [torch.cuda.Event(), torch.cuda.Event()]      # main: the batch's masks are written
    masks_home = [torch.cuda.Event(), torch.cuda.Event()]      # copy stream: they are on the host, the buffer is free again

    def upload(k):            # batch k -> bufs[k % 2] on the copy stream, after the batch that used that buffer was normalised
        n = min(B, K - k * B)
        with torch.cuda.stream(copy_stream):
            if k >= 2:
                copy_stream.wait_event(consumed[k % 2])
            i = 0
            while i < n:      # the pool is a ring: at most two contiguous runs per batch, one PCIe copy each
                src = (n_prime + k * B + i) % pool
                run = min(n - i, pool - src)
                bufs[k % 2][i:i + run].copy_(frames_host[src:src + run], non_blocking=True)
                i += run
            copied[k % 2].record(copy_stream)
        return n

    def download(k, n):       # the masks of batch k -> pinned host memory, one copy, on the copy stream
        masks_done[k % 2].record(main)
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(masks_done[k % 2])
            masks_host[k * B:k * B + n].copy_(mask_dev[k % 2][:n], non_blocking=True)
            masks_home[k % 2].record(copy_stream)

    n_batches = (K + B - 1) // B

    def fresh_video():
        # a fresh video, primed (untimed) on the clip's first n_prime frames to frame_idx >= 17: the leg then continues the clip,
        # its masks hold the three objects of the annotation instead of whatever hundreds of timed steps left of them
        eng.begin_video(ann)
        with torch.no_grad():
            pf = net(clip[:n_prime].contiguous(memory_format=torch.channels_last))
        for i in range(n_prime):
            eng.step(pf[i][None], want_pred=False, want_mask=False)

    def run(nb):
        upload(0)
        for k in range(nb):
            n = min(B, K - k * B)
            if k + 1 < nb:
                upload(k + 1)
            main.wait_event(copied[k % 2])
            with torch.no_grad():
                x = ds.normalize_on_device(bufs[k % 2][:n]).to(enc_dtype).contiguous(memory_format=torch.channels_last)
                consumed[k % 2].record(main)
                feats = net(x)
            if k >= 2:
                main.wait_event(masks_home[k % 2])
            for i in range(n):
                eng.step(feats[i][None], want_pred=False, mask_out=mask_dev[k % 2][i])
            download(k, n)

    # one untimed batch through the whole pipeline first (the copy stream, the pinned buffers' first DMA, the normalisation's
    # kernels and look-up table: first-use costs of ~10 ms that a 20-step run would otherwise report as throughput)
    fresh_video()
    run(1)
    fence()
    masks_host.fill_(255)
    fresh_video()
    fence()
    t0 = time.perf_counter()
    run(n_batches)
    fence()
    dt = time.perf_counter() - t0
    # every mask is on the host (no 255 left), the last batch equals what the device holds, and the labels did not collapse
    last_k = n_batches - 1
    n_last = min(B, K - last_k * B)
    landed = not bool((masks_host == 255).any())
    same = bool(torch.equal(masks_host[last_k * B:last_k * B + n_last], mask_dev[last_k % 2][:n_last].cpu()))
    if not (landed and same):
        raise SystemExit(f'end_to_end: masks missing on the host (all landed: {landed}, last batch equal to the device copy: {same})')
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device='cpu' if on_gloo else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    hist = torch.bincount(masks_host.reshape(-1).to(torch.int64), minlength=4)
    return {'value': world * K / dt, 'unit': 'frames/s', 'steps': K, 'ms_per_step': dt / K * 1e3,
            'bytes_over_pcie_per_frame': H * W * 3 + H * W,
            'what': 'uint8 HWC frames in pinned host memory -> H2D (copy stream, one batch ahead) -> ToTensor + Normalize on the '
                    'device -> encoder -> propagation -> mask -> D2H into pinned host memory (one copy per batch and direction, copy stream); clock stops with the last mask on the host',
            'masks_checked': 'all K masks landed (buffer pre-filled with 255), last batch equals the device copy',
            'mask_checksum': int(masks_host.to(torch.int64).sum()), 'mask_nonzero_pixels': int((masks_host != 0).sum()),
            'mask_class_histogram': [int(v) for v in hist]}


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv, script=None, extra_env=None):
    """`python bench.py --gpus N` without a launcher around it: start the N rank processes (the same command line, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, rendezvous on 127.0.0.1) from a parent that has NOT touched
    the GPU - fresh children, never an exec of a process that has initialised HIP.  Rank r uses the r-th visible device
    (LOCAL_RANK = r; an inherited HIP_VISIBLE_DEVICES is kept as it is, so the ordinals index into it).  Rank 0's stdout (the
    single JSON line) is passed through; the exit code is the first non-zero one of the ranks."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script or str(Path(__file__).resolve())] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for pr in procs[1:]:
        pr.wait()
        rc = rc or pr.returncode
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


def stub_main(args, rank, world):
    """`--workload stub_cpu`: the rank protocol of the bench (rendezvous, fence, K timed steps, max-over-ranks, one JSON line from
    rank 0) around a trivial CPU step under gloo - what tests/test_bench_launcher.py drives on a box without GPUs."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('gloo')
    x = torch.ones(64, 64)
    for _ in range(args.warmup):
        x = (x @ x) / 64.0
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = (x @ x) / 64.0
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        print(json.dumps({'metric': 'stub steps/sec', 'value': world * args.steps / dt, 'unit': 'steps/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                          'config': {'workload': 'stub_cpu'}, 'checksum': float(x.sum())}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=384)
    ap.add_argument('--warmup', type=int, default=64)
    ap.add_argument('--workload', default='davis480p_r50_dense', choices=sorted(WORKLOADS) + ['stub_cpu'])
    ap.add_argument('--encoder-dtype', default='f16', choices=['bf16', 'f16', 'f32'],
                    help='encoder precision; f16 = the reference (torch.cuda.amp.autocast, src/utils/inference_utils.py:35,52)')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help='barrier / max-reduce transport for N > 1 (nccl = RCCL; gloo only to rehearse several ranks on one GPU)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-end-to-end', action='store_true', help='skip the host-to-host leg (uint8 frames in pinned memory -> masks in pinned memory)')
    ap.add_argument('--no-encoder-graph', action='store_true', help='run the encoder as eager kernel launches')
    ap.add_argument('--no-miopen-find', action='store_true',
                    help='take MIOpen\'s immediate-mode convolution algorithms instead of letting it time its solvers in the warm-up')
    ap.add_argument('--prime', type=int, default=20, help='untimed frames that fill the reference history')
    ap.add_argument('--encoder-batch', type=int, default=96,
                    help='frames encoded per encoder call (features do not depend on the propagated labels)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher around us: become one.  Nothing in this process has touched a GPU yet (importing torch does not).
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    # the dispatch-attached event pairs that time the propagation kernel inside the loop cost ~1.7 us per launch they ride on:
    # in a long run every 4th launch of the timed region carries them (96 of the default 384), in a short one (<= 64 steps, the
    # driver's 20) EVERY launch does; `roofline.kernel_launches_timed` says how many
    os.environ.setdefault('VOSPROP_TIMING_STRIDE', '1' if args.steps <= 64 else '4')
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: start me as `python bench.py --gpus N` (I launch the ranks) '
                         f'or under torch.distributed.run with --nproc-per-node N')
    if args.workload == 'stub_cpu':
        return stub_main(args, rank, world)
    # VOSPROP_BENCH_DEVICES="0,0": device ordinal per local rank (rehearsing the N > 1 protocol on a box with fewer GPUs; RCCL
    # refuses two ranks on one device, so that rehearsal runs with --dist-backend gloo)
    devmap = [int(d) for d in os.environ.get('VOSPROP_BENCH_DEVICES', '').split(',') if d.strip() != '']
    local_dev = devmap[local % len(devmap)] if devmap else local
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    on_gloo = args.dist_backend == 'gloo'
    if world > 1:
        import torch.distributed as dist
        if on_gloo:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
    local = local_dev

    vos = importlib.import_module('semi-supervised-vos_amd')
    vos_net = importlib.import_module('semi-supervised-vos_amd.vos_net')
    wl = WORKLOADS[args.workload]
    cfg = dict(frame_range=40, sigma1=8.0, sigma2=21.0, temperature=1.0)
    H, W = wl['H'], wl['W']
    Hd, Wd = vos.feature_map_size(H, W)
    enc_dtype = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[args.encoder_dtype]

    torch.manual_seed(0)
    net = vos_net.VOSNet(wl['model'])
    if hasattr(net, 'bn256'):
        # random-init weights, with the embedding head's BatchNorm gain at 0.01: torch's default gain of 1 leaves embeddings of norm
        # ~270 on this clip, i.e. logits of ~7e4 at temperature 1 - a one-hot soft-max in which every label history collapses to one
        # class within a frame or two (tools/clip_probe.py) and the masks check nothing.  Norm ~3 keeps the soft-max soft.
        with torch.no_grad():
            net.bn256.weight.mul_(0.01)
    model_state = {k: v.clone() for k, v in net.state_dict().items()}
    # the features leave the encoder as channels-last f16 / bf16 and are handed over as they are: the propagation kernel reads the
    # target frame in place (f16 -> bf16 as it is loaded) and combine_kernel carries the ring copy - what the CLI does
    net.prepare_for_inference(dev, enc_dtype, miopen_find=not args.no_miopen_find)
    if not args.no_encoder_graph:
        net = vos_net.GraphedEncoder(net, max_graphs=8)     # the look-ahead batch forward as one HIP graph launch per shape

    # one distinct frame per step of the whole run (primed + warm-up + timed), resident in HBM, capped at 1 024 frames (cycled beyond
    # that): a look-ahead batch is then a VIEW of consecutive frames - round 2 / early round 3 cycled 32 frames and GATHERED every
    # batch (index_select + a channels-last copy, 3.8 us per frame of the timed region that no real pipeline has: there the batch
    # arrives in its buffer by DMA)
    pool = min(1024, max(args.prime, 17) + args.warmup + args.steps + 1)
    clip, ann = synthetic_clip(H, W, pool, seed=rank, device=dev)
    clip = clip.to(enc_dtype).contiguous(memory_format=torch.channels_last)
    eng = vos.PropagationEngine(Hd, Wd, device=local, ref_num=wl['ref_num'], frame_range=cfg['frame_range'],
                                sigma1=cfg['sigma1'], sigma2=cfg['sigma2'], temperature=cfg['temperature'],
                                topk=wl['topk'], materialise=wl.get('materialise', False))
    eng.begin_video(ann)

    keep_feats, keep_cls, keep_masks = [], [], []

    B = max(1, args.encoder_batch)
    feat_buf = {'f': None, 'pos': 0}

    def frames_of(i, n):
        """frames i .. i+n-1 of the clip (a ring of `pool` frames): a view when they are consecutive in memory, a gather at the wrap"""
        a = i % pool
        if a + n <= pool:
            return clip[a:a + n]
        idx = torch.arange(i, i + n, device=dev) % pool
        return clip.index_select(0, idx).contiguous(memory_format=torch.channels_last)

    def encode_next(i, end):
        """Encoder look-ahead: the features of frames i .. min(i+B, end)-1 in one call (never a frame past `end`: a phase only
        encodes what it uses); the propagation stays strictly sequential."""
        if feat_buf['f'] is None or feat_buf['pos'] == feat_buf['f'].shape[0]:
            with torch.no_grad():
                feat_buf['f'] = net(frames_of(i, min(i + B, end) - i))
            feat_buf['pos'] = 0
        f = feat_buf['f'][feat_buf['pos']]
        feat_buf['pos'] += 1
        return f

    # first output row / column whose ATen nearest source index is i / j: the mask read there IS the low-resolution class map
    up_r = [next(y for y in range(H) if min(int(np.floor(np.float32(y) * (np.float32(Hd) / np.float32(H)))), Hd - 1) == i) for i in range(Hd)]
    up_c = [next(x for x in range(W) if min(int(np.floor(np.float32(x) * (np.float32(Wd) / np.float32(W)))), Wd - 1) == j) for j in range(Wd)]

    def one_frame(i, keep=False, end=1 << 60):
        # every step - kept or timed - runs the SAME kernel form: mask only (pred_out_dev == NULL: no softmax denominators), the
        # target frame read in place; `mask_parity` therefore checks the form that is timed
        feats = encode_next(i, end)[None]
        _, mask = eng.step(feats, want_pred=False, want_mask=True)
        if keep:
            keep_feats.append(feats.float().cpu())
            m = None if mask is None else mask.cpu().numpy()
            keep_cls.append(None if m is None else torch.from_numpy(np.ascontiguousarray(m[np.ix_(up_r, up_c)])).reshape(-1).long())
            keep_masks.append(m)
        return mask

    fi = 0
    n_prime = max(args.prime, 17)
    for _ in range(n_prime):
        one_frame(fi, keep=(rank == 0 and not args.no_cpu_baseline and world == 1), end=n_prime)
        fi += 1
    feat_buf['f'] = None
    for _ in range(args.warmup):
        one_frame(fi, end=n_prime + args.warmup)
        fi += 1
    # the timed region encodes exactly the K frames it propagates: full batches of B and one tail batch of K % B frames.  Its
    # batch shapes are run once here, untimed, so that MIOpen's one-off algorithm search (and the HIP-graph capture) for a
    # new batch size is not inside the timing
    for n_warm in sorted({min(B, args.steps), args.steps % B} - {0}, reverse=True):
        with torch.no_grad():
            net(frames_of(0, n_warm))
    feat_buf['f'] = None   # the timed region starts with an empty look-ahead buffer: it pays for every frame it uses
    t_end = fi + args.steps

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    eng.timing_begin()      # HIP events around every propagation-kernel launch of the timed region, on its stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mask = one_frame(fi, end=t_end)
        fi += 1
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device='cpu' if on_gloo else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- dominant hand-written kernel alone: HIP events on the launch stream, N=9 propagation of the last step ----
    st = eng.last_stats()
    prop_us, timed_launches = eng.timing_read()          # the kernel as it ran inside the timed loop
    b2b_us = eng.time_last_propagation(iters=50)         # and re-run back to back (warm caches, steady clocks)
    if not timed_launches:                                # top-k runs two passes + a selection: only the back-to-back timer covers it
        prop_us = b2b_us
    achieved = st['flops'] / (prop_us * 1e-6) / 1e12
    hbm_bound = bool(wl.get('materialise'))          # the materialised-affinity variant is priced against the HBM roof
    achieved_gbs = st['bytes'] / (prop_us * 1e-6) / 1e9
    # propagation-only frames/s (push + propagate + combine + label pack + mask), encoder excluded
    with torch.no_grad():
        feats = net(clip[0:1]).detach()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(50):
        eng.step(feats, want_pred=False, want_mask=True)
    torch.cuda.synchronize()
    prop_fps = 50 / (time.perf_counter() - t1)

    # the encoder alone: the look-ahead batch replayed 5 times back to back (same graph, same input shape as the timed region)
    xb = frames_of(0, min(B, args.steps))
    with torch.no_grad():
        net(xb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            net(xb)
        e1.record()
    torch.cuda.synchronize()
    encoder_us = e0.elapsed_time(e1) / 5 / xb.shape[0] * 1e3
    del xb

    # ---- end to end, host to host: uint8 frames in pinned host memory -> H2D -> normalise -> encoder ->
    # propagate -> mask -> D2H into pinned host memory.  Reported beside `value`, never as `value`.
    end_to_end = None
    if not args.no_end_to_end and wl['topk'] == 0 and not wl.get('materialise'):
        end_to_end = end_to_end_leg(args, wl, dev, clip, pool, net, eng, enc_dtype, fence, world, on_gloo, ann, n_prime)

    # HBM-side traffic of the same kernel: PMC counters cannot be read from inside this process (rocprofv3 collects them, in
    # their own passes), so the line carries the committed measurement for this workload (tools/traffic_pmc.sh) or null
    # A profile is only quoted while it describes the kernels that ran: it carries the hash of the kernel sources it was taken
    # with (kernel_source_hash below); after any edit of csrc/prop_*.h / common.h the line says traffic: null until the profile is
    # re-taken.
    traffic, traffic_src, pmc = None, None, {}
    prof = Path(__file__).resolve().parent / 'profiles'
    src_hash = kernel_source_hash()
    tj = prof / f'r04_prop_kernel_traffic_{args.workload}.json'
    if tj.exists():
        try:
            t = json.loads(tj.read_text())
            if t.get('kernel_source_hash') == src_hash:
                traffic = float(t['traffic_bytes_per_launch'])
                traffic_src = f'profiles/{tj.name}: ' + t['how']
                pmc = {k: t[k] for k in ('mfma_busy_frac', 'l2_hit_rate', 'hbm_gb_per_s', 'counters_source') if k in t}
            else:
                traffic_src = (f'profiles/{tj.name} was taken with kernel sources {t.get("kernel_source_hash")}, this build is '
                               f'{src_hash}: not quoted')
        except Exception:
            traffic = None
    if rank == 0:
        out = {
            'metric': '480p frames/sec at 1/2/4/8 MI355X; mask IoU delta vs CPU ref',
            'value': world * args.steps / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': args.workload, 'image': [H, W], 'feature_map': [Hd, Wd], 'ref_num': wl['ref_num'],
                       'frame_range': cfg['frame_range'], 'topk': wl['topk'], 'encoder': wl['model'],
                       'encoder_dtype': args.encoder_dtype,
                       'feature_handoff': (f'channels-last {args.encoder_dtype} as the encoder left it: target frame read in place (f16 '
                                           'converted to bf16 in the kernel prologue), ring copy inside combine_kernel'
                                           if enc_dtype != torch.float32 and not wl['topk'] and not wl.get('materialise')
                                           else 'push kernel into the ring per frame'),
                       'step_form': 'mask only (pred_out_dev = NULL)', 'encoder_batch': B, 'encoder_weights': 'random-init, BatchNorm folded', 'objects': 3,
                       'videos_per_gpu': 1, 'sharding': 'whole videos per GPU, no collective'},
            'propagation_only_frames_per_s_per_gpu': prop_fps, 'encoder_us_per_frame': encoder_us, 'end_to_end': end_to_end,
            # the kernel the ENGINE reports it launched (vosprop_stats.kernel_id, set where the launch is decided) - never re-derived here
            'roofline': {'kernel': st['kernel'], 'kernel_id': st['kernel_id'],
                         'bound': 'hbm' if hbm_bound else 'mfma', 'achieved': achieved_gbs if hbm_bound else achieved,
                         'peak': HBM_PEAK_GBS if hbm_bound else MFMA_BF16_PEAK_TFLOPS, 'unit': 'GB/s' if hbm_bound else 'TFLOP/s',
                         'frac': achieved_gbs / HBM_PEAK_GBS if hbm_bound else achieved / MFMA_BF16_PEAK_TFLOPS,
                         'traffic': traffic, 'traffic_source': traffic_src, 'kernel_us': prop_us,
                         'kernel_launches_timed': timed_launches, 'kernel_us_back_to_back': b2b_us, 'flops_per_launch': st['flops'],
                         'algorithmic_bytes_per_launch': st['bytes'], 'workgroups': st['workgroups'], 'pmc': pmc,
                         'kernel_source_hash': src_hash},
        }
        if world == 1 and not args.no_cpu_baseline:
            T0 = len(keep_feats)
            fh = torch.cat(keep_feats, 0)
            ann_cls = torch.from_numpy(np.ascontiguousarray(ann)).long()
            from oracle import vos_oracle as vo
            src_r, src_c = vo.nearest_src_index(Hd, H), vo.nearest_src_index(Wd, W)
            cls0 = ann_cls[src_r][:, src_c].reshape(-1)
            cls_hist = torch.stack([cls0] + [c for c in keep_cls[1:]], 0)
            frames_cpu = clip[0:8].float().cpu().contiguous()
            out['cpu_baseline'] = cpu_baseline(wl, cfg, model_state, fh, cls_hist, ann, frames_cpu)
            out['mask_parity'] = mask_parity(wl, cfg, ann, fh, keep_masks)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
