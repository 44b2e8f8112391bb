#!/usr/bin/env python3
"""Headline benchmark: 480p frames/sec of the `main.py inference` hot loop (encoder -> label propagation -> mask).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (N > 1: launched by torch.distributed.run; videos shard across ranks with no data-path
collective - every rank runs its own synthetic clip, "weak" scaling).  A step = one frame of BASELINE.json
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
}


def synthetic_clip(H, W, n_frames, seed, device):
    """Low-frequency noise field drifting ~0.15 sigma per frame, ImageNet-normalised scale.
    Returns (n_frames,3,H,W) f32 on `device` and the first-frame annotation (H,W) u8 with 3 objects (d = 4)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    lo = torch.randn(3, 16, 28, generator=g)
    frames = []
    for _ in range(n_frames):
        lo = (1 - 0.15 ** 2) ** 0.5 * lo + 0.15 * torch.randn(3, 16, 28, generator=g)
        frames.append(torch.nn.functional.interpolate(lo[None], size=(H, W), mode='bilinear', align_corners=False)[0])
    clip = torch.stack(frames).to(device)
    yy, xx = np.mgrid[0:H, 0:W]
    ann = np.zeros((H, W), np.uint8)
    ann[(yy - 0.35 * H) ** 2 / (0.18 * H) ** 2 + (xx - 0.3 * W) ** 2 / (0.12 * W) ** 2 <= 1] = 1
    ann[int(0.55 * H):int(0.85 * H), int(0.5 * W):int(0.7 * W)] = 2
    ann[(yy - 0.3 * H) ** 2 / (0.12 * H) ** 2 + (xx - 0.78 * W) ** 2 / (0.1 * W) ** 2 <= 1] = 3
    return clip, ann


def cpu_baseline(wl, cfg, model_state, feats_hist, labels_hist_cls, ann, frames_cpu, n_time=4):
    """The oracle (torch-CPU restatement of the reference, oracle/vos_oracle.py) + the same encoder on the host
    cores, on a bounded sample: `n_time` frames at frame_idx >= 20 (N = 9)."""
    from oracle import vos_oracle as vo
    vos_net = importlib.import_module('semi-supervised-vos_amd.vos_net')
    # the GPU box gives one GPU's job a 16-core share (os.cpu_count() reports the whole host)
    threads = min(16, len(os.sched_getaffinity(0)))
    torch.set_num_threads(threads)
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
    times = []
    with torch.no_grad():
        for i in range(n_time + 1):
            t0 = time.perf_counter()
            f = net(frames_cpu[i:i + 1])
            vo.rollout_step(st, f, cfg['frame_range'], wl['ref_num'], cfg['temperature'])
            times.append(time.perf_counter() - t0)
    dt = float(np.mean(times[1:]))
    return {'value': 1.0 / dt, 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'sample': f'{n_time} frames (after 1 untimed) of {wl["model"]} encoder + oracle predict at frame_idx '
                      f'{T0 + 1}..{T0 + n_time}, N={wl["ref_num"]}, fp32, torch {threads} threads'}


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
            'iou_delta': round(1.0 - min(iou), 5),
            'what': 'engine masks vs oracle (torch-CPU restatement of the reference) on identical bf16 encoder features'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=384)
    ap.add_argument('--warmup', type=int, default=64)
    ap.add_argument('--workload', default='davis480p_r50_dense', choices=sorted(WORKLOADS))
    ap.add_argument('--encoder-dtype', default='bf16', choices=['bf16', 'f16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-encoder-graph', action='store_true', help='run the encoder as eager kernel launches')
    ap.add_argument('--no-miopen-find', action='store_true',
                    help='take MIOpen\'s immediate-mode convolution algorithms instead of letting it time its solvers in the warm-up')
    ap.add_argument('--prime', type=int, default=20, help='untimed frames that fill the reference history')
    ap.add_argument('--encoder-batch', type=int, default=64,
                    help='frames encoded per encoder call (features do not depend on the propagated labels)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)

    vos = importlib.import_module('semi-supervised-vos_amd')
    vos_net = importlib.import_module('semi-supervised-vos_amd.vos_net')
    wl = WORKLOADS[args.workload]
    cfg = dict(frame_range=40, sigma1=8.0, sigma2=21.0, temperature=1.0)
    H, W = wl['H'], wl['W']
    Hd, Wd = vos.feature_map_size(H, W)
    enc_dtype = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[args.encoder_dtype]

    torch.manual_seed(0)
    net = vos_net.VOSNet(wl['model'])
    model_state = {k: v.clone() for k, v in net.state_dict().items()}
    net.prepare_for_inference(dev, enc_dtype, miopen_find=not args.no_miopen_find)
    if not args.no_encoder_graph:
        net = vos_net.GraphedEncoder(net, max_graphs=8)     # the look-ahead batch forward as one HIP graph launch per shape

    pool = 32                                    # distinct frames, cycled
    clip, ann = synthetic_clip(H, W, pool, seed=rank, device=dev)
    clip = clip.to(enc_dtype).contiguous(memory_format=torch.channels_last)
    eng = vos.PropagationEngine(Hd, Wd, device=local, ref_num=wl['ref_num'], frame_range=cfg['frame_range'],
                                sigma1=cfg['sigma1'], sigma2=cfg['sigma2'], temperature=cfg['temperature'],
                                topk=wl['topk'])
    eng.begin_video(ann)

    keep_feats, keep_cls, keep_masks = [], [], []

    B = max(1, args.encoder_batch)
    feat_buf = {'f': None, 'pos': 0}

    def encode_next(i, end):
        """Encoder look-ahead: the features of frames i .. min(i+B, end)-1 in one call (never a frame past `end`: a phase only
        encodes what it uses); the propagation stays strictly sequential."""
        if feat_buf['f'] is None or feat_buf['pos'] == feat_buf['f'].shape[0]:
            idx = torch.arange(i, min(i + B, end), device=dev) % pool
            with torch.no_grad():
                feat_buf['f'] = net(clip.index_select(0, idx).contiguous(memory_format=torch.channels_last))
            feat_buf['pos'] = 0
        f = feat_buf['f'][feat_buf['pos']]
        feat_buf['pos'] += 1
        return f

    def one_frame(i, keep=False, end=1 << 60):
        feats = encode_next(i, end)[None]
        pred, mask = eng.step(feats, want_pred=keep, want_mask=True)
        if keep:
            keep_feats.append(feats.float().cpu())
            keep_cls.append(None if pred is None else pred.argmax(0).cpu())
            keep_masks.append(None if mask is None else mask.cpu().numpy())
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
        idx = torch.arange(n_warm, device=dev) % pool
        with torch.no_grad():
            net(clip.index_select(0, idx).contiguous(memory_format=torch.channels_last))
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
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- dominant hand-written kernel alone: HIP events on the launch stream, N=9 propagation of the last step ----
    st = eng.last_stats()
    prop_us, timed_launches = eng.timing_read()          # the kernel as it ran inside the timed loop
    b2b_us = eng.time_last_propagation(iters=50)         # and re-run back to back (warm caches, steady clocks)
    if not timed_launches:                                # top-k runs two passes + a selection: only the back-to-back timer covers it
        prop_us = b2b_us
    achieved = st['flops'] / (prop_us * 1e-6) / 1e12
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
    idx = torch.arange(min(B, args.steps), device=dev) % pool
    xb = clip.index_select(0, idx).contiguous(memory_format=torch.channels_last)
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

    # HBM-side traffic of the same kernel: PMC counters cannot be read from inside this process (rocprofv3 collects them, in
    # their own passes), so the line carries the committed measurement for this workload (tools/traffic_pmc.sh) or null
    traffic, traffic_src, pmc = None, None, {}
    tj = Path(__file__).resolve().parent / 'profiles' / 'r01_prop_kernel_traffic.json'
    if args.workload == 'davis480p_r50_dense' and tj.exists():
        try:
            t = json.loads(tj.read_text())
            traffic = float(t['traffic_bytes_per_launch'])
            traffic_src = 'profiles/r01_prop_kernel_traffic.json: ' + t['how']
            pmc = {k: t[k] for k in ('mfma_busy_frac', 'l2_hit_rate', 'hbm_gb_per_s', 'counters_source') if k in t}
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
                       'encoder_dtype': args.encoder_dtype, 'encoder_batch': B, 'encoder_weights': 'random-init, BatchNorm folded', 'objects': 3,
                       'videos_per_gpu': 1, 'sharding': 'whole videos per GPU, no collective'},
            'propagation_only_frames_per_s_per_gpu': prop_fps, 'encoder_us_per_frame': encoder_us,
            'roofline': {'kernel': 'prop_bf16_kernel', 'bound': 'mfma', 'achieved': achieved,
                         'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / MFMA_BF16_PEAK_TFLOPS,
                         'traffic': traffic, 'traffic_source': traffic_src, 'kernel_us': prop_us,
                         'kernel_launches_timed': timed_launches, 'kernel_us_back_to_back': b2b_us, 'flops_per_launch': st['flops'],
                         'algorithmic_bytes_per_launch': st['bytes'], 'workgroups': st['workgroups'], 'pmc': pmc},
        }
        if world == 1 and not args.no_cpu_baseline:
            T0 = len(keep_feats)
            fh = torch.cat(keep_feats, 0)
            ann_cls = torch.from_numpy(np.ascontiguousarray(ann)).long()
            from oracle import vos_oracle as vo
            src_r, src_c = vo.nearest_src_index(Hd, H), vo.nearest_src_index(Wd, W)
            cls0 = ann_cls[src_r][:, src_c].reshape(-1)
            cls_hist = torch.stack([cls0] + [c for c in keep_cls[1:]], 0)
            frames_cpu = clip[0:6].float().cpu().contiguous()
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
