"""The `inference_single` frame loop of the reference (src/utils/inference_utils.py:23-87) on top of the engine.

Same signature and observable behaviour (frame 0 seeds the history, masks of a video are written as 00001.png...
with the first annotation's palette when the video changes), but no module globals: the per-video state lives in a
PropagationEngine, the (N*HW)xHW affinity is never formed, histories are a ring in HBM instead of torch.cat.
"""
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

from . import engine as _engine
from .config import Config
from .datasets import normalize_on_device
from .io_pipeline import AsyncMaskWriter
from .utils import save_predictions


def _read_annotation(path):
    from PIL import Image
    ann = Image.open(path)
    return np.asarray(ann), ann.getpalette(), ann


class EncoderNotReproducible(RuntimeError):
    """--deterministic: an encoder batch did not repeat bit for bit (inference.py turns it into a click.ClickException)."""


def encoded_branches(models, loader, device, encoder_dtype, batch, resize=None, align_videos=False):
    """Encoder look-ahead for every strategy: yields ([features (1,C,H_d,W_d) per branch], video_name) in loader order, but
    runs each encoder on `batch` consecutive frames at a time - the features do not depend on the propagated labels (only the
    labels are sequential) nor on the video, and the encoder is ~3.5x cheaper per frame at batch 16 than at batch 1 on
    MI355X.  A batch runs across video boundaries and, on the GPU, a short one (end of the data, change of frame size) is padded
    to `batch` frames: every new batch size costs MIOpen look-ups, a GEMM plan and a graph capture - far more than encoding a
    few frames too many - so the encoder only ever sees ONE shape per frame size.  `models`: one encoder per branch.  A loader item carries one tensor (every branch sees it) or one
    tensor per branch (the flip / 2-scale datasets).  `resize(H, W) -> (h, w)`: nearest pre-scaling of the input, the
    3-scale strategy's (reference inference_utils.py:523-526).  align_videos (`--deterministic`): a batch also ends where the video
    changes, so the position of a frame inside its encoder batch depends on its index in its video only - whichever other videos
    the process was given (a library GEMM may split tiles differently along the batch; measured position-independent on this
    stack, but nothing promises it).  In that mode the FIRST batch of every distinct (branch, input shape) is also encoded twice
    more and compared bitwise (`EncoderNotReproducible` on a mismatch): the reproducibility of the library's kernels depends on the
    problem size, and scaled branches or a dataset with mixed frame sizes reach sizes the start-up check never saw."""
    nb = len(models)
    verified = set()
    pend = [[] for _ in range(nb)]
    names = []
    copy_stream = [torch.cuda.Stream(device)]
    pad = torch.device(device).type == 'cuda' and batch > 1

    def flush():
        feats = []
        for b in range(nb):
            # One small copy per frame into a device batch, on a separate copy stream: a host-side torch.cat of 16 frames costs
            # 18-38 ms (measured), and on the compute stream a copy from non-pinned memory is synchronous - it would make the
            # host wait for all queued GPU work before it can enqueue the next batch (GPU 57 % busy in the CLI, measured).
            first = pend[b][0]
            compute = torch.cuda.current_stream(device)
            with torch.cuda.stream(copy_stream[0]):
                n = len(pend[b])
                x = torch.empty((batch if pad else n,) + tuple(first.shape[1:]), dtype=first.dtype, device=device)
                for i, t in enumerate(pend[b]):
                    x[i:i + 1].copy_(t, non_blocking=True)
                if x.shape[0] > n:
                    x[n:].zero_()       # padding frames: encoded and dropped (samples are independent: BatchNorm is folded)
                copied = torch.cuda.Event()
                copied.record(copy_stream[0])
            compute.wait_event(copied)
            x.record_stream(compute)
            if x.dtype == torch.uint8:      # raw (B,H,W,3) frames from the decode workers: ToTensor + Normalize on the GPU
                x = normalize_on_device(x)
            if resize is not None:
                x = F.interpolate(x, size=resize(x.shape[-2], x.shape[-1]), mode='nearest')
            if encoder_dtype is not None:
                x = x.to(encoder_dtype)
            with torch.no_grad():
                xc = x.contiguous(memory_format=torch.channels_last)
                f = models[b](xc)
                key = (b, tuple(xc.shape), xc.dtype)
                if align_videos and torch.device(device).type == 'cuda' and key not in verified:
                    ref = f.clone()
                    if not (torch.equal(models[b](xc), ref) and torch.equal(models[b](xc), ref)):
                        raise EncoderNotReproducible(f'--deterministic: the encoder is NOT bit-reproducible for input {tuple(xc.shape)} '
                                                     f'({xc.dtype}) on this software stack (a library kernel accumulates with atomics); '
                                                     'see DESIGN.md section 7.1 / tools/determinism_probe.py')
                    verified.add(key)
                    f = ref
            if any(models[c] is models[b] for c in range(b + 1, nb)):
                f = f.clone()       # a graph-replaying encoder reuses its output buffer: the next branch would overwrite it
            feats.append(f)
        if hasattr(loader, 'recycle'):   # ShmFrameLoader: the slots may be reused once the copies have completed (one stream, in
            uniq = {id(t): t for b in range(nb) for t in pend[b]}     # order: the last event covers them all; 'multimodel'
            loader.recycle(list(uniq.values()), copied)               # shows the same tensor to both branches)
        for b in range(nb):
            pend[b].clear()
        out = [([f[i:i + 1] for f in feats], names[i]) for i in range(len(names))]
        names.clear()
        return out

    for input, (name,) in loader:
        ins = list(input) if isinstance(input, (list, tuple)) else [input] * nb
        if len(ins) != nb:
            raise ValueError(f'loader item carries {len(ins)} inputs, the strategy has {nb} branches')
        if names and (len(names) == batch or any(ins[b].shape != pend[b][-1].shape for b in range(nb))
                      or (align_videos and name != names[-1])):
            yield from flush()
        for b in range(nb):
            pend[b].append(ins[b])
        names.append(name)
    if names:
        yield from flush()


def encoded_frames(model, loader, device, encoder_dtype, batch, align_videos=False):
    """Single-branch form: yields (features (1,C,H_d,W_d), video_name)."""
    for feats, name in encoded_branches([model], loader, device, encoder_dtype, batch, align_videos=align_videos):
        yield feats[0], name


def inference_single(model, inference_loader, total_len, annotation_dir, last_video, save, sigma_1, sigma_2,
                     frame_range, ref_num, temperature, probability_propagation, disable, encoder_dtype=None,
                     stats=None, encoder_batch=32, png_workers=2, precision=0, align_videos=False):
    """precision: VOSPROP_PREC_* of the propagation (0 = bf16 MFMA, 1 = the f32 parity path).
    stats (optional dict) receives {'frames', 'videos', 'seconds'} for the fps report."""
    import time
    from tqdm import tqdm
    device = Config.DEVICE
    if device.type != 'cuda':
        raise _engine.VospropError("--device cpu: the propagation engine is HIP-only (no CPU fallback)")
    eng = None
    masks = []
    palette = None
    frame_idx = 0
    n_frames = 0
    videos = 0
    t0 = time.perf_counter()

    writer = AsyncMaskWriter(save, png_workers)

    def flush(video):
        if masks:
            writer.submit(video, palette, masks)     # D2H + PNG encoding proceed while the next video is processed
            masks.clear()

    stream = encoded_frames(model, inference_loader, device, encoder_dtype, max(1, encoder_batch), align_videos)
    for features, current_video in tqdm(stream, total=total_len, disable=disable):
        if current_video != last_video:
            flush(last_video)
            frame_idx = 0
        if frame_idx == 0:
            label, palette, ann_img = _read_annotation(Path(annotation_dir) / current_video / '00000.png')
            H_d, W_d = features.shape[-2:]
            if eng is None or (eng.feat_h, eng.feat_w) != (H_d, W_d):
                if eng is not None:
                    eng.close()
                eng = _engine.PropagationEngine(H_d, W_d, device=device.index or 0, ref_num=ref_num,
                                                frame_range=frame_range, sigma1=sigma_1, sigma2=sigma_2,
                                                temperature=temperature, probability=probability_propagation,
                                                precision=precision)
            eng.begin_video(label)
            if save is not None:   # reference predict.py:120-126: the annotation becomes 00000.png of the output
                out_dir = Path(save) / current_video
                out_dir.mkdir(parents=True, exist_ok=True)
                ann_img.save(out_dir / '00000.png')
            eng.step(features)
            videos += 1
        else:
            _, mask = eng.step(features, want_pred=False, want_mask=True)
            masks.append(mask)
        last_video = current_video
        frame_idx += 1
        n_frames += 1
    flush(last_video)
    torch.cuda.synchronize()
    writer.close()
    if eng is not None:
        eng.close()
    if stats is not None:
        stats.update(frames=n_frames, videos=videos, seconds=time.perf_counter() - t0)


# ---------------------------------------------------------------------------------------------------------------------
# Multi-branch strategies (reference src/utils/inference_utils.py:90-595).  Each one runs two (or, 3-scale, three
# sequential) INDEPENDENT propagation chains - own features, own label history, own map size, no feedback from the
# fused result - and fuses the up-sampled outputs per frame.  A chain is one PropagationEngine.
# The reference's behaviour is kept as it is, including four things that look unintended:
#   * label mode fuses class-INDEX maps with an element-wise maximum (:178, :400, :506, :594);
#   * 'vert-flip' un-flips its second branch with fliplr, like 'hor-flip' (:282);
#   * in probability mode that fliplr acts on a (1,d,H,W) tensor, i.e. it reverses the CLASS axis (:166, :282);
#   * 'hor-2-scale' mirrors the second image but propagates the un-mirrored first labels on it (:326 passes '2-scale').
# ---------------------------------------------------------------------------------------------------------------------
REDUCTIONS = {'maximum': lambda x, y: torch.maximum(x, y),       # reference inference_utils.py:18-20
              'minimum': lambda x, y: torch.minimum(x, y),
              'mean': lambda x, y: (x + y) / 2.0}

THREE_SCALE_OUTPUT = (480, 910)     # hard-coded in the reference (inference_utils.py:574)


def lowres_class_map(label, H_d, W_d):
    """Class map of get_labels(label, d, H, W, H_d, W_d) (reference predict.py:92-96): the nearest resize of a one-hot
    stack is the one-hot of the nearest-resized index map."""
    t = torch.from_numpy(np.ascontiguousarray(label).astype(np.float32))[None, None]
    return F.interpolate(t, size=(H_d, W_d), mode='nearest')[0, 0].to(torch.uint8).numpy()


def scaled_map_size(H, W, scale=None):
    """reference predict.py:109-110 (scale None) and :138-139, :148-149."""
    k = Config.SCALE if scale is None else Config.SCALE * scale
    return int(np.ceil(H * k)), int(np.ceil(W * k))


class _Chain:
    """One propagation chain of a strategy: the engine is (re)built when the feature-map size changes."""

    def __init__(self, device, **engine_kw):
        self.device, self.kw, self.eng = device, engine_kw, None

    def begin(self, features, label, d, map_hw, out_hw):
        H_d, W_d = (int(v) for v in features.shape[-2:])
        if (H_d, W_d) != tuple(map_hw):
            raise _engine.VospropError(f'feature map {H_d}x{W_d} does not match the {map_hw[0]}x{map_hw[1]} label map the '
                                       'reference derives from the annotation for this strategy (predict.py:109-110,138-139)')
        if self.eng is None or (self.eng.feat_h, self.eng.feat_w) != (H_d, W_d):
            self.close()
            self.eng = _engine.PropagationEngine(H_d, W_d, device=self.device.index or 0, **self.kw)
        self.eng.begin_video_labels(lowres_class_map(label, H_d, W_d), d, out_hw)
        self.eng.step(features)

    def step(self, features, probability):
        """-> (H,W) u8 class map in label mode, (1,d,H,W) f32 up-sampled prediction in probability mode."""
        e = self.eng
        if not probability:
            return e.step(features, want_pred=False, want_mask=True)[1]
        pred, _ = e.step(features, want_pred=True, want_mask=False)
        return F.interpolate(pred.view(1, e.d, e.feat_h, e.feat_w), size=(e.H, e.W), mode='nearest')

    def close(self):
        if self.eng is not None:
            self.eng.close()
            self.eng = None


# strategy -> (transform of the first label per branch, does branch 2 use the scaled map, un-flip of branch 2's output)
_FLIP_W = lambda a: np.ascontiguousarray(a[:, ::-1])
_FLIP_H = lambda a: np.ascontiguousarray(a[::-1, :])
_TWO_BRANCH = {
    'hor-flip': dict(label2=_FLIP_W, scaled=False, unflip='fliplr'),      # :90-187
    'vert-flip': dict(label2=_FLIP_H, scaled=False, unflip='fliplr'),     # :196-298 (fliplr, sic)
    '2-scale': dict(label2=None, scaled=True, unflip=None),               # :300-413
    'hor-2-scale': dict(label2=None, scaled=True, unflip='hflip'),        # same function, flip_pred=True (:389-390)
    'multimodel': dict(label2=None, scaled=False, unflip=None),           # :416-511
}


def fuse_two(a, b, probability, reduction_str, unflip):
    """Per-frame fusion, reference :163-178 / :389-400.  Label mode: a, b (H,W) u8 -> (H,W) u8.  Probability mode:
    a, b (1,d,H,W) f32 -> (H,W) u8 (argmax after the reference's cast to half)."""
    if unflip == 'fliplr':
        b = b.flip(1)               # torch.fliplr: axis 1 - W of an (H,W) map, the class axis of (1,d,H,W)
    elif unflip == 'hflip':
        b = b.flip(-1)
    if probability:
        return torch.argmax(REDUCTIONS[reduction_str](a, b).half(), 1)[0].to(torch.uint8)
    return torch.maximum(a, b)


def _inference_two_branch(strategy, models, inference_loader, total_len, annotation_dir, last_video, save, sigma_1, sigma_2,
                          frame_range, ref_num, temperature, probability_propagation, scale, reduction_str, disable,
                          encoder_dtype=None, stats=None, encoder_batch=32, png_workers=2, precision=0, align_videos=False):
    import time
    from tqdm import tqdm
    spec = _TWO_BRANCH[strategy]
    device = Config.DEVICE
    if device.type != 'cuda':
        raise _engine.VospropError("--device cpu: the propagation engine is HIP-only (no CPU fallback)")
    if probability_propagation and reduction_str not in REDUCTIONS:
        raise ValueError(f'unknown fusion {reduction_str!r}')
    kw = dict(ref_num=ref_num, frame_range=frame_range, sigma1=sigma_1, sigma2=sigma_2, temperature=temperature,
              probability=probability_propagation, precision=precision)
    chains = [_Chain(device, **kw), _Chain(device, **kw)]
    masks, palette, frame_idx, n_frames, videos = [], None, 0, 0, 0
    t0 = time.perf_counter()

    writer = AsyncMaskWriter(save, png_workers)

    def flush(video):
        if masks:
            writer.submit(video, palette, masks)
            masks.clear()

    stream = encoded_branches(models, inference_loader, device, encoder_dtype, max(1, encoder_batch), align_videos=align_videos)
    for feats, current_video in tqdm(stream, total=total_len, disable=disable):
        if current_video != last_video:
            flush(last_video)
            frame_idx = 0
        if frame_idx == 0:
            label, palette, ann_img = _read_annotation(Path(annotation_dir) / current_video / '00000.png')
            H, W = label.shape
            d = int(label.max()) + 1
            chains[0].begin(feats[0], label, d, scaled_map_size(H, W), (H, W))
            lab2 = label if spec['label2'] is None else spec['label2'](label)
            chains[1].begin(feats[1], lab2, d, scaled_map_size(H, W, scale if spec['scaled'] else None), (H, W))
            if save is not None:
                out_dir = Path(save) / current_video
                out_dir.mkdir(parents=True, exist_ok=True)
                ann_img.save(out_dir / '00000.png')
            videos += 1
        else:
            a = chains[0].step(feats[0], probability_propagation)
            b = chains[1].step(feats[1], probability_propagation)
            masks.append(fuse_two(a, b, probability_propagation, reduction_str, spec['unflip']))
        last_video = current_video
        frame_idx += 1
        n_frames += 1
    flush(last_video)
    torch.cuda.synchronize()
    writer.close()
    for c in chains:
        c.close()
    if stats is not None:
        stats.update(frames=n_frames, videos=videos, seconds=time.perf_counter() - t0)


def inference_hor_flip(model, inference_loader, total_len, annotation_dir, last_video, save, sigma_1, sigma_2,
                       frame_range, ref_num, temperature, probability_propagation, reduction_str, disable, **engine_opts):
    """reference inference_utils.py:90-187"""
    _inference_two_branch('hor-flip', [model, model], inference_loader, total_len, annotation_dir, last_video, save,
                          sigma_1, sigma_2, frame_range, ref_num, temperature, probability_propagation, None,
                          reduction_str, disable, **engine_opts)


def inference_ver_flip(model, inference_loader, total_len, annotation_dir, last_video, save, sigma_1, sigma_2,
                       frame_range, ref_num, temperature, probability_propagation, reduction_str, disable, **engine_opts):
    """reference inference_utils.py:196-298"""
    _inference_two_branch('vert-flip', [model, model], inference_loader, total_len, annotation_dir, last_video, save,
                          sigma_1, sigma_2, frame_range, ref_num, temperature, probability_propagation, None,
                          reduction_str, disable, **engine_opts)


def inference_2_scale(model, inference_loader, total_len, annotation_dir, last_video, save, sigma_1, sigma_2,
                      frame_range, ref_num, temperature, probability_propagation, scale, reduction_str, flip_pred,
                      disable, **engine_opts):
    """reference inference_utils.py:300-413 ('2-scale', and 'hor-2-scale' with flip_pred=True)"""
    _inference_two_branch('hor-2-scale' if flip_pred else '2-scale', [model, model], inference_loader, total_len,
                          annotation_dir, last_video, save, sigma_1, sigma_2, frame_range, ref_num, temperature,
                          probability_propagation, scale, reduction_str, disable, **engine_opts)


def inference_multimodel(model, additional_model, inference_loader, total_len, annotation_dir, last_video, save,
                         sigma_1, sigma_2, frame_range, ref_num, temperature, probability_propagation, reduction_str,
                         disable, **engine_opts):
    """reference inference_utils.py:416-511"""
    _inference_two_branch('multimodel', [model, additional_model], inference_loader, total_len, annotation_dir, last_video,
                          save, sigma_1, sigma_2, frame_range, ref_num, temperature, probability_propagation, None,
                          reduction_str, disable, **engine_opts)


def inference_3_scale(model, inference_loader, total_len, annotation_dir, last_video, save, sigma_1, sigma_2,
                      frame_range, ref_num, temperature, probability_propagation, scale, disable, encoder_dtype=None,
                      stats=None, encoder_batch=32, output_size=THREE_SCALE_OUTPUT, png_workers=2, precision=0,
                      align_videos=False):
    """reference inference_utils.py:514-595: three full passes over the loader at input scales [0.9, 1.0, scale] (nearest
    pre-scaling of the normalised image), each a single chain whose class maps are produced at `output_size` (the
    reference hard-codes 480x910 whatever the video size); the saved mask is the element-wise maximum of the three class
    maps.  Class maps are kept on the host between passes, as the reference does (one byte per pixel)."""
    import time
    from tqdm import tqdm
    device = Config.DEVICE
    if device.type != 'cuda':
        raise _engine.VospropError("--device cpu: the propagation engine is HIP-only (no CPU fallback)")
    chain = _Chain(device, ref_num=ref_num, frame_range=frame_range, sigma1=sigma_1, sigma2=sigma_2,
                   temperature=temperature, probability=probability_propagation, precision=precision)
    per_video, palettes, order = {}, {}, []
    n_frames, videos = 0, 0
    t0 = time.perf_counter()
    for s in (0.9, 1.0, scale):
        def resize(H, W, _s=s):
            return int(np.ceil(H * _s)), int(np.ceil(W * _s))
        masks, frame_idx, prev = [], 0, None

        def flush(video):
            if masks:
                per_video.setdefault(video, []).append(torch.stack(masks).cpu().numpy())
                masks.clear()

        stream = encoded_branches([model], inference_loader, device, encoder_dtype, max(1, encoder_batch), resize=resize,
                                  align_videos=align_videos)
        for feats, current_video in tqdm(stream, total=total_len, disable=disable):
            if prev is not None and current_video != prev:
                flush(prev)
                frame_idx = 0
            if frame_idx == 0:
                label, palette, ann_img = _read_annotation(Path(annotation_dir) / current_video / '00000.png')
                H, W = label.shape
                chain.begin(feats[0], label, int(label.max()) + 1, scaled_map_size(H, W, s), output_size)
                if current_video not in palettes:
                    palettes[current_video] = palette
                    order.append(current_video)
                    videos += 1
                if save is not None:
                    out_dir = Path(save) / current_video
                    out_dir.mkdir(parents=True, exist_ok=True)
                    ann_img.save(out_dir / '00000.png')
            else:
                # nearest up-sample + argmax in both modes (:574-576), fused in the engine's mask output
                masks.append(chain.eng.step(feats[0], want_pred=False, want_mask=True)[1])
            prev = current_video
            frame_idx += 1
            n_frames += 1
        flush(prev)
    torch.cuda.synchronize()
    chain.close()
    writer = AsyncMaskWriter(save, png_workers)
    for video in order:
        frames = per_video.get(video)
        if frames and len(frames) == 3:
            writer.submit(video, palettes[video], np.maximum(np.maximum(frames[0], frames[1]), frames[2]))
    writer.close()
    if stats is not None:
        stats.update(frames=n_frames, videos=videos, seconds=time.perf_counter() - t0)
