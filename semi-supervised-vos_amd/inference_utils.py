"""The `inference_single` frame loop of the reference (src/utils/inference_utils.py:23-87) on top of the engine.

Same signature and observable behaviour (frame 0 seeds the history, masks of a video are written as 00001.png...
with the first annotation's palette when the video changes), but no module globals: the per-video state lives in a
PropagationEngine, the (N*HW)xHW affinity is never formed, histories are a ring in HBM instead of torch.cat.
"""
from pathlib import Path

import numpy as np
import torch

from . import engine as _engine
from .config import Config
from .utils import save_predictions


def _read_annotation(path):
    from PIL import Image
    ann = Image.open(path)
    return np.asarray(ann), ann.getpalette(), ann


def encoded_frames(model, loader, device, encoder_dtype, batch):
    """Encoder look-ahead: yields (features (1,C,H_d,W_d), video_name) in loader order, but runs the encoder on up to `batch`
    consecutive frames of one video at a time - the features do not depend on the propagated labels (only the labels are
    sequential), and the encoder is ~3.5x cheaper per frame at batch 16 than at batch 1 on MI355X."""
    pend, names = [], []

    def flush():
        x = torch.cat(pend).to(device, non_blocking=True)
        if encoder_dtype is not None:
            x = x.to(encoder_dtype)
        with torch.no_grad():
            f = model(x.contiguous(memory_format=torch.channels_last))
        out = [(f[i:i + 1], names[i]) for i in range(len(names))]
        pend.clear()
        names.clear()
        return out

    for input, (name,) in loader:
        if names and (name != names[-1] or len(names) == batch or input.shape != pend[-1].shape):
            yield from flush()
        pend.append(input)
        names.append(name)
    if names:
        yield from flush()


def inference_single(model, inference_loader, total_len, annotation_dir, last_video, save, sigma_1, sigma_2,
                     frame_range, ref_num, temperature, probability_propagation, disable, encoder_dtype=None,
                     stats=None, encoder_batch=16):
    """stats (optional dict) receives {'frames', 'videos', 'seconds'} for the fps report."""
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

    def flush(video):
        if masks:
            save_predictions(torch.stack(masks).cpu().numpy(), palette, save, video)
            masks.clear()

    stream = encoded_frames(model, inference_loader, device, encoder_dtype, max(1, encoder_batch))
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
                                                temperature=temperature, probability=probability_propagation)
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
    if eng is not None:
        eng.close()
    if stats is not None:
        stats.update(frames=n_frames, videos=videos, seconds=time.perf_counter() - t0)
