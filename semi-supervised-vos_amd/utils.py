"""Checkpoint loading and PNG output with the reference's semantics (src/utils/utils.py:34-42,59-100)."""
import os
from pathlib import Path

import numpy as np
import torch

from .config import Config
from .predict import index_to_onehot  # noqa: F401  (re-exported, as in the reference's utils)


def save_prediction(prediction, palette, save_path, save_name, video_name):
    """reference utils.py:34-42: int32 array -> 'I' -> 'L' -> palette -> 'P' PNG."""
    from PIL import Image
    img = Image.fromarray(prediction)
    img = img.convert('L')
    img.putpalette(palette)
    img = img.convert('P')
    video_path = Path(save_path) / video_name
    video_path.mkdir(parents=True, exist_ok=True)
    img.save((video_path / (save_name + '.png')).absolute())


def save_predictions(predictions, palette, save, video_name):
    """reference utils.py:97-100: frames are written as 00001.png ... (1-based, 5 digits)."""
    for idx, prediction in enumerate(predictions, start=1):
        save_prediction(np.asarray(prediction).astype(np.int32), palette, save, str(idx).zfill(5), video_name)


def load_model(model, checkpoint):
    """reference utils.py:71-94: accepts {'state_dict': ...} or a raw state dict; keys saved from an
    nn.DataParallel wrapper ('module.' prefix) are accepted too (the reference retries through DataParallel;
    here the prefix is stripped).  A missing file is an error (the reference calls exit(-1))."""
    if checkpoint is None:
        return model
    if not os.path.isfile(checkpoint):
        raise FileNotFoundError(f"no checkpoint found at '{checkpoint}'")
    ckpt = torch.load(checkpoint, map_location=Config.DEVICE)
    sd = ckpt['state_dict'] if isinstance(ckpt, dict) and 'state_dict' in ckpt else ckpt
    if any(k.startswith('module.') for k in sd):
        sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
    model.load_state_dict(sd)
    return model
