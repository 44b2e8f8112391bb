"""Input contract of the reference's InferenceDataset (src/utils/datasets.py:111-167) without torchvision:
sorted video directories, sorted frames, JPEG -> RGB -> [0,1] CHW f32 -> ImageNet normalisation.
Items are (tensor (3,H,W), video_name); the DataLoader's batch of 1 turns that into ((1,3,H,W), (name,)).  The flip and
2-scale strategies return a PAIR of tensors per frame (datasets.py:148-162): the frame and its mirrored / flipped /
Lanczos-rescaled copy (PIL's ANTIALIAS is the old name of LANCZOS)."""
from pathlib import Path

import numpy as np
import torch
import torch.utils.data

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
STRATEGIES = ('single', 'hor-flip', 'vert-flip', '2-scale', 'multimodel', 'hor-2-scale', '3-scale')
_EXT = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif', '.tiff', '.webp')


def list_videos(root):
    """{video_name: [frame paths]} in the order torchvision's ImageFolder walks them (sorted dirs, sorted files)."""
    root = Path(root)
    out = {}
    for vdir in sorted(p for p in root.iterdir() if p.is_dir()):
        frames = sorted(f for f in vdir.rglob('*') if f.is_file() and f.suffix.lower() in _EXT)
        if frames:
            out[vdir.name] = frames
    return out


def raw_image(img):
    """PIL image -> (H,W,3) uint8 tensor: the decode workers' output on the fast path (4x less data through the loader's
    queues than normalised f32; `normalize_on_device` finishes the job on the GPU with the same f32 operations)."""
    return torch.from_numpy(np.array(img.convert('RGB'), dtype=np.uint8))


_LUT = {}


def normalize_on_device(x):
    """(B,H,W,3) uint8 on the GPU -> (B,3,H,W) f32, bit-identical to `normalize_image`: a byte has 256 values, so the
    per-channel results of ToTensor + Normalize are tabulated on the host with the host's arithmetic and looked up on the
    device (device-side f32 division is not correctly rounded, a computed version differs in the last bit)."""
    key = str(x.device)
    lut = _LUT.get(key)
    if lut is None:
        v = np.arange(256, dtype=np.float32) / 255.0
        tab = (v[None, :] - np.asarray(IMAGENET_MEAN, np.float32)[:, None]) / np.asarray(IMAGENET_STD, np.float32)[:, None]
        lut = _LUT[key] = torch.from_numpy(np.ascontiguousarray(tab.astype(np.float32))).to(x.device)
    idx = x.permute(0, 3, 1, 2).to(torch.int64)                       # (B,3,H,W)
    return torch.gather(lut[None, :, :].expand(x.shape[0], 3, 256), 2, idx.flatten(2)).view(idx.shape)


def normalize_image(img):
    """PIL image -> (3,H,W) f32, ToTensor + Normalize (datasets.py:128-131,147)."""
    a = np.asarray(img.convert('RGB'), dtype=np.float32) / 255.0
    a = (a - np.asarray(IMAGENET_MEAN, np.float32)) / np.asarray(IMAGENET_STD, np.float32)
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))


class InferenceDataset(torch.utils.data.Dataset):
    def __init__(self, root, transform=None, target_transform=None, disable=False, inference_strategy='single',
                 scale=None, videos=None, raw_uint8=False):
        if inference_strategy not in STRATEGIES:
            raise ValueError(f"unknown inference strategy '{inference_strategy}'")
        self.inference_strategy = inference_strategy
        self.scale = scale
        self._to_tensor = raw_image if raw_uint8 else normalize_image   # raw_uint8: [engine] fast path, see raw_image()
        vids = list_videos(root)
        if videos is not None:           # a shard: subset of the video names, reference order kept
            vids = {k: v for k, v in vids.items() if k in set(videos)}
        self.videos = vids
        self.imgs = [(p, name) for name, frames in vids.items() for p in frames]
        self.img_bytes = [Path(p).read_bytes() for p, _ in self.imgs]   # preloaded, as the reference does (:133-135)

    def __getitem__(self, index):
        from io import BytesIO
        from PIL import Image
        from PIL import ImageOps
        img = Image.open(BytesIO(self.img_bytes[index])).convert('RGB')
        name = self.imgs[index][1]
        normalize_image = self._to_tensor
        normalized = normalize_image(img)
        st = self.inference_strategy
        if st == 'hor-flip':                                   # reference datasets.py:148-151
            return (normalized, normalize_image(ImageOps.mirror(img))), name
        if st == 'vert-flip':                                  # :152-155
            return (normalized, normalize_image(ImageOps.flip(img))), name
        if st in ('2-scale', 'hor-2-scale'):                   # :156-162: Lanczos resize to ceil(size * scale)
            size2 = tuple(int(v) for v in np.ceil(np.array(img.size) * self.scale))
            if st == 'hor-2-scale':
                img = ImageOps.mirror(img)
            return (normalized, normalize_image(img.resize(size2, Image.LANCZOS))), name
        return normalized, name                                # single, multimodel, 3-scale

    def __len__(self):
        return len(self.imgs)
