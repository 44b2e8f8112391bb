"""Python face of the stateful engine (include/vosprop.h) - device plumbing only (torch tensors carry the
HBM pointers and the HIP stream); all arithmetic happens in libvosprop.so."""
import ctypes
import math

import numpy as np
import torch

from . import _native
from ._native import VospropError

DT_CODES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
PREC_BF16, PREC_F32 = 0, 1
LAYOUT_HWC = 0x10       # VOSPROP_LAYOUT_HWC
SCALE = 0.125           # reference src/config.py:12
CONTINUOUS_FRAME = 4    # reference src/config.py:13


def feature_map_size(H, W):
    """reference src/model/predict.py:109-110"""
    return int(math.ceil(H * SCALE)), int(math.ceil(W * SCALE))


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class PropagationEngine:
    """One engine context per (GPU, stream): feature/label ring in HBM + the fused propagation kernel.

    Stands in for the per-video state that the reference's `inference_single` keeps in module globals
    (feats_history, label_history, weight_dense, weight_sparse, d; src/utils/inference_utils.py:25).
    """

    def __init__(self, feat_h, feat_w, device=None, ref_num=9, frame_range=40, sigma1=8.0, sigma2=21.0,
                 temperature=1.0, probability=False, topk=0, precision=PREC_BF16, ring_capacity=0, materialise=False):
        L = _native.lib()
        if not torch.cuda.is_available():
            raise VospropError('no HIP device visible: the propagation engine has no CPU path')
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else
                                   (device if isinstance(device, int) else torch.device(device).index or 0))
        cfg = _native.Config()
        L.vosprop_default_config(ctypes.byref(cfg), int(feat_h), int(feat_w))
        cfg.device = self.device.index
        cfg.ref_num, cfg.frame_range = int(ref_num), int(frame_range)
        cfg.sigma1, cfg.sigma2, cfg.temperature = float(sigma1), float(sigma2), float(temperature)
        cfg.probability, cfg.topk, cfg.precision = int(bool(probability)), int(topk), int(precision)
        cfg.ring_capacity = int(ring_capacity)
        cfg.materialise = int(bool(materialise))
        self.cfg = cfg
        self._L = L
        self._ctx = ctypes.c_void_p()
        rc = L.vosprop_create(ctypes.byref(self._ctx), ctypes.byref(cfg))
        if rc != 0:
            self._ctx = ctypes.c_void_p()
            raise VospropError(f'vosprop_create failed with code {rc}')
        self.feat_h, self.feat_w = int(feat_h), int(feat_w)
        self.HW = self.feat_h * self.feat_w
        self.d = None
        self.H = self.W = None

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, '_ctx', None) is not None and self._ctx.value:
            self._L.vosprop_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self._L.vosprop_last_error(self._ctx)
            raise VospropError(f'{what} failed ({rc}): {msg.decode() if msg else ""}')

    # -- video API --------------------------------------------------------------------------------
    def begin_video(self, first_label, sync=False):
        """first_label: (H, W) integer class map of the first annotation (00000.png).  Returns d.  Enqueued on the current
        stream of the engine's device (vosprop_begin_video_on); sync=True uses the device-synchronising ABI call."""
        lab = np.ascontiguousarray(np.asarray(first_label), dtype=np.uint8)
        if lab.ndim != 2:
            raise ValueError('first_label must be (H, W)')
        d = ctypes.c_int(0)
        if sync:
            rc = self._L.vosprop_begin_video(self._ctx, lab.ctypes.data_as(ctypes.c_void_p), lab.shape[0], lab.shape[1],
                                             ctypes.byref(d))
        else:
            rc = self._L.vosprop_begin_video_on(self._ctx, lab.ctypes.data_as(ctypes.c_void_p), lab.shape[0], lab.shape[1],
                                                ctypes.byref(d), _stream_ptr(self.device))
        self._check(rc, 'vosprop_begin_video')
        self.d, self.H, self.W = d.value, lab.shape[0], lab.shape[1]
        return self.d

    def begin_video_labels(self, cls_lowres, d, out_hw):
        """Start a video from an already down-sampled class map (feat_h, feat_w) - the multi-scale / flip strategies.
        Masks of step() are up-sampled to out_hw."""
        lab = np.ascontiguousarray(np.asarray(cls_lowres), dtype=np.uint8)
        if lab.shape != (self.feat_h, self.feat_w):
            raise ValueError(f'label map must be ({self.feat_h},{self.feat_w}), got {lab.shape}')
        rc = self._L.vosprop_begin_video_labels_on(self._ctx, lab.ctypes.data_as(ctypes.c_void_p), int(d), int(out_hw[0]),
                                                   int(out_hw[1]), _stream_ptr(self.device))
        self._check(rc, 'vosprop_begin_video_labels')
        self.d, self.H, self.W = int(d), int(out_hw[0]), int(out_hw[1])
        return self.d

    @property
    def frame_index(self):
        return self._L.vosprop_frame_index(self._ctx)

    def step(self, features, want_pred=True, want_mask=True, mask_out=None):
        """features: (C,H_d,W_d) or (1,C,H_d,W_d) tensor on this engine's GPU (f32 / f16 / bf16).
        Frame 0 returns (None, None); later frames return (prediction (d,HW) f32 | None, mask (H,W) u8 | None).
        mask_out: optional contiguous (H,W) uint8 tensor on this GPU the mask is written into (e.g. a slice of a batch buffer
        that goes back to the host in one copy) instead of a fresh tensor."""
        if self.frame_index < 0:
            raise VospropError('step() before begin_video()')
        f = features[0] if features.dim() == 4 else features
        if f.device != self.device:
            raise VospropError(f'features live on {f.device}, engine on {self.device}')
        if tuple(f.shape) != (256, self.feat_h, self.feat_w):
            raise ValueError(f'features must be (256,{self.feat_h},{self.feat_w}), got {tuple(f.shape)}')
        if f.dtype not in DT_CODES:
            raise ValueError(f'unsupported feature dtype {f.dtype}')
        # a slice of a channels_last batch is already pixel-major (H_d*W_d, C): hand it over as it is (VOSPROP_LAYOUT_HWC)
        hwc = f.stride(0) == 1 and f.stride(2) == f.shape[0] and f.stride(1) == f.shape[0] * f.shape[2]
        if not hwc:
            f = f.contiguous()
        first = self.frame_index == 0
        pred = mask = None
        if not first:
            if want_pred:
                pred = torch.empty((self.d, self.HW), dtype=torch.float32, device=self.device)
            if mask_out is not None:
                if (mask_out.dtype != torch.uint8 or tuple(mask_out.shape) != (self.H, self.W) or mask_out.device != self.device
                        or not mask_out.is_contiguous()):
                    raise ValueError(f'mask_out must be a contiguous ({self.H},{self.W}) uint8 tensor on {self.device}')
                mask = mask_out
            elif want_mask:
                mask = torch.empty((self.H, self.W), dtype=torch.uint8, device=self.device)
        rc = self._L.vosprop_step(self._ctx, ctypes.c_void_p(f.data_ptr()), DT_CODES[f.dtype] | (LAYOUT_HWC if hwc else 0),
                                  ctypes.c_void_p(pred.data_ptr()) if pred is not None else None,
                                  ctypes.c_void_p(mask.data_ptr()) if mask is not None else None,
                                  _stream_ptr(self.device))
        self._check(rc, 'vosprop_step')
        return pred, mask

    # -- stateless operator -----------------------------------------------------------------------
    def predict(self, ref, target, ref_label, frame_idx, frame_range, ref_num, temperature, sigma1, sigma2,
                probability_propagation):
        """The reference's predict() (src/model/predict.py:19-71) with the weight matrices given by their sigmas.
        An engine created with topk=k applies the top-k variant (label mode only)."""
        T = ref.shape[0]
        d = ref_label.shape[0]
        ref = ref.contiguous()
        target = target.contiguous().to(ref.dtype)
        lab = ref_label.to(torch.float32).contiguous()
        if tuple(lab.shape) != (d, T, self.HW):
            raise ValueError(f'ref_label must be (d,{T},{self.HW}), got {tuple(lab.shape)}')
        out = torch.empty((d, self.HW), dtype=torch.float32, device=self.device)
        rc = self._L.vosprop_predict(self._ctx, ctypes.c_void_p(ref.data_ptr()), ctypes.c_void_p(target.data_ptr()),
                                     DT_CODES[ref.dtype], ctypes.c_void_p(lab.data_ptr()), T, d, int(frame_idx),
                                     int(frame_range), int(ref_num), float(temperature), float(sigma1), float(sigma2),
                                     int(bool(probability_propagation)), ctypes.c_void_p(out.data_ptr()),
                                     _stream_ptr(self.device))
        self._check(rc, 'vosprop_predict')
        return out

    # -- measurement ------------------------------------------------------------------------------
    def last_stats(self):
        st = _native.Stats()
        self._check(self._L.vosprop_last_stats(self._ctx, ctypes.byref(st)), 'vosprop_last_stats')
        out = {k: getattr(st, k) for k, _ in st._fields_ if k != 'reserved_'}
        out['kernel'] = self._L.vosprop_kernel_name(st.kernel_id).decode()      # what the engine launched (set where it decides)
        return out

    def topk_overflows(self):
        """(dump slots, combine groups, select candidates): how often a capacity limit of the top-k kernels dropped candidates since
        the engine's first top-k step (include/vosprop.h vosprop_topk_overflows); waits for the current stream."""
        out = (ctypes.c_uint * 3)()
        self._check(self._L.vosprop_topk_overflows(self._ctx, out, _stream_ptr(self.device)), 'vosprop_topk_overflows')
        return tuple(int(v) for v in out)

    def timing_begin(self):
        """Start bracketing every dense propagation-kernel launch with HIP events on its stream (in-situ timing)."""
        self._check(self._L.vosprop_timing_begin(self._ctx), 'vosprop_timing_begin')

    def timing_read(self):
        """-> (mean kernel duration in us, number of launches) since timing_begin(); waits for those launches."""
        us, n = ctypes.c_double(0.0), ctypes.c_int(0)
        self._check(self._L.vosprop_timing_read(self._ctx, ctypes.byref(us), ctypes.byref(n)), 'vosprop_timing_read')
        return us.value, n.value

    def time_last_propagation(self, iters=20):
        """Mean duration (us) of the propagation kernel alone, HIP events on the launch stream."""
        us = ctypes.c_double(0.0)
        rc = self._L.vosprop_time_last_propagation(self._ctx, int(iters), _stream_ptr(self.device), ctypes.byref(us))
        self._check(rc, 'vosprop_time_last_propagation')
        return us.value


def sample_frames_list(frame_idx, take_range, num_refs):
    """reference sample_frames (src/model/predict.py:74-89) through the C ABI (host-side, exact)."""
    buf = (ctypes.c_int * (max(int(num_refs), int(frame_idx), 1) + 4))()
    n = _native.lib().vosprop_sample_frames(int(frame_idx), int(take_range), int(num_refs), buf)
    if n < 0:     # num_refs < 3 past frame num_refs: the reference's np.linspace(.., num < 0) raises ValueError
        raise ValueError(f'Number of samples, {int(num_refs) - 3}, must be non-negative.')
    return [buf[i] for i in range(n)]
