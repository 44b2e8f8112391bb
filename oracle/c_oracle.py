"""ctypes binding of oracle/libvos_oracle.so (the plain-C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


def build():
    subprocess.run(['make', '-s', '-C', str(_HERE), 'libvos_oracle.so'], check=True)


def lib():
    global _LIB
    if _LIB is None:
        so = _HERE / 'libvos_oracle.so'
        if not so.exists():
            build()
        L = ctypes.CDLL(str(so))
        f32p = np.ctypeslib.ndpointer(np.float32, flags='C_CONTIGUOUS')
        u8p = np.ctypeslib.ndpointer(np.uint8, flags='C_CONTIGUOUS')
        i32p = np.ctypeslib.ndpointer(np.int32, flags='C_CONTIGUOUS')
        L.vos_oracle_sample_frames.argtypes = [ctypes.c_int] * 3 + [i32p]
        L.vos_oracle_sample_frames.restype = ctypes.c_int
        L.vos_oracle_spatial_weight.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_float, f32p]
        L.vos_oracle_spatial_weight.restype = None
        L.vos_oracle_get_labels.argtypes = [u8p] + [ctypes.c_int] * 5 + [f32p]
        L.vos_oracle_get_labels.restype = None
        L.vos_oracle_predict.argtypes = ([f32p, f32p, f32p] + [ctypes.c_int] * 8 + [ctypes.c_float] * 3
                                         + [ctypes.c_int, f32p])
        L.vos_oracle_predict.restype = ctypes.c_int
        L.vos_oracle_argmax_upsample.argtypes = [f32p] + [ctypes.c_int] * 5 + [u8p, ctypes.c_void_p]
        L.vos_oracle_argmax_upsample.restype = None
        _LIB = L
    return _LIB


def sample_frames(frame_idx, take_range, num_refs):
    out = np.zeros(max(num_refs, frame_idx), dtype=np.int32)
    n = lib().vos_oracle_sample_frames(frame_idx, take_range, num_refs, out)
    return out[:n].tolist()


def spatial_weight(H, W, sigma):
    out = np.empty((H * W, H * W), dtype=np.float32)
    lib().vos_oracle_spatial_weight(H, W, sigma, out)
    return out


def get_labels(label, Hd, Wd, d):
    label = np.ascontiguousarray(label, dtype=np.uint8)
    H, W = label.shape
    out = np.empty((d, Hd * Wd), dtype=np.float32)
    lib().vos_oracle_get_labels(label, H, W, Hd, Wd, d, out)
    return out


def predict(ref, target, ref_label, frame_idx, take_range, ref_num, temperature, sigma1, sigma2, prob):
    ref = np.ascontiguousarray(ref, dtype=np.float32)
    T, C, Hd, Wd = ref.shape
    target = np.ascontiguousarray(target, dtype=np.float32).reshape(C, Hd * Wd)
    ref_label = np.ascontiguousarray(ref_label, dtype=np.float32)
    d = ref_label.shape[0]
    assert ref_label.shape == (d, T, Hd * Wd)
    out = np.empty((d, Hd * Wd), dtype=np.float32)
    rc = lib().vos_oracle_predict(ref.reshape(T, C, Hd * Wd), target, ref_label, T, C, Hd, Wd, d, frame_idx,
                                  take_range, ref_num, temperature, sigma1, sigma2, int(bool(prob)), out)
    if rc != 0:
        raise RuntimeError(f'vos_oracle_predict failed rc={rc}')
    return out


def argmax_upsample(pred, Hd, Wd, H, W):
    pred = np.ascontiguousarray(pred, dtype=np.float32)
    d = pred.shape[0]
    cls = np.empty(Hd * Wd, dtype=np.uint8)
    mask = np.empty((H, W), dtype=np.uint8)
    lib().vos_oracle_argmax_upsample(pred, d, Hd, Wd, H, W, cls, mask.ctypes.data)
    return cls, mask
