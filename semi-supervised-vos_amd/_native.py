"""Loader (and in-tree builder) of libvosprop.so - the hand-written HIP engine behind include/vosprop.h.

There is NO fallback: if the shared library is missing or cannot be loaded, every entry point of the
package raises.  `build()` compiles it in-tree with hipcc for gfx950 (cross-compiles without a GPU).
"""
import ctypes
import os
import shutil
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_SRC = _PKG / 'csrc'
LIB_PATH = Path(os.environ['VOSPROP_LIB']) if os.environ.get('VOSPROP_LIB') else _PKG / 'libvosprop.so'   # override: dev ablation builds only
_LIB = None


class VospropError(RuntimeError):
    pass


class Config(ctypes.Structure):
    """Mirror of `vosprop_config` (include/vosprop.h)."""
    _fields_ = [('abi_version', ctypes.c_int), ('device', ctypes.c_int), ('feat_h', ctypes.c_int),
                ('feat_w', ctypes.c_int), ('channels', ctypes.c_int), ('ref_num', ctypes.c_int),
                ('frame_range', ctypes.c_int), ('sigma1', ctypes.c_float), ('sigma2', ctypes.c_float),
                ('temperature', ctypes.c_float), ('probability', ctypes.c_int), ('topk', ctypes.c_int),
                ('precision', ctypes.c_int), ('ring_capacity', ctypes.c_int), ('materialise', ctypes.c_int),
                ('reserved', ctypes.c_int * 7)]


class Stats(ctypes.Structure):
    """Mirror of `vosprop_stats`."""
    _fields_ = [('n_ref', ctypes.c_int), ('hw', ctypes.c_int), ('workgroups', ctypes.c_int),
                ('tiles_per_wg', ctypes.c_int), ('flops', ctypes.c_double), ('bytes', ctypes.c_double),
                ('kernel_id', ctypes.c_int), ('reserved_', ctypes.c_int)]


# vosprop_stats.kernel_id (include/vosprop.h VOSPROP_KERNEL_*)
KERNEL_DENSE, KERNEL_MASK, KERNEL_TOPK, KERNEL_F32, KERNEL_MATERIALISED = 1, 2, 3, 4, 5


# every symbol include/vosprop.h declares: name -> (restype, argtypes)
_vp = ctypes.c_void_p
SYMBOLS = {
    'vosprop_default_config': (None, [ctypes.POINTER(Config), ctypes.c_int, ctypes.c_int]),
    'vosprop_version': (ctypes.c_char_p, []),
    'vosprop_create': (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(Config)]),
    'vosprop_destroy': (None, [_vp]),
    'vosprop_last_error': (ctypes.c_char_p, [_vp]),
    'vosprop_begin_video': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    'vosprop_begin_video_labels': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'vosprop_begin_video_on': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), _vp]),
    'vosprop_begin_video_labels_on': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    'vosprop_step': (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
    'vosprop_timing_begin': (ctypes.c_int, [_vp]),
    'vosprop_timing_read': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    'vosprop_bias_act': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    'vosprop_bias_relu_maxpool': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, _vp]),
    'vosprop_pointwise_conv': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, _vp]),
    'vosprop_frame_index': (ctypes.c_int, [_vp]),
    'vosprop_set_deterministic': (ctypes.c_int, [ctypes.c_int]),
    'vosprop_predict': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_int, _vp, _vp]),
    'vosprop_sample_frames': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    'vosprop_last_stats': (ctypes.c_int, [_vp, ctypes.POINTER(Stats)]),
    'vosprop_kernel_name': (ctypes.c_char_p, [ctypes.c_int]),
    'vosprop_topk_overflows': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint), _vp]),
    'vosprop_time_last_propagation': (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.POINTER(ctypes.c_double)]),
}

# -fno-slp-vectorize: packed f32 VALU instructions do not overlap with MFMAs on gfx950 (tools/ubench_slot.hip); without the
# flag hipcc packs the softmax sums into v_pk_add_f32 and the shipped kernel is 5 % slower (measured)
HIPCC_FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-shared', '-fno-slp-vectorize', '-mllvm', '-amdgpu-mfma-vgpr-form=1']
# hipBLASLt serves the encoder's pointwise convolutions (csrc/pointwise.h); in a torch process the copy torch has already loaded
# (same SONAME) is the one that binds
LINK_FLAGS = ['-L/opt/rocm/lib', '-lhipblaslt', '-Wl,-rpath,/opt/rocm/lib']


def build(force=False, verbose=False):
    """Compile csrc/engine.hip -> libvosprop.so with hipcc for gfx950 (in-tree, so it travels to the GPU box)."""
    srcs = sorted(_SRC.glob('*.hip')) + sorted(_SRC.glob('*.h')) + [_PKG.parent / 'include' / 'vosprop.h']
    if LIB_PATH.exists() and not force:
        if all(LIB_PATH.stat().st_mtime >= s.stat().st_mtime for s in srcs):
            return LIB_PATH
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    cmd = [hipcc] + HIPCC_FLAGS + ['-o', str(LIB_PATH), str(_SRC / 'engine.hip')] + LINK_FLAGS
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True, cwd=str(_SRC))
    return LIB_PATH


def lib():
    """The loaded library; raises VospropError (never falls back) when it is not there."""
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            raise VospropError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                               '(hipcc --offload-arch=gfx950); there is no CPU fallback for the propagation path')
        try:
            L = ctypes.CDLL(str(LIB_PATH))
        except OSError as e:
            raise VospropError(f'cannot load {LIB_PATH}: {e}') from e
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB
