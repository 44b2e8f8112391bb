"""MI355X-native label-propagation engine: a drop-in for the per-frame propagation step of
hynekdav/semi-supervised-VOS (`predict`, src/model/predict.py:19-71, inside the `inference_single`
loop, src/utils/inference_utils.py:23-87).  The arithmetic lives in hand-written HIP
(csrc/, C ABI in include/vosprop.h); this package is the host-side mirror of the reference's
operator interface.  No CPU fallback exists: without libvosprop.so / a HIP device it raises."""
from . import _native
from ._native import VospropError, build
from .engine import PropagationEngine, feature_map_size, sample_frames_list, PREC_BF16, PREC_F32
from .config import Config
from . import inference_utils  # noqa: E402,F401

__all__ = ['Config', 'PropagationEngine', 'VospropError', 'build', 'feature_map_size', 'sample_frames_list',
           'PREC_BF16', 'PREC_F32']
