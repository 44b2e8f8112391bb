"""Global constants of the reference (src/config.py:10-14).  DEVICE is mutable, as in the reference
(src/inference.py:57-58 overwrites it from --device); 'cuda' means the HIP device under ROCm."""
import multiprocessing

import torch


class Config(object):
    DEVICE = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
    SCALE = 0.125
    CONTINUOUS_FRAME = 4
    CPU_COUNT = max(multiprocessing.cpu_count(), 1)
