import importlib
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests' / 'golden'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def _built_library():
    """The suite needs libvosprop.so (and the oracle's C restatement): build them once if this is a fresh checkout and a
    compiler is present - the same thing `__graft_entry__.build()` does.  The product itself never builds on import: without
    the library it raises (test_missing_library_is_an_error)."""
    import shutil
    native = importlib.import_module('semi-supervised-vos_amd._native')
    if not native.LIB_PATH.exists() and (shutil.which('hipcc') or Path('/opt/rocm/bin/hipcc').exists()):
        native.build()
    yield


@pytest.fixture(scope='session')
def goldens():
    import numpy as np
    return np.load(ROOT / 'tests' / 'golden' / 'reference_goldens.npz')


@pytest.fixture(scope='session')
def vos():
    """The product package (directory name has a hyphen, so it is imported by string)."""
    return importlib.import_module('semi-supervised-vos_amd')
