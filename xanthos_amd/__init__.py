"""xanthos_amd -- the Xanthos monthly PET -> runoff -> routing hot path on AMD MI355X (gfx950).

Public surface mirrors the reference package (xanthos/__init__.py:1-3): ``Xanthos``, ``run_model``, ``ConfigReader``,
``ConfigRunner``, ``Calibrate``.  Compute goes through hand-written HIP kernels behind the C-ABI in
``include/xanthos_hip.h`` (``libxanthos_hip.so``); there is no CPU fallback.
"""
from .ini_reader import ConfigReader, ValidationException        # noqa: F401
from .configurations import ConfigRunner                          # noqa: F401
from .model import Xanthos, run_model                             # noqa: F401
from .calibrate.calibrate_abcd import Calibrate                   # noqa: F401

__version__ = '0.1.0'
