"""Model registry mirror of reference src/dlwpbench/models/__init__.py:4-15.

The reference scripts resolve a backbone with `from models import *` followed by
`eval(cfg.model.type)(**cfg.model)` (scripts/train.py:18,54; scripts/evaluate.py:34,140), so the
drop-in is a package that exports the same CLASS NAMES.  Put `dlwp_benchmark_amd/` in front of
`sys.path` as `models` or import from here directly.
"""
from .fno import FNO2DModule
from .pangu import PanguWeather
from .spectral import SpectralConv2d
from .swin import SwinTransformer

__all__ = ["FNO2DModule", "PanguWeather", "SpectralConv2d", "SwinTransformer"]
