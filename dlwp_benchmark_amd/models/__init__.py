"""Model registry mirror of reference src/dlwpbench/models/__init__.py:4-15.

The reference scripts resolve a backbone with `from models import *` followed by
`eval(cfg.model.type)(**cfg.model)` (scripts/train.py:18,54; scripts/evaluate.py:34,140), so the
drop-in is a package that exports the same CLASS NAMES (live names of :6-12 plus the commented
FNO2DModule / ConvLSTM of :4-5 that the north star covers).  Put this package in front of `sys.path`
as `models` or import from here directly.
"""
from .fno import FNO2DModule, TFNO2DModule
from .fourcastnet import AFNONet, FourCastNet
from .pangu import PanguWeather
from .spectral import SpectralConv2d
from .swin import SwinTransformer, SwinTransformerHPX
from .diffusion import DiffModernUNet, DiffMUNetHPX
from .unet import ConvLSTM, ConvLSTMHPX, HEALPixLayer, HEALPixPadding, ModernUNet, MUNetHPX, UNet, UNetHPX

__all__ = ["DiffModernUNet", "DiffMUNetHPX", "FNO2DModule", "TFNO2DModule", "ConvLSTMHPX", "ModernUNet", "FourCastNet", "AFNONet", "PanguWeather", "SpectralConv2d", "SwinTransformer", "SwinTransformerHPX", "UNet",
           "UNetHPX", "MUNetHPX", "ConvLSTM", "HEALPixPadding", "HEALPixLayer"]
