"""Shim package: put this directory's parent (`<repo>/shim`) in front of the reference's `src/dlwpbench` on PYTHONPATH
and the reference scripts resolve their backbones on MI355X without a single edit:

    from models import *                                   # scripts/train.py:18, scripts/evaluate.py:34
    model = eval(cfg.model.type)(**cfg.model).to(device)   # scripts/train.py:54, scripts/evaluate.py:140

It re-exports the registry of dlwp_benchmark_amd.models under the reference's package name `models`
(reference src/dlwpbench/models/__init__.py:4-15).  Exercised by tests/test_registry_cpu.py with
every configs/model/*.yaml of the reference whose `type` is on the hot path.
"""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)

from dlwp_benchmark_amd.models import *          # noqa: E402,F401,F403
from dlwp_benchmark_amd.models import __all__    # noqa: E402,F401
