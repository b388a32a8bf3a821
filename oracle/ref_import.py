"""TEST INFRASTRUCTURE ONLY (oracle side) -- never imported by the product package.

Imports the *reference's own* Python model files from /root/reference (read-only) so that
golden vectors can be generated from the real reference code.  Follows the recipe recorded in
SURVEY.md section 8c: absent third-party modules get in-memory stand-ins that carry NO hot-path
arithmetic (DropPath is identity in eval mode; trunc_normal_ is torch's own initialiser; the
neuralop / torch_harmonics classes raise if anybody tries to construct them).  Nothing is
written to /root/reference (PYTHONDONTWRITEBYTECODE is forced on).

The reference cannot travel to the GPU box: this module is only ever used by
oracle/make_golden.py in the build container; the fixtures it emits are committed under
tests/golden/.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("DLWP_REFERENCE_ROOT", "/root/reference")
REF_PKG = os.path.join(REF_ROOT, "src", "dlwpbench")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_PKG, "models"))


def _install_standins():
    import torch

    sys.dont_write_bytecode = True

    # (i) timm.models.layers : DropPath / to_2tuple / trunc_normal_
    #     used at swin_transformer.py:18, panguweather.py:16, fourcastnet.py:19
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        timm_models = types.ModuleType("timm.models")
        timm_layers = types.ModuleType("timm.models.layers")

        class DropPath(torch.nn.Module):
            """Stochastic depth; identity in eval mode (the only mode parity is defined in)."""

            def __init__(self, drop_prob=0.0, scale_by_keep=True):
                super().__init__()
                self.drop_prob = drop_prob

            def forward(self, x):
                if self.training and self.drop_prob > 0.0:
                    raise RuntimeError("oracle stand-in DropPath only supports eval mode")
                return x

        def to_2tuple(x):
            if isinstance(x, (tuple, list)):
                return tuple(x)
            return (x, x)

        timm_layers.DropPath = DropPath
        timm_layers.to_2tuple = to_2tuple
        timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
        timm.models = timm_models
        timm_models.layers = timm_layers
        sys.modules["timm"] = timm
        sys.modules["timm.models"] = timm_models
        sys.modules["timm.models.layers"] = timm_layers

    # (ii) hydra.utils.instantiate : imported at unet.py:6, never called on the hot path
    if "hydra" not in sys.modules:
        hydra = types.ModuleType("hydra")
        hydra_utils = types.ModuleType("hydra.utils")

        def instantiate(*a, **k):
            raise RuntimeError("hydra is not installed (oracle stand-in)")

        hydra_utils.instantiate = instantiate
        hydra.utils = hydra_utils
        sys.modules["hydra"] = hydra
        sys.modules["hydra.utils"] = hydra_utils

    # (iii) neuralop / torch_harmonics : imported at module top of fno.py:7-8 and
    #       fourcastnet.py:17-18; constructing them is an error (third-party arithmetic is
    #       NOT available -> FNO2d parity is "unpinned")
    def _raiser(name):
        class _Absent:
            def __init__(self, *a, **k):
                raise RuntimeError(f"{name} is not installed (oracle stand-in)")

        _Absent.__name__ = name
        return _Absent

    if "neuralop" not in sys.modules:
        neuralop = types.ModuleType("neuralop")
        neuralop_models = types.ModuleType("neuralop.models")
        neuralop_models.FNO = _raiser("FNO")
        neuralop_models.TFNO = _raiser("TFNO")
        neuralop.models = neuralop_models
        sys.modules["neuralop"] = neuralop
        sys.modules["neuralop.models"] = neuralop_models
    if "torch_harmonics" not in sys.modules:
        th_ = types.ModuleType("torch_harmonics")
        th_ex = types.ModuleType("torch_harmonics.examples")
        th_sfno = types.ModuleType("torch_harmonics.examples.sfno")
        th_sfno.SphericalFourierNeuralOperatorNet = _raiser("SphericalFourierNeuralOperatorNet")
        th_.examples = th_ex
        th_ex.sfno = th_sfno
        sys.modules["torch_harmonics"] = th_
        sys.modules["torch_harmonics.examples"] = th_ex
        sys.modules["torch_harmonics.examples.sfno"] = th_sfno

    # (iv) numpy.lib.arraypad (fourcastnet.py:12, unused import; module is gone in numpy 2)
    import numpy as np

    if "numpy.lib.arraypad" not in sys.modules:
        ap = types.ModuleType("numpy.lib.arraypad")
        ap.pad = np.pad
        sys.modules["numpy.lib.arraypad"] = ap

    # (v) a bare namespace package "models" so that models/__init__.py (which cannot import,
    #     SURVEY.md header) is not executed while absolute "models.xxx" imports resolve.
    if REF_PKG not in sys.path:
        sys.path.insert(0, REF_PKG)
    if "models" not in sys.modules or getattr(sys.modules["models"], "__oracle_ns__", False) is False:
        ns = types.ModuleType("models")
        ns.__path__ = [os.path.join(REF_PKG, "models")]
        ns.__oracle_ns__ = True
        sys.modules["models"] = ns
        for sub in ("convlstm", "unet", "fourcastnet", "swintransformer", "panguweather", "fno"):
            m = types.ModuleType(f"models.{sub}")
            m.__path__ = [os.path.join(REF_PKG, "models", sub)]
            sys.modules[f"models.{sub}"] = m
            setattr(ns, sub, m)


def load_reference():
    """Returns a dict of the reference modules on the hot path."""
    import importlib

    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    _install_standins()
    out = {}
    out["utils"] = importlib.import_module("utils")
    out["convlstm"] = importlib.import_module("models.convlstm.convlstm")
    out["unet"] = importlib.import_module("models.unet.unet")
    out["fourcastnet"] = importlib.import_module("models.fourcastnet.fourcastnet")
    out["swin"] = importlib.import_module("models.swintransformer.swin_transformer")
    out["pangu"] = importlib.import_module("models.panguweather.panguweather")
    return out


def load_reference_diffusion():
    """models/diffusion_models/modern_unet/modern_unet.py (DiffModernUNet, DiffMUNetHPX).  AS SHIPPED the module cannot be
    imported: its line 4 asks `utils` for `ConditionalHEALPixLayer`, which utils/__init__.py does not export (the class exists,
    utils/healpix.py:117).  The oracle re-exports the reference's OWN class on the in-memory `utils` module -- no arithmetic is
    replaced -- and imports the file unchanged."""
    import importlib

    ref = load_reference()
    ref["utils"].ConditionalHEALPixLayer = importlib.import_module("utils.healpix").ConditionalHEALPixLayer
    for sub in ("diffusion_models", "diffusion_models.modern_unet"):
        name = f"models.{sub}"
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF_PKG, "models", *sub.split("."))]
            sys.modules[name] = m
    return importlib.import_module("models.diffusion_models.modern_unet.modern_unet")


if __name__ == "__main__":
    mods = load_reference()
    for k, v in mods.items():
        print(k, v.__file__)
