"""The drop-in boundary as the reference scripts use it: `from models import *` through the
shipped shim package, then `eval(cfg.model.type)(**cfg.model)` for every configs/model/*.yaml of the reference whose
`type` names a hot-path backbone.  Construction only (CPU): no compute call is made without a GPU.

The YAML text is read from /root/reference in the build container; the test skips where the reference is absent
(the GPU box).  OmegaConf interpolations are resolved with the values of configs/data/weatherbench.yaml."""
import glob
import importlib
import os
import sys

import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CFG = "/root/reference/src/dlwpbench/configs/model"

HOT_PATH_TYPES = {"FourCastNet", "PanguWeather", "SwinTransformer", "SwinTransformerHPX", "UNet", "UNetHPX", "ModernUNet",
                  "MUNetHPX", "FNO2DModule", "TFNO2DModule", "ConvLSTM", "ConvLSTMHPX",           # SURVEY.md 8b "Registry"
                  "DiffModernUNet", "DiffMUNetHPX"}                                               # 8f row f4 (models/__init__.py:15)
INTERP = {"${data.height}": 32, "${data.width}": 64, "${training.batch_size}": 4, "${device}": "cpu"}


def _shim():
    """`from models import *` with <repo>/shim in front of sys.path."""
    shim = os.path.join(ROOT, "shim")
    old = sys.modules.pop("models", None)
    sys.path.insert(0, shim)
    try:
        mod = importlib.import_module("models")
        assert os.path.dirname(mod.__file__) == os.path.join(shim, "models"), mod.__file__
        ns = {}
        exec("from models import *", ns)
        return ns
    finally:
        sys.path.remove(shim)
        sys.modules.pop("models", None)
        if old is not None:
            sys.modules["models"] = old


def test_shim_exports_every_registry_name():
    ns = _shim()
    missing = HOT_PATH_TYPES - set(ns)
    assert not missing, f"shim/models does not export {sorted(missing)}"
    for name in HOT_PATH_TYPES:
        assert isinstance(ns[name], type) and issubclass(ns[name], torch.nn.Module), name


def _resolve(v):
    if isinstance(v, str) and v in INTERP:
        return INTERP[v]
    if isinstance(v, list):
        return [_resolve(x) for x in v]
    return v


def _reference_yamls():
    return sorted(glob.glob(os.path.join(REF_CFG, "*.yaml")))


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference configs only exist in the build container")
@pytest.mark.parametrize("path", _reference_yamls(), ids=lambda p: os.path.basename(p))
def test_reference_model_yaml_constructs(path):
    cfg = {k: _resolve(v) for k, v in yaml.safe_load(open(path).read()).items()}
    if cfg.get("type") not in HOT_PATH_TYPES:
        pytest.skip(f"type {cfg.get('type')} is out of the hot-path scope (SURVEY.md section 2)")
    ns = _shim()
    th = torch                                                      # noqa: F841  (configs say "th.nn.GELU()")
    model = eval(cfg["type"], dict(ns))(**cfg)                      # scripts/train.py:54
    assert isinstance(model, torch.nn.Module)
    assert sum(p.numel() for p in model.parameters()) > 0           # train.py:56 prints the count
    assert model.eval() is model and model.train() is model         # SURVEY 8b "Module protocol"
    assert model.context_size == cfg.get("context_size", model.context_size)
    sd = model.state_dict()
    model.load_state_dict(sd, strict=True)                          # evaluate.py:148-149


def test_modernunet_healpix_is_munethpx():
    ns = _shim()
    m = ns["ModernUNet"](constant_channels=2, prescribed_channels=1, prognostic_channels=3, hidden_channels=[16, 8],
                         context_size=1, norm=True, mesh="healpix")
    assert type(m) is ns["MUNetHPX"]


def test_fno_rejects_unsupported_kwargs():
    ns = _shim()
    with pytest.raises(NotImplementedError):
        ns["FNO2DModule"](n_modes=[12, 12], max_n_modes=[16, 16])
    ns["FNO2DModule"](n_modes=[12, 12], max_n_modes=[12, 12], bias=False)   # `bias` is swallowed by the reference too


def test_tfno_state_dict_layout_and_dense_reconstruction():
    """TFNO2DModule (fno.py:109-146): Tucker factors in the tensor library's parameter layout; the dense weight the
    kernels consume equals the explicit mode products."""
    ns = _shim()
    m = ns["TFNO2DModule"](n_modes=[8, 6], constant_channels=0, prescribed_channels=0, prognostic_channels=1,
                           hidden_channels=8, lifting_channels=16, projection_channels=16, n_layers=2, rank=0.5,
                           context_size=1)
    keys = set(m.state_dict())
    for l in range(2):
        assert f"fno.fno_blocks.convs.weight.{l}.core" in keys
        for i in range(4):
            assert f"fno.fno_blocks.convs.weight.{l}.factors.factor_{i}" in keys
    t = m.fno.fno_blocks.convs.weight[0]
    assert tuple(t.shape) == (8, 8, 8, 4)
    for p in t.parameters():
        torch.nn.init.normal_(p)
    core = torch.view_as_complex(t.core.detach())
    fs = [torch.view_as_complex(f.detach()) for f in t.factors]
    want = torch.einsum("abcd,ia,jb,kc,ld->ijkl", core, *fs)
    assert torch.allclose(t.dense(), want, atol=1e-5)
    n_dense = 8 * 8 * 8 * 4
    n_fact = core.numel() + sum(f.numel() for f in fs)
    assert n_fact <= 0.6 * n_dense          # rank=0.5: about half the dense parameter count
