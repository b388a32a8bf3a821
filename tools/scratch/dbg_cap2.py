import json, sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import dlwp_benchmark_amd.models as M
from dlwp_benchmark_amd.sharding import CapturedStep, ShardedRollout
from dlwp_benchmark_amd.weights import fill_by_spec
from helpers import load_golden
from oracle.make_golden import MODEL_CASES, model_inputs
DEV = torch.device("cuda:0")
tag = "swin_e32_32x64"
family, cfg, (batch, frames), gain = MODEL_CASES[tag]
g = load_golden(f"model_{tag}")
sd, _ = fill_by_spec(json.loads(str(g["param_spec"])), gain=gain)
c, p, x = model_inputs(tag, cfg, batch, frames)
c, p, x = (t.to(DEV) if t is not None else None for t in (c, p, x))
for trial in range(6):
    model = M.SwinTransformer(**cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(DEV).eval()
    runner = ShardedRollout(model, gather=False)
    want = runner(constants=c, prescribed=p, prognostic=x).clone()
    cap = CapturedStep(lambda cc, pp, gg: runner(constants=cc, prescribed=pp, prognostic=gg), model=model)
    res = []
    for i in range(4):
        got = cap(c, p, x)
        torch.cuda.synchronize()
        eq = torch.equal(got, want)
        d = (got - want).abs()
        res.append((eq, float(d.max()), float((d > 0).float().mean()), bool(torch.isnan(got).any())))
    print(trial, res)
