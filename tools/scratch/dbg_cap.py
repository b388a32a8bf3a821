import json, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
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
model = M.SwinTransformer(**cfg)
model.load_state_dict(sd, strict=False)
model = model.to(DEV).eval()
c, p, x = (t.to(DEV) if t is not None else None for t in (c, p, x))
runner = ShardedRollout(model, gather=False)
a1 = runner(constants=c, prescribed=p, prognostic=x).clone()
a2 = runner(constants=c, prescribed=p, prognostic=x).clone()
print("eager(graphs default) repeat equal:", torch.equal(a1, a2), getattr(model, "_use_graphs", None))
model.set_step_graphs(False)
b1 = runner(constants=c, prescribed=p, prognostic=x).clone()
b2 = runner(constants=c, prescribed=p, prognostic=x).clone()
print("eager(no step graphs) equal to default:", torch.equal(a1, b1), "repeat:", torch.equal(b1, b2), (a1 - b1).abs().max().item())
cap = CapturedStep(lambda cc, pp, gg: runner(constants=cc, prescribed=pp, prognostic=gg), model=model)
for i in range(4):
    got = cap(c, p, x)
    torch.cuda.synchronize()
    d = (got - b1).abs()
    print(i, "captured equal:", torch.equal(got, b1), d.max().item(), (d > 0).float().mean().item(), [int(v) for v in torch.nonzero(d.flatten(2).amax(2))[:6].flatten()])
