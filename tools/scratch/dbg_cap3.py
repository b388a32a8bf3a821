import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from helpers import fno_std_fn
from dlwp_benchmark_amd.models import FNO2DModule
from dlwp_benchmark_amd.weights import fill_state_dict
from dlwp_benchmark_amd.sharding import CapturedStep, ShardedRollout
from dlwp_benchmark_amd.synthetic import navier_stokes
DEV = torch.device("cuda:0")
NS_KW = dict(n_modes=[12, 12], constant_channels=0, prescribed_channels=0, prognostic_channels=1,
             hidden_channels=32, lifting_channels=256, projection_channels=256, n_layers=4, context_size=1)
for B in (4, 16, 32):
    m = FNO2DModule(**NS_KW)
    fill_state_dict(m, std_fn=fno_std_fn(0.85), gain=0.85)
    m = m.to(DEV).eval()
    m.set_execution_form(check="deferred")
    runner = ShardedRollout(m, gather=False)
    _, _, prog = navier_stokes(B, 21, 64, 64, seed=1234)
    prog = prog.to(DEV)
    want = runner(constants=None, prescribed=None, prognostic=prog).clone()
    again = runner(constants=None, prescribed=None, prognostic=prog).clone()
    cap = CapturedStep(lambda c, p, g: runner(constants=c, prescribed=p, prognostic=g), model=m)
    res = [torch.equal(want, again)]
    for i in range(6):
        got = cap(None, None, prog)
        torch.cuda.synchronize()
        d = (got - want).abs()
        res.append((torch.equal(got, want), float(d.max())))
    m.verify()
    print(B, res, m.fused_timeouts(), m.range_reruns())
