"""PPO policy-gradient call (gradient kernel + partial-sum kernel) of the default 50-50 relu policy over batch sizes,
for 34 and 42 inputs (profiles/tools/mlp_sizes.py); PDS_LIB selects an A/B build of the library."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phoenix_drone_simulation_amd.fused import FusedMLP
from phoenix_drone_simulation_amd.ppo import _mlp
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for D in (34, 42):
  for B in (16, 2048, 32768, 65536, 131072, 262144, 524288, 1048576):
    H, A = 50, 4
    net = _mlp([D, H, H, A], "relu").cuda(); fm = FusedMLP(net, "relu")
    x = torch.randn(B, D, device="cuda"); act = torch.randn(B, A, device="cuda"); adv = torch.randn(B, device="cuda")
    lp = torch.randn(B, device="cuda") - 4; ls = torch.full((A,), math.log(0.3), device="cuda")
    t_f = timeit(lambda: fm.ppo_grad(x, act, adv, lp, ls, 0.2))
    print(f"D {D} B {B:8d}: ppo_grad (grad kernel + reduce) {t_f*1e3:9.1f} us")
