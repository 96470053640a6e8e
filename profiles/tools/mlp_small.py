"""Fixed cost of the trainer's MLP calls: small batches, where the prologue / epilogue of the kernels and the launches
dominate (profiles/tools/mlp_small.py): policy gradient, value gradient on a gathered mini-batch, both forwards."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phoenix_drone_simulation_amd.fused import FusedMLP
from phoenix_drone_simulation_amd.ppo import _mlp
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
D, A = 34, 4
pi = _mlp([D, 50, 50, A], "relu").cuda(); fpi = FusedMLP(pi, "relu")
vf = _mlp([D, 64, 64, 1], "tanh").cuda(); fvf = FusedMLP(vf, "tanh")
for B in (16, 4096, 32768, 65536):
    N = 16 * B
    x = torch.randn(N, D, device="cuda"); act = torch.randn(B, A, device="cuda"); adv = torch.randn(B, device="cuda")
    lp = torch.randn(B, device="cuda") - 4; ls = torch.full((A,), math.log(0.3), device="cuda")
    tgt = torch.randn(N, device="cuda"); idx = torch.randperm(N, device="cuda")[:B]
    xb = x[:B].contiguous()
    print(f"B {B:6d}: policy grad {timeit(lambda: fpi.ppo_grad(xb, act, adv, lp, ls, 0.2)):6.1f} us   value grad (gathered) {timeit(lambda: fvf.value_grad(x, tgt, idx)):6.1f} us"
          f"   policy forward {timeit(lambda: fpi.forward(xb)):6.1f} us   value forward {timeit(lambda: fvf.forward(xb)):6.1f} us")
