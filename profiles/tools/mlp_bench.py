"""Fused MFMA kernels of csrc/pds_mlp.hip vs the PyTorch op chain they replace (1 GPU)."""
import math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phoenix_drone_simulation_amd.fused import FusedMLP
from phoenix_drone_simulation_amd.ppo import _mlp

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for B in (65536, 524288, 2097152, 8388608):
    D, H, A = 34, 50, 4
    net = _mlp([D, H, H, A], "relu").cuda(); fm = FusedMLP(net, "relu")
    x = torch.randn(B, D, device="cuda"); act = torch.randn(B, A, device="cuda"); adv = torch.randn(B, device="cuda")
    lp = torch.randn(B, device="cuda") - 4; ls = torch.full((A,), math.log(0.3), device="cuda")
    t_f = timeit(lambda: fm.ppo_grad(x, act, adv, lp, ls, 0.2))
    def torch_iter():
        for p in fm.params: p.grad = None
        d = torch.distributions.Normal(net(x), torch.exp(ls))
        r = torch.exp(d.log_prob(act).sum(-1) - lp)
        (-(torch.min(r * adv, adv * torch.clamp(r, 0.8, 1.2))).mean()).backward()
    t_t = timeit(torch_iter, 5)
    t_fw = timeit(lambda: fm.forward(x))
    with torch.no_grad(): t_tw = timeit(lambda: net(x), 5)
    # flops: fwd 2*(D*H + H*H + H*A), bwd dX 2*(A*H + H*H), dW 2*(A*H + H*H + H*D) per sample
    fl = 2 * (D * H + H * H + H * A) + 2 * (A * H + H * H) + 2 * (A * H + H * H + H * D)
    print(f"B {B:8d}: ppo_grad fused {t_f*1e3:9.1f} us ({B*fl/t_f/1e9:6.2f} TFLOP/s useful)  torch autograd {t_t*1e3:9.1f} us  x{t_t/t_f:5.1f} | forward fused {t_fw*1e3:8.1f} us  torch {t_tw*1e3:8.1f} us  x{t_tw/t_fw:4.1f}")
