"""csrc/pds_mlp.hip (the trainer's fused MFMA kernels) against plain PyTorch fp32 / autograd."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(d_in, h1, h2, d_out, act, seed):
    from phoenix_drone_simulation_amd.ppo import _mlp
    torch.manual_seed(seed)
    return _mlp([d_in, h1, h2, d_out], act).cuda()


@pytest.mark.parametrize("d_in,h1,h2,d_out,act,B", [
    (34, 50, 50, 4, "relu", 1000), (40, 50, 50, 4, "relu", 32), (42, 64, 64, 1, "tanh", 4097),
    (48, 50, 50, 4, "tanh", 77), (17, 33, 7, 8, "relu", 1), (64, 64, 64, 1, "tanh", 300),
])
def test_forward_matches_torch(d_in, h1, h2, d_out, act, B):
    from phoenix_drone_simulation_amd.fused import FusedMLP
    net = _net(d_in, h1, h2, d_out, act, 1)
    fm = FusedMLP(net, act)
    x = torch.randn(B, d_in, device="cuda") * 2
    mean, std = torch.randn(d_in, device="cuda"), torch.rand(d_in, device="cuda") + 0.5
    with torch.no_grad():
        ref = net(x)
        ref_s = net((x - mean) / (std + 1e-5))
    assert torch.allclose(fm.forward(x), ref, rtol=1e-5, atol=2e-6)
    assert torch.allclose(fm.forward(x, mean=mean, std=std, eps=1e-5), ref_s, rtol=1e-5, atol=5e-6)
    idx = torch.randint(0, B, (max(B // 2, 1),), device="cuda")
    assert torch.allclose(fm.forward(x, index=idx), ref[idx], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("d_in,h,act,B", [(34, 50, "relu", 5000), (40, 50, "relu", 31), (42, 64, "tanh", 2048 + 5)])
def test_ppo_policy_grad_matches_autograd(d_in, h, act, B):
    from phoenix_drone_simulation_amd.fused import FusedMLP
    A, clip = 4, 0.2
    net = _net(d_in, h, h, A, act, 2)
    fm = FusedMLP(net, act)
    torch.manual_seed(5)
    x = torch.randn(B, d_in, device="cuda")
    log_std = torch.full((A,), math.log(0.3), device="cuda") + 0.1 * torch.randn(A, device="cuda")
    with torch.no_grad():
        mu0 = net(x)
        act_t = mu0 + torch.exp(log_std) * torch.randn(B, A, device="cuda")
        logp_old = torch.distributions.Normal(mu0, torch.exp(log_std)).log_prob(act_t).sum(-1)
        logp_old = logp_old + 0.3 * torch.randn(B, device="cuda")  # ratios on both sides of the clip range
    adv = torch.randn(B, device="cuda")
    stats = fm.ppo_grad(x, act_t, adv, logp_old, log_std, clip).clone()
    got = fm.flat_grad.clone()
    # reference: compute_loss_pi (algs/ppo/ppo.py:22-40) through autograd
    for p in fm.params:
        p.grad = None
    d = torch.distributions.Normal(net(x), torch.exp(log_std))
    ratio = torch.exp(d.log_prob(act_t).sum(-1) - logp_old)
    loss = -(torch.min(ratio * adv, adv * torch.clamp(ratio, 1 - clip, 1 + clip))).mean()
    loss.backward()
    want = torch.cat([p.grad.reshape(-1) for p in fm.params])
    scale = float(want.abs().max())
    assert torch.allclose(got, want, rtol=2e-4, atol=2e-6 * max(scale, 1.0)), float((got - want).abs().max())
    assert abs(float(stats[0]) / B - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    assert abs(float(stats[1]) / B - float(ratio.mean())) < 1e-5 * float(ratio.mean())
    kl = (0.5 * (d.mean - act_t) ** 2 / d.stddev ** 2).mean()
    assert abs(float(stats[2]) / (B * A) - float(kl)) < 1e-5 * float(kl)
    assert float(stats[3]) == B


@pytest.mark.parametrize("d_in,B,use_index", [(34, 3000, False), (42, 4096, True), (40, 17, True)])
def test_value_grad_matches_autograd(d_in, B, use_index):
    from phoenix_drone_simulation_amd.fused import FusedMLP
    net = _net(d_in, 64, 64, 1, "tanh", 3)
    fm = FusedMLP(net, "tanh")
    x = torch.randn(B, d_in, device="cuda")
    target = torch.randn(B, device="cuda")
    idx = torch.randperm(B, device="cuda")[: max(B // 4, 1)] if use_index else None
    stats = fm.value_grad(x, target, idx).clone()
    got = fm.flat_grad.clone()
    for p in fm.params:
        p.grad = None
    xs, ts = (x[idx], target[idx]) if use_index else (x, target)
    loss = torch.nn.functional.mse_loss(net(xs).squeeze(-1), ts)
    loss.backward()
    want = torch.cat([p.grad.reshape(-1) for p in fm.params])
    assert torch.allclose(got, want, rtol=2e-4, atol=2e-6 * max(float(want.abs().max()), 1.0)), float((got - want).abs().max())
    n = xs.shape[0]
    assert abs(float(stats[0]) / n - float(loss)) < 1e-5 * max(1.0, float(loss))


def test_gradients_are_deterministic_and_size_limits_are_checked():
    from phoenix_drone_simulation_amd.fused import FusedMLP
    net = _net(34, 50, 50, 4, "relu", 4)
    fm = FusedMLP(net, "relu")
    B = 100000
    x = torch.randn(B, 34, device="cuda"); a = torch.randn(B, 4, device="cuda")
    adv = torch.randn(B, device="cuda"); lp = torch.randn(B, device="cuda") - 5; ls = torch.zeros(4, device="cuda")
    fm.ppo_grad(x, a, adv, lp, ls, 0.2); g1 = fm.flat_grad.clone()
    fm.ppo_grad(x, a, adv, lp, ls, 0.2); g2 = fm.flat_grad.clone()
    assert torch.equal(g1, g2)
    with pytest.raises((ValueError, NotImplementedError)):
        FusedMLP(_net(70, 50, 50, 4, "relu", 0), "relu")
