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
    # round 6: more than 64 inputs (observation_history_size >= 4): the K-tiled first layer of csrc/pds_mlp_wide.hip, one
    # shape per input-tile count it is built for (<= 96 / 128 / 160 / 192) and the widths the three tasks produce
    (68, 50, 50, 4, "relu", 1000), (80, 32, 32, 4, "relu", 33), (96, 64, 64, 1, "tanh", 4097), (102, 48, 48, 4, "relu", 500),
    (136, 64, 64, 1, "tanh", 777), (160, 64, 64, 4, "relu", 2049), (192, 64, 64, 1, "tanh", 300), (65, 50, 50, 4, "relu", 17),
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


@pytest.mark.parametrize("d_in,h,act,B", [(34, 50, "relu", 5000), (40, 50, "relu", 31), (42, 64, "tanh", 2048 + 5),
                                          (48, 50, "relu", 777), (64, 33, "tanh", 1500), (17, 16, "relu", 100),
                                          # round 6: the K-tiled kernels (64 < d_in <= 192)
                                          (68, 50, "relu", 5000), (80, 32, "relu", 31), (136, 64, "relu", 2048 + 5),
                                          (160, 48, "tanh", 777), (192, 64, "relu", 1500), (120, 64, "relu", 70001),
                                          # round 6: batches of the size at which ppo_split_kernel runs its weight-gradient role on
                                          # split-bf16 MFMAs (>= 65 536 samples; both input-step forms, a ragged last tile)
                                          (34, 50, "relu", 200003), (42, 50, "relu", 163840), (48, 50, "relu", 300017)])
def test_ppo_policy_grad_matches_autograd(d_in, h, act, B):
    from phoenix_drone_simulation_amd.fused import FusedMLP
    A, clip = (6 if d_in == 17 else 4), 0.2  # 6 outputs: the log-prob sum spans two lane groups
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


def _ppo_case_inputs(d_in, h, A, act, B, seed):
    net = _net(d_in, h, h, A, act, seed)
    torch.manual_seed(5 + B % 7)
    x = torch.randn(B, d_in, device="cuda")
    log_std = torch.full((A,), math.log(0.3), device="cuda") + 0.1 * torch.randn(A, device="cuda")
    with torch.no_grad():
        mu0 = net(x)
        act_t = mu0 + torch.exp(log_std) * torch.randn(B, A, device="cuda")
        logp_old = torch.distributions.Normal(mu0, torch.exp(log_std)).log_prob(act_t).sum(-1)
        logp_old = logp_old + 0.3 * torch.randn(B, device="cuda")
    return net, x, act_t, torch.randn(B, device="cuda"), logp_old, log_std


def _ppo_grad_case(d_in, h, A, act, B, seed=2):
    """fused PPO-clip gradient and statistics of a d_in-h-h-A net against compute_loss_pi (algs/ppo/ppo.py:22-40) through
    autograd in FLOAT64: the loss has kinks (clip range, relu), and a sample that sits on one within f32 rounding takes
    the other branch in an f32 reference as easily as in the kernel -- measured: torch's own f32 gradient is 6e-6 off
    the f64 one at B = 70 001 where the kernel is 9e-9 off.  Such a sample moves an element by (its gradient) / B,
    which the tolerance of the large batches allows for; dropped or repeated tiles are the additivity test's job."""
    import copy
    from phoenix_drone_simulation_amd.fused import FusedMLP
    clip = 0.2
    net, x, act_t, adv, logp_old, log_std = _ppo_case_inputs(d_in, h, A, act, B, seed)
    fm = FusedMLP(net, act)
    stats = fm.ppo_grad(x, act_t, adv, logp_old, log_std, clip).clone()
    got = fm.flat_grad.clone().double()
    net64 = copy.deepcopy(net).double()
    d = torch.distributions.Normal(net64(x.double()), torch.exp(log_std.double()))
    ratio = torch.exp(d.log_prob(act_t.double()).sum(-1) - logp_old.double())
    loss = -(torch.min(ratio * adv.double(), adv.double() * torch.clamp(ratio, 1 - clip, 1 + clip))).mean()
    loss.backward()
    want = torch.cat([p.grad.reshape(-1) for p in net64.parameters()])
    scale = float(want.abs().max())
    atol = 2e-6 * max(scale, 1.0) + (10.0 / B if B > 20000 else 0.0)
    assert torch.allclose(got, want, rtol=2e-4, atol=atol), float((got - want).abs().max())
    assert abs(float(stats[0]) / B - float(loss)) < 2e-5 * max(1.0, abs(float(loss)))
    assert abs(float(stats[1]) / B - float(ratio.mean())) < 2e-5 * float(ratio.mean())
    assert float(stats[3]) == B


@pytest.mark.parametrize("d_in,B,cut", [(34, 300007, 123457), (42, 70001, 16 * 1000), (42, 1000003, 500001)])
def test_wave_role_policy_gradient_is_additive_over_batch_splits(d_in, B, cut):
    """every tile is handed from a forward wave to its weight-gradient wave exactly once: the gradient of a batch equals
    the sample-weighted sum of the gradients of its two parts up to the rounding of the partial sums (a sample's own
    arithmetic does not depend on the batch around it, so the loss's kinks cancel) -- a dropped, repeated or half-written
    tile of 16 samples shows up 16 / B ~ 1e-5..2e-4 relative; the per-sample count is exact"""
    from phoenix_drone_simulation_amd.fused import FusedMLP
    net, x, act_t, adv, logp_old, log_std = _ppo_case_inputs(d_in, 50, 4, "relu", B, 11)
    fm = FusedMLP(net, "relu")

    def grad(lo, hi):
        st = fm.ppo_grad(x[lo:hi].contiguous(), act_t[lo:hi].contiguous(), adv[lo:hi].contiguous(),
                         logp_old[lo:hi].contiguous(), log_std, 0.2).clone().double()
        assert float(st[3]) == hi - lo
        return fm.flat_grad.clone().double() * (hi - lo), st
    full, st_full = grad(0, B)
    a, st_a = grad(0, cut)
    b, st_b = grad(cut, B)
    err = float((full - (a + b)).abs().max()) / float(full.abs().max())
    assert err < 2e-6, err
    assert torch.allclose(st_full[:3], st_a[:3] + st_b[:3], rtol=1e-5)


@pytest.mark.parametrize("B", [1, 15, 16, 17, 64 * 3 + 1, 16 * 1024 * 3, 70001, 300007])
def test_wave_role_policy_gradient_over_batch_shapes(B):
    """ppo_split_kernel (round 3; default policy 50-50 relu): a batch smaller than one tile, ragged tails, fewer tiles than
    wave pairs, exactly 3 tiles per pair of every block, and batches where every pair reuses its two LDS image sets
    several times (the full / empty hand-over of the two roles)"""
    _ppo_grad_case(34, 50, 4, "relu", B)
    _ppo_grad_case(42, 50, 4, "relu", B, seed=3)


@pytest.mark.parametrize("d_in,A", [(33, 4), (47, 4), (20, 4), (3, 2), (42, 1), (42, 6), (34, 8), (40, 5)])
def test_wave_role_policy_gradient_over_net_shapes(d_in, A):
    """input widths on either side of the 34-input specialisation, the widest one the kernel takes (47: the ones column
    of dW1 is the last column of the third input tile), tiny inputs; action dimensions 1..8 (> 4: the second k-step of
    the output-layer backward GEMM, log-prob sums across lane groups)"""
    _ppo_grad_case(d_in, 50, A, "relu", 5000 + d_in)
    _ppo_grad_case(d_in, 50, A, "relu", 40000 + A, seed=7)


@pytest.mark.parametrize("d_in,B,use_index", [(34, 3000, False), (42, 4096, True), (40, 17, True),
                                              (68, 3000, False), (80, 4096, True), (136, 17, True), (160, 40000, True),
                                              (192, 5000, False)])
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
    # size limits: d_in <= 192 (round 6: the K-tiled kernels; 64 before), hidden <= 64, d_out <= 8
    FusedMLP(_net(192, 64, 64, 8, "relu", 0), "relu")
    for shape in ((193, 50, 50, 4), (34, 65, 50, 4), (34, 50, 65, 4), (34, 50, 50, 9)):
        with pytest.raises((ValueError, NotImplementedError)):
            FusedMLP(_net(*shape, "relu", 0), "relu")
    # the wide kernels are deterministic too (one partial per wave, summed in a fixed order)
    netw = _net(136, 64, 64, 4, "relu", 4)
    fw = FusedMLP(netw, "relu")
    xw = torch.randn(B, 136, device="cuda")
    fw.ppo_grad(xw, a, adv, lp, ls, 0.2); g1 = fw.flat_grad.clone()
    fw.ppo_grad(xw, a, adv, lp, ls, 0.2); g2 = fw.flat_grad.clone()
    assert torch.equal(g1, g2)


def test_adam_step_matches_torch_adam():
    from phoenix_drone_simulation_amd.fused import FusedMLP
    net = _net(34, 50, 50, 4, "relu", 7)
    ref = _net(34, 50, 50, 4, "relu", 7)
    fm = FusedMLP(net, "relu")
    opt = torch.optim.Adam(ref.parameters(), lr=3e-4)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for k in range(25):
        flat = torch.randn(fm.flat_grad.numel(), device="cuda", generator=g) * (0.1 if k % 3 else 10.0)
        fm.flat_grad.copy_(flat)
        off = 0
        for p in ref.parameters():
            p.grad = flat[off:off + p.numel()].view_as(p).clone(); off += p.numel()
        lr = 3e-4 * (1 - k / 50)
        for gr in opt.param_groups:
            gr["lr"] = lr
        opt.step(); fm.adam_step(lr)
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), float((a - b).abs().max())


def test_gaussian_sample_and_rollout_record():
    from phoenix_drone_simulation_amd.fused import gaussian_sample, rollout_record
    n, d = 200000, 4
    mu = torch.randn(n, d, device="cuda")
    log_std = torch.tensor([-1.0, -0.5, 0.0, 0.3], device="cuda")
    a, lp = torch.empty(n, d, device="cuda"), torch.empty(n, device="cuda")
    gaussian_sample(mu, log_std, a, lp, seed=3, call=1)
    z = (a - mu) / torch.exp(log_std)
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1) < 0.01
    assert abs(float((z ** 3).mean())) < 0.03 and abs(float((z ** 4).mean()) - 3) < 0.06
    c = torch.corrcoef(z.T)
    assert float((c - torch.eye(d, device="cuda")).abs().max()) < 0.01
    want = torch.distributions.Normal(mu, torch.exp(log_std)).log_prob(a).sum(-1)
    assert torch.allclose(lp, want, atol=1e-4)
    a2, lp2 = torch.empty_like(a), torch.empty_like(lp)
    gaussian_sample(mu, log_std, a2, lp2, seed=3, call=1)
    assert torch.equal(a, a2)                         # counter-based: reproducible
    gaussian_sample(mu, log_std, a2, lp2, seed=3, call=2)
    assert not torch.equal(a, a2)
    gaussian_sample(mu[100:], log_std, a2[100:], lp2[100:], seed=3, call=1, id_base=100)
    assert torch.equal(a[100:], a2[100:])             # keyed by the global sample id (sharding)
    gaussian_sample(mu, log_std, a2, lp2, seed=3, call=5, deterministic=True)
    assert torch.equal(a2, mu)
    # bookkeeping
    T = 3
    rew_buf = torch.zeros(T, n, device="cuda"); term_buf = torch.zeros(T, n, dtype=torch.uint8, device="cuda")
    trunc_buf = torch.zeros_like(term_buf)
    ep_ret, ep_len, stats = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(3, device="cuda")
    er, el, st = ep_ret.clone(), ep_len.clone(), torch.zeros(3, device="cuda", dtype=torch.float64)
    for t in range(T):
        r = torch.randn(n, device="cuda")
        te = (torch.rand(n, device="cuda") < 0.1).to(torch.uint8); tr = (torch.rand(n, device="cuda") < 0.05).to(torch.uint8)
        rollout_record(r, te, tr, rew_buf[t], term_buf[t], trunc_buf[t], ep_ret, ep_len, stats)
        er += r; el += 1
        done = (te | tr).bool()
        st += torch.stack([(er * done).sum().double(), (el * done).sum().double(), done.sum().double()])
        er = torch.where(done, torch.zeros_like(er), er); el = torch.where(done, torch.zeros_like(el), el)
        assert torch.equal(rew_buf[t], r) and torch.equal(term_buf[t], te) and torch.equal(trunc_buf[t], tr)
    assert torch.allclose(ep_ret, er) and torch.equal(ep_len, el)
    assert torch.allclose(stats.double(), st, rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 1000, 4096, 65536, 524288 + 3])
def test_permutation_is_a_permutation_and_keyed(n):
    """pds_permutation (csrc/pds_train.hip): the value net's mini-batch shuffle in one launch -- every index exactly once
    for domain sizes on both sides of a power of two, reproducible per (seed, call), different between calls"""
    from phoenix_drone_simulation_amd.fused import random_permutation
    p = random_permutation(n, 7, 1, "cuda")
    assert p.dtype == torch.int64 and p.shape == (n,)
    assert torch.equal(torch.sort(p).values, torch.arange(n, device="cuda"))
    assert torch.equal(p, random_permutation(n, 7, 1, "cuda"))
    if n >= 17:
        assert not torch.equal(p, random_permutation(n, 7, 2, "cuda"))
        assert not torch.equal(p, random_permutation(n, 8, 1, "cuda"))


def test_permutation_statistics():
    """over 400 keys: where element 0 goes is uniform over the 50 positions (chi-square), neighbours do not stay
    neighbours, and the mean displacement is that of a uniform shuffle, n / 3"""
    from phoenix_drone_simulation_amd.fused import random_permutation
    n, calls = 50, 400
    ps = torch.stack([random_permutation(n, 3, c, "cuda") for c in range(calls)]).cpu()
    counts = torch.bincount(ps[:, 0], minlength=n).double()
    chi2 = float(((counts - calls / n) ** 2 / (calls / n)).sum())
    assert chi2 < 100.0, chi2            # 49 degrees of freedom: mean 49, P(> 100) ~ 2e-5
    adjacent = float(((ps[:, 1:] - ps[:, :-1]).abs() == 1).double().mean())
    assert adjacent < 0.08, adjacent     # uniform shuffle: 2 / n = 0.04
    big = random_permutation(1 << 18, 5, 9, "cuda").double()
    disp = float((big - torch.arange(1 << 18, device="cuda").double()).abs().mean()) / (1 << 18)
    assert abs(disp - 1.0 / 3.0) < 0.01, disp


@pytest.mark.parametrize("which", ["policy", "value"])
def test_gradient_call_with_adam_step_equals_gradient_then_adam_step_bitwise(which):
    """pds_*_grad_step (round 3): Adam applied by the partial-sum kernel -- parameters, both moment buffers and the
    gradient after 4 steps equal, bit for bit, the route through pds_adam_step (which test_adam_... pins to torch)"""
    import copy
    from phoenix_drone_simulation_amd.fused import FusedMLP
    B = 20000
    if which == "policy":
        net_a = _net(34, 50, 50, 4, "relu", 21)
    else:
        net_a = _net(34, 64, 64, 1, "tanh", 22)
    net_b = copy.deepcopy(net_a)
    act = "relu" if which == "policy" else "tanh"
    fa, fb = FusedMLP(net_a, act), FusedMLP(net_b, act)
    torch.manual_seed(3)
    x = torch.randn(B, 34, device="cuda")
    a4 = torch.randn(B, 4, device="cuda"); adv = torch.randn(B, device="cuda"); lp = torch.randn(B, device="cuda") - 4
    ls = torch.full((4,), math.log(0.4), device="cuda"); tgt = torch.randn(B, device="cuda")
    idx = torch.randperm(B, device="cuda")[:5000]
    for it in range(4):
        lr = 3e-4 * (it + 1)
        if which == "policy":
            fa.ppo_grad(x, a4, adv, lp, ls, 0.2); fa.adam_step(lr)
            fb.ppo_grad(x, a4, adv, lp, ls, 0.2, adam_lr=lr)
        else:
            fa.value_grad(x, tgt, idx); fa.adam_step(lr)
            fb.value_grad(x, tgt, idx, adam_lr=lr)
        assert torch.equal(fa.flat_grad, fb.flat_grad), it
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):
        assert torch.equal(pa, pb)
    assert torch.equal(fa.exp_avg, fb.exp_avg) and torch.equal(fa.exp_avg_sq, fb.exp_avg_sq)
    assert fa.adam_steps == fb.adam_steps == 4
    assert not torch.equal(next(iter(net_a.parameters())), next(iter(_net(34, 50 if which == "policy" else 64, 50 if which == "policy" else 64, 4 if which == "policy" else 1, act, 21 if which == "policy" else 22).parameters())))


def test_round3_entry_points_reject_bad_arguments():
    """pds_permutation / pds_*_grad_step through the C ABI: PDS_EINVAL (not a launch) for an empty range, a missing
    output, an optimiser block without state or with step 0"""
    import ctypes as C
    from phoenix_drone_simulation_amd import native
    from phoenix_drone_simulation_amd.fused import FusedMLP, _ptr
    lib = native.load()
    out = torch.empty(8, dtype=torch.int64, device="cuda")
    assert lib.pds_permutation(_ptr(out), 0, 1, 1, None) == native.EINVAL
    assert lib.pds_permutation(None, 8, 1, 1, None) == native.EINVAL
    assert lib.pds_permutation(_ptr(out), 8, 1, 1, None) == native.OK
    net = _net(34, 50, 50, 4, "relu", 4)
    fm = FusedMLP(net, "relu"); fm._bind(); fm._adam_state()
    B = 64
    x = torch.randn(B, 34, device="cuda"); a = torch.randn(B, 4, device="cuda")
    adv = torch.randn(B, device="cuda"); lp = torch.randn(B, device="cuda") - 5; ls = torch.zeros(4, device="cuda")
    before = [p.detach().clone() for p in net.parameters()]

    def call(opt):
        return lib.pds_ppo_policy_grad_step(C.byref(fm.m), _ptr(x), _ptr(a), _ptr(adv), _ptr(lp), _ptr(ls), B, 0.2,
                                            _ptr(fm.flat_grad), _ptr(fm.stats), _ptr(fm.workspace), opt, None)
    bad_step = native.Adam(_ptr(fm.exp_avg), _ptr(fm.exp_avg_sq), 0, 1e-3, 0.9, 0.999, 1e-8)
    no_state = native.Adam(None, _ptr(fm.exp_avg_sq), 1, 1e-3, 0.9, 0.999, 1e-8)
    assert call(C.byref(bad_step)) == native.EINVAL
    assert call(C.byref(no_state)) == native.EINVAL
    torch.cuda.synchronize()
    for p, q in zip(net.parameters(), before):
        assert torch.equal(p, q)  # nothing was launched
    assert call(None) == native.OK


def _feistel_permutation_numpy(n, seed, call):
    """the algorithm include/pds.h documents for pds_permutation, restated with numpy on the oracle's Philox4x32-10:
    6 Feistel rounds (murmur3 finaliser of the keyed right half) over 2^(2 * half_bits) >= n, cycle-walked into [0, n)"""
    import numpy as np
    from oracle import oracle as po
    bits = 1
    while bits < 63 and (1 << bits) < n:
        bits += 1
    half = (bits + 1) // 2
    mask = (1 << half) - 1
    k0 = po.philox4x32_10((0, 0x7065726D, call & 0xFFFFFFFF, call >> 32), (seed & 0xFFFFFFFF, seed >> 32))
    k1 = po.philox4x32_10((1, 0x7065726D, call & 0xFFFFFFFF, call >> 32), (seed & 0xFFFFFFFF, seed >> 32))
    keys = [int(v) for v in list(k0) + list(k1)[:2]]

    def f(r, key):
        h = (r * np.uint64(0x9E3779B1) + np.uint64(key)) & np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(16)
        return h

    def once(x):
        L, R = x >> np.uint64(half), x & np.uint64(mask)
        for key in keys:
            L, R = R, L ^ (f(R, key) & np.uint64(mask))
        return (L << np.uint64(half)) | R
    x = once(np.arange(n, dtype=np.uint64))
    while True:
        out = x >= np.uint64(n)
        if not out.any():
            return x.astype(np.int64)
        x[out] = once(x[out])


@pytest.mark.parametrize("n,seed,call", [(1000, 7, 1), (4097, 2 ** 40 + 5, 3), (65536, 0, 2 ** 33)])
def test_permutation_equals_its_documented_algorithm(n, seed, call):
    import numpy as np
    from phoenix_drone_simulation_amd.fused import random_permutation
    got = random_permutation(n, seed, call, "cuda").cpu().numpy()
    assert np.array_equal(got, _feistel_permutation_numpy(n, seed, call))


_FORMS_CHILD = r"""
import math, sys, torch
sys.path.insert(0, sys.argv[1])
from phoenix_drone_simulation_amd.fused import FusedMLP
from phoenix_drone_simulation_amd.ppo import _mlp
B, D, H, A = 262144, 34, 50, 4
torch.manual_seed(3)
net = _mlp([D, H, H, A], "relu").cuda(); fm = FusedMLP(net, "relu")
x = torch.randn(B, D, device="cuda"); log_std = torch.full((A,), math.log(0.3), device="cuda")
with torch.no_grad():
    mu0 = net(x); act = mu0 + torch.exp(log_std) * torch.randn(B, A, device="cuda")
    lp = torch.distributions.Normal(mu0, torch.exp(log_std)).log_prob(act).sum(-1) + 0.3 * torch.randn(B, device="cuda")
adv = torch.randn(B, device="cuda")
fm.ppo_grad(x, act, adv, lp, log_std, 50.0); got = fm.flat_grad.double().clone()  # (clip range never reached)
net64 = _mlp([D, H, H, A], "relu").cuda().double(); net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
d = torch.distributions.Normal(net64(x.double()), torch.exp(log_std.double()))
r = torch.exp(d.log_prob(act.double()).sum(-1) - lp.double())
(-(torch.min(r * adv.double(), adv.double() * torch.clamp(r, -49.0, 51.0))).mean()).backward()
want = torch.cat([p.grad.reshape(-1) for p in net64.parameters()])
torch.save(got.cpu(), sys.argv[2])
print("ERR", float(want.abs().max()), float(((got - want) ** 2).sum().sqrt() / (want ** 2).sum().sqrt()))
"""


def test_split_bf16_weight_gradient_role_is_no_less_accurate_than_the_f32_form(tmp_path):
    """ppo_split_kernel's weight-gradient role (dZ1, dW2, dW1) runs on v_mfma_f32_16x16x32_bf16 with every operand in three bf16
    pieces (six products: exact, one f32 rounding per 32 terms) from 65 536 samples on -- and layers 1 and 2 of the forward role with it --, on v_mfma_f32_16x16x4_f32 below
    (csrc/pds_mlp.hip, PDS_SPLIT_BF16).  The same 262 144-sample policy gradient through both forms (PDS_BF16_MIN_SAMPLES, read
    once per process: two child processes) against float64 autograd.  Measured: relative L2 error 9.4e-4 for the f32 form --
    two or three of the 13 M layer-1 relu units take the other branch than in float64, because v_mfma_f32_16x16x4_f32 leaves
    ~1e-7 on a pre-activation, and each such sample moves the gradient by O(1 / B) -- and 3.7e-7 for the bf16 form, whose
    layer 1 is accurate enough to flip none on this batch (the count is a Poisson number: the bars do not rely on zero).  Bars:
    both forms within 2e-3 of float64, the bf16 form not worse than 1.02 x the f32 form, and the two results differ (they are
    different kernels) by no more than 2e-3 of the largest entry."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    errs, grads = {}, {}
    for name, thr in (("bf16", "0"), ("f32", "99999999999")):
        out = str(tmp_path / (name + ".pt"))
        r = subprocess.run([sys.executable, "-c", _FORMS_CHILD, root, out], env=dict(os.environ, PDS_BF16_MIN_SAMPLES=thr), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("ERR")][0].split()
        errs[name] = (float(line[1]), float(line[2]))
        grads[name] = torch.load(out)
    (scale, rb), (_, rf) = errs["bf16"], errs["f32"]
    diff = float((grads["bf16"] - grads["f32"]).abs().max())
    assert 0.0 < diff <= 2e-3 * scale, (diff, scale, errs)
    assert rb <= 2e-3 and rf <= 2e-3 and rb <= rf * 1.02, errs
