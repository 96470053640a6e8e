"""Host side of csrc/pds_mlp.hip: the trainer's dense work as fused HIP kernels
on the f32 matrix cores -- MLP forward for the rollout, PPO-clip policy gradient and value-regression
gradient for the update.  Each call replaces an autograd op chain of the reference trainer:

    FusedMLP.forward      ActorCritic.step / MLPGaussianActor.net / MLPCritic.net  algs/core.py:228-311,370-393
    FusedMLP.ppo_grad     compute_loss_pi + backward                                algs/ppo/ppo.py:22-40
    FusedMLP.value_grad   compute_loss_v + backward                                 algs/iwpg/iwpg.py:272-275

The gradients are written straight into the `.grad` storage of the torch parameters (one flat
buffer, torch parameter order), so the optimiser step stays torch.optim.Adam like the reference's."""
import ctypes as C

import torch
import torch.nn as nn

from . import native

_ACT = {"relu": 0, "tanh": 1}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL = _NullCtx()


def _on(t):
    """Context that makes t's device current if it is not (a process that holds envs / networks on several
    GPUs must not launch on, or take the stream of, whichever device happens to be current)."""
    return _NULL if t.device.index == torch.cuda.current_device() else torch.cuda.device(t.device)


class FusedMLP:
    """View of `nn.Sequential(Linear, act, Linear, act, Linear[, Identity])` as a struct pds_mlp."""

    def __init__(self, net, activation):
        lin = [l for l in net if isinstance(l, nn.Linear)]
        if len(lin) != 3 or activation not in _ACT:
            raise NotImplementedError("fused kernels cover 2 hidden layers with relu or tanh")
        self.lin = lin
        self.params = [p for l in lin for p in (l.weight, l.bias)]
        for p in self.params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise ValueError("fused kernels need contiguous float32 parameters on the HIP device")
        self.lib = native.load()
        m = native.Mlp()
        m.d_in, m.h1, m.h2, m.d_out = lin[0].in_features, lin[0].out_features, lin[1].out_features, lin[2].out_features
        m.activation = _ACT[activation]
        self.m = m
        self._bind()
        n = self.lib.pds_mlp_param_count(C.byref(m))
        if n < 0:
            raise ValueError("layer sizes outside the fused kernels' range (d_in <= 192, h1, h2 <= 64, d_out <= 8)")
        dev = self.params[0].device
        self.flat_grad = torch.zeros(n, device=dev)
        off = 0
        for p in self.params:  # .grad of every parameter is a view into the flat buffer the kernel fills
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.stats = torch.zeros(4, device=dev)
        self.workspace = torch.empty(self.lib.pds_mlp_workspace_floats(C.byref(m)), device=dev)

    def _bind(self):
        m, l = self.m, self.lin
        m.w1, m.b1, m.w2, m.b2, m.w3, m.b3 = (p.data_ptr() for p in (l[0].weight, l[0].bias, l[1].weight, l[1].bias,
                                                                       l[2].weight, l[2].bias))

    @staticmethod
    def _stream(t=None):
        """The caller's current stream ON THE TENSOR'S DEVICE (the kernels of pds_mlp / pds_train / pds_gae run
        on the current device: `_on(t)` makes that the tensor's one for the duration of the call)."""
        return C.c_void_p(torch.cuda.current_stream(t.device if t is not None else None).cuda_stream)

    def forward(self, x, index=None, mean=None, std=None, eps=1e-5, out=None):
        """y[B, d_out] = net(standardise(x[index]))."""
        self._bind()
        B = x.shape[0] if index is None else index.shape[0]
        y = out if out is not None else torch.empty(B, self.m.d_out, device=x.device)
        with _on(x):
            rc = self.lib.pds_mlp_forward(C.byref(self.m), _ptr(x), _ptr(index), B, _ptr(mean), _ptr(std), float(eps),
                                          _ptr(y), self._stream(x))
        if rc != native.OK:
            raise RuntimeError(f"pds_mlp_forward -> {rc}")
        return y

    def _adam_state(self):
        if not hasattr(self, "exp_avg"):
            self.exp_avg = torch.zeros_like(self.flat_grad)
            self.exp_avg_sq = torch.zeros_like(self.flat_grad)
            self.adam_steps = 0

    def _adam_arg(self, adam_lr, betas, eps):
        """pds_adam for a gradient call that also steps (None: gradient only)"""
        if adam_lr is None:
            return None
        self._adam_state()
        self.adam_steps += 1
        return C.byref(native.Adam(_ptr(self.exp_avg), _ptr(self.exp_avg_sq), self.adam_steps, float(adam_lr),
                                   float(betas[0]), float(betas[1]), float(eps)))

    def ppo_grad(self, x, act, adv, logp_old, log_std, clip_ratio, adam_lr=None, betas=(0.9, 0.999), eps=1e-8):
        """Fills the parameters' .grad with d loss_pi / d theta; returns the stats tensor
        [sum(-min(..)), sum(ratio), sum(0.5 z^2), B] (device, no sync).  adam_lr: also take the torch.optim.Adam step
        with this learning rate, in the same two launches (same bits as ppo_grad + adam_step)."""
        self._bind()
        opt = self._adam_arg(adam_lr, betas, eps)
        with _on(x):
            rc = self.lib.pds_ppo_policy_grad_step(C.byref(self.m), _ptr(x), _ptr(act), _ptr(adv), _ptr(logp_old),
                                                   _ptr(log_std), x.shape[0], float(clip_ratio), _ptr(self.flat_grad),
                                                   _ptr(self.stats), _ptr(self.workspace), opt, self._stream(x))
        if rc != native.OK:
            raise RuntimeError(f"pds_ppo_policy_grad -> {rc}")
        return self.stats

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8):
        """torch.optim.Adam.step for this network's parameters from the flat gradient, in one launch."""
        self._adam_state()
        self.adam_steps += 1
        self._bind()
        with _on(self.flat_grad):
            rc = self.lib.pds_adam_step(C.byref(self.m), _ptr(self.flat_grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                        self.adam_steps, float(lr), float(betas[0]), float(betas[1]), float(eps),
                                        self._stream(self.flat_grad))
        if rc != native.OK:
            raise RuntimeError(f"pds_adam_step -> {rc}")

    def value_grad(self, x, target, index=None, adam_lr=None, betas=(0.9, 0.999), eps=1e-8):
        """Fills .grad with d mse(net(x[index]), target[index]) / d theta; stats[0] = sum of squared errors.
        adam_lr: as in ppo_grad."""
        self._bind()
        B = x.shape[0] if index is None else index.shape[0]
        opt = self._adam_arg(adam_lr, betas, eps)
        with _on(x):
            rc = self.lib.pds_value_grad_step(C.byref(self.m), _ptr(x), _ptr(index), _ptr(target), B, _ptr(self.flat_grad),
                                              _ptr(self.stats), _ptr(self.workspace), opt, self._stream(x))
        if rc != native.OK:
            raise RuntimeError(f"pds_value_grad -> {rc}")
        return self.stats


def random_permutation(n, seed, call, device):
    """int64 tensor p with p[i] = a pseudo-random permutation of range(n) keyed by (seed, call): one launch
    (include/pds.h pds_permutation) where torch.randperm sorts."""
    out = torch.empty(int(n), dtype=torch.int64, device=device)
    with _on(out):
        rc = native.load().pds_permutation(_ptr(out), int(n), int(seed) & (2 ** 64 - 1), int(call), FusedMLP._stream(out))
    if rc != native.OK:
        raise RuntimeError(f"pds_permutation -> {rc}")
    return out


def counter_add(counter, inc):
    """counter (int64 device tensor of one element) += inc, stream-ordered (captured-rollout call counter)."""
    with _on(counter):
        rc = native.load().pds_counter_add(_ptr(counter), int(inc), FusedMLP._stream(counter))
    if rc != native.OK:
        raise RuntimeError(f"pds_counter_add -> {rc}")


def fused_rollout(env, fm_pi, fm_v, T, mean, std, eps, log_std, seed, call_offset, deterministic, obs_buf, act_buf, logp_buf,
                  val_buf, rew_buf, term_buf, trunc_buf, cost_buf, fval_buf, last_val, ep_ret, ep_len, stats, call_base=None):
    """ONE launch for the T closed-loop steps of a rollout (include/pds.h pds_rollout, csrc/pds_rollout.h): obs_buf is
    [T + 1, N, D] with o(0) in row 0.  Raises NotImplementedError for env configurations the kernel is not built
    for (the per-step path gives the same bits)."""
    fm_pi._bind(); fm_v._bind()
    with _on(obs_buf):
        rc = env.lib.pds_rollout(env._handle, int(T), C.byref(fm_pi.m), C.byref(fm_v.m), _ptr(mean), _ptr(std), float(eps),
                                 _ptr(log_std), int(seed), _ptr(call_base), int(call_offset), int(bool(deterministic)),
                                 _ptr(obs_buf), _ptr(act_buf), _ptr(logp_buf), _ptr(val_buf), _ptr(rew_buf), _ptr(term_buf),
                                 _ptr(trunc_buf), _ptr(cost_buf), _ptr(fval_buf), _ptr(last_val), _ptr(ep_ret), _ptr(ep_len),
                                 _ptr(stats), FusedMLP._stream(obs_buf))
    if rc != native.OK:
        native.check(env._handle, rc, "pds_rollout")


def fused_rollout_history(env, fm_pi, T, H, mean, std, eps, log_std, seed, call_offset, deterministic, obs_buf, act_buf, logp_buf,
                          rew_buf, term_buf, trunc_buf, cost_buf, fin_rows, fin_step, ep_ret, ep_len, stats, call_base=None):
    """ONE launch for the T closed-loop steps of a rollout with observation_history_size = H != 2 (include/pds.h
    pds_rollout_history, csrc/pds_rollout_hist.h): obs_buf is [T + 1, N, H * half] with the current histories in row 0; the critic
    is NOT in the kernel (the caller evaluates V over obs_buf and over the final histories in fin_rows [slots, N, H * half],
    whose fin_step [slots, N] (int32, preset to -1) names the step each belongs to).  Raises NotImplementedError for env
    configurations the kernel is not built for (the per-step path gives the same bits)."""
    fm_pi._bind()
    with _on(obs_buf):
        rc = env.lib.pds_rollout_history(env._handle, int(T), int(H), C.byref(fm_pi.m), _ptr(mean), _ptr(std), float(eps),
                                         _ptr(log_std), int(seed), _ptr(call_base), int(call_offset), int(bool(deterministic)),
                                         _ptr(obs_buf), _ptr(act_buf), _ptr(logp_buf), _ptr(rew_buf), _ptr(term_buf),
                                         _ptr(trunc_buf), _ptr(cost_buf), _ptr(fin_rows), _ptr(fin_step), int(fin_rows.shape[0]),
                                         _ptr(ep_ret), _ptr(ep_len), _ptr(stats), FusedMLP._stream(obs_buf))
    if rc != native.OK:
        native.check(env._handle, rc, "pds_rollout_history")


def gaussian_sample(mu, log_std, act_out, logp_out, seed, call, id_base=0, deterministic=False, call_base=None):
    """act_out[n, d] = mu + exp(log_std) * z, logp_out[n] = log N(act | mu, sigma) summed over d.  The Philox
    call counter is `call` (+ the int64 device word `call_base` when given: hipGraph-capturable form)."""
    with _on(mu):
        rc = native.load().pds_gaussian_sample_dev(_ptr(mu), _ptr(log_std), mu.shape[0], mu.shape[1], int(seed),
                                                   _ptr(call_base), int(call), int(id_base), int(bool(deterministic)),
                                                   _ptr(act_out), _ptr(logp_out), FusedMLP._stream(mu))
    if rc != native.OK:
        raise RuntimeError(f"pds_gaussian_sample -> {rc}")


def rollout_record(rew, term, trunc, rew_buf_t, term_buf_t, trunc_buf_t, ep_ret, ep_len, stats):
    """One step of the rollout bookkeeping (see include/pds.h pds_rollout_record)."""
    with _on(rew):
        rc = native.load().pds_rollout_record(_ptr(rew), _ptr(term), _ptr(trunc), rew.shape[0], _ptr(rew_buf_t),
                                              _ptr(term_buf_t), _ptr(trunc_buf_t), _ptr(ep_ret), _ptr(ep_len), _ptr(stats),
                                              FusedMLP._stream(rew))
    if rc != native.OK:
        raise RuntimeError(f"pds_rollout_record -> {rc}")
