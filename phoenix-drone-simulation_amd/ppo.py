"""On-device PPO for the batched envs: the CALLER of the hot path.

Mirrors the reference trainer on the path `train.py --alg ppo` (same hyper-parameter names and
defaults, same state_dict keys, same update order), re-shaped for a lockstep [T, N] rollout:

  IWPGAlgorithm.roll_out / update / update_value_net / update_policy_net   algs/iwpg/iwpg.py:350-485
  ProximalPolicyOptimizationAlgorithm.compute_loss_pi                      algs/ppo/ppo.py:22-40
  ActorCritic, MLPGaussianActor, MLPCritic                                 algs/core.py:226-412
  Buffer.finish_path / calculate_adv_and_value_targets (GAE)               algs/core.py:461-533
  OnlineMeanStd (Chan parallel update)                                     utils/online_mean_std.py:6-95
  mpi_avg_grads (per-parameter Allreduce)                                  utils/mpi_tools.py:30-36

MI355X-first differences: all tensors stay in HBM; GAE is one HIP kernel over [T, N] (pds_gae)
instead of one scipy lfilter per path; ranks = GPUs, gradients are averaged with ONE flattened
all-reduce per optimiser step over RCCL instead of one MPI Allreduce per parameter tensor; envs that
finish are auto-reset inside pds_step and bootstrap from `final_obs`.
PyTorch is used for what it is good at here: the two tiny MLPs and Adam.
"""
import ctypes as C
import math

import time

import torch
import torch.distributed as dist
import torch.nn as nn

from . import native
from .fused import counter_add, gaussian_sample, random_permutation, rollout_record


# Collectives are skipped when the job has one rank -- unless this is set (tests: the RCCL calls of the multi-GPU path executed
# on a 1-GPU box under torch.distributed.run --nproc-per-node 1; an all-reduce over one rank leaves its operand as it is).
FORCE_COLLECTIVES = False


def _collectives():
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


class OnlineMeanStd(nn.Module):
    """utils/online_mean_std.py:6-95 with identical parameter names (mean, std, count)."""

    def __init__(self, epsilon=1e-5, shape=()):
        super().__init__()
        self.mean = nn.Parameter(torch.zeros(*shape), requires_grad=False)
        self.std = nn.Parameter(torch.ones(*shape), requires_grad=False)
        self.count = nn.Parameter(torch.zeros(1), requires_grad=False)
        self.eps = epsilon
        self.bound = 10
        self.shape = shape

    def forward(self, x, subtract_mean=True, clip=False):
        y = (x - self.mean) / (self.std + self.eps) if subtract_mean else x / (self.std + self.eps)
        return torch.clamp(y, -self.bound, self.bound) if clip else y

    @torch.no_grad()
    def update(self, x):
        """Chan et al. parallel update; batch statistics are averaged over ranks (equal batch sizes)."""
        if self.shape[0] == 1:
            x = x.reshape(-1, 1)
        world = dist.get_world_size() if dist.is_initialized() else 1
        n_B = x.shape[0] * world
        n_A = self.count.clone()
        n_AB = self.count + n_B
        batch_mean = torch.mean(x, dim=0)
        if _collectives():
            dist.all_reduce(batch_mean)
            batch_mean /= world
        delta = batch_mean - self.mean
        mean_new = self.mean + delta * n_B / n_AB
        batch_var = torch.mean((x - mean_new) ** 2, dim=0)
        if _collectives():
            dist.all_reduce(batch_var)
            batch_var /= world
        M2_AB = n_A * torch.square(self.std) + n_B * batch_var + delta ** 2 * (n_A * n_B / n_AB)
        # in place: the fused kernels (and a captured rollout graph) hold the addresses of these tensors
        self.mean.data.copy_(mean_new)
        self.count.data.copy_(n_AB)
        self.std.data.copy_(torch.sqrt(M2_AB / n_AB))


def _mlp(sizes, activation):
    act = {"relu": nn.ReLU, "tanh": nn.Tanh, "identity": nn.Identity, "sigmoid": nn.Sigmoid,
           "softplus": nn.Softplus}[activation]
    layers = []
    for j in range(len(sizes) - 1):
        lin = nn.Linear(sizes[j], sizes[j + 1])
        nn.init.kaiming_uniform_(lin.weight, a=math.sqrt(5))  # algs/core.py initialize_layer default
        layers += [lin, act() if j < len(sizes) - 2 else nn.Identity()]
    return nn.Sequential(*layers)


class GaussianActor(nn.Module):
    """MLPGaussianActor (algs/core.py:226-290): state-independent log_std, annealed 0.5 -> 0.01."""

    def __init__(self, obs_dim, act_dim, hidden_sizes=(50, 50), activation="relu"):
        super().__init__()
        self.log_std = nn.Parameter(torch.full((act_dim,), math.log(0.5)), requires_grad=False)
        self.net = _mlp([obs_dim] + list(hidden_sizes) + [act_dim], activation)

    def dist(self, obs):
        return torch.distributions.Normal(self.net(obs), torch.exp(self.log_std))

    def forward(self, obs, act=None):
        d = self.dist(obs)
        return d, (d.log_prob(act).sum(-1) if act is not None else None)

    def set_log_std(self, frac):
        self.log_std.data.fill_(math.log(0.499 * frac + 0.01))  # algs/core.py:268-276


class Critic(nn.Module):
    def __init__(self, obs_dim, hidden_sizes=(64, 64), activation="tanh"):
        super().__init__()
        self.net = _mlp([obs_dim] + list(hidden_sizes) + [1], activation)

    def forward(self, obs):
        return torch.squeeze(self.net(obs), -1)


class ActorCritic(nn.Module):
    """algs/core.py:313-412; state_dict keys are the reference's (obs_oms.*, pi.log_std, pi.net.*,
    v.net.*, ret_oms.*), so `torch_save/model.pt` checkpoints interchange."""

    def __init__(self, obs_dim, act_dim, ac_kwargs=None, use_standardized_obs=True, use_scaled_rewards=True):
        super().__init__()
        ac_kwargs = ac_kwargs or {"pi": {"hidden_sizes": (50, 50), "activation": "relu"},
                                  "val": {"hidden_sizes": (64, 64), "activation": "tanh"}}
        self.obs_oms = OnlineMeanStd(shape=(obs_dim,)) if use_standardized_obs else None
        self.pi = GaussianActor(obs_dim, act_dim, **ac_kwargs["pi"])
        self.v = Critic(obs_dim, **ac_kwargs["val"])
        self.ret_oms = OnlineMeanStd(shape=(1,)) if use_scaled_rewards else None

    @classmethod
    def from_reference_state_dict(cls, sd, pi_activation="relu", val_activation="tanh"):
        """ActorCritic with the layer sizes of a reference `torch_save/model.pt` state_dict
        (utils/loggers.py:382-407), loaded strictly.  The checkpoints bundled under experiments/
        come from an older reference version and also hold a cost critic `c.net.*`, which no
        algorithm of the path reads: those keys are dropped."""
        sd = {k: torch.as_tensor(v) for k, v in sd.items() if not k.startswith("c.")}

        def hidden(prefix):
            idx = sorted(int(k.split(".")[2]) for k in sd if k.startswith(prefix + ".net.") and k.endswith(".weight"))
            return tuple(int(sd[f"{prefix}.net.{i}.weight"].shape[0]) for i in idx[:-1])

        first = sd["pi.net.0.weight"]
        last = sd["pi.net.%d.weight" % max(int(k.split(".")[2]) for k in sd if k.startswith("pi.net."))]
        ac = cls(int(first.shape[1]), int(last.shape[0]),
                 ac_kwargs={"pi": {"hidden_sizes": hidden("pi"), "activation": pi_activation},
                            "val": {"hidden_sizes": hidden("v"), "activation": val_activation}},
                 use_standardized_obs="obs_oms.mean" in sd, use_scaled_rewards="ret_oms.mean" in sd)
        ac.load_state_dict(sd, strict=True)
        return ac

    @torch.no_grad()
    def step(self, obs):
        """(action, value, log_prob) for raw observations [N, D] (algs/core.py:370-393)."""
        if self.obs_oms is not None:
            obs = self.obs_oms(obs)
        v = self.v(obs)
        d = self.pi.dist(obs)
        a = d.sample() if self.training else d.mean
        return a, v, d.log_prob(a).sum(-1)

    @torch.no_grad()
    def value(self, obs):
        return self.v(self.obs_oms(obs) if self.obs_oms is not None else obs)

    def update(self, frac):
        self.pi.set_log_std(1 - frac)


def ppo_loss(ac, data, clip_ratio=0.2, entropy_coef=0.0):
    """algs/ppo/ppo.py:22-40."""
    d, logp = ac.pi(data["obs"], data["act"])
    ratio = torch.exp(logp - data["log_p"])
    clip_adv = data["adv"] * torch.clamp(ratio, 1 - clip_ratio, 1 + clip_ratio)
    loss = -(torch.min(ratio * data["adv"], clip_adv)).mean()
    loss = loss - entropy_coef * d.entropy().mean()
    info = dict(kl=(0.5 * (d.mean - data["act"]) ** 2 / d.stddev ** 2).mean(), ent=d.entropy().mean(),
                ratio=ratio.mean())
    return loss, info


def value_loss(ac, obs, ret):
    """IWPGAlgorithm.compute_loss_v."""
    return ((ac.v(obs) - ret) ** 2).mean()


def gae(rew, val, terminated, truncated, final_val, last_val, gamma, lam, rew_scale=0.0, rew_clip=10.0):
    """pds_gae on [T, N] device tensors -> (adv, target_v, discounted_ret)."""
    T, N = rew.shape
    out = [torch.empty_like(rew) for _ in range(3)]
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    with torch.cuda.device(rew.device):  # pds_gae launches on the current device
        rc = native.load().pds_gae(p(rew.contiguous()), p(val.contiguous()), p(terminated.contiguous()),
                                   p(truncated.contiguous()), p(final_val), p(last_val.contiguous()),
                                   C.c_float(gamma), C.c_float(lam), C.c_float(rew_scale), C.c_float(rew_clip),
                                   T, N, p(out[0]), p(out[1]), p(out[2]),
                                   C.c_void_p(torch.cuda.current_stream(rew.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"pds_gae failed ({rc})")
    return out


def avg_grads(module):
    """mpi_avg_grads (utils/mpi_tools.py:30-36) as ONE flattened RCCL all-reduce."""
    if not _collectives():
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat /= dist.get_world_size()
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


class PPOTrainer:
    """IWPGAlgorithm + PPO loss on a DroneVecEnv.  steps_per_epoch = rollout_len * env.num_envs per rank."""

    def __init__(self, env, rollout_len=32, epochs=300, gamma=0.99, lam=0.95, clip_ratio=0.2,
                 entropy_coef=0.01, use_entropy=False, pi_lr=3e-4, vf_lr=1e-3, train_pi_iterations=80,
                 train_v_iterations=5, num_mini_batches=16, target_kl=0.01, use_kl_early_stopping=False,
                 use_linear_lr_decay=True, use_exploration_noise_anneal=True, use_reward_scaling=True,
                 use_standardized_obs=True, use_max_grad_norm=False, max_grad_norm=0.5, ac_kwargs=None,
                 seed=0, fused=None, graph_rollout=None, fused_rollout=None, reset_each_rollout=False,
                 overlap_value_update=True, log_reference_columns=False):
        self.env, self.T, self.N = env, int(rollout_len), env.num_envs
        self.overlap_value_update, self._side_stream = bool(overlap_value_update), None
        # also log what the reference's progress.csv carries per epoch besides EpRet / EpLen (Values/V/Mean, Misc/RewScaleMean,
        # Misc/RewScaleStddev: algs/iwpg/iwpg.py:524-563) -- three host syncs per epoch, for comparisons of whole runs
        self.log_reference_columns = bool(log_reference_columns)
        self.epochs, self.gamma, self.lam, self.clip_ratio = epochs, gamma, lam, clip_ratio
        self.entropy_coef = entropy_coef if use_entropy else 0.0
        self.train_pi_iterations, self.train_v_iterations = train_pi_iterations, train_v_iterations
        self.num_mini_batches, self.target_kl = num_mini_batches, target_kl
        self.use_kl_early_stopping, self.use_linear_lr_decay = use_kl_early_stopping, use_linear_lr_decay
        self.use_exploration_noise_anneal, self.use_reward_scaling = use_exploration_noise_anneal, use_reward_scaling
        self.use_standardized_obs, self.use_max_grad_norm, self.max_grad_norm = use_standardized_obs, use_max_grad_norm, max_grad_norm
        rank = dist.get_rank() if dist.is_initialized() else 0
        torch.manual_seed(seed + 10000 * rank)  # algs/iwpg/iwpg.py:124-127
        dev = env.device
        self.ac = ActorCritic(env.obs_dim, env.act_dim, ac_kwargs, use_standardized_obs, use_reward_scaling).to(dev)
        if _collectives():  # sync_params, utils/mpi_tools.py:39-44
            for p in self.ac.parameters():
                dist.broadcast(p.data, 0)
        self.pi_opt = torch.optim.Adam(self.ac.pi.net.parameters(), lr=pi_lr)
        self.vf_opt = torch.optim.Adam(self.ac.v.parameters(), lr=vf_lr)
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(self.pi_opt, lambda e: 1 - e / epochs) if use_linear_lr_decay else None
        # fused MFMA kernels (csrc/pds_mlp.hip) for network inference in the rollout and for the loss
        # gradients of the update; the PyTorch op chains below stay as the reference path (fused=False)
        kw = ac_kwargs or {"pi": {"hidden_sizes": (50, 50), "activation": "relu"},
                           "val": {"hidden_sizes": (64, 64), "activation": "tanh"}}
        self._pi_activation = kw["pi"]["activation"]
        self.fused = (dev.type == "cuda") if fused is None else bool(fused)
        self._sample_seed, self._sample_calls = (seed + 10000 * rank) & 0xFFFFFFFFFFFFFFFF, 0
        self._perm_calls = 0  # call counter of the mini-batch shuffles (pds_permutation)
        # tests: a callable B -> int64 index tensor that replaces the value net's shuffle (the reference's np.random.shuffle
        # sequences are replayed through it: tests/golden/update.npz)
        self.perm_fn = None
        # hipGraph capture of the whole rollout (fused path, even rollout length so that the env's two output
        # sets line up from replay to replay): the env step is capturable (tick / parity in device memory), the
        # sampling call counter gets a device word that the graph advances once per replay
        if graph_rollout is None:  # default: on where it pays (small batches are launch-bound) and is possible
            graph_rollout = env.num_envs <= 262144 and getattr(env, "observation_history_size", 2) == 2
        self.graph_rollout = bool(graph_rollout) and self.fused and (rollout_len % 2 == 0)
        self._graph, self._graph_stats, self._call_base = None, None, None
        # ONE launch per rollout (csrc/pds_rollout.h: networks + sampling + env step + bookkeeping for all T steps,
        # the env state in registers, the observation tile in LDS); None = use it when the env configuration has an
        # instantiation (found out at the first rollout), False = the per-step kernels (same bits)
        # (round 3 fell back to the per-step kernels above 262 144 envs, where one tile per block lost -- 2^20 x 8: 9.1 vs 6.3 ms;
        #  with two tiles per block where there are more tiles than CUs the one-launch form wins at every size: 5.6 vs 6.1 ms)
        self.fused_rollout = fused_rollout
        if self.fused:
            from .fused import FusedMLP
            try:
                self.fm_pi = FusedMLP(self.ac.pi.net, kw["pi"]["activation"])
                self.fm_v = FusedMLP(self.ac.v.net, kw["val"]["activation"])
            except (ValueError, NotImplementedError) as e:
                # outside the fused kernels' range (layer sizes, depths, activations: fused.FusedMLP says which): the networks
                # run as PyTorch ops on the HIP envs (same update, the path the fused kernels are tested against) -- an order
                # of magnitude slower, so never silently; asked for explicitly, it is an error
                if fused:
                    raise
                import warnings
                warnings.warn(f"PPOTrainer: the fused MFMA kernels do not cover these networks ({e}); using PyTorch ops",
                              RuntimeWarning, stacklevel=2)
                self.fused = False
                self.graph_rollout = False
                self.fm_pi = self.fm_v = None
        if self.fused:
            # Adam runs in pds_adam_step; the torch optimisers only carry the learning rate (LambdaLR)
            self.pi_opt._opt_called = True
        T, N, D = self.T, self.N, env.obs_dim
        f = dict(device=dev, dtype=torch.float32)
        self._obs_buf = torch.zeros(T + 1, N, D, **f)  # row T: o(T), written by the fused rollout
        self.obs_buf = self._obs_buf[:T]
        self.cost_buf = torch.zeros(T, N, **f)
        self._val_store = torch.zeros(T + 1, N, **f)   # V(o(0 .. T)): rows 0 .. T - 1 = val_buf, row T = V(o(T))
        self.val_buf, self._last_val_buf = self._val_store[:T], self._val_store[T]
        self.act_buf = torch.zeros(T, N, 4, **f)
        self.rew_buf, self.logp_buf, self.fval_buf = (torch.zeros(T, N, **f) for _ in range(3))
        self._fin_rows = self._fin_step = None         # pds_rollout_history's slot list (allocated at its first call)
        self.term_buf = torch.zeros(T, N, device=dev, dtype=torch.uint8)
        self.trunc_buf = torch.zeros(T, N, device=dev, dtype=torch.uint8)
        self.ep_ret = torch.zeros(N, **f)
        self.ep_len = torch.zeros(N, **f)
        self.obs, _ = env.reset()
        # reset_each_rollout: IWPGAlgorithm.roll_out starts every epoch with `o, _ = self.env.reset()` and drops the
        # episode that the previous epoch cut (algs/iwpg/iwpg.py:352-353, 382-385).  With thousands of lockstep envs
        # that would throw away the unfinished episode of every env each epoch, so the default is to carry on
        # (the cut path is bootstrapped with V either way); True mirrors the reference -- including its bootstrap of a path
        # that terminated on the epoch's last step, see roll_out -- e.g. for the learning-curve comparison against its own
        # trainer (tests/test_trainer.py, tests/golden/learning_curve.json) and the element-for-element replay of its
        # rollouts (tests/golden/rollout.npz, oracle/refgen/check_rollout_logic.py).
        self.reset_each_rollout = bool(reset_each_rollout)
        if self.reset_each_rollout:
            graph_rollout = False  # (a captured rollout holds the address of the previous observation)
            self.graph_rollout = False
        self.epoch = 0
        self.log = []

    def roll_out(self):
        """algs/iwpg/iwpg.py:350-385 over all envs at once.  Returns per-epoch episode statistics."""
        self.ac.train()
        if self.reset_each_rollout and self.epoch > 0:
            self.obs, _ = self.env.reset()
            self.ep_ret.zero_()
            self.ep_len.zero_()
        stats = self._roll_out_dispatch()
        if self.reset_each_rollout:
            # the reference's epoch-end cut: `if truncated or epoch_ended: v = V(o)` (algs/iwpg/iwpg.py:374-379) also for a path
            # that TERMINATED on the epoch's last step -- it bootstraps with V(its last observation), not with 0.  pds_gae
            # lets the cut win over the termination; V(final_obs) of the last step is in fval_buf[T - 1] on every path.
            self.trunc_buf[self.T - 1] |= self.term_buf[self.T - 1]
        return stats

    def _roll_out_dispatch(self):
        if self.fused and self.fused_rollout is not False:
            try:
                if getattr(self.env, "observation_history_size", 2) != 2:
                    return self._roll_out_fused_history()
                return self._roll_out_fused()
            except NotImplementedError:
                if self.fused_rollout:  # asked for explicitly
                    raise
                self.fused_rollout = False
        if self.graph_rollout:
            return self._roll_out_graph()
        return self._roll_out_eager()

    @torch.no_grad()
    def _roll_out_fused(self):
        """The whole rollout in one launch (pds_rollout).  Same draws as the per-step path: step t samples with
        call counter `_sample_calls + t + 1`."""
        from .fused import fused_rollout
        mean, std, eps = self._oms()
        stats = torch.zeros(3, device=self.obs.device)
        self._obs_buf[0].copy_(self.obs)
        fused_rollout(self.env, self.fm_pi, self.fm_v, self.T, mean, std, eps, self.ac.pi.log_std, self._sample_seed,
                      self._sample_calls, not self.ac.training, self._obs_buf, self.act_buf, self.logp_buf, self.val_buf,
                      self.rew_buf, self.term_buf, self.trunc_buf, self.cost_buf, self.fval_buf, self._last_val_buf,
                      self.ep_ret, self.ep_len, stats)
        self._sample_calls += self.T
        self.fused_rollout = True
        self.obs = self._obs_buf[self.T]  # (a view: copied into row 0 before the next launch overwrites it)
        self.last_val = self._last_val_buf
        return stats

    @torch.no_grad()
    def _roll_out_fused_history(self):
        """observation_history_size != 2: actor, sampling, env step and history update in one launch (pds_rollout_history), then
        the critic over the whole rollout in one pds_mlp_forward (V(o(0..T)): val_buf and last_val) and one over the final
        histories of the paths that bootstrap with V.  Same draws and same bits as the per-step path."""
        from .fused import fused_rollout_history
        env, T, N = self.env, self.T, self.N
        H, D = env.observation_history_size, env.obs_dim
        mean, std, eps = self._oms()
        stats = torch.zeros(3, device=self.obs.device)
        if self._fin_rows is None:
            slots = T // int(env.cfg.max_episode_steps) + 2
            self._fin_rows = torch.zeros(slots, N, D, device=self.obs.device)
            self._fin_step = torch.empty(slots, N, dtype=torch.int32, device=self.obs.device)
            self._fval_store = torch.zeros(T * N + 1, device=self.obs.device)  # (+ 1: where the unused slots' values go)
            self.fval_buf = self._fval_store[:T * N].view(T, N)
            self._env_ids = torch.arange(N, device=self.obs.device, dtype=torch.int64)
        self._obs_buf[0].copy_(self.obs)
        self._fin_step.fill_(-1)
        fused_rollout_history(env, self.fm_pi, T, H, mean, std, eps, self.ac.pi.log_std, self._sample_seed, self._sample_calls,
                              not self.ac.training, self._obs_buf, self.act_buf, self.logp_buf, self.rew_buf, self.term_buf,
                              self.trunc_buf, self.cost_buf, self._fin_rows, self._fin_step, self.ep_ret, self.ep_len, stats)
        self._sample_calls += T
        self.fused_rollout = True
        # the critic, off the kernel: V(o(t)) for t = 0 .. T in one pass over obs_buf ...
        self.fm_v.forward(self._obs_buf.view((T + 1) * N, D), mean=mean, std=std, eps=eps, out=self._val_store.view(-1, 1))
        # ... and V(final history) of the paths that bootstrap with it -> fval_buf[fin_step, env]
        vfin = self.fm_v.forward(self._fin_rows.view(-1, D), mean=mean, std=std, eps=eps).view(-1, N)
        step = self._fin_step.to(torch.int64)
        target = torch.where(step >= 0, step * N + self._env_ids, torch.full_like(step, T * N))
        self._fval_store.scatter_(0, target.view(-1), vfin.reshape(-1))
        self.obs = self._obs_buf[T]
        env.adopt_history(self.obs)
        self.last_val = self._last_val_buf
        return stats

    def _roll_out_graph(self):
        """The same launches, captured once (7 per step x T) and replayed per epoch: one host call per rollout."""
        if self._graph is None:
            dev = self.env.device
            self._call_base = torch.full((1,), self._sample_calls, dtype=torch.int64, device=dev)
            self._graph_stats = torch.zeros(3, device=dev)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._roll_out_eager(stats=self._graph_stats)
                counter_add(self._call_base, self.T)
            self._graph = g
        self._graph.replay()
        self._sample_calls += self.T
        return self._graph_stats

    def _roll_out_eager(self, stats=None):
        o = self.obs
        capturing = stats is not None
        if capturing:
            stats.zero_()
        else:
            stats = torch.zeros(3, device=o.device)
        for t in range(self.T):
            if self.fused:
                a, logp = self._fused_step(o, t)
            else:
                a, v, logp = self.ac.step(o)
                self.val_buf[t].copy_(v)
            next_o, r, term, trunc, info = self.env.step(a)
            self.obs_buf[t].copy_(o)
            # V(final obs) for the TimeLimit bootstrap; evaluated for every row to stay sync-free
            # (rows of envs that did not finish are ignored by pds_gae)
            if self.fused:  # action / log-prob were sampled straight into their buffer rows
                self._fused_value(info["final_obs"], out=self.fval_buf[t])
                rollout_record(r, term.view(torch.uint8), trunc.view(torch.uint8), self.rew_buf[t], self.term_buf[t],
                               self.trunc_buf[t], self.ep_ret, self.ep_len, stats)
                o = next_o
                continue
            self.act_buf[t].copy_(a)
            self.rew_buf[t].copy_(r); self.logp_buf[t].copy_(logp)
            self.term_buf[t].copy_(term.view(torch.uint8)); self.trunc_buf[t].copy_(trunc.view(torch.uint8))
            self.fval_buf[t].copy_(self.ac.value(info["final_obs"]))
            self.ep_ret += r
            self.ep_len += 1
            done = term | trunc
            stats += torch.stack([(self.ep_ret * done).sum(), (self.ep_len * done).sum(), done.sum().float()])
            self.ep_ret = torch.where(done, torch.zeros_like(self.ep_ret), self.ep_ret)
            self.ep_len = torch.where(done, torch.zeros_like(self.ep_len), self.ep_len)
            o = next_o
        self.obs = o
        self.last_val = self._fused_value(o) if self.fused else self.ac.value(o)
        return stats

    def _oms(self):
        oms = self.ac.obs_oms
        return (oms.mean, oms.std, oms.eps) if oms is not None else (None, None, 0.0)

    @torch.no_grad()
    def _fused_value(self, obs, out=None):
        mean, std, eps = self._oms()
        y = self.fm_v.forward(obs, mean=mean, std=std, eps=eps, out=None if out is None else out.view(-1, 1))
        return y.view(-1)

    @torch.no_grad()
    def _fused_step(self, obs, t):
        """ActorCritic.step (algs/core.py:370-393) on the fused kernels: V(o) -> val_buf[t]; action
        a = mu + sigma * z with z ~ N(0, 1) and its log-probability -sum(0.5 z^2 + log sigma + 0.5 log 2 pi)."""
        mean, std, eps = self._oms()
        self._fused_value(obs, out=self.val_buf[t])
        mu = self.fm_pi.forward(obs, mean=mean, std=std, eps=eps)
        if self._call_base is not None and torch.cuda.is_current_stream_capturing():
            # captured: call = device word (advanced by T per replay) + step index
            gaussian_sample(mu, self.ac.pi.log_std, self.act_buf[t], self.logp_buf[t], self._sample_seed, t + 1,
                            id_base=self.env.env_id_base, deterministic=not self.ac.training, call_base=self._call_base)
        else:
            self._sample_calls += 1
            gaussian_sample(mu, self.ac.pi.log_std, self.act_buf[t], self.logp_buf[t], self._sample_seed, self._sample_calls,
                            id_base=self.env.env_id_base, deterministic=not self.ac.training)
        return self.act_buf[t], self.logp_buf[t]

    def update(self):
        """algs/iwpg/iwpg.py:398-485."""
        ac, T, N = self.ac, self.T, self.N
        scale = 0.0
        if self.use_reward_scaling:
            scale = float(1.0 / (ac.ret_oms.std.item() + ac.ret_oms.eps))
        adv, target_v, disc_ret = gae(self.rew_buf, self.val_buf, self.term_buf, self.trunc_buf, self.fval_buf,
                                      self.last_val, self.gamma, self.lam, scale, float(ac.ret_oms.bound if ac.ret_oms else 10))
        raw_obs = self.obs_buf.reshape(T * N, -1)
        obs = ac.obs_oms(raw_obs) if self.use_standardized_obs else raw_obs  # pre_process_data
        data = dict(obs=obs, act=self.act_buf.reshape(T * N, -1), adv=adv.reshape(-1),
                    log_p=self.logp_buf.reshape(-1), target_v=target_v.reshape(-1))
        # ---- value net: train_v_iterations x num_mini_batches shuffled mini-batches
        B = T * N
        mbs = B // self.num_mini_batches
        if self.fused:
            return self._fused_update(data, raw_obs, disc_ret, B, mbs)
        loss_v_before = value_loss(ac, data["obs"], data["target_v"]).item()
        for _ in range(self.train_v_iterations):
            perm = self.perm_fn(B) if self.perm_fn is not None else torch.randperm(B, device=obs.device)
            for s in range(0, mbs * self.num_mini_batches, mbs):
                idx = perm[s:s + mbs]
                self.vf_opt.zero_grad()
                lv = value_loss(ac, data["obs"][idx], data["target_v"][idx])
                lv.backward()
                avg_grads(ac.v)
                self.vf_opt.step()
        # ---- policy net: full-batch PPO-clip steps
        with torch.no_grad():
            loss_pi_before, _ = ppo_loss(ac, data, self.clip_ratio, self.entropy_coef)
            p_dist = ac.pi.dist(data["obs"])
        stop_iter = self.train_pi_iterations
        for i in range(self.train_pi_iterations):
            self.pi_opt.zero_grad()
            loss_pi, pi_info = ppo_loss(ac, data, self.clip_ratio, self.entropy_coef)
            loss_pi.backward()
            if self.use_max_grad_norm:
                torch.nn.utils.clip_grad_norm_(ac.pi.parameters(), self.max_grad_norm)
            avg_grads(ac.pi.net)
            self.pi_opt.step()
            if self.use_kl_early_stopping:
                with torch.no_grad():
                    kl = torch.distributions.kl.kl_divergence(p_dist, ac.pi.dist(data["obs"])).mean()
                    if _collectives():
                        dist.all_reduce(kl); kl /= dist.get_world_size()
                if kl.item() > self.target_kl:
                    stop_iter = i + 1
                    break
        # ---- running statistics from RAW data, after the update (update_running_statistics)
        if self.use_standardized_obs:
            ac.obs_oms.update(raw_obs)
        if self.use_reward_scaling:
            ac.ret_oms.update(disc_ret.reshape(-1))
        return dict(loss_pi=float(loss_pi_before), loss_v=loss_v_before, stop_iter=stop_iter,
                    entropy=float(pi_info["ent"].detach()), ratio=float(pi_info["ratio"].detach()))

    def _fused_update(self, data, raw_obs, disc_ret, B, mbs):
        """The same update with the loss gradients from csrc/pds_mlp.hip: one fused pass over the batch
        per iteration writes d loss / d theta into the parameters' .grad (a flat buffer, so the
        gradient averaging over ranks is one RCCL all-reduce of it); Adam stays torch's."""
        ac = self.ac
        world = dist.get_world_size() if dist.is_initialized() else 1

        multi = _collectives()  # several ranks (or FORCE_COLLECTIVES): gradients are averaged between the gradient and Adam

        def average(fm):
            if multi:
                dist.all_reduce(fm.flat_grad)
                fm.flat_grad /= world

        obs, target_v = data["obs"].contiguous(), data["target_v"].contiguous()
        with torch.no_grad():
            loss_v_before = ((self.fm_v.forward(obs).view(-1) - target_v) ** 2).mean()

        def value_steps():
            """one mini-batch step of the value net per next()"""
            for _ in range(self.train_v_iterations):
                # one elementwise launch (pds_permutation) where torch.randperm sorts (~160 us at 2^19 samples, 5 x per epoch)
                self._perm_calls += 1
                perm = (self.perm_fn(B) if self.perm_fn is not None else
                        random_permutation(B, self._sample_seed ^ 0x5045524D, self._perm_calls, obs.device))
                for s in range(0, mbs * self.num_mini_batches, mbs):
                    if not multi:  # the Adam step rides on the gradient's partial-sum kernel (same bits, one launch less)
                        self.fm_v.value_grad(obs, target_v, index=perm[s:s + mbs], adam_lr=self.vf_opt.param_groups[0]["lr"])
                    else:
                        self.fm_v.value_grad(obs, target_v, index=perm[s:s + mbs])
                        average(self.fm_v)
                        self.fm_v.adam_step(self.vf_opt.param_groups[0]["lr"])
                    yield

        # The two updates share no state (two networks, two optimisers, read-only batch): single process, the value net's 80
        # mini-batch steps -- each 10 us of work behind ~27 us of fixed latency -- run on a second stream next to the policy
        # net's 80 full-batch steps instead of in front of them, and the host feeds the two streams ALTERNATELY (one value
        # step per policy step: enqueueing the whole value path first keeps the policy stream empty for its 2.4 ms of host
        # time).  Same launches, same bits.  (Several ranks: the two nets' all-reduces would have to be issued in one order
        # on every rank: kept sequential.)
        side = None
        vgen = value_steps()
        if not multi and self.overlap_value_update and obs.is_cuda:
            main = torch.cuda.current_stream(obs.device)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=obs.device)  # (a high-priority stream: no gain, profiles/r04_ppo_overlap.txt)
            side = self._side_stream
            side.wait_stream(main)
        else:
            for _ in vgen:
                pass

        def feed_value_stream(steps):
            if side is None:
                return
            with torch.cuda.stream(side):
                for _ in range(steps):
                    if next(vgen, StopIteration) is StopIteration:
                        break
        act, adv, logp_old = data["act"].contiguous(), data["adv"].contiguous(), data["log_p"].contiguous()
        log_std = ac.pi.log_std
        # entropy of Normal(., sigma): sum(0.5 + 0.5 log 2 pi + log sigma), independent of the network
        ent = float((0.5 + 0.5 * math.log(2 * math.pi) + log_std).sum())
        if self.use_kl_early_stopping:
            with torch.no_grad():
                mu_old = self.fm_pi.forward(obs)
        first = None
        stop_iter = self.train_pi_iterations
        v_total = self.train_v_iterations * self.num_mini_batches
        try:
            for i in range(self.train_pi_iterations):
                # the value steps due by now: spread evenly over the policy iterations
                feed_value_stream((i + 1) * v_total // self.train_pi_iterations - i * v_total // self.train_pi_iterations)
                ride = not multi and not self.use_max_grad_norm  # Adam inside the gradient call (same bits)
                stats = self.fm_pi.ppo_grad(obs, act, adv, logp_old, log_std, self.clip_ratio,
                                            adam_lr=self.pi_opt.param_groups[0]["lr"] if ride else None)
                if first is None:
                    first = stats.clone()
                if not ride:
                    if self.use_max_grad_norm:
                        torch.nn.utils.clip_grad_norm_(ac.pi.net.parameters(), self.max_grad_norm)
                    average(self.fm_pi)
                    self.fm_pi.adam_step(self.pi_opt.param_groups[0]["lr"])  # lr follows the LambdaLR schedule
                if self.use_kl_early_stopping:
                    with torch.no_grad():  # KL(N(mu_old, s) || N(mu_new, s)) = sum (mu_old - mu_new)^2 / (2 s^2)
                        kl = (((mu_old - self.fm_pi.forward(obs)) ** 2) / (2 * torch.exp(2 * log_std))).sum(-1).mean()
                        if multi:
                            dist.all_reduce(kl); kl /= world
                    if kl.item() > self.target_kl:
                        stop_iter = i + 1
                        break
            feed_value_stream(v_total)  # (whatever an early stop of the policy loop has left; not on an exception)
        finally:
            # whatever happens in the policy loop (a non-finite KL, KeyboardInterrupt): the value steps already enqueued on the
            # side stream read `obs`, `target_v` and the value net's tensors, which were allocated on the main stream -- the
            # main stream waits for them before anything can be freed or reused.  Nothing more is enqueued here: after an
            # exception the value net keeps the steps it had got, and a second error cannot hide the first.
            if side is not None:
                torch.cuda.current_stream(obs.device).wait_stream(side)
        if self.use_standardized_obs:
            ac.obs_oms.update(raw_obs)
        if self.use_reward_scaling:
            ac.ret_oms.update(disc_ret.reshape(-1))
        f = first.tolist()
        return dict(loss_pi=f[0] / f[3] - self.entropy_coef * ent, loss_v=float(loss_v_before), stop_iter=stop_iter,
                    entropy=ent, ratio=f[1] / f[3])

    def save_checkpoint(self, log_dir, activation=None):
        """The artefacts the reference's logger leaves per run (utils/loggers.py:382-407,
        utils/export.py:83-98): `torch_save/model.pt` = ActorCritic.state_dict() with the reference's
        keys, and the firmware JSON of the actor next to it."""
        import os
        from .policy_io import convert_actor_critic_to_json
        os.makedirs(os.path.join(log_dir, "torch_save"), exist_ok=True)
        path = os.path.join(log_dir, "torch_save", "model.pt")
        torch.save({k: v.detach().cpu() for k, v in self.ac.state_dict().items()}, path)
        convert_actor_critic_to_json(self.ac, os.path.join(log_dir, "model.json"), activation or self._pi_activation)
        return path

    def learn_one_epoch(self):
        t0 = time.time()
        if self.use_exploration_noise_anneal:
            self.ac.update(frac=self.epoch / self.epochs)
        stats = self.roll_out()
        extra = {}
        if self.log_reference_columns:
            extra["values_v_mean"] = float(self.val_buf.mean())
            with torch.no_grad():  # the policy's means over the epoch's batch before the update (for the reference's `KL` column)
                obs_std = self.ac.obs_oms(self.obs_buf.reshape(self.T * self.N, -1)) if self.use_standardized_obs else self.obs_buf.reshape(self.T * self.N, -1)
                mu_old = self.ac.pi.net(obs_std)
        info = self.update()
        if self.log_reference_columns:
            with torch.no_grad():  # torch_kl of update_policy_net (algs/iwpg/iwpg.py:437-439): KL(p_old || p_new), mean over batch and action dims
                mu_new = self.ac.pi.net(obs_std)
                extra["kl"] = float((((mu_old - mu_new) ** 2) / (2 * torch.exp(2 * self.ac.pi.log_std))).mean())
        if self.log_reference_columns and self.ac.ret_oms is not None:  # (the logger reads them after update_running_statistics)
            extra["rew_scale_mean"] = float(self.ac.ret_oms.mean)
            extra["rew_scale_std"] = float(self.ac.ret_oms.std)
        info.update(extra)
        bad_here = not (math.isfinite(info["loss_pi"]) and math.isfinite(info["loss_v"]))
        if _collectives():
            # the losses are rank-local (the shard that holds a NaN env sees it first): decide TOGETHER, or the
            # other ranks would walk into the next all-reduce and hang until the RCCL timeout
            flag = torch.tensor([1.0 if bad_here else 0.0], device=self.env.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            bad_any = bool(flag.item() > 0)
        else:
            bad_any = bad_here
        if bad_any:
            # the reference only guards NPG (algs/npg/npg.py:118,126); a poisoned batch would train NaNs
            bad = self.env.count_nonfinite() if hasattr(self.env, "count_nonfinite") else "?"
            raise FloatingPointError(f"non-finite loss in epoch {self.epoch + 1}: {bad} envs hold a NaN/Inf state "
                                     "(e.g. DroneTakeOffSimpleEnv-v0 with domain randomisation: the reference's "
                                     "explicit Euler step overflows on an env that never terminates)")
        if self.scheduler is not None:
            self.scheduler.step()
        if _collectives():
            dist.all_reduce(stats)
        s = stats.tolist()
        world = dist.get_world_size() if dist.is_initialized() else 1
        self._t_total = getattr(self, "_t_total", 0.0) + (time.time() - t0)
        info.update(epoch=self.epoch + 1, ep_ret=s[0] / max(s[2], 1.0), ep_len=s[1] / max(s[2], 1.0),
                    episodes=s[2], fps=self.T * self.N * world / (time.time() - t0),
                    noise_std=float(torch.exp(self.ac.pi.log_std[0])), lr=self.pi_opt.param_groups[0]["lr"],
                    total_env_steps=(self.epoch + 1) * self.T * self.N * world, time=self._t_total)
        self.log.append(info)
        self.epoch += 1
        return info

    def write_progress_csv(self, path):
        """The per-epoch log in the column names of the reference's progress.csv (utils/loggers.py;
        IWPGAlgorithm.log, algs/iwpg/iwpg.py:524-563) -- the subset of its columns this trainer tracks."""
        cols = [("Epoch", "epoch"), ("EpRet/Mean", "ep_ret"), ("EpLen/Mean", "ep_len"), ("Loss/Pi", "loss_pi"),
                ("Loss/Value", "loss_v"), ("Entropy", "entropy"), ("Misc/StopIter", "stop_iter"), ("PolicyRatio", "ratio"),
                ("LR", "lr"), ("Misc/ExplorationNoiseStd", "noise_std"), ("TotalEnvSteps", "total_env_steps"),
                ("Time", "time"), ("FPS", "fps")]
        with open(path, "w") as f:
            f.write(",".join(c for c, _ in cols) + "\n")
            for row in self.log:
                f.write(",".join(str(row.get(k, "")) for _, k in cols) + "\n")

    def learn(self, epochs=None, verbose=False):
        for _ in range(epochs or self.epochs):
            info = self.learn_one_epoch()
            if verbose:
                print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in info.items()})
        return self.ac, self.env


def train_runs_side_by_side(env_id, seeds, num_envs, rollout_len, epochs, threads=8, env_kwargs=None, trainer_kwargs=None):
    """Independent PPO runs for `seeds` on ONE GPU, `threads` of them at a time (one Python thread and one HIP stream per run) --
    the role of the reference's Benchmark loop over `num_runs` seeds, which spreads them over MPI cores (benchmark.py:60-130).
    A run at the reference's layout (one env, 32 000 steps per epoch) keeps one of the 256 CUs busy, so runs overlap almost for
    free: 12.4 s -> 1.9 s per 40-epoch run with 8 threads -- PROVIDED the HIP runtime has enough hardware queues: export
    GPU_MAX_HW_QUEUES=16 before the process first touches the GPU (the default of 4 serialises streams that share a queue:
    5.2 s per run).  On the FUSED path results are bit-identical to running the seeds one after the other (every random draw is
    keyed by the run's seed; only the construction of a trainer is serialised, because torch.manual_seed and the networks'
    initialisation use torch's global generator).  Where the networks fall back to PyTorch ops the update draws from torch's
    global generator (randperm, Normal.sample), so runs would depend on the threads' interleaving: that case runs the seeds one
    after the other.  Rollouts are never captured into a hipGraph here (a capture in one thread is invalidated by allocations
    and launches of the others).  -> {seed: PPOTrainer.log (list of per-epoch dicts)}"""
    import threading
    from .envs import make
    env_kwargs, trainer_kwargs = dict(env_kwargs or {}), dict(trainer_kwargs or {})
    trainer_kwargs["graph_rollout"] = False
    build, pick, serial = threading.Lock(), threading.Lock(), threading.Lock()
    pending, logs, errors = list(seeds), {}, []

    def worker():
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                while True:
                    with pick:
                        if not pending or errors:
                            return
                        seed = pending.pop(0)
                    env = None
                    try:
                        with build:
                            env = make(env_id, num_envs=num_envs, seed=seed, **env_kwargs)
                            tr = PPOTrainer(env, rollout_len=rollout_len, epochs=epochs, seed=seed, **trainer_kwargs)
                            torch.cuda.current_stream().synchronize()
                        if tr.fused:
                            tr.learn()
                        else:  # torch's global generator is in play: one run at a time, re-seeded as a lone run would be
                            with serial:
                                torch.manual_seed(seed)
                                tr.learn()
                        torch.cuda.current_stream().synchronize()
                        logs[seed] = tr.log
                    finally:
                        if env is not None:
                            env.close()
        except BaseException as e:  # noqa: BLE001  (re-raised in the caller's thread)
            errors.append(e)

    ts = [threading.Thread(target=worker) for _ in range(max(1, min(int(threads), len(pending))))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errors:
        raise errors[0]
    return logs
