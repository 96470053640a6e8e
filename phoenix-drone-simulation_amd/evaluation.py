"""Batched counterpart of EnvironmentEvaluator (utils/evaluation.py:15-117): one evaluation episode per
env of the batch, all in lockstep on the device.

The reference runs `num_evaluations` episodes one after the other (`eval_once`: reset, act
deterministically until terminated or truncated, sum reward and info['cost']) and writes one value
per line to `returns.csv` / `costs.csv`.  Here every env of the vector env plays exactly one
episode: the accumulators of an env freeze at its first `terminated | truncated`, and the loop ends
after `max_episode_steps` steps at the latest (TimeLimit)."""
import os

import torch


def _as_policy(policy):
    if hasattr(policy, "step") and hasattr(policy, "pi"):  # ActorCritic: deterministic in eval mode
        return lambda obs: policy.step(obs)[0]
    return policy


@torch.no_grad()
def evaluate(env, policy, log_dir=None, log_costs=True):
    """-> (returns [N], ep_lengths [N], costs [N]) float32 CPU tensors, one episode per env.
    `policy`: ActorCritic (evaluated with exploration noise off), JsonPolicy or any obs -> action
    callable on device tensors."""
    was_training = getattr(policy, "training", False)
    if hasattr(policy, "eval"):
        policy.eval()  # disable exploration noise (evaluation.py:63)
    act = _as_policy(policy)
    n, dev = env.num_envs, env.device
    ret = torch.zeros(n, device=dev); cost = torch.zeros(n, device=dev); length = torch.zeros(n, device=dev)
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    obs, _ = env.reset()
    for _ in range(env._max_episode_steps):
        obs, r, term, trunc, info = env.step(act(obs).contiguous())
        ret += torch.where(alive, r, torch.zeros_like(r))
        cost += torch.where(alive, info["cost"], torch.zeros_like(r))
        length += alive.float()
        alive &= ~(term | trunc)
    if hasattr(policy, "train") and was_training:
        policy.train()  # back to train mode (evaluation.py:91)
    ret, length, cost = ret.cpu(), length.cpu(), cost.cpu()
    if log_dir is not None:
        os.makedirs(log_dir, exist_ok=True)
        with open(os.path.join(log_dir, "returns.csv"), "w") as f:
            f.write("\n".join(str(float(x)) for x in ret) + "\n")
        if log_costs:
            with open(os.path.join(log_dir, "costs.csv"), "w") as f:
                f.write("\n".join(str(float(x)) for x in cost) + "\n")
    return ret, length, cost


@torch.no_grad()
def get_batch(env, policy, steps):
    """TrajectoryGenerator.get_batch (utils/trajectory_generator.py:84-118) for the whole batch:
    X[t] = the observation the policy acted on (standardised when `policy` carries scaling
    parameters, as `obs_rms(x)` there), Y[t] = the observation the step returned -- for an env that
    finished at t that is its terminal observation (`final_obs`), and X[t+1] its reset observation,
    exactly as the reference resets after appending y.  Returns (X, Y) of shape [steps, N, D]."""
    if hasattr(policy, "eval"):
        policy.eval()
    act = _as_policy(policy)
    n, d, dev = env.num_envs, env.obs_dim, env.device
    X = torch.empty(steps, n, d, device=dev); Y = torch.empty(steps, n, d, device=dev)
    mean, std, eps = getattr(policy, "mean", None), getattr(policy, "std", None), getattr(policy, "eps", 0.0)
    obs, _ = env.reset()
    for t in range(steps):
        X[t] = (obs - mean) / (std + eps) if mean is not None else obs
        obs, r, term, trunc, info = env.step(act(obs).contiguous())
        done = (term | trunc).unsqueeze(-1)
        Y[t] = torch.where(done, info["final_obs"], obs)
    return X, Y
