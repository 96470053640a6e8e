"""Policy JSON interchange with the reference / the CrazyFlie firmware format.

Format written by utils/export.py:23-80 (`dump_json`) and read by utils/utils.py:56-111,309-337
(`build_mlp_network`, `load_network_json`): {"check_sum": sum(net(ones)), "scaling_parameters":
[mean[D], std[D]] (the observation standardisation of ActorCritic.obs_oms, export.py:88-92),
"activation": name, "0".."k": {"type": "standard", "weights": [out][in], "biases": [out]}}.
The 153 trained policies bundled with the reference (experiments/07_*/models) use this format, so
policies trained on the GPU simulator can be flown on the firmware and vice versa.
Sparse `csrproduct` layers (utils/utils.py:78-98) are not supported."""
import json

import numpy as np
import torch
import torch.nn as nn

_ACT = {"relu": nn.ReLU, "tanh": nn.Tanh, "identity": nn.Identity, "sigmoid": nn.Sigmoid, "softplus": nn.Softplus}


class JsonPolicy(nn.Module):
    """Deterministic policy: net((obs - mean) / (std + eps)) -- the eval-mode path of
    ActorCritic.step (algs/core.py:370-393) for an exported actor."""

    def __init__(self, net, scaling_parameters, activation, eps=1e-5):
        super().__init__()
        self.net = net
        self.activation = activation
        sp = torch.as_tensor(np.asarray(scaling_parameters), dtype=torch.float32)
        self.register_buffer("mean", sp[0].clone())
        self.register_buffer("std", sp[1].clone())
        self.eps = eps

    @torch.no_grad()
    def forward(self, obs):
        return self.net((obs - self.mean) / (self.std + self.eps))


def load_network_json(path, check=True):
    """utils/utils.py:309-337; returns a JsonPolicy. Verifies `check_sum` (sum of net(ones))."""
    with open(path) as f:
        data = json.load(f)
    act = _ACT[data["activation"]]
    layers, i = [], 0
    while str(i) in data:
        entry = data[str(i)]
        if entry["type"] != "standard":
            raise NotImplementedError(f"layer type {entry['type']!r} (only 'standard' dense layers)")
        w = torch.tensor(entry["weights"], dtype=torch.float32)
        b = torch.tensor(entry["biases"], dtype=torch.float32).reshape(-1)
        lin = nn.Linear(w.shape[1], w.shape[0])
        lin.weight.data, lin.bias.data = w, b
        layers += [lin, act()]
        i += 1
    layers[-1] = nn.Identity()
    net = nn.Sequential(*layers)
    pol = JsonPolicy(net, data["scaling_parameters"], data["activation"])
    if check and "check_sum" in data:
        with torch.no_grad():
            s = float(net(torch.ones(pol.mean.shape[0])).sum())
        if not np.isclose(s, float(data["check_sum"]), rtol=1e-4, atol=1e-4):
            raise ValueError(f"check_sum mismatch: {s} vs {data['check_sum']}")
    return pol


def dump_json(activation, scaling_parameters, neural_network, path):
    """utils/export.py:23-80."""
    net = neural_network.cpu() if hasattr(neural_network, "cpu") else neural_network
    sp = np.asarray(scaling_parameters, dtype=np.float64)
    with torch.no_grad():
        out = net(torch.ones(sp.shape[1], dtype=torch.float32))
    data = {"check_sum": str(out.numpy().sum()), "scaling_parameters": sp.tolist(), "activation": activation}
    k = 0
    for layer in net:
        if isinstance(layer, nn.Linear):
            data[str(k)] = {"type": "standard", "weights": layer.weight.detach().cpu().numpy().tolist(),
                            "biases": layer.bias.detach().cpu().numpy().tolist()}
            k += 1
    with open(path, "w") as f:
        json.dump(data, f)
    return data


def convert_actor_critic_to_json(ac, path, activation="relu"):
    """utils/export.py:83-98: scaling parameters = obs_oms mean / std, network = pi.net."""
    sp = np.stack([ac.obs_oms.mean.detach().cpu().numpy(), ac.obs_oms.std.detach().cpu().numpy()])
    import copy
    return dump_json(activation, sp, copy.deepcopy(ac.pi.net).cpu(), path)
