"""Stand-ins with ONLY the attributes the reference's classes have (super_sac/agent.py:13-130, nets/mlps.py,
popart.py:8-20, experiments/gym/train_gym.py:18-28) -- no ``arena``, no ``action_size``, no device PopArt struct --
so the GPU box (where /root/reference does not exist) can still check that the update functions accept an agent
that was NOT built by super_sac_amd.Agent.  Written from the attribute lists alone; holds no arithmetic.
"""
import torch
from torch import nn


class _Mlp(nn.Module):
    def __init__(self, head, in_dim, hidden, out_dim):
        super().__init__()
        self.fc1 = nn.Linear(in_dim, hidden)
        self.fc2 = nn.Linear(hidden, hidden)
        setattr(self, head, nn.Linear(hidden, out_dim))


class ForeignCritic(_Mlp):
    def __init__(self, state_size, action_size, hidden_size, discrete):
        super().__init__("out", state_size if discrete else state_size + action_size, hidden_size,
                         action_size if discrete else 1)


class ForeignActor(_Mlp):
    def __init__(self, kind, state_size, action_size, hidden_size, lo, hi):
        head, out = {"stochastic": ("fc3", 2 * action_size), "deterministic": ("out", action_size),
                     "discrete": ("act_p", action_size)}[kind]
        super().__init__(head, state_size, hidden_size, out)
        self.log_std_low, self.log_std_high = lo, hi
        self.dist_impl = "deterministic" if kind == "deterministic" else "pyd"


class ForeignEnsemble(nn.Module):
    """agent.Critic: just `.nets`"""

    def __init__(self, nets):
        super().__init__()
        self.nets = nn.ModuleList(nets)


class ForeignPopArt:
    def __init__(self, beta=1e-4, min_steps=1000):
        self.mu, self.nu, self.w, self.b = torch.zeros(1), torch.zeros(1), torch.ones(1), torch.zeros(1)
        self.beta, self._t, self._stable, self.min_steps = beta, 1, False, min_steps

    def to(self, dev):
        return self


class ForeignEncoder(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.have_at_least_one_param = nn.Linear(1, 1)
        self._dim = dim

    @property
    def embedding_dim(self):
        return self._dim

    def forward(self, obs_dict):
        return obs_dict["obs"]


class ForeignAgent:
    def __init__(self, cfg, n_critics):
        kind = cfg["actor"]
        self.encoder = ForeignEncoder(cfg["obs"])
        self.actors = [ForeignActor(kind, cfg["obs"], cfg["act"], cfg["hidden"], cfg["lo"], cfg["hi"])
                       for _ in range(cfg["E"])]
        self.critics = [ForeignEnsemble([ForeignCritic(cfg["obs"], cfg["act"], cfg["hidden"], cfg["discrete"])
                                         for _ in range(n_critics)]) for _ in range(cfg["E"])]
        self.ensemble_size, self.num_critics = cfg["E"], n_critics
        self.popart = [ForeignPopArt() if cfg["popart"] else False for _ in range(cfg["E"])]
        self.discrete, self.ucb_bonus = cfg["discrete"], 0.0
        self.adv_estimator = object()

    @property
    def ensemble(self):
        return zip(self.actors, self.critics)

    def to(self, dev):
        self.encoder = self.encoder.to(dev)
        self.actors = [a.to(dev) for a in self.actors]
        self.critics = [c.to(dev) for c in self.critics]

    def train(self):
        pass
