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


# ---------------------------------------------------------------------------------------------------------------------
# Stand-ins for the other two objects the reference's training scripts build themselves (experiments/gym/train_gym.py:84,
# experiments/dmc/train_dmc_from_pixels.py:62,88): a numpy replay buffer and an augmentation sequence, again carrying
# ONLY the attribute names of the reference's classes (replay.py:10-28,140-155; augmentations.py:20-24,176-181,220-231,
# 489-492).  The adoption logic (super_sac_amd/adopt.py) matches augmentations by class NAME, hence the names below.
import numpy as np


class _Tree:
    def __init__(self, values):
        self._value = values
        self._capacity = len(values) // 2


class _NumpyStorage:
    def __init__(self, size, s, a, r, s1, d):
        n = len(a)
        self.size, self._next_idx, self._max_filled = size, n % size, n
        self.action_stack = np.zeros((size,) + a.shape[1:], np.float32)
        self.reward_stack = np.zeros((size, 1), np.float32)
        self.done_stack = np.zeros((size, 1), np.uint8)
        self.action_stack[:n], self.reward_stack[:n], self.done_stack[:n] = a, r.reshape(n, 1), d.reshape(n, 1)
        self.s_stack, self.s1_stack, self.s_dtypes = {}, {}, {}
        for k in s:
            self.s_dtypes[k] = s[k].dtype
            self.s_stack[k] = np.zeros((size,) + s[k].shape[1:], s[k].dtype)
            self.s1_stack[k] = np.zeros((size,) + s[k].shape[1:], s[k].dtype)
            self.s_stack[k][:n], self.s1_stack[k][:n] = s[k], s1[k]

    def __len__(self):
        return self._max_filled


class ForeignReplayBuffer:
    """attribute-for-attribute what a reference ReplayBuffer holds after load_experience(s, a, r, s1, d)"""

    def __init__(self, size, s, a, r, s1, d, alpha=0.6, beta=1.0):
        self._maxsize, self.alpha, self.beta = size, alpha, beta
        self._storage = _NumpyStorage(size, s, np.asarray(a), np.asarray(r), s1, np.asarray(d))
        cap = 1
        while cap < size:
            cap *= 2
        n = len(self._storage)
        ssum, smin = np.zeros(2 * cap), np.full(2 * cap, np.inf)
        ssum[cap:cap + n] = smin[cap:cap + n] = 1.0 ** alpha   # every pushed row at the initial max priority
        for node in range(cap - 1, 0, -1):
            ssum[node] = ssum[2 * node] + ssum[2 * node + 1]
            smin[node] = min(smin[2 * node], smin[2 * node + 1])
        self._it_sum, self._it_min = _Tree(ssum), _Tree(smin)
        self._max_priority = 1.0
        self.total_sample_calls = 0

    def __len__(self):
        return len(self._storage)


class Drqv2Aug:
    def __init__(self, batch_size, pad=4, noise=False):
        self.batch_size, self.pad, self.noise = batch_size, pad, noise
        self.shift = torch.zeros(batch_size, 1, 1, 2, dtype=torch.int64)


class DrqNoNoiseAug:
    def __init__(self, batch_size, pad=4, noise=False):
        self.batch_size, self.pad, self.noise = batch_size, pad, noise
        self.w1 = torch.zeros(batch_size, dtype=torch.int64)
        self.h1 = torch.zeros(batch_size, dtype=torch.int64)
        self.pad_func = nn.ReflectionPad2d(pad)


class IdentityAug:
    def __init__(self, batch_size):
        self.batch_size = batch_size


class ForeignAugmentationSequence:
    def __init__(self, aug_list, keys=None):
        self.aug_list, self.keys = aug_list, keys
