"""Agent container mirroring super_sac/agent.py:13-130 (constructor arguments, attributes and
the ensemble layout).  The networks are parameter containers (nets.py); the engine packs each
ensemble member's critics into one arena at first use.
"""
import copy
import os

import torch
from torch import nn

from . import engine, nets, popart, rng
from . import device as _default_device


class Critic(nn.Module):
    """`num_critics` Q-networks of one ensemble member (agent.py:13-40)."""

    def __init__(self, critic_network_cls, critic_kwargs, num_critics):
        super().__init__()
        self.nets = nn.ModuleList([critic_network_cls(**critic_kwargs) for _ in range(num_critics)])
        self.features = None
        self.num_critics = num_critics

    def arena(self, dev):
        return engine.bind_arena(self, "nets", list(self.nets), dev)

    def forward(self, *args, subset=None, return_min=True):
        """Q(s,a) of a random subset (REDQ target) or of all nets, in ONE ensemble launch per
        layer instead of the reference's Python loop (agent.py:34)."""
        x = args[0] if len(args) == 1 else torch.cat(args, dim=-1)
        engine.require_gpu(x)
        x = x.contiguous().float()
        if subset is not None:
            assert subset > 0 and subset <= self.num_critics
            ids = rng.draw_subset(self.num_critics, subset)
        else:
            ids = list(range(self.num_critics))
        ar = self.arena(x.device)
        ws = self.__dict__.setdefault("_ssac_ws", engine.Workspace(x.device))
        idt = torch.tensor(ids, dtype=torch.int32, device=x.device)
        _, h2, y = engine.mlp_forward(ar, x, x.shape[1], 0, x.shape[0], ws, "fwd", net_ids=idt,
                                      n_sel=len(ids))
        self.features = h2.clone()
        if return_min:
            return y.min(0).values
        return tuple(y[k].clone() for k in range(len(ids)))


class Agent:
    def __init__(self, act_space_size, encoder, actor_network_cls, critic_network_cls, discrete=False,
                 ensemble_size=3, num_critics=2, ucb_bonus=0.0, hidden_size=256,
                 auto_rescale_targets=True, log_std_low=-10.0, log_std_high=2.0, adv_method=None,
                 beta_dist=False):
        assert hasattr(encoder, "embedding_dim")
        assert not beta_dist, "Beta policies are outside the accelerated path"
        actor_kwargs = {"state_size": encoder.embedding_dim, "action_size": act_space_size,
                        "hidden_size": hidden_size}
        critic_kwargs = dict(actor_kwargs)
        if not discrete:
            actor_kwargs.update({"log_std_low": log_std_low, "log_std_high": log_std_high,
                                 "dist_impl": "pyd"})
        self.encoder = encoder
        self.actors = [actor_network_cls(**actor_kwargs) for _ in range(ensemble_size)]
        self.critics = [Critic(critic_network_cls, critic_kwargs, num_critics)
                        for _ in range(ensemble_size)]
        self.ensemble_size = ensemble_size
        self.num_critics = num_critics
        self.popart = ([popart.PopArtLayer() for _ in range(ensemble_size)]
                       if auto_rescale_targets else [False for _ in range(ensemble_size)])
        self.discrete = discrete
        self.ucb_bonus = ucb_bonus
        self.act_space_size = act_space_size
        # agent.py:102-122
        from .adv_estimator import AdvantageEstimator
        if discrete:
            self.adv_estimator = AdvantageEstimator(self, discrete=True,
                                                    discrete_method=adv_method if adv_method else "indirect")
        else:
            self.adv_estimator = AdvantageEstimator(self, discrete=False,
                                                    continuous_method=adv_method if adv_method else "mean")

    @property
    def ensemble(self):
        return zip(self.actors, self.critics)

    def _modules(self):
        yield self.encoder
        yield from self.actors
        yield from self.critics
        for p in self.popart:
            if p:
                yield p

    def to(self, dev):
        for i, a in enumerate(self.actors):
            self.actors[i] = a.to(dev)
        self.encoder = self.encoder.to(dev)
        for i, p in enumerate(self.popart):
            if p:
                self.popart[i] = p.to(dev)
        for i, c in enumerate(self.critics):
            self.critics[i] = c.to(dev)

    def eval(self):
        for m in self._modules():
            m.eval()

    def train(self):
        for m in self._modules():
            m.train()

    # same per-module files as agent.py:172-202 (inverse/contrastive models are out of scope)
    def save(self, path):
        torch.save(self.encoder.state_dict(), os.path.join(path, "encoder.pt"))
        for i, p in enumerate(self.popart):
            if p:
                torch.save(p.state_dict(), os.path.join(path, f"popart{i}.pt"))
        for i, c in enumerate(self.critics):
            torch.save(c.state_dict(), os.path.join(path, f"critic{i}.pt"))
        for i, a in enumerate(self.actors):
            torch.save(a.state_dict(), os.path.join(path, f"actor{i}.pt"))

    def load(self, path):
        _load = lambda name: torch.load(os.path.join(path, name), map_location=_default_device)
        self.encoder.load_state_dict(_load("encoder.pt"))
        for i, p in enumerate(self.popart):
            if p:
                p.load_state_dict(_load(f"popart{i}.pt"))
        for i, c in enumerate(self.critics):
            c.load_state_dict(_load(f"critic{i}.pt"))
        for i, a in enumerate(self.actors):
            a.load_state_dict(_load(f"actor{i}.pt"))
