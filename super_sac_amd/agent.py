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
        # agent.py:112-127: models of the Markov state-abstraction update (learning.py:266-341)
        self.inverse_model = (nets.DiscreteInverseModel(**actor_kwargs) if discrete
                              else nets.ContinuousInverseModel(**actor_kwargs))
        self.contrastive_model = nets.ContrastiveModel(state_size=encoder.embedding_dim, hidden_size=hidden_size)

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
        yield self.inverse_model
        yield self.contrastive_model

    def to(self, dev):
        for i, a in enumerate(self.actors):
            self.actors[i] = a.to(dev)
        self.encoder = self.encoder.to(dev)
        for i, p in enumerate(self.popart):
            if p:
                self.popart[i] = p.to(dev)
        for i, c in enumerate(self.critics):
            self.critics[i] = c.to(dev)
        self.inverse_model = self.inverse_model.to(dev)
        self.contrastive_model = self.contrastive_model.to(dev)

    def eval(self):
        for m in self._modules():
            m.eval()

    def train(self):
        for m in self._modules():
            m.train()

    # ---- acting path on the engine's weights (agent.py:204-327) ----------------------
    def _process_obs(self, obs, num_envs=1):
        dev = next(self.actors[0].parameters()).device
        unsq = (lambda t: t.unsqueeze(0)) if num_envs == 1 else (lambda t: t)
        return {k: unsq(torch.from_numpy(v)).float().to(dev) for k, v in obs.items()}

    def _process_act(self, act, num_envs=1):
        act = act.squeeze(0) if num_envs == 1 else act
        if not self.discrete:
            act.clamp_(-1.0, 1.0)
        return act.cpu().numpy()

    def _state_rep(self, obs, rolling):
        return (self.encoder.forward_rolling(obs) if rolling else self.encoder.forward(obs)).contiguous()

    def _mean_action(self, actor, out):
        """dist.mean of a continuous actor: tanh(mu) (SquashedNormal.mean, distributions.py:99-104) or the
        deterministic actor's tanh(out) (mlps.py:91-92), by the det-action kernel without noise."""
        from ._lib import check, lib
        n, A = out.shape[0], actor.action_size
        act = torch.empty(n, A, device=out.device)
        check(lib.ssac_det_action_fwd(out.data_ptr(), out.shape[1], 0, 0.0, 0, 0.0, 0.0, n, A,
                                      act.data_ptr(), A, 0, engine.stream()))
        return act

    def _sample(self, actor, out):
        """dist.sample() of one actor on the rows of `out` (device generator for the noise / the draw)."""
        from ._lib import check, lib
        n = out.shape[0]
        if self.discrete:
            return torch.multinomial(torch.softmax(out, dim=-1), 1).squeeze(-1)  # Categorical.sample()
        A = actor.action_size
        act = torch.empty(n, A, device=out.device)
        if getattr(actor, "dist_impl", None) == "deterministic":
            return self._mean_action(actor, out)  # ContinuousDeterministic.sample() = loc
        eps = rng.draw_normal((n, A), out.device)
        check(lib.ssac_tanh_normal_fwd(out.data_ptr(), 2 * A, eps.data_ptr(), n, A, float(actor.log_std_low),
                                       float(actor.log_std_high), act.data_ptr(), A, 0, 0, engine.stream()))
        return act

    def forward(self, state, from_cpu=True, num_envs=1, rolling=False):
        """greedy action: mean over the ensemble's actors of dist.mean (continuous) or argmax of the mean
        action probabilities (discrete), agent.py:204-246."""
        engine.require_gpu()
        if from_cpu:
            # one C call per step where the agent allows it (identity encoder, fused-kernel shapes): acting.py
            from . import acting
            fast = acting.act(self, state, num_envs, sample=False, rolling=rolling)
            if fast is not None:
                return fast[0]
            state = self._process_obs(state, num_envs=num_envs)
        s_rep = self._state_rep(state, rolling)
        outs = [actor.raw_forward(s_rep) for actor in self.actors]
        if self.discrete:
            probs = torch.stack([torch.softmax(o, dim=-1) for o in outs], dim=0).mean(0)
            act = torch.argmax(probs, dim=-1, keepdim=True)
        else:
            act = torch.stack([self._mean_action(a, o) for a, o in zip(self.actors, outs)], dim=0).mean(0)
        if from_cpu:
            act = self._process_act(act, num_envs=num_envs)
        return act

    def sample_action(self, obs, from_cpu=True, num_envs=1, return_dist=False, rolling=False):
        """exploration action (agent.py:248-315): a random actor's sample, or with ucb_bonus > 0 the SUNRISE
        rule -- one candidate per actor, argmax over candidates of mean_c Q_c + bonus * std_c Q_c, the critics'
        values coming from one ensemble-Q launch per member on the stacked candidates."""
        engine.require_gpu()
        if from_cpu:
            from . import acting
            fast = acting.act(self, obs, num_envs, sample=True, return_dist=return_dist, rolling=rolling)
            if fast is not None:
                return fast if return_dist else fast[0]
            obs = self._process_obs(obs, num_envs)
        s_rep = self._state_rep(obs, rolling)
        n = s_rep.shape[0]
        if self.ucb_bonus > 0:
            outs = [actor.raw_forward(s_rep) for actor in self.actors]
            cands = torch.stack([self._sample(a, o) for a, o in zip(self.actors, outs)], dim=0)
            dist_out = rng.choice(outs)  # `random.choice(act_dists)`: consumed for logging only
            E = len(self.actors)
            if self.discrete:
                # q of the specific candidate action of every actor, for every critic member
                q_all = torch.stack([critic(s_rep) for critic in self.critics], dim=0)       # (Ec, n, A)
                q = torch.stack([q_all.gather(-1, cands[a].view(1, n, 1).expand(len(self.critics), n, 1))
                                 for a in range(E)], dim=1).squeeze(-1)                      # (Ec, Ea, n)
            else:
                x = torch.cat((s_rep.unsqueeze(0).expand(E, n, s_rep.shape[1]), cands), dim=-1)
                x = x.reshape(E * n, -1).contiguous()
                q = torch.stack([critic(x).view(E, n) for critic in self.critics], dim=0)     # (Ec, Ea, n)
            ucb = q.mean(0) + self.ucb_bonus * q.std(0)
            best = torch.argmax(ucb, dim=0)                                                   # (n,)
            act = cands[best, torch.arange(n, device=cands.device)]
            if self.discrete:
                act = act.unsqueeze(-1)
        else:
            actor = rng.choice(self.actors)
            dist_out = actor.raw_forward(s_rep)
            act = self._sample(actor, dist_out)
            if self.discrete:
                act = act.unsqueeze(-1)
        if from_cpu:
            act = self._process_act(act, num_envs)
        if return_dist:
            return act, dist_out  # the chosen actor's raw head output (distribution parameters)
        return act

    # The reference's on-disk layout, file for file (agent.py:172-202): a directory written here loads into the
    # reference's Agent.load and vice versa (oracle/check_reference_compat.py).  The reference's PopArt layer has no
    # parameters or buffers, so its popart{i}.pt is an EMPTY state_dict and the statistics are silently lost
    # (popart.py:11-16); here they go into popart{i}_stats.pt beside it.
    def save(self, path):
        torch.save(self.encoder.state_dict(), os.path.join(path, "encoder.pt"))
        for i, p in enumerate(self.popart):
            if p:
                torch.save(p.state_dict(), os.path.join(path, f"popart{i}.pt"))
                torch.save(p.stats_dict(), os.path.join(path, f"popart{i}_stats.pt"))
        for i, c in enumerate(self.critics):
            torch.save(c.state_dict(), os.path.join(path, f"critic{i}.pt"))
        for i, a in enumerate(self.actors):
            torch.save(a.state_dict(), os.path.join(path, f"actor{i}.pt"))
        torch.save(self.inverse_model.state_dict(), os.path.join(path, "inverse.pt"))
        torch.save(self.contrastive_model.state_dict(), os.path.join(path, "contrastive.pt"))

    def load(self, path):
        _load = lambda name: torch.load(os.path.join(path, name), map_location=_default_device)
        self.encoder.load_state_dict(_load("encoder.pt"))
        for i, p in enumerate(self.popart):
            if p:
                sd = _load(f"popart{i}.pt")
                if "state" in sd:
                    # checkpoint of an earlier revision of this package: the statistics were a persistent buffer
                    p.load_stats_dict({"state": sd["state"]})
                else:
                    p.load_state_dict(sd)
                if os.path.exists(os.path.join(path, f"popart{i}_stats.pt")):  # (absent in reference checkpoints)
                    p.load_stats_dict(_load(f"popart{i}_stats.pt"))
        for i, c in enumerate(self.critics):
            c.load_state_dict(_load(f"critic{i}.pt"))
        for i, a in enumerate(self.actors):
            a.load_state_dict(_load(f"actor{i}.pt"))
        for name, model in (("inverse.pt", self.inverse_model), ("contrastive.pt", self.contrastive_model)):
            if os.path.exists(os.path.join(path, name)):  # (absent in checkpoints of earlier revisions of this package)
                model.load_state_dict(_load(name))
        engine.sync_shadows(self)  # (bf16 mode: the shadows follow the freshly loaded masters)
