"""Advantage estimator of the AFBC / AWAC actor update (adv_estimator.py:8-90), on the engine's kernels.

A(s, a) = Q(s, a) - V(s) with Q = min over ALL critics of the member (then ``popart(q)`` when the member
has a PopArt layer, adv_estimator.py:31-36) and, for continuous actions, V(s) = mean (``"mean"``) or max
(``"max"``) of Q over n = 4 actions sampled from the member's policy (adv_estimator.py:58-79).  The data
action and the 4 sampled actions are evaluated by ONE ensemble-Q launch on a stacked (5B x (S+A)) batch.
"""
import ctypes as C

import torch

from . import engine, rng
from . import learning_utils as lu
from ._lib import check, lib

N_SAMPLES = 4  # adv_estimator.py:58 (n=4)


class AdvantageEstimator:
    def __init__(self, agent, discrete_method="indirect", continuous_method="mean", discrete=False):
        assert continuous_method in ["mean", "max"]
        assert discrete_method in ["indirect", "direct"]
        self.agent = agent
        self.cont_method = continuous_method
        self.discrete = discrete
        self.discrete_method = discrete_method

    def __call__(self, obs, action, ensemble_idx):
        return self.forward(obs, action, ensemble_idx)

    def forward(self, obs, action, ensemble_idx):
        out = self.evaluate(obs, action, ensemble_idx)
        return out["adv"].view(-1, 1)

    def evaluate(self, obs, action, ensemble_idx, want=("adv",), log_ptr=0):
        """dict with the requested (B,) device tensors among adv / mask / prio (one launch chain)."""
        engine.require_gpu()
        agent, i = self.agent, ensemble_idx
        actor, critic, popart = agent.actors[i], agent.critics[i], agent.popart[i]
        s_rep = lu.encode(agent.encoder, obs)
        B, S = s_rep.shape
        dev = s_rep.device
        ws = lu.agent_ws(agent, dev)
        st = engine.stream()
        if self.discrete:
            if self.discrete_method != "indirect":
                raise NotImplementedError("dueling-architecture advantage (adv_estimator.py:37-39)")
            nA = agent.act_space_size
            E = len(agent.actors)
            logits = ws.get("adv.logits", (E, B, nA))
            for m, ac in enumerate(agent.actors):  # V(s) uses the mean probabilities of ALL ensemble actors
                arena = engine.bind_arena(ac, "self", [ac], dev)
                _, _, out = engine.mlp_forward(arena, s_rep, lu._row_stride(s_rep), 0, B, ws, f"adv.a{m}", save=False)
                logits[m].copy_(out[0])
            c_arena = critic.arena(dev)
            _, _, q = engine.mlp_forward(c_arena, s_rep, lu._row_stride(s_rep), 0, B, ws, f"adv.c{i}", save=False)
            res = {k_: ws.get(f"adv.{k_}{i}", (B,)) for k_ in want}
            check(lib.ssac_adv_filter_discrete(q.data_ptr(), c_arena.n_nets, B, nA, logits.data_ptr(), E,
                                               action.data_ptr(), action.stride(0), popart.ptr if popart else 0,
                                               res["adv"].data_ptr() if "adv" in res else 0,
                                               res["mask"].data_ptr() if "mask" in res else 0,
                                               res["prio"].data_ptr() if "prio" in res else 0, log_ptr, st))
            return res
        A = actor.action_size
        n = N_SAMPLES
        # stacked batch: block 0 = (s, a_data), blocks 1..n = (s, a_k)
        X = ws.get(f"adv.x{i}", ((n + 1) * B, S + A))
        Xv = X.view(n + 1, B, S + A)
        Xv[:, :, :S].copy_(s_rep.unsqueeze(0).expand(n + 1, B, S))   # device plumbing
        Xv[0, :, S:].copy_(action)
        a_arena = engine.bind_arena(actor, "self", [actor], dev)
        _, _, aout = engine.mlp_forward(a_arena, s_rep, lu._row_stride(s_rep), 0, B, ws, f"adv.a{i}", save=False)
        deterministic = lu.actor_kind(actor) == "deterministic"
        for k in range(n):
            blk = Xv[k + 1]
            if deterministic:
                # ContinuousDeterministic.sample() is its loc (distributions.py:113-114): no draw, n equal actions
                check(lib.ssac_det_action_fwd(aout.data_ptr(), A, 0, 0.0, 0, 0.0, 0.0, B, A, blk.data_ptr(), S + A, S, st))
                continue
            eps = rng.draw_normal((B, A), dev)  # dist.sample(): one normal draw per sampled action
            check(lib.ssac_tanh_normal_fwd(aout.data_ptr(), 2 * A, eps.data_ptr(), B, A,
                                           float(actor.log_std_low), float(actor.log_std_high),
                                           blk.data_ptr(), S + A, S, 0, st))
        c_arena = critic.arena(dev)
        _, _, q = engine.mlp_forward(c_arena, X, S + A, 0, (n + 1) * B, ws, f"adv.c{i}", save=False)
        res = {k_: ws.get(f"adv.{k_}{i}", (B,)) for k_ in want}
        check(lib.ssac_adv_filter(q.data_ptr(), c_arena.n_nets, B, n, popart.ptr if popart else 0,
                                  1 if self.cont_method == "max" else 0,
                                  res["adv"].data_ptr() if "adv" in res else 0,
                                  res["mask"].data_ptr() if "mask" in res else 0,
                                  res["prio"].data_ptr() if "prio" in res else 0, log_ptr, st))
        return res
