"""Adoption of a FOREIGN agent: an object built by the reference's own ``super_sac.Agent`` (agent.py:43-130) -- or
anything with the same attributes -- handed to this engine's update functions after ``super_sac_amd.install``.

The engine needs a handful of things the reference's classes do not carry; ``adopt_agent`` adds them in place, once:

  critics[i]        ``arena(dev)``: the packed-ensemble view of ``critics[i].nets`` (engine.MlpArena.adopt re-points
                    every ``nn.Parameter.data`` of fc1 / fc2 / out at it; the modules stay the source of truth)
  actors[i]         ``action_size`` (read off the head layer: fc3 -> out/2, out / act_p -> out)
  popart[i]         the reference layer keeps mu / nu / w / b as plain tensors (popart.py:8-20); it is replaced by
                    the device-resident ``popart.PopArtLayer`` initialised from those values
  encoder           ``ssac_identity_key`` when the encoder returns one of its inputs untouched (the scripts' identity
                    encoders, experiments/gym/train_gym.py:18-28), found by probing with the first batch
  adv_estimator     replaced by ``adv_estimator.AdvantageEstimator`` (same call signature, adv_estimator.py:82-90)
  act_space_size    read off the actor head

Nothing is copied: weights stay in the torch modules' (re-pointed) parameters, so ``agent.save / load``,
``copy.deepcopy(agent)`` and acting through the modules keep working.
"""
import types

import torch

from . import engine


def _head(module):
    return engine.MlpArena.linear_triples(module)[2]


def action_size(actor):
    a = getattr(actor, "action_size", None)
    if a is not None:
        return a
    out = _head(actor).out_features
    return out // 2 if hasattr(actor, "fc3") else out  # tanh-normal heads emit (mu, raw log std)


def _critic_arena(self, dev):
    return engine.bind_arena(self, "nets", list(self.nets), dev)


def adopt_agent(agent, device=None):
    """idempotent; returns the agent"""
    if agent.__dict__.get("_ssac_adopted"):
        return agent
    from . import popart as popart_mod
    from .adv_estimator import AdvantageEstimator
    if device is None:
        device = next(agent.actors[0].parameters()).device
    for actor in agent.actors:
        if not hasattr(actor, "action_size"):
            actor.action_size = action_size(actor)
    for critic in agent.critics:
        assert hasattr(critic, "nets"), "critics[i] must hold its Q-networks in `.nets` (agent.py:16-19)"
        if not hasattr(critic, "arena"):
            critic.arena = types.MethodType(_critic_arena, critic)
        if not hasattr(critic, "num_critics"):
            critic.num_critics = len(critic.nets)
    for i, p in enumerate(agent.popart):
        if p and not hasattr(p, "ptr"):
            mine = popart_mod.PopArtLayer(beta=float(p.beta), min_steps=int(p.min_steps), init_nu=float(p.nu))
            st = mine._read()
            st.mu, st.nu, st.w, st.b = float(p.mu), float(p.nu), float(p.w), float(p.b)
            st.t, st.stable = int(p._t), int(bool(p._stable))
            mine._write(st)
            agent.popart[i] = mine.to(device)
    if not hasattr(agent, "act_space_size"):
        agent.act_space_size = action_size(agent.actors[0])
    if not hasattr(agent.adv_estimator, "evaluate"):
        agent.adv_estimator = AdvantageEstimator(
            agent, discrete=bool(agent.discrete),
            discrete_method=getattr(agent.adv_estimator, "discrete_method", "indirect"),
            continuous_method=getattr(agent.adv_estimator, "cont_method", "mean"))
    agent.__dict__["_ssac_adopted"] = True
    return agent


def probe_identity(encoder, obs_dict):
    """identity encoders hand one of their inputs back untouched; remember which (cached on the encoder)"""
    if "ssac_identity_key" in encoder.__dict__ or getattr(encoder, "ssac_identity_key", None) is not None:
        return getattr(encoder, "ssac_identity_key")
    if encoder.__dict__.get("_ssac_probed"):
        return None
    encoder.__dict__["_ssac_probed"] = True
    from . import conv_encoder
    if conv_encoder.find_conv_module(encoder) is not None:
        return None  # a pixel encoder: runs on the HIP convolution engine
    with torch.no_grad():
        out = encoder(obs_dict)
    for key, val in obs_dict.items():
        if isinstance(out, torch.Tensor) and out.data_ptr() == val.data_ptr() and out.shape == val.shape:
            encoder.__dict__["ssac_identity_key"] = key
            return key
    return None
