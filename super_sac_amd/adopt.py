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

The other two objects the training scripts build with the reference's classes are adopted the same way, in place:

  augmenter         a reference ``AugmentationSequence`` (augmentations.py:20-41) whose ``aug_list`` holds DrQ-family /
                    identity augmentations (:165-293, :489-503, matched by class NAME along the MRO): every element and
                    the sequence itself change class to this package's; batch size, pad, noise flag and the CURRENT
                    randomisation (``shift`` / ``w1, h1``) are kept, no host draw is consumed
  buffer            a reference ``ReplayBuffer`` (replay.py:140-190: numpy ``ReplayBufferStorage`` + float64 segment
                    trees): the filled rows move into an HBM-resident storage, the trees' arrays become the
                    ``PrioritySampler``'s, counters carry over, and the object becomes a ``replay.ReplayBuffer`` --
                    later ``buffer.push(...)`` calls of the collection loop land on the device ring
"""
import types

import numpy as np
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


# ------------------------------------------------------------------------------------------ augmenters
_AUG_NAMES = ("Drqv2Aug", "DrqNoNoiseAug", "LargeDrqNoNoiseAug", "LargeDrqAug", "DrqAug", "IdentityAug")


def _own_aug_class(obj):
    from . import augmentations as A
    for klass in type(obj).__mro__:
        if klass.__name__ in _AUG_NAMES:
            return getattr(A, klass.__name__)
    return None


def adopt_augmenter(augmenter):
    """idempotent; returns the augmenter (class-swapped in place when it was built by the reference's classes)"""
    from . import augmentations as A
    if isinstance(augmenter, A.AugmentationSequence):
        return augmenter
    aug_list = getattr(augmenter, "aug_list", None)
    if aug_list is None:
        raise TypeError(f"{type(augmenter).__name__}: expected an AugmentationSequence (augmentations.py:20-41)")
    swaps = []
    for aug in aug_list:
        if isinstance(aug, (A._ShiftAug, A.IdentityAug)):
            swaps.append(None)
            continue
        mine = _own_aug_class(aug)
        if mine is None:
            raise NotImplementedError(
                f"augmentation {type(aug).__name__!r} has no HIP path; the update engine runs the DrQ family "
                f"({', '.join(_AUG_NAMES)}) -- the other augmentations of super_sac/augmentations.py are out of scope")
        swaps.append(mine)
    for aug, mine in zip(aug_list, swaps):
        if mine is None:
            continue
        aug.__dict__.pop("pad_func", None)   # (nn.ReflectionPad2d of the reference's DrqAug: the kernel pads itself)
        aug.__class__ = mine
        aug._shift_dev = None
        if not hasattr(aug, "noise"):
            aug.noise = False
        aug._adopt_state()
    augmenter.__class__ = A.AugmentationSequence
    return augmenter


# ------------------------------------------------------------------------------------------ replay buffers
def adopt_buffer(buffer, device=None):
    """idempotent; returns the buffer.  A reference-built (numpy) ReplayBuffer becomes a device-resident one in place."""
    from . import replay as R
    from . import device as default_device
    if isinstance(buffer, R.ReplayBuffer):
        return buffer
    need = ("_maxsize", "_storage", "alpha", "beta", "_it_sum", "_it_min", "_max_priority")
    missing = [n for n in need if not hasattr(buffer, n)]
    if missing:
        raise TypeError(
            f"{type(buffer).__name__} is neither a super_sac_amd.replay.ReplayBuffer nor a reference ReplayBuffer "
            f"(replay.py:140-190; missing {missing}).  Build the buffer after `super_sac_amd.install(super_sac)` -- "
            "it rebinds super_sac.replay.ReplayBuffer -- or pass a super_sac_amd.replay.ReplayBuffer.")
    dev = torch.device(device) if device is not None else torch.device(default_device)
    old = buffer._storage
    per = (R.DevicePrioritySampler(buffer._maxsize, buffer.alpha, buffer.beta, dev) if dev.type == "cuda"
           else R.PrioritySampler(buffer._maxsize, buffer.alpha, buffer.beta))
    sum_v, min_v = np.asarray(buffer._it_sum._value, np.float64), np.asarray(buffer._it_min._value, np.float64)
    assert sum_v.shape == (2 * per.cap,) == min_v.shape, "segment trees of an unexpected capacity"
    per.load_state(sum_v, min_v, buffer._max_priority)   # (an np.float32 stays one: it decides the power a push takes)
    storage = None
    if old is not None:
        n = int(old._max_filled)
        storage = R.ReplayBufferStorage(int(old.size), {k: v[0] for k, v in old.s_stack.items()}, old.action_stack[0], dev)
        for dst, src in ([(storage.s_stack[k], old.s_stack[k]) for k in old.s_stack]
                         + [(storage.s1_stack[k], old.s1_stack[k]) for k in old.s1_stack]
                         + [(storage.action_stack, old.action_stack), (storage.reward_stack, old.reward_stack),
                            (storage.done_stack, old.done_stack)]):
            if n:
                dst[:n].copy_(torch.from_numpy(np.ascontiguousarray(src[:n])).to(dst.dtype))
        storage._next_idx, storage._max_filled = int(old._next_idx), n
    calls = int(getattr(buffer, "total_sample_calls", 0))
    for name in ("_it_sum", "_it_min", "_max_priority"):
        buffer.__dict__.pop(name, None)
    buffer.__class__ = R.ReplayBuffer
    buffer._storage, buffer._per, buffer.device = storage, per, dev
    buffer._stager = R._IndexStager(dev) if storage is not None else None
    buffer.total_sample_calls = calls
    return buffer
