"""Checkpoint + TRUE resume.

``Agent.save`` / ``Agent.load`` keep the reference's on-disk layout -- one ``state_dict`` file per module
(agent.py:172-202: encoder.pt, critic{i}.pt, actor{i}.pt, popart{i}.pt) -- so a checkpoint written here loads
into the reference and vice versa (PopArt statistics are a registered buffer here; the reference silently drops
them, popart.py:11-16).  What the reference never saves, and a run needs to continue bit-for-bit, goes beside
those files in ``engine_state.pt``:

  * the target agent's modules                         (``target/`` sub-directory, same layout)
  * Adam state of every optimizer the engine stepped   (device control block: step count, bias corrections;
                                                        first/second moments in the arenas' layout)
  * ``log_alpha`` tensors
  * host RNG streams (torch CPU, torch device, numpy, Python ``random``) and the log-ring position
  * optionally the replay buffer: storage arrays, write cursor, PER trees, sample counter

``load_training_state`` restores all of it in place; updates issued afterwards reproduce the uninterrupted run
exactly (tests/test_hip_checkpoint.py).
"""
import os
import random

import numpy as np
import torch

from . import engine
from . import learning_utils as lu

STATE_FILE = "engine_state.pt"


def _optimizer_state(opt):
    grp = getattr(opt, "_ssac_adam", None)
    out = {"torch": opt.state_dict()}
    if grp is not None:
        out["ctl"] = grp.ctl.dev.cpu()
        out["moments"] = {k: (m.cpu(), v.cpu()) for k, (m, v) in grp.moments.items()}
    return out


def _load_optimizer_state(opt, st, device):
    opt.load_state_dict(st["torch"])
    if "ctl" in st:
        grp = engine.adam_group(opt, device)
        grp.ctl.dev.copy_(st["ctl"])
        for k, (m, v) in st["moments"].items():
            cur = grp.moments.get(k)
            if cur is None or cur[0].numel() != m.numel():
                grp.moments[k] = (m.to(device).clone(), v.to(device).clone())
            else:  # keep addresses (recorded launch lists point at them)
                cur[0].copy_(m)
                cur[1].copy_(v)


def buffer_state(buffer):
    st = buffer._storage
    out = {"maxsize": buffer._maxsize, "total_sample_calls": buffer.total_sample_calls, "storage": None,
           "per": {"sum": buffer._per.sum_tree.copy(), "min": buffer._per.min_tree.copy(),
                   "max_priority": buffer._per._max_priority,
                   "max_priority_is_f32": bool(getattr(buffer._per, "_max_priority_is_f32", False))}}
    if st is not None:
        out["storage"] = {"action": st.action_stack.cpu(), "reward": st.reward_stack.cpu(),
                          "done": st.done_stack.cpu(), "s": {k: v.cpu() for k, v in st.s_stack.items()},
                          "s1": {k: v.cpu() for k, v in st.s1_stack.items()}, "next_idx": st._next_idx,
                          "max_filled": st._max_filled}
    return out


def load_buffer_state(buffer, state):
    from .replay import ReplayBufferStorage, _IndexStager
    assert state["maxsize"] == buffer._maxsize, "checkpointed buffer has a different capacity"
    buffer.total_sample_calls = state["total_sample_calls"]
    if hasattr(buffer._per, "_max_priority_is_f32"):
        buffer._per.load_state(state["per"]["sum"], state["per"]["min"], state["per"]["max_priority"],
                               bool(state["per"].get("max_priority_is_f32", False)))
    else:
        buffer._per.load_state(state["per"]["sum"], state["per"]["min"], state["per"]["max_priority"])
    s = state["storage"]
    if s is None:
        return
    if buffer._storage is None:
        ex_s = {k: v[0].numpy() for k, v in s["s"].items()}
        buffer._storage = ReplayBufferStorage(buffer._maxsize, ex_s, s["action"][0].numpy(), buffer.device)
        buffer._stager = _IndexStager(buffer.device)
    st = buffer._storage
    st.action_stack.copy_(s["action"])
    st.reward_stack.copy_(s["reward"])
    st.done_stack.copy_(s["done"])
    for k in s["s"]:
        st.s_stack[k].copy_(s["s"][k])
        st.s1_stack[k].copy_(s["s1"][k])
    st._next_idx, st._max_filled = s["next_idx"], s["max_filled"]


def save_training_state(path, agent, target_agent=None, optimizers=None, log_alphas=None, buffer=None):
    """optimizers: dict name -> torch optimizer (or list of optimizers, e.g. the per-member alpha optimizers)."""
    os.makedirs(path, exist_ok=True)
    torch.cuda.synchronize()
    agent.save(path)
    if target_agent is not None:
        os.makedirs(os.path.join(path, "target"), exist_ok=True)
        target_agent.save(os.path.join(path, "target"))
    dev = next(agent.actors[0].parameters()).device
    state = {"optimizers": {}, "log_alphas": None, "buffer": None}
    for name, opt in (optimizers or {}).items():
        state["optimizers"][name] = ([_optimizer_state(o) for o in opt] if isinstance(opt, (list, tuple))
                                     else _optimizer_state(opt))
    if log_alphas is not None:
        state["log_alphas"] = [la.detach().cpu() for la in log_alphas]
    if buffer is not None:
        state["buffer"] = buffer_state(buffer)
    state["rng"] = {"torch_cpu": torch.get_rng_state(), "torch_dev": torch.cuda.get_rng_state(dev),
                    "numpy": np.random.get_state(), "python": random.getstate()}
    state["log_ring"] = lu.ring_for(dev).k
    state["noise"] = {"agent": agent.__dict__.get("_ssac_noise")}
    torch.save(state, os.path.join(path, STATE_FILE))


def load_training_state(path, agent, target_agent=None, optimizers=None, log_alphas=None, buffer=None):
    agent.load(path)
    if target_agent is not None:
        target_agent.load(os.path.join(path, "target"))
    dev = next(agent.actors[0].parameters()).device
    state = torch.load(os.path.join(path, STATE_FILE), map_location="cpu", weights_only=False)
    for name, opt in (optimizers or {}).items():
        st = state["optimizers"][name]
        if isinstance(opt, (list, tuple)):
            for o, s in zip(opt, st):
                _load_optimizer_state(o, s, dev)
        else:
            _load_optimizer_state(opt, st, dev)
    if log_alphas is not None:
        for la, saved in zip(log_alphas, state["log_alphas"]):
            la.data.copy_(saved)
    if buffer is not None and state["buffer"] is not None:
        load_buffer_state(buffer, state["buffer"])
    r = state["rng"]
    torch.set_rng_state(r["torch_cpu"])
    torch.cuda.set_rng_state(r["torch_dev"], dev)
    np.random.set_state(r["numpy"])
    random.setstate(r["python"])
    lu.ring_for(dev).k = state["log_ring"]
    if state.get("noise", {}).get("agent") is not None:
        agent.__dict__["_ssac_noise"] = list(state["noise"]["agent"])
    # The chained actor update tags its hand-off granules with (update number + 1) of the noise stream just restored.  A
    # load that REWINDS the counter inside one process would let granules of the earlier pass carry a tag the counter
    # reaches again: zero the hand-off buffers (tag 0 is never used), so a consumer can only ever accept this pass's values.
    ws = agent.__dict__.get("_ssac_ws")
    for key, buf in (getattr(ws, "_bufs", None) or {}).items():
        if "handoff" in key[0]:
            buf.zero_()
    torch.cuda.synchronize()
    return state
