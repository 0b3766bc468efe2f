"""The acting path as ONE C call per environment step (agent.py:204-315; csrc/ssac_act.hip).

``Agent.forward`` / ``Agent.sample_action`` with a numpy observation of an identity-encoder agent whose networks fit the
fused MLP kernels take this path: the first call at a given (rule, num_envs) RECORDS the rule's launches -- every actor's
forward (+ tanh-normal sample from the engine's Philox stream, in the kernel), SUNRISE's ensemble-Q passes on the stacked
candidates, the rule's reduction (UCB arg-max / mean of mean actions / categorical draw / arg-max of mean probabilities) and
the publish step -- into a launch list of an ``ssac_act`` plan; every later call is ``ssac_act_run``: the observation is
written straight into device-visible memory, the list is re-issued, the action arrives in pinned host memory.  No torch op,
no hipMemcpy, no stream synchronisation.  Measured (bench.py ``secondary.acting``): see profiles/r6_acting.md.

What stays on the general path (agent.py's eager code, unchanged): pixel encoders, injected noise (a hook on
``rng.draw_normal`` -- the parity tests), discrete UCB, networks outside the fused kernels' shapes (hidden > 256),
``from_cpu=False`` callers.  Host RNG contract: the Python ``random`` draws of the reference (``random.choice`` of the
acting actor / of the logged distribution) are consumed exactly as before."""
import ctypes as C
import weakref

import numpy as np
import torch

from . import _lib, engine, rng
from . import learning_utils as lu
from ._lib import check, lib

ENABLED = True
_PLANS = weakref.WeakKeyDictionary()   # agent -> {(rule, num_envs, bonus): _Plan}
_SERIAL = [0]
_STREAM_SALT = 0x41C7A11D5EEDB00C      # the acting noise stream: the agent's engine seed under another key


class _Plan:
    def __init__(self, agent, rule, n, dev, pixel_shape=None):
        self.rule, self.n, self.dev = rule, n, dev
        self.S = agent.encoder.embedding_dim
        self.pixel_shape = pixel_shape    # (C, H, W) of a uint8 observation that goes through the pixel encoder, or None
        self.key = agent.encoder.ssac_identity_key if pixel_shape is None else getattr(agent.encoder, "ssac_obs_key", "obs")
        self.discrete = bool(agent.discrete)
        self.A = agent.act_space_size
        self.out_floats = n if self.discrete else n * self.A
        obs_bytes = 4 * n * self.S if pixel_shape is None else n * int(np.prod(pixel_shape))
        self.handle = lib.ssac_act_create(obs_bytes, self.out_floats)
        if not self.handle:
            raise RuntimeError("libssac_hip: " + lib.ssac_last_error().decode())
        self.obs_dev = lib.ssac_act_obs(self.handle)
        self.counter = lib.ssac_act_counter(self.handle)
        _SERIAL[0] += 1
        self.serial = _SERIAL[0]
        self.result = np.empty(self.out_floats, np.float32)
        self.bufs = []        # device tensors the recorded launches point at
        self.outs = []        # per actor: its head output (n x out_dim), what return_dist hands back
        self.sig = None
        self.lists = {}       # which-actor -> list index

    def buf(self, *shape):
        t = torch.zeros(*shape, device=self.dev)
        self.bufs.append(t)
        return t

    def rng_for(self, agent, member):
        seed = (lu.noise_stream(agent, self.dev)[0] ^ _STREAM_SALT) & (2 ** 64 - 1)
        return _lib.Rng(seed, self.counter, (self.serial << 48) + (member << 40))

    def __del__(self):
        try:
            if self.handle:
                lib.ssac_act_destroy(self.handle)
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass


def _signature(agent, with_critics):
    sig = [a.fc1.weight.data_ptr() for a in agent.actors]
    if not lu.is_identity(agent.encoder):
        conv = _conv_module(agent)
        sig.append(conv.conv1.weight.data_ptr() if conv is not None else 0)
    if with_critics:
        sig += [c.nets[0].fc1.weight.data_ptr() for c in agent.critics]
    return tuple(sig)


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _conv_module(agent):
    """the convolutional module inside the agent's encoder wrapper (found once per encoder object)"""
    enc = agent.encoder
    if "_ssac_conv_mod" not in enc.__dict__:
        from . import conv_encoder
        enc.__dict__["_ssac_conv_mod"] = conv_encoder.find_conv_module(enc)
    return enc.__dict__["_ssac_conv_mod"]


def _pixel_obs(agent, obs, num_envs):
    """(key, (C, H, W)) when the observation is a uint8 image batch for a pixel encoder of this package, else None"""
    key = getattr(agent.encoder, "ssac_obs_key", None)
    if key is None:
        return None
    conv = _conv_module(agent)
    v = obs.get(key)
    if conv is None or not isinstance(v, np.ndarray) or v.dtype != np.uint8:
        return None
    shape = tuple(v.shape[-3:])
    if len(shape) != 3 or v.size != num_envs * int(np.prod(shape)) or shape[0] != conv.conv1.in_channels:
        return None
    return key, shape


def _eligible(agent, obs, num_envs, sample, rolling):
    if not (ENABLED and isinstance(obs, dict) and engine.CAPTURE is None):
        return False
    if lu.is_identity(agent.encoder):
        v = obs.get(agent.encoder.ssac_identity_key)
        if not isinstance(v, np.ndarray) or v.size != num_envs * agent.encoder.embedding_dim:
            return False
    elif rolling or _pixel_obs(agent, obs, num_envs) is None:
        return False          # (a rolling encoder keeps state between calls: the general path)
    E = len(agent.actors)
    if E > 8 or agent.act_space_size > 64:
        return False
    kind = lu.actor_kind(agent.actors[0])
    if sample and kind == "stochastic" and not rng.normal_is_stock():
        return False          # injected noise: the general path draws it through the hook
    if sample and agent.ucb_bonus > 0 and (agent.discrete or kind != "stochastic" or E < 2):
        return False
    return True


def _arena_ok(agent, dev, with_critics):
    """every network the rule touches has a launchable forward: the fused MLP kernel, or -- networks outside its shapes, e.g.
    DrQv2's hidden-1024 actor -- the per-layer GEMM family (three launches); the UCB rule's sampling kernels are fused-only"""
    arenas = [engine.bind_arena(a, "self", [a], dev) for a in agent.actors]
    if with_critics:
        return all(a.fused for a in arenas) and all(c.arena(dev).fused for c in agent.critics)
    return all(a.fused for a in arenas) or lu.actor_kind(agent.actors[0]) != "stochastic"


def _fwd_desc(plan, desc, n_nets, fused, hidden, out_dim, x_ptr, ldx, n_rows, y):
    """y (n_nets x n_rows x out) = MLP(x) for every net of `desc` (shared input rows): one fused launch, or three per-layer
    launches through plan-owned activations (mlps.py:123-129)"""
    st = engine.stream()
    if fused:
        check(lib.ssac_mlp3_fwd_fused(C.byref(desc), 0, n_nets, x_ptr, ldx, 0, n_rows, 0, 0, y.data_ptr(), st))
        return
    h1, h2 = plan.buf(n_nets, n_rows, hidden), plan.buf(n_nets, n_rows, hidden)
    check(lib.ssac_mlp_layer_fwd(C.byref(desc), 0, 0, n_nets, x_ptr, ldx, 0, n_rows, h1.data_ptr(), hidden, n_rows * hidden, 1, st))
    check(lib.ssac_mlp_layer_fwd(C.byref(desc), 1, 0, n_nets, h1.data_ptr(), hidden, n_rows * hidden, n_rows, h2.data_ptr(), hidden,
                                 n_rows * hidden, 1, st))
    check(lib.ssac_mlp_layer_fwd(C.byref(desc), 2, 0, n_nets, h2.data_ptr(), hidden, n_rows * hidden, n_rows, y.data_ptr(), out_dim,
                                 n_rows * out_dim, 0, st))


def _fwd(plan, arena, x_ptr, ldx, n_rows, y):
    _fwd_desc(plan, arena.desc(), arena.n_nets, arena.fused, arena.hidden, arena.out_dim, x_ptr, ldx, n_rows, y)


class _Pack:
    """the members' networks of one role in ONE plan-owned arena: the members of an ensemble are separate allocations (one arena
    per actor, one per member's critics), so a forward of all of them is one launch PER MEMBER -- ~11 us each for a 16-row
    tile.  A plan copies them into a pack at the head of every call (one multi-tensor launch, a few MB: ~5 us) and runs ONE
    fused launch over the pack instead."""

    def __init__(self, plan, arenas):
        a0 = arenas[0]
        assert all((a.in_dim, a.hidden, a.out_dim, a.stride) == (a0.in_dim, a0.hidden, a0.out_dim, a0.stride) for a in arenas)
        self.n_nets = sum(a.n_nets for a in arenas)
        self.buf = plan.buf(self.n_nets * a0.stride)
        self.srcs = [a.params for a in arenas]
        self.desc = _lib.MlpDesc(self.buf.data_ptr(), a0.stride, self.n_nets, a0.in_dim, a0.hidden, a0.out_dim)
        self.out_dim, self.hidden, self.fused = a0.out_dim, a0.hidden, all(a.fused for a in arenas)

    def segments(self):
        off, out = 0, []
        for s in self.srcs:
            out.append((self.buf.data_ptr() + 4 * off, s.data_ptr(), s.numel()))
            off += s.numel()
        return out


def _pack_launch(packs):
    """one launch: every pack <- its members' arenas (ssac_polyak_multi with tau = 1: t <- 0 t + 1 s)"""
    segs = [sg for p in packs for sg in p.segments()]
    n = len(segs)
    t = (C.c_void_p * n)(*[s[0] for s in segs])
    s_ = (C.c_void_p * n)(*[s[1] for s in segs])
    cnt = (C.c_int64 * n)(*[s[2] for s in segs])
    check(lib.ssac_polyak_multi(t, s_, cnt, n, 1.0, engine.stream()))


def _record(plan, agent, which):
    """record the launches of plan.rule for acting actor `which` (None: the rule involves every actor)"""
    dev, n, S, A = plan.dev, plan.n, plan.S, plan.A
    st = engine.stream()
    kind = lu.actor_kind(agent.actors[0])
    E = len(agent.actors)
    arenas = [engine.bind_arena(a, "self", [a], dev) for a in agent.actors]
    if not plan.outs:
        plan.outs = [plan.buf(n, ar.out_dim) for ar in arenas]
        plan.res = plan.buf(max(plan.out_floats, 1))
        plan.logp = plan.buf(max(n * E, 1))
    check(lib.ssac_record_begin())
    try:
        x_in = plan.obs_dev      # what the actors read: the observation itself, or the encoder's output
        if plan.pixel_shape is not None:
            # the pixel encoder's forward straight from the uint8 observation buffer (the cast and the /255 normalisation
            # happen in the first layer's patch gather), into a buffer of this plan -- the same launches as lu.encode
            plan.srep = plan.buf(n, S)
            plan.conv_engine.forward_ptr(plan.obs_dev, (n,) + plan.pixel_shape, 1, plan.srep.data_ptr(), S, False)
            x_in = plan.srep.data_ptr()
        if plan.rule == "forward":
            if E > 1:   # one launch over the packed actors; plan.outs[e] are views of its output
                pa = _Pack(plan, arenas)
                packed_out = plan.buf(E, n, pa.out_dim)
                plan.outs = [packed_out[e] for e in range(E)]
                _pack_launch([pa])
                _fwd_desc(plan, pa.desc, E, pa.fused, pa.hidden, pa.out_dim, x_in, S, n, packed_out)
            else:
                _fwd(plan, arenas[0], x_in, S, n, plan.outs[0])
            if plan.discrete:
                check(lib.ssac_act_discrete(_ptr_array(plan.outs), E, A, n, A, 0, None, plan.res.data_ptr(), st))
            else:
                check(lib.ssac_act_mean_tanh(_ptr_array(plan.outs), E, arenas[0].out_dim, n, A, plan.res.data_ptr(), st))
        elif plan.rule == "ucb":
            # agent.py:262-300: a candidate per actor on the rows of x = [s | a_k] (E n rows), every member's critics on x,
            # mean + bonus * std over the members, arg-max over the candidates
            # Six launches whatever the ensemble size: the members' networks copied into two packs, ONE forward of all actors,
            # the candidates (tanh-normal samples from the engine's Philox stream, member e at offset e << 40 of the plan's
            # stream) beside the state columns, ONE ensemble-Q forward of every member's critics on the E n stacked rows, the
            # rule's reduction, the publish step
            x = plan.buf(E * n, S + A)
            c_arenas = [c.arena(dev) for c in agent.critics]
            pa, pc = _Pack(plan, arenas), _Pack(plan, c_arenas)
            packed_out = plan.buf(E, n, pa.out_dim)
            plan.outs = [packed_out[e] for e in range(E)]
            N = c_arenas[0].n_nets
            q = plan.buf(pc.n_nets, E * n, 1)
            _pack_launch([pa, pc])
            check(lib.ssac_mlp3_fwd_fused(C.byref(pa.desc), 0, E, x_in, S, 0, n, 0, 0, packed_out.data_ptr(), st))
            r = plan.rng_for(agent, 0)
            a0 = agent.actors[0]
            assert all((float(a_.log_std_low), float(a_.log_std_high)) == (float(a0.log_std_low), float(a0.log_std_high))
                       for a_ in agent.actors)
            check(lib.ssac_act_candidates(packed_out.data_ptr(), E, n, A, x_in, S, S, float(a0.log_std_low),
                                          float(a0.log_std_high), C.byref(r), 1 << 40, x.data_ptr(), S + A, st))
            check(lib.ssac_mlp3_fwd_fused(C.byref(pc.desc), 0, pc.n_nets, x.data_ptr(), S + A, 0, E * n, 0, 0, q.data_ptr(), st))
            qs = [q[c * N:(c + 1) * N] for c in range(len(c_arenas))]
            check(lib.ssac_ucb_select(_ptr_array(qs), len(qs), N, E, n, float(agent.ucb_bonus), x.data_ptr(), S + A, S, A,
                                      plan.res.data_ptr(), st))
        else:   # "sample": one actor's draw (agent.py:301-309)
            actor, ar, out = agent.actors[which], arenas[which], plan.outs[which]
            if plan.discrete:
                _fwd(plan, ar, x_in, S, n, out)
                r = plan.rng_for(agent, which)
                check(lib.ssac_act_discrete(_ptr_array([out]), 1, A, n, A, 1, C.byref(r), plan.res.data_ptr(), st))
            elif kind == "stochastic":
                # (tanh-normal samples lie inside (-1, 1): _process_act's clamp is the identity)
                r = plan.rng_for(agent, which)
                check(lib.ssac_actor_sample_fused(
                    C.byref(ar.desc()), x_in, S, n, 0, float(actor.log_std_low), float(actor.log_std_high),
                    plan.res.data_ptr(), A, 0, plan.logp.data_ptr(), 0, 0, out.data_ptr(), C.byref(r), st))
            else:   # deterministic actor: sample() = loc = tanh(out) (distributions.py:107-114)
                _fwd(plan, ar, x_in, S, n, out)
                check(lib.ssac_act_mean_tanh(_ptr_array([out]), 1, ar.out_dim, n, A, plan.res.data_ptr(), st))
        check(lib.ssac_act_publish(plan.handle, plan.res.data_ptr(), plan.out_floats, st))
    finally:
        lst = lib.ssac_record_end()
    # (the recording pass issued the launches too -- on whatever the observation buffer held -- and advanced the device-side
    #  call counter: ssac_act_add_list drains the device and re-reads the count)
    idx = lib.ssac_act_add_list(plan.handle, lst)
    if idx < 0:
        raise RuntimeError("libssac_hip: " + lib.ssac_last_error().decode())
    plan.lists[which] = idx


def act(agent, obs, num_envs, sample, return_dist=False, rolling=False):
    """the fast path's answer -- (action as numpy, dist_out or None) -- or None when the call is not eligible"""
    if not _eligible(agent, obs, num_envs, sample, rolling):
        return None
    dev = next(agent.actors[0].parameters()).device
    ucb = bool(sample and agent.ucb_bonus > 0)
    rule = "ucb" if ucb else ("sample" if sample else "forward")
    plans = _PLANS.setdefault(agent, {})
    pkey = (rule, num_envs, float(agent.ucb_bonus) if ucb else 0.0)
    plan = plans.get(pkey)
    sig = _signature(agent, ucb)
    if plan is None or plan.sig != sig:
        if not _arena_ok(agent, dev, ucb):
            return None
        if len(plans) > 12:
            plans.clear()
        pix = None if lu.is_identity(agent.encoder) else _pixel_obs(agent, obs, num_envs)
        eng = None
        if pix is not None:
            from . import conv_encoder
            eng = conv_encoder.conv_engine(agent.encoder, dev)   # (the module's own engine: its parameters live in its arena)
            if eng is None:
                return None
        plan = plans[pkey] = _Plan(agent, rule, num_envs, dev, pixel_shape=None if pix is None else pix[1])
        plan.conv_engine = eng
        plan.sig = _signature(agent, ucb)   # (binding the arenas may have re-pointed the parameters)
    # the reference's host draws, in its order: random.choice(act_dists) under UCB (for the logged distribution),
    # random.choice(self.actors) otherwise (agent.py:262, 301)
    which = None
    if rule == "ucb":
        which_dist = rng.choice(range(len(agent.actors)))
    elif rule == "sample":
        which = which_dist = rng.choice(range(len(agent.actors)))
    if which not in plan.lists:
        _record(plan, agent, which)
    v = np.ascontiguousarray(obs[plan.key], dtype=np.float32 if plan.pixel_shape is None else np.uint8)
    rc = lib.ssac_act_run(plan.handle, plan.lists[which], v.ctypes.data, v.nbytes, plan.result.ctypes.data, plan.out_floats,
                          engine.stream())
    if rc:
        raise RuntimeError("libssac_hip: " + lib.ssac_last_error().decode())
    n, A = num_envs, plan.A
    if plan.discrete:
        out = plan.result.astype(np.int64).reshape(n, 1)
    else:
        out = plan.result.reshape(n, A).copy()
    if num_envs == 1:
        out = out[0]
    dist_out = None
    if return_dist:
        dist_out = plan.outs[which_dist].clone()   # the chosen actor's raw head output (distribution parameters)
    return out, dist_out
