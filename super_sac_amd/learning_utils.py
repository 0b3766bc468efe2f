"""Update helpers with the reference's names and signatures (super_sac/learning_utils.py):
``sample_move_and_augment`` (:174), ``compute_td_targets`` (:298), ``compute_backup_weights``
(:357), ``soft_update`` / ``hard_update`` (:160-167), ``GaussianExplorationNoise`` (:18-66).

Every arithmetic step is a kernel of libssac_hip.so; this file only sequences launches and
keeps the reference's host-RNG order (index draw -> augmentation draw -> action noise ->
REDQ subset).
"""
import ctypes as C
import os
import math

import numpy as np
import torch

from . import augmentations, engine, nets, rng
from . import _lib
from ._lib import check, lib

LOG_WIDTH = 64
MAX_MEMBERS = 8
# slot layout of one update's log block (floats)
L_CRITIC_LOSS, L_TD_ERR, L_CRITIC_GN, L_ENC_GN = 0, 1, 2, 3
L_TD0 = 4            # + 3*i : mean, std, entropy bonus of member i
L_BW = 28            # bellman weights mean, max, min, std
L_ACTOR_LOSS, L_ACTOR_GN = 32, 33
L_ALPHA0 = 34        # + 2*i : alpha_loss_i, alpha_i
L_ADVW, L_BC_TOTAL, L_BC_GN = 50, 51, 52   # AFBC: adv_weights_mean, overall loss, actor grad norm
L_BC0 = 53           # + i : filtered BC loss of member i
L_ENC_INV = 60       # encoder invariance constraint (learning_utils.py:401-409)
L_ACT_INV = 61       # action invariance constraint (learning_utils.py:272-285; not a log key of the reference)
# Markov state-abstraction update (its own log block): losses inverse / contrastive / smoothness / total, then the
# gradient norms of the contrastive model, the inverse model and the encoder; two scratch words for the head kernels
L_MK_LOSS, L_MK_GN_CON, L_MK_GN_INV, L_MK_GN_ENC, L_MK_RAW, L_MK_CON, L_MK_SMOOTH, L_MK_TMP = 40, 44, 45, 46, 47, 48, 49, 56


class LogRing:
    """Ring of device log blocks.  Update functions hand out 0-dim views of a block as log
    values, so nothing synchronises until the caller reads them (the reference blocks on
    ``.item()`` several times per update: learning.py:132, learning_utils.py:351-353)."""

    def __init__(self, device, slots=512):
        self.buf = torch.zeros(slots, LOG_WIDTH, dtype=torch.float32, device=device)
        self.k = 0

    def advance(self):
        self.k = (self.k + 1) % self.buf.shape[0]
        return self.k

    def next(self, adam=None):
        """fresh zeroed block; the same launch advances `adam`'s step when given."""
        self.k = (self.k + 1) % self.buf.shape[0]
        blk = self.buf[self.k]
        check(lib.ssac_begin_update(blk.data_ptr(), LOG_WIDTH, 0 if adam is None else adam.ctl.ptr, 0,
                                    engine.stream()))
        return blk


_rings = {}

# Deferred log finalisation of recorded updates (csrc/ssac_critic_logs.h): the newest recorded update's log block is
# written to its ring slot by the NEXT update's first launch -- or, when somebody looks at one of its values first, by a
# flush launch.  `pending` = (log-ring slot, ssac_deferred_logs struct) of that newest update, per device ring.
DEFERRED_LOGS = True


def flush_pending_logs(owner):
    """write the newest recorded update's log block to its ring slot now (owner: the recording's state object, which
    carries `pending` = that update's log-ring slot, and `deferred` = its ssac_deferred_logs struct)"""
    slot_i = owner.__dict__.get("pending")
    if slot_i is not None:
        owner.pending = None
        check(lib.ssac_deferred_logs_flush(C.byref(owner.deferred), slot_i, engine.stream()))


class LazyLog(torch.Tensor):
    """a log value of a recorded update: a 0-dim view of its slot of the log ring that makes sure the slot has been
    written before anything reads it (any torch operation on it, float(), .item(), ...)"""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        for a in args:
            if isinstance(a, LazyLog):
                owner = a.__dict__.get("_ssac_owner")
                if owner is not None and owner.__dict__.get("pending") == a.__dict__.get("_ssac_slot"):
                    flush_pending_logs(owner)
        return super().__torch_function__(func, types, args, kwargs or {})


def lazy_views(owner, ring, slot_i, index):
    """{log key: LazyLog view} of ring slot slot_i"""
    out = {}
    blk = ring.buf[slot_i]
    for k_, i in index.items():
        v = blk[i].as_subclass(LazyLog)
        v.__dict__["_ssac_owner"], v.__dict__["_ssac_slot"] = owner, slot_i
        out[k_] = v
    return out


def draw_normal(shape, device):
    """action-noise draw site (honours an active graph capture)."""
    if engine.CAPTURE is not None:
        return engine.CAPTURE.normals.pop(0)
    return rng.draw_normal(shape, device)


IN_KERNEL_NOISE = True


def noise_stream(agent, device):
    """[seed, critic-update draws so far, online-actor-update number] of the agent's engine noise stream (Philox4x32-10, include/ssac_hip.h: ssac_rng).  The seed
    is drawn once from torch's device generator, so `torch.manual_seed` still determines the run."""
    ns = agent.__dict__.get("_ssac_noise")
    if ns is None:
        seed = int(torch.randint(0, 2 ** 62, (1,), device=device, dtype=torch.int64))
        ns = agent.__dict__["_ssac_noise"] = [seed, 0, 0]
    if len(ns) < 3:   # (a checkpoint written before the actor updates numbered themselves)
        ns.append(0)
    return ns


def draw_subset(num_critics, k):
    if engine.CAPTURE is not None:
        return list(engine.CAPTURE.ids)
    return rng.draw_subset(num_critics, k)


def log_block(device, adam=None):
    if engine.CAPTURE is not None:
        cap = engine.CAPTURE
        blk = cap.logblk
        if cap.defer_begin and cap.feed and cap.pending_begin is None:
            cap.pending_begin = (blk, 0 if adam is None else adam.ctl.ptr)  # folded into the replay gather
            return blk
        check(lib.ssac_begin_update(blk.data_ptr(), LOG_WIDTH, 0 if adam is None else adam.ctl.ptr,
                                    cap.feed, engine.stream()))
        return blk
    return ring_for(device).next(adam)


def ring_for(device):
    ring = _rings.get(device)
    if ring is None:
        ring = _rings[device] = LogRing(device)
    return ring


_ones = {}


def unit_weight(device):
    """imp_weights of the uniform path: torch.ones(1) (learning_utils.py:180), allocated once."""
    t = _ones.get(device)
    if t is None:
        t = _ones[device] = torch.ones(1, device=device)
    return t


def agent_ws(agent, device):
    ws = agent.__dict__.get("_ssac_ws")
    if ws is None or ws.device != device:
        ws = agent.__dict__["_ssac_ws"] = engine.Workspace(device)
    return ws


class GaussianExplorationNoise:
    """learning_utils.py:18-66.  Inside the updates only ``current_scale`` is read; the noise
    itself is drawn on the device and applied by ``ssac_det_action_fwd``."""

    def __init__(self, action_space, start_scale=1.0, final_scale=0.1, steps_annealed=1000, eps=1e-6):
        assert start_scale >= final_scale
        self.action_space = action_space
        self.start_scale, self.final_scale = start_scale, final_scale
        self.steps_annealed = steps_annealed
        self.current_scale = start_scale
        self._scale_slope = (start_scale - final_scale) / steps_annealed
        self.eps = eps

    def sample(self, action, clip=None, update_schedule=False):
        assert isinstance(action, np.ndarray), "device-side noise is applied inside the update kernels"
        noise = self.current_scale * np.random.randn(*action.shape)
        if clip is not None:
            noise = np.clip(noise, -clip, clip)
        out = np.clip(action + noise, self.action_space.low + self.eps, self.action_space.high - self.eps)
        if update_schedule:
            self.current_scale = max(self.current_scale - self._scale_slope, self.final_scale)
        return out


# ------------------------------------------------------------------------------------------
def _polyak_tensor(t, s, tau):
    check(lib.ssac_polyak(t.data_ptr(), s.data_ptr(), t.numel(), float(tau), engine.stream()))


# soft_update right behind a recorded critic update: no launch -- the update's own weight-gradient launch, still
# waiting in the queue, applies the target update in its Adam epilogue (ssac_late_polyak in include/ssac_hip.h)
LATE_POLYAK = True


def soft_update(target, source, tau):
    """theta_bar <- (1-tau) theta_bar + tau theta over all parameters (learning_utils.py:160-162);
    one launch over the packed arena when both sides are packed ensembles."""
    plan = target.__dict__.get("_ssac_polyak")
    last = source.__dict__.pop("_ssac_last_step", None)  # the recorded update issued last on these critics, if any
    if plan is not None and plan[0] is source:
        # packed ensembles seen before: one C call (the arenas are re-validated by pointer every 256 calls)
        _, ta, sa, tmods, smods, calls = plan
        plan[5] = calls + 1
        if (calls & 255) or (ta.is_bound(tmods) and sa.is_bound(smods)):
            if (last is not None and LATE_POLYAK and last.late_arenas is not None and last.late_arenas[0] is ta
                    and last.late_arenas[1] is sa and last.gs.fast is last and last.gs.path == "fast"):
                rc = lib.ssac_step_polyak(last.handle, float(tau))
                if rc == 1:
                    return  # served (or certain to be) by the update's own weight-gradient launch
                if rc < 0:  # no decision within 2 ms (a stalled queue): settle it the slow way
                    torch.cuda.synchronize()
                    if lib.ssac_step_polyak_done(last.handle) == 1:
                        return
            if ta.shadow is not None:
                check(lib.ssac_bf16_polyak(C.byref(ta.desc()), C.byref(sa.desc()), float(tau), ta.shadow.data_ptr(),
                                           engine.stream()))
            else:
                check(lib.ssac_polyak(ta.params.data_ptr(), sa.params.data_ptr(), ta.params.numel(), float(tau),
                                      engine.stream()))
            return
        del target.__dict__["_ssac_polyak"]
    if hasattr(target, "arena") and hasattr(source, "arena"):
        dev = next(source.parameters()).device
        ta, sa = target.arena(dev), source.arena(dev)
        if ta.params.numel() == sa.params.numel():
            if ta.shadow is not None:  # bf16 mode: the target's shadow is refreshed by the same launch
                check(lib.ssac_bf16_polyak(C.byref(ta.desc()), C.byref(sa.desc()), float(tau), ta.shadow.data_ptr(),
                                           engine.stream()))
            else:
                _polyak_tensor(ta.params, sa.params, tau)
            target.__dict__["_ssac_polyak"] = [source, ta, sa, list(target.nets), list(source.nets), 1]
            return
    # any other module (a pixel encoder: conv / fc / norm tensors): all of its parameters in ONE launch
    pairs = list(zip(target.parameters(), source.parameters()))
    if not pairs:
        return
    n = len(pairs)
    tp_, sp_, cn_ = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_int64 * n)()
    for j, (tp, sp) in enumerate(pairs):
        if not (tp.is_cuda and tp.data.is_contiguous() and sp.data.is_contiguous() and tp.dtype == torch.float32
                and sp.dtype == torch.float32 and tp.numel() == sp.numel()):
            raise RuntimeError("soft_update expects contiguous float32 device parameters of equal size")
        tp_[j], sp_[j], cn_[j] = tp.data_ptr(), sp.data_ptr(), tp.numel()
    check(lib.ssac_polyak_multi(tp_, sp_, cn_, n, float(tau), engine.stream()))


def hard_update(target, source):
    soft_update(target, source, 1.0)


# ------------------------------------------------------------------------------------------
class _Batch:
    """device buffers of one sampled minibatch: xsa = [s | a], x1sa = [s' | (a' written later)]."""
    __slots__ = ("B", "S", "A", "xsa", "x1sa", "r", "d", "key", "pixel", "pending")


# actor sample -> target critics chained per workgroup, beside the critics' forward + unscaled backward: ONE launch
CHAIN_LAUNCH = True
# the chained launch in its producer / consumer form (the actor once per tile, the target critics start before a' exists);
# a module constant that tests flip to compare the two forms
CHAIN_PC = True
CHAIN_SPLIT = True   # ... with the target critics' fc2 split by columns over 2 / 4 consumer workgroups while the launch stays one round


def parallel_shard_of(agent):
    from . import parallel
    return parallel.shard_of(agent)


# the replay gather of a critic update rides in the merged actor / critic-forward launch (ssac_gather): no gather launch
FOLD_GATHER = True


def ensure_gathered(bt):
    """issue the replay gather of a batch whose gather was deferred (sample_move_and_augment(_defer_gather=True)) and
    has not been taken over by the merged launch"""
    if bt is None or getattr(bt, "pending", None) is None:
        return
    p, bt.pending = bt.pending, None
    S, A, B = p["S"], p["A"], bt.B
    if p["begin"] is not None:
        blk, ctl = p["begin"]
        check(lib.ssac_gather_transition_begin(p["s"], p["s1"], p["dtype"], S, p["act"], A, p["rew"], p["done"], B,
                                               bt.xsa.data_ptr(), S + A, bt.x1sa.data_ptr(), S + A, bt.r.data_ptr(),
                                               bt.d.data_ptr(), p["feed"], blk.data_ptr(), LOG_WIDTH, ctl,
                                               engine.stream()))
    else:
        check(lib.ssac_gather_transition(p["s"], p["s1"], p["dtype"], S, p["act"], A, p["rew"], p["done"],
                                         p["idx"].data_ptr(), B, bt.xsa.data_ptr(), S + A, bt.x1sa.data_ptr(), S + A,
                                         bt.r.data_ptr(), bt.d.data_ptr(), engine.stream()))


def gather_struct(bt):
    """the ssac_gather of a batch with a deferred gather (consumes it): handed to the merged launch"""
    p, bt.pending = bt.pending, None
    S, A = p["S"], p["A"]
    cap = engine.CAPTURE
    g = _lib.Gather(p["s"], p["s1"], p["act"], p["rew"], p["done"], S, A, p["idx"].data_ptr(), 0,
                    bt.xsa.data_ptr(), S + A, bt.x1sa.data_ptr(), S + A, bt.r.data_ptr(), bt.d.data_ptr(), 0, 0, -1, 0,
                    -1, 0)
    if p["begin"] is not None:
        blk, ctl = p["begin"]
        g.idx, g.feed, g.logs, g.n_logs, g.ctl = 0, p["feed"], blk.data_ptr(), LOG_WIDTH, ctl
        if cap is not None and cap.tick_ptr:
            g.rng_word = (cap.tick_ptr - cap.idx_dev.data_ptr()) // 4
        if cap is not None:
            g.ids_word = (cap.ids_dev.data_ptr() - cap.idx_dev.data_ptr()) // 4
    g._keep = (p["idx"],)
    return g


def _host_view(idx):
    """replay_dict["priority_idxs"]: numpy for a host-drawn index vector, the lazy host view of a device-drawn one"""
    return idx.numpy() if torch.is_tensor(idx) else idx


def sample_move_and_augment(buffer, batch_size, augmenter, aug_mix, per=True, _defer_gather=False,
                            _invariance=False):
    assert len(buffer) >= batch_size
    if not hasattr(buffer, "draw_uniform_indices") or not hasattr(augmenter, "single_shift"):
        # objects built by the reference's own classes (the shipped scripts construct them before main.super_sac
        # runs: experiments/gym/train_gym.py:84, experiments/dmc/train_dmc_from_pixels.py:62,88) are adopted in place
        from . import adopt
        adopt.adopt_buffer(buffer)
        adopt.adopt_augmenter(augmenter)
    st = buffer._storage
    dev = st.device
    if per:
        # replay.py:163-177: the uniforms from numpy's global generator on the host; total mass, prefix-sum descent and
        # importance weights on the float64 trees in HBM (csrc/ssac_per.hip) -- no synchronisation
        idx_cpu, idx, w = buffer.draw_per_indices(batch_size)
        imp_weights = w if torch.is_tensor(w) else torch.from_numpy(w).to(dev)  # float64 (B,), as the reference hands it over
    else:
        idx_cpu, idx = buffer.draw_uniform_indices(batch_size)
        imp_weights = unit_weight(dev)
    B = batch_size
    keys = list(st.s_stack.keys())
    A = int(np.prod(st.action_stack.shape[1:]))
    r = torch.empty(B, 1, device=dev)
    d = torch.empty(B, 1, device=dev)
    # one randomisation per call, shared by s and s' (augmentations.py:28-29)
    augmenter.change_randomization_params()
    bt = _Batch()
    bt.B, bt.A, bt.pending = B, A, None
    vec = len(keys) == 1 and st.s_stack[keys[0]].dim() == 2
    if vec:
        key = keys[0]
        S = st.s_stack[key].shape[1]
        xsa = torch.empty(B, S + A, device=dev)
        x1sa = torch.empty(B, S + A, device=dev)
        src = st.s_stack[key]
        cap = engine.CAPTURE
        begin = None
        if cap is not None and cap.pending_begin:
            begin, cap.pending_begin = cap.pending_begin, ()  # this gather also does ssac_begin_update's work
        bt.xsa, bt.x1sa, bt.r, bt.d = xsa, x1sa, r, d
        bt.pending = dict(s=src.data_ptr(), s1=st.s1_stack[key].data_ptr(), act=st.action_stack.data_ptr(),
                          rew=st.reward_stack.data_ptr(), done=st.done_stack.data_ptr(), idx=idx, S=S, A=A,
                          begin=begin, feed=cap.feed if begin is not None else 0,
                          dtype=1 if src.dtype == torch.uint8 else 0)
        if not (_defer_gather and FOLD_GATHER and src.dtype == torch.float32):
            ensure_gathered(bt)
        o, o1, a = {key: xsa[:, :S]}, {key: x1sa[:, :S]}, xsa[:, S:]
        bt.S, bt.xsa, bt.x1sa, bt.key, bt.pixel = S, xsa, x1sa, key, False
        assert augmenter.is_identity(), "image augmentations need image observations"
    else:
        shift_aug = augmenter.single_shift()
        k_aug = int(batch_size * aug_mix)
        o, o1 = {}, {}
        for key in keys:
            src, src1 = st.s_stack[key], st.s1_stack[key]
            if src.dim() == 4 and shift_aug is not None:
                n, (c, h, w) = B, src.shape[1:]
                assert h == w
                for dst_dict, s_arr in ((o, src), (o1, src1)):
                    out = torch.empty(B, c, h, w, device=dev)
                    nz = rng.draw_normal((B, c, h, w), dev) if shift_aug.noise else None
                    shift_aug.apply(s_arr, idx, B, c, h, k_aug, out, nz)
                    dst_dict[key] = out
            else:
                assert augmenter.is_identity(), "unsupported augmentation on the accelerated path"
                o[key] = st.gather_field(src, idx, B)
                o1[key] = st.gather_field(src1, idx, B)
        a = st.gather_field(st.action_stack, idx, B)
        if a.dim() < 2:
            a = a.unsqueeze(1)
        st.gather_field(st.reward_stack, idx, B, dst=r, ld=1)
        st.gather_field(st.done_stack, idx, B, dst=d, ld=1)
        bt.S, bt.xsa, bt.x1sa, bt.key, bt.pixel = None, None, None, None, True
        if _invariance:
            # the invariance constraints (learning_utils.py:272-285, 401-409) look at the FULLY augmented and the
            # un-augmented observation batch (same rows, same randomisation); where the mix already is one of them
            # (aug_mix 1 / 0) that tensor is shared
            ao, oo = {}, {}
            for key in keys:
                src = st.s_stack[key]
                if src.dim() == 4 and shift_aug is not None:
                    c, h, w = src.shape[1:]
                    for dst_dict, n_aug in ((ao, B), (oo, 0)):
                        if n_aug == k_aug:
                            dst_dict[key] = o[key]
                            continue
                        assert not shift_aug.noise, "the noise of DrqAug would have to be re-used across the passes"
                        out = torch.empty(B, c, h, w, device=dev)
                        shift_aug.apply(src, idx, B, c, h, n_aug, out, None)
                        dst_dict[key] = out
                else:
                    ao[key] = oo[key] = o[key]
            inv_obs = ((ao, None), (oo, None))
    bt.r, bt.d = r, d
    if _invariance and vec:
        inv_obs = ((o, None), (o, None))   # identity augmentation: augmented == original == the batch
    if _invariance:
        return {"primary_batch": (o, a, r, o1, d), "augmented_obs": inv_obs[0], "original_obs": inv_obs[1],
                "priority_idxs": _host_view(idx_cpu), "imp_weights": imp_weights, "_ssac": bt}
    return {"primary_batch": (o, a, r, o1, d), "augmented_obs": None, "original_obs": None,
            "priority_idxs": _host_view(idx_cpu), "imp_weights": imp_weights, "_ssac": bt}


# ------------------------------------------------------------------------------------------
def encode(encoder, obs_dict, dst=None, save=False):
    """state representation of the batch: identity encoders return the (strided) view; pixel
    encoders run the HIP conv engine (conv_encoder.py) and write into `dst[:, :emb]` when given."""
    key = getattr(encoder, "ssac_identity_key", None)
    if key is not None:
        return obs_dict[key]
    from . import conv_encoder
    img = obs_dict[getattr(encoder, "ssac_obs_key", "obs")]
    eng = conv_encoder.conv_engine(encoder, img.device)
    if eng is None:
        raise NotImplementedError(f"{type(encoder).__name__}: this encoder has no HIP path")
    if dst is None:
        dst = torch.empty(img.shape[0], eng.emb, device=img.device)
    eng.forward(img, dst, dst.stride(0), save)
    return dst[:, :eng.emb]


def is_identity(encoder):
    return getattr(encoder, "ssac_identity_key", None) is not None


def ensure_adopted(agent, buffer=None):
    """first contact with an agent: add what a foreign (reference-built) agent lacks (adopt.py) and find out whether
    its encoder is an identity map, probing it with one row of the buffer's observations"""
    if agent.__dict__.get("_ssac_adopted") and (buffer is None or agent.encoder.__dict__.get("_ssac_probed")
                                                 or is_identity(agent.encoder)):
        return
    from . import adopt
    adopt.adopt_agent(agent)
    if buffer is not None and not hasattr(buffer, "draw_uniform_indices"):
        adopt.adopt_buffer(buffer, next(agent.actors[0].parameters()).device)
    st = getattr(buffer, "_storage", None) if buffer is not None else None
    if st is not None and not is_identity(agent.encoder):
        adopt.probe_identity(agent.encoder, {k: v[:1].float() for k, v in st.s_stack.items()})


def _row_stride(t):
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major (B, F) device tensor"
    return t.stride(0)


def _concat_buffer(ws, tag, s_rep, A):
    """(B, S+A) buffer whose first S columns hold s_rep (copy is device plumbing)."""
    B, S = s_rep.shape
    buf = ws.get(tag, (B, S + A))
    buf[:, :S].copy_(s_rep)
    return buf


def actor_kind(actor):
    if isinstance(actor, nets.DiscreteActor) or hasattr(actor, "act_p"):
        return "discrete"
    if getattr(actor, "dist_impl", None) == "deterministic":
        return "deterministic"
    return "stochastic"


def _upload_ids(ws, ids, device, tag):
    if engine.CAPTURE is not None:
        return engine.CAPTURE.ids_dev
    stager = ws.__dict__.setdefault("_stager", None)
    if stager is None:
        from .replay import _IndexStager
        stager = ws.__dict__["_stager"] = _IndexStager(device)
    return stager.upload(torch.tensor(ids, dtype=torch.int32), tag=tag)


def compute_td_targets(logs, replay_dict, agent, target_agent, ensemble_idx, ensemble_n, log_alphas,
                       pop, gamma, random_process, noise_clip, discrete=False, _slot=None, _defer=False,
                       _co_forward=None, _co_backward=None, _log_idx=None):
    """learning_utils.py:298-354.  With ``_defer`` (critic_update's fused path; continuous actions, no PopArt)
    the final elementwise step -- and its three log values -- is not launched here: the returned ``td`` buffer
    carries a ``_ssac_spec`` (ssac_td_spec) and the critic launch evaluates the targets into it.
    ``_co_forward`` = (critic arena, X, ldx, h1, h2, q): when the actor runs as the fused sample launch, the online
    critics' forward rides in the SAME launch (ssac_actor_sample_critic_fwd); replay_dict["_co_fwd"] is then True.
    ``_co_backward`` = (critic arena, h1, h2, act, ld_act, dz2u, dz1u): after such a forward, the TD-independent half
    of the critics' backward pass rides in the target critics' launch (ssac_target_fwd_critic_bwdu);
    replay_dict["_co_bwd"] is then True."""
    o, a, r, o1, d = replay_dict["primary_batch"]
    i = ensemble_idx
    dev = r.device
    ws = agent_ws(agent, dev)
    slot = _slot if _slot is not None else log_block(dev)
    actor = agent.actors[i]
    popart = agent.popart[i]
    log_alpha = log_alphas[i]
    B = r.shape[0]
    kind = actor_kind(actor)
    st = engine.stream()

    x1_pre = None
    if not is_identity(target_agent.encoder) and kind != "discrete":
        emb = target_agent.encoder.embedding_dim
        x1_pre = ws.get(f"td.x1.{i}", (B, emb + actor.action_size))
    s1_rep = encode(target_agent.encoder, o1, dst=x1_pre)
    S = s1_rep.shape[1]
    a_arena = engine.bind_arena(actor, "self", [actor], dev)
    fuse_sample = kind == "stochastic" and a_arena.fused and random_process is None
    # actor -> target critics chained inside one launch with the critics' forward + TD-independent backward
    chain = ({} if (CHAIN_LAUNCH and fuse_sample and _co_forward is not None and _co_backward is not None
                    and (parallel_shard_of(target_agent) is None or engine.CAPTURE is not None)
                    and target_agent.critics[i].arena(dev).fused_dbuf
                    and target_agent.critics[i].arena(dev).out_dim == 1) else None)
    if not (fuse_sample and _co_forward is not None):
        ensure_gathered(replay_dict.get("_ssac"))  # (a deferred replay gather rides in the merged launch only)
    if not fuse_sample:
        _, _, aout = engine.mlp_forward(a_arena, s1_rep, _row_stride(s1_rep), 0, B, ws, f"td.a{i}",
                                        save=False)
    t_arena = target_agent.critics[i].arena(dev)
    from . import parallel
    shard = parallel.shard_of(target_agent)
    N = t_arena.n_nets if shard is None else shard.num_critics  # the subset is drawn over the GLOBAL ensemble
    assert 0 < ensemble_n <= N
    bt = replay_dict.get("_ssac")
    logp = ws.get(f"td.logp{i}", (B,))
    use_entropy = 0
    if kind == "discrete":
        ids = draw_subset(N, ensemble_n)
        q1, n_q = _subset_q(ws, shard, t_arena, ids, s1_rep, _row_stride(s1_rep), B, dev, f"td.c{i}")
        lp_ptr, qd = aout.data_ptr(), t_arena.out_dim
        a_s1 = None
    else:
        A = actor.action_size
        if bt is not None and bt.x1sa is not None and s1_rep.data_ptr() == bt.x1sa.data_ptr():
            x1 = bt.x1sa
        elif x1_pre is not None:
            x1 = x1_pre
        else:
            x1 = _concat_buffer(ws, f"td.x1.{i}", s1_rep, A)
        if kind == "stochastic":
            if fuse_sample and IN_KERNEL_NOISE and rng.normal_is_stock():
                # the noise comes from the agent's Philox stream inside the launch: draw number = host count (eager)
                # or capture-time count + the device-resident update counter (recorded launch list)
                ns = noise_stream(agent, dev)
                cap = engine.CAPTURE
                rs = _lib.Rng(ns[0], cap.tick_ptr, cap.noise_offset) if cap is not None else _lib.Rng(ns[0], 0, ns[1])
                if cap is None:
                    ns[1] += 1
                _actor_sample(a_arena, s1_rep, B, 0, actor, x1, S, A, logp, C.addressof(rs), st, _co_forward,
                              replay_dict, chain=chain, rng_keep=rs)
                eps = None
            else:
                eps = draw_normal((B, A), dev)
            if eps is None:
                pass
            elif fuse_sample:
                # actor forward + sample + log pi: ONE launch, a' lands in the [s'|a'] buffer
                _actor_sample(a_arena, s1_rep, B, eps.data_ptr(), actor, x1, S, A, logp, 0, st, _co_forward,
                              replay_dict, chain=chain, rng_keep=eps)
            else:
                check(lib.ssac_tanh_normal_fwd(aout.data_ptr(), 2 * A, eps.data_ptr(), B, A,
                                               float(actor.log_std_low), float(actor.log_std_high),
                                               x1.data_ptr(), S + A, S, logp.data_ptr(), st))
            use_entropy = 1
            if random_process is not None:
                # exploration noise on the sampled action replaces the entropy term (learning_utils.py:331-335)
                noise = draw_normal((B, A), dev)
                check(lib.ssac_exploration_noise(x1.data_ptr(), S + A, S, noise.data_ptr(),
                                                 float(random_process.current_scale),
                                                 float(noise_clip) if noise_clip is not None else 0.0, B, A, st))
                use_entropy = 0
        else:
            if random_process is not None:
                noise = draw_normal((B, A), dev)
                check(lib.ssac_det_action_fwd(aout.data_ptr(), A, 0, 0.0, noise.data_ptr(),
                                              float(random_process.current_scale),
                                              float(noise_clip) if noise_clip is not None else 0.0, B, A,
                                              x1.data_ptr(), S + A, S, st))
            else:
                # no exploration process: a' = loc, and the entropy term is the log-density of Normal(loc, 1e-4) at its
                # own mean (learning_utils.py:336-338 on distributions.py:107-114)
                check(lib.ssac_det_action_fwd(aout.data_ptr(), A, 0, 0.0, 0, 0.0, 0.0, B, A, x1.data_ptr(), S + A, S, st))
                check(lib.ssac_det_logprob(0, B, A, logp.data_ptr(), st))
                use_entropy = 1
        ids = draw_subset(N, ensemble_n)
        cob = _co_backward if replay_dict.get("_co_fwd") else None
        q1, n_q = _subset_q(ws, shard, t_arena, ids, x1, S + A, B, dev, f"td.c{i}", co_backward=cob,
                            replay_dict=replay_dict)
        lp_ptr, qd = logp.data_ptr(), 1
        a_s1 = x1[:, S:]
    td = torch.empty(B, 1, device=dev)
    if _defer and qd == 1 and not popart:
        td._ssac_spec = _lib.TdSpec(q1.data_ptr(), lp_ptr, r.data_ptr(), d.data_ptr(), log_alpha.data_ptr(),
                                    td.data_ptr(), float(gamma), n_q, use_entropy, int(getattr(q1, "_ssac_parts", 1)))
        td._ssac_logs = slot[L_TD0 + 3 * i:]
        td._ssac_keep = (q1, logp, r, d, log_alpha)
    else:
        assert getattr(q1, "_ssac_parts", 1) == 1, "partial target Q needs the in-launch TD target (ssac_td_spec.n_parts)"
        check(lib.ssac_td_target(q1.data_ptr(), n_q, B, qd, lp_ptr, r.data_ptr(), d.data_ptr(),
                                 log_alpha.data_ptr(), use_entropy, float(gamma),
                                 popart.ptr if popart else 0, 1 if (popart and pop) else 0,
                                 td.data_ptr(), slot[L_TD0 + 3 * i:].data_ptr(), st))
    li = i if _log_idx is None else _log_idx   # (member-sharded ranks: the member's GLOBAL index names its logs)
    logs[f"td_targets/mean_td_target_{li}"] = slot[L_TD0 + 3 * i]
    logs[f"td_targets/std_td_target_{li}"] = slot[L_TD0 + 3 * i + 1]
    logs[f"td_targets/entropy_bonus_{li}"] = slot[L_TD0 + 3 * i + 2]
    replay_dict["_subset"] = ids
    replay_dict["_x1"] = None if kind == "discrete" else x1  # [s' | a'] as fed to the target critics (DR3)
    if kind == "discrete":
        a_s1 = aout[0]  # logits; the reference returns probs here, only used by dr3
    return td, (s1_rep, a_s1)


def _actor_sample(a_arena, s1_rep, B, eps_ptr, actor, x1, S, A, logp, rng_ptr, st, co_forward, replay_dict,
                  chain=None, rng_keep=None):
    """actor forward + tanh-normal sample + log pi in ONE launch (a' lands in the [s'|a'] buffer), optionally with
    the online critics' forward as extra workgroups of the same launch."""
    bt = replay_dict.get("_ssac")
    if co_forward is not None and chain is not None:
        # the whole TD-independent part of the update is ONE launch (ssac_chain_update), issued by _subset_q once the
        # REDQ subset is known: nothing is launched here
        chain.update(a_arena=a_arena, s1_rep=s1_rep, eps_ptr=eps_ptr, actor=actor, x1=x1, S=S, A=A, logp=logp,
                     rng_ptr=rng_ptr, rng_keep=rng_keep, co_forward=co_forward)
        replay_dict["_chain"] = chain
        replay_dict["_co_fwd"] = True
        return
    if co_forward is not None:
        c_arena, X, ldx, h1, h2, q = co_forward
        gth = None
        if (bt is not None and bt.pending is not None and x1.data_ptr() == bt.x1sa.data_ptr()
                and X.data_ptr() == bt.xsa.data_ptr()):
            gth = gather_struct(bt)  # the workgroups fetch their rows from the replay arrays themselves
        else:
            ensure_gathered(bt)
        with engine._timed("dual_fwd") as tm:
            for _ in range(tm.reps):  # 1, except under bench.py's live kernel timing (the launch is idempotent)
                check(lib.ssac_actor_sample_critic_fwd(
                    C.byref(a_arena.desc()), s1_rep.data_ptr(), _row_stride(s1_rep), B, eps_ptr,
                    float(actor.log_std_low), float(actor.log_std_high), x1.data_ptr(), S + A, S, logp.data_ptr(),
                    rng_ptr, C.byref(c_arena.desc()), X.data_ptr(), ldx, h1.data_ptr(), h2.data_ptr(),
                    q.data_ptr(), C.byref(gth) if gth is not None else 0, st))
        replay_dict["_co_fwd"] = True
        return
    ensure_gathered(bt)
    check(lib.ssac_actor_sample_fused(C.byref(a_arena.desc()), s1_rep.data_ptr(), _row_stride(s1_rep), B, eps_ptr,
                                      float(actor.log_std_low), float(actor.log_std_high), x1.data_ptr(), S + A, S,
                                      logp.data_ptr(), 0, 0, 0, rng_ptr, st))


# critic-sharded ranks: the exchange of the subset's target Q rides in the chained launch (a tail workgroup behind its
# target-critic workgroups) instead of being a launch of its own behind it; a module constant that tests flip to compare
FUSE_XCHG = True


def _launch_chain(ch, co_backward, t_arena, ids_ptr, n, ws, tag, B, replay_dict, allow_split=True, xchg=None):
    """ssac_chain_update: the deferred actor sample (ch, from _actor_sample), the target critics of the n subset slots
    and the online critics' forward + TD-independent backward, ONE launch; returns the target outputs (n, B, 1)"""
    c_arena, h1, h2, act, ld_act, dz2u, dz1u = co_backward
    _, Xc, ldxc, _, _, qc = ch["co_forward"]
    bt = replay_dict.get("_ssac")
    x1, S, A, actor = ch["x1"], ch["S"], ch["A"], ch["actor"]
    gth = None
    if (bt is not None and bt.pending is not None and x1.data_ptr() == bt.x1sa.data_ptr()
            and Xc.data_ptr() == bt.xsa.data_ptr() and bt.pending["dtype"] == 0):
        gth = gather_struct(bt)
    else:
        ensure_gathered(bt)
    # column-split target critics (producer / consumer form, fp32 family): every (slot, tile) gets `splits` consumer
    # workgroups and the slot's value arrives as that many partial sums -- read through ssac_td_spec.n_parts
    # (inside a recording the hand-off's tag must change per replay: it then comes from the input ring's update counter,
    # which the launch reaches through the folded gather or the deferred-log struct -- a recording with neither keeps the
    # one-workgroup form)
    cap_ = engine.CAPTURE
    pc_ok = CHAIN_PC and (cap_ is None or (gth is not None and gth.feed) or (cap_.feed and cap_.deferred is not None))
    splits = 1
    if pc_ok and CHAIN_SPLIT and allow_split and c_arena.shadow is None:
        splits = int(lib.ssac_chain_target_splits(C.byref(ch["a_arena"].desc()), C.byref(t_arena.desc()),
                                                  C.byref(c_arena.desc()), B, n))
    q1 = ws.get(tag + ".y", (n * splits, B, 1))
    q1._ssac_parts = splits
    s1_rep = ch["s1_rep"]
    cap = engine.CAPTURE
    dl_ptr = 0
    if cap is not None and cap.feed and cap.deferred is not None:
        # recorded update: one extra workgroup of this launch writes the PREVIOUS update's log block to its ring slot
        dl_ptr = C.addressof(cap.deferred)
        cap.deferred_chain = True
    if c_arena.shadow is not None:
        # bf16-operand mode: the same launch on v_mfma_f32_32x32x16_bf16, fed from the arenas' bf16 shadows
        a_arena = ch["a_arena"]
        assert a_arena.shadow is not None and t_arena.shadow is not None, \
            "bf16 mode: set_precision must cover the actor and the target agent too"
        bf = c_arena.bf_buffers(ws, "cu", B)
        # producer / consumer form (csrc/ssac_bf16.hip, bf_chain_pc_kernel), as below for the fp32 family
        ho = ws.get(tag + ".handoff", (B * A,), dtype=torch.int64, zero=True) if (pc_ok and A <= 8) else None
        with engine._timed("chain") as tm:
            for _ in range(tm.reps):
                check(lib.ssac_bf16_chain_update(
                    C.byref(a_arena.desc()), a_arena.shadow.data_ptr(), s1_rep.data_ptr(), _row_stride(s1_rep), B,
                    ch["eps_ptr"], float(actor.log_std_low), float(actor.log_std_high), x1.data_ptr(), S + A, S,
                    ch["logp"].data_ptr(), ch["rng_ptr"], C.byref(t_arena.desc()), t_arena.shadow.data_ptr(), ids_ptr, n,
                    q1.data_ptr(), C.byref(c_arena.desc()), c_arena.shadow.data_ptr(), Xc.data_ptr(), ldxc, qc.data_ptr(),
                    bf["h1t"].data_ptr(), bf["h2t"].data_ptr(), bf["dz2t"].data_ptr(), bf["dz1t"].data_ptr(),
                    bf["xt"].data_ptr(), C.byref(gth) if gth is not None else 0, dl_ptr,
                    ho.data_ptr() if ho is not None else 0, engine.stream()))
        replay_dict["_co_bwd"] = True
        return q1
    skip_dz2 = bool(replay_dict.pop("_dz2_optional", False))
    # (the head rows W3 as THIS launch sees them: the weight-gradient launch that rebuilds dz2u from h2 cannot read the
    #  arena's W3 -- its own head workgroups are updating it while the fc2 tiles run)
    w3s = ws.get(tag + ".w3s", (c_arena.n_nets, c_arena.hidden))
    # producer / consumer form of the launch (csrc/ssac_fused.hip, fused_chain_pc_kernel): the actor once per tile, a'
    # handed to the tile's target-critic workgroups through tagged granules in this buffer (zeroed once: tag 0 is never used)
    ho = ws.get(tag + ".handoff", (B * A,), dtype=torch.int64, zero=True) if pc_ok else None
    with engine._timed("chain") as tm:
        for _ in range(tm.reps):  # 1, except under bench.py's live kernel timing (the launch is idempotent)
            check(lib.ssac_chain_update(
                C.byref(ch["a_arena"].desc()), s1_rep.data_ptr(), _row_stride(s1_rep), B, ch["eps_ptr"],
                float(actor.log_std_low), float(actor.log_std_high), x1.data_ptr(), S + A, S,
                ch["logp"].data_ptr(), ch["rng_ptr"], C.byref(t_arena.desc()), ids_ptr, n,
                q1.data_ptr(), C.byref(c_arena.desc()), Xc.data_ptr(), ldxc, h1.data_ptr(), h2.data_ptr(),
                qc.data_ptr(), 0 if skip_dz2 else dz2u.data_ptr(), dz1u.data_ptr(), w3s.data_ptr(),
                C.byref(gth) if gth is not None else 0, dl_ptr, ho.data_ptr() if ho is not None else 0, splits,
                xchg.handle if (xchg is not None and ho is not None) else 0, engine.stream()))
    q1._ssac_xchg_done = xchg is not None and ho is not None   # (the launch reduced q1 over the ranks itself)
    if skip_dz2:
        replay_dict["_dz2_skipped"] = w3s   # (the weight-gradient launch rebuilds dz2u from h2 and this W3 copy)
    replay_dict["_co_bwd"] = True
    return q1


def _subset_q(ws, shard, t_arena, ids, X, ldx, B, dev, tag, co_backward=None, replay_dict=None):
    """target-critic outputs for the REDQ subset `ids`: (q, n) with q of shape (n, B, out).
    Sharded: forward of the locally owned subset members, elementwise min, MIN all-reduce of the
    (B x out) partial (the one real exchange step of the critic update), n = 1."""
    if shard is None:
        ids_dev = _upload_ids(ws, ids, dev, "sub")
        ch = replay_dict.pop("_chain", None) if replay_dict is not None else None
        if ch is not None:
            q1 = _launch_chain(ch, co_backward, t_arena, ids_dev.data_ptr(), len(ids), ws, tag, B, replay_dict)
            return q1, len(ids)
        if co_backward is not None and t_arena.fused_dbuf:
            c_arena, h1, h2, act, ld_act, dz2u, dz1u = co_backward
            q1 = ws.get(tag + ".y", (len(ids), B, t_arena.out_dim))
            with engine._timed("dual_bwd") as tm:
                for _ in range(tm.reps):  # 1, except under bench.py's live kernel timing (idempotent launch)
                    check(lib.ssac_target_fwd_critic_bwdu(
                        C.byref(t_arena.desc()), ids_dev.data_ptr(), len(ids), X.data_ptr(), ldx, B, q1.data_ptr(),
                        C.byref(c_arena.desc()), h1.data_ptr(), h2.data_ptr(), act.data_ptr(), ld_act,
                        dz2u.data_ptr(), dz1u.data_ptr(), engine.stream()))
            replay_dict["_co_bwd"] = True
            return q1, len(ids)
        _, _, q1 = engine.mlp_forward(t_arena, X, ldx, 0, B, ws, tag, net_ids=ids_dev, n_sel=len(ids),
                                      save=False)
        return q1, len(ids)
    from . import parallel
    O = t_arena.out_dim
    qpart = ws.get(tag + ".qpart", (1, B, O))
    cap = engine.CAPTURE
    if cap is not None:
        # recorded launch sequence: a fixed shape for every subset draw -- all len(ids) slots are forwarded, the
        # per-update device id block holds the LOCAL index of the members this rank owns and -1 for the others
        # (their outputs are +inf), and the collective runs between two recorded segments
        n = len(ids)
        ch = replay_dict.pop("_chain", None) if replay_dict is not None else None
        if ch is not None:
            # (partial sums only when the one-shot exchange will carry them: it sums a slot's parts before sending)
            x_ = parallel._exchange if (FUSE_XCHG and parallel._exchange is not None and ch["a_arena"].shadow is None
                                        and n * B <= parallel._exchange.max_floats) else None
            q1 = _launch_chain(ch, co_backward, t_arena, cap.ids_dev.data_ptr(), n, ws, tag, B, replay_dict,
                               allow_split=parallel._exchange is not None, xchg=x_)
            if getattr(q1, "_ssac_xchg_done", False):
                parallel.check_exchange()   # (of the exchanges issued so far: a host load, as all_reduce_min_owned does)
                return q1, n
        elif co_backward is not None and t_arena.fused_dbuf:
            # ... and the TD-independent half of the local critics' backward pass rides in the same launch
            c_arena, h1, h2, act, ld_act, dz2u, dz1u = co_backward
            q1 = ws.get(tag + ".y", (n, B, O))
            check(lib.ssac_target_fwd_critic_bwdu(
                C.byref(t_arena.desc()), cap.ids_dev.data_ptr(), n, X.data_ptr(), ldx, B, q1.data_ptr(),
                C.byref(c_arena.desc()), h1.data_ptr(), h2.data_ptr(), act.data_ptr(), ld_act, dz2u.data_ptr(),
                dz1u.data_ptr(), engine.stream()))
            replay_dict["_co_bwd"] = True
        else:
            _, _, q1 = engine.mlp_forward(t_arena, X, ldx, 0, B, ws, tag, net_ids=cap.ids_dev, n_sel=n, save=False)
        # MIN all-reduce of all n slots at once (n x B x O floats; slots this rank does not own hold +inf): the minimum
        # over the slots is taken where the TD target is evaluated, so no local min launch is needed
        if parallel.one_shot_ready(q1):
            # ONE recorded launch (csrc/ssac_xchg.hip): the update stays one launch list; only the subset members'
            # owners send (the id block, mirrored to device memory by the update's first launch, names them)
            parallel.all_reduce_min_owned(q1, cap.ids_dev, n, int(getattr(q1, "_ssac_parts", 1)))
        else:
            assert getattr(q1, "_ssac_parts", 1) == 1
            cap.collective(lambda: parallel.all_reduce_min(q1))
        return q1, n
    local = shard.local_subset(ids)
    if local:
        ids_dev = _upload_ids(ws, local, dev, f"sub{len(local)}")
        _, _, q1 = engine.mlp_forward(t_arena, X, ldx, 0, B, ws, tag, net_ids=ids_dev, n_sel=len(local),
                                      save=False)
        _min_over_nets(q1, len(local), B * O, qpart)
    else:
        qpart.fill_(float("inf"))
    parallel.all_reduce_min(qpart)
    return qpart, 1


# ------------------------------------------------------------------------------------------
# Member-sharded ranks (parallel.MemberShard, SURVEY 8(e) "SUNRISE variant"): a member another rank owns still
# CONSUMES its host draws here -- in the order the reference makes them -- so that the generators of every rank (and
# the in-kernel noise counter of this agent) stay in step with the owner's.  `actor` = any local actor (all members
# share class and shape).
# ------------------------------------------------------------------------------------------
def skip_td_draws(agent, actor, B, dev, n_critics, ensemble_n, random_process):
    """the draws of compute_td_targets for one member: a' noise, exploration noise, then the REDQ subset"""
    kind = actor_kind(actor)
    if kind != "discrete":
        A = actor.action_size
        if kind == "stochastic":
            fused = engine.bind_arena(actor, "self", [actor], dev).fused and random_process is None
            if fused and IN_KERNEL_NOISE and rng.normal_is_stock():
                noise_stream(agent, dev)[1] += 1
            else:
                draw_normal((B, A), dev)
        if random_process is not None:
            draw_normal((B, A), dev)
    draw_subset(n_critics, ensemble_n)


def skip_actor_draws(actor, B, dev, random_process):
    """the draws of online_actor_update for one member: rsample noise, then exploration noise"""
    if actor_kind(actor) == "discrete":
        return
    A = actor.action_size
    draw_normal((B, A), dev)
    if random_process is not None:
        draw_normal((B, A), dev)


def skip_alpha_draws(agent, actor, B, dev):
    """the draw of alpha_update for one member: the policy sample behind log pi"""
    if actor_kind(actor) != "stochastic":
        return
    if engine.bind_arena(actor, "self", [actor], dev).fused and IN_KERNEL_NOISE and rng.normal_is_stock():
        noise_stream(agent, dev)[1] += 1
    else:
        draw_normal((B, actor.action_size), dev)


def member_sharded_sunrise_weights(logs, replay_dicts, agent, target_agent, member_shard, weight_temp, discrete, slot):
    """compute_backup_weights("sunrise") on a member-sharded rank (learning_utils.py:372-382): replay_dicts = the batch
    of EVERY global member (this rank gathered them all: same index draws everywhere).  This rank's members' target
    critics score all E batches, the all-gather completes the (batch, member, row) table, and the weights of the
    members this rank owns are formed from it.  Returns {global member index: (B, 1) weights}."""
    from . import parallel
    ms = member_shard
    E, El = ms.ensemble_size, ms.n_local
    a0 = replay_dicts[0]["primary_batch"][1]
    dev = a0.device
    ws = agent_ws(agent, dev)
    st = engine.stream()
    B = a0.shape[0]
    table = ws.get("bw.table", (E, E, B))   # [batch of member i][member k][row]
    table.zero_()
    for ig, rd in enumerate(replay_dicts):
        o, a = rd["primary_batch"][0], rd["primary_batch"][1]
        ensure_gathered(rd.get("_ssac"))
        s_rep = encode(target_agent.encoder, o)
        S = s_rep.shape[1]
        bt = rd.get("_ssac")
        if discrete:
            x, ldx = s_rep, _row_stride(s_rep)
        elif bt is not None and bt.xsa is not None and s_rep.data_ptr() == bt.xsa.data_ptr():
            x, ldx = bt.xsa, bt.xsa.stride(0)
        else:
            x = _concat_buffer(ws, f"bw.x{ig}", s_rep, a.shape[1])
            x[:, S:].copy_(a)
            ldx = x.stride(0)
        for k in range(El):
            ar = target_agent.critics[k].arena(dev)
            _, _, q = engine.mlp_forward(ar, x, ldx, 0, B, ws, f"bw.c{k}", save=False)
            check(lib.ssac_ensemble_min_select(q.data_ptr(), ar.n_nets, B, ar.out_dim,
                                               a.data_ptr() if discrete else 0, a.stride(0) if discrete else 0,
                                               table[ig, ms.lo + k].data_ptr(), st))
    parallel.all_gather_blocks(table)
    out = {}
    for ig in range(ms.lo, ms.hi):
        w = torch.empty(B, 1, device=dev)
        check(lib.ssac_sunrise_weights(table[ig].data_ptr(), E, B, float(weight_temp), w.data_ptr(),
                                       slot[L_BW:].data_ptr(), st))
        out[ig] = w
    for j, nm in enumerate(("mean", "max", "min", "std")):   # (the LAST owned member's statistics stay in the block)
        logs[f"bellman_weights/{nm}"] = slot[L_BW + j]
    return out


def member_sharded_softmax_scores(replay_dict, agent, target_agent, member_shard, row):
    """The "softmax" backup weights on a member-sharded rank, first half (learning_utils.py:383-393), for the batch of ONE
    global member -- owned or not: EVERY member k's online actor samples a'_k on the target encoder's s' and member k's online
    critics score it.  The draws are made for all E members in member order on every rank (shape-only for the tanh-normal
    policy; a categorical draw on a uniform stand-in for a member another rank holds), this rank's members fill their rows
    of `row` (E x B: [member k][batch row]); the all-gather over the ranks completes the table
    (member_sharded_softmax_finish)."""
    ms = member_shard
    o1 = replay_dict["primary_batch"][3]
    ensure_gathered(replay_dict.get("_ssac"))
    s1_rep = encode(target_agent.encoder, o1)
    B, S = s1_rep.shape
    dev = s1_rep.device
    ws = agent_ws(agent, dev)
    st = engine.stream()
    lds = _row_stride(s1_rep)
    for kg in range(ms.ensemble_size):
        k = ms.local(kg)
        actor = agent.actors[0 if k is None else k]
        kind = actor_kind(actor)
        if kind == "discrete":
            if k is None:   # (a stand-in draw of the same shape keeps the hooks / the generator in step with the owner's)
                rng.draw_categorical(torch.zeros(B, agent.critics[0].arena(dev).out_dim, device=dev))
                continue
            a_arena, c_arena = engine.bind_arena(actor, "self", [actor], dev), agent.critics[k].arena(dev)
            _, _, aout = engine.mlp_forward(a_arena, s1_rep, lds, 0, B, ws, f"bw.a{k}", save=False)
            a1 = rng.draw_categorical(aout[0]).to(torch.float32).view(B, 1)
            _, _, q = engine.mlp_forward(c_arena, s1_rep, lds, 0, B, ws, f"bw.c{k}", save=False)
            check(lib.ssac_ensemble_min_select(q.data_ptr(), c_arena.n_nets, B, c_arena.out_dim, a1.data_ptr(), 1,
                                               row[kg].data_ptr(), st))
        else:
            assert kind == "stochastic", "softmax backup weights sample from a stochastic policy"
            A = actor.action_size
            eps = rng.draw_normal((B, A), dev)   # actor_k(s1_rep).sample(): every rank makes every member's draw
            if k is None:
                continue
            a_arena, c_arena = engine.bind_arena(actor, "self", [actor], dev), agent.critics[k].arena(dev)
            _, _, aout = engine.mlp_forward(a_arena, s1_rep, lds, 0, B, ws, f"bw.a{k}", save=False)
            x1 = _concat_buffer(ws, f"bw.x1.{k}", s1_rep, A)
            check(lib.ssac_tanh_normal_fwd(aout.data_ptr(), 2 * A, eps.data_ptr(), B, A, float(actor.log_std_low),
                                           float(actor.log_std_high), x1.data_ptr(), S + A, S, 0, st))
            _, _, q = engine.mlp_forward(c_arena, x1, S + A, 0, B, ws, f"bw.c{k}", save=False)
            check(lib.ssac_ensemble_min_select(q.data_ptr(), c_arena.n_nets, B, 1, 0, 0, row[kg].data_ptr(), st))


def member_sharded_softmax_finish(logs, table, member_shard, weight_temp, slot):
    """second half: all-gather of the (batch of member i, member k, row) table -- a SUM of disjointly filled blocks -- then
    B softmax_b(-std_k T) for the members this rank owns.  Returns {global member index: (B, 1) weights}."""
    from . import parallel
    ms = member_shard
    E, B = table.shape[1], table.shape[2]
    parallel.all_gather_blocks(table)
    out = {}
    for ig in range(ms.lo, ms.hi):
        w = torch.empty(B, 1, device=table.device)
        check(lib.ssac_softmax_weights(table[ig].data_ptr(), E, B, float(weight_temp), w.data_ptr(), slot[L_BW:].data_ptr(),
                                       engine.stream()))
        out[ig] = w
    for j, nm in enumerate(("mean", "max", "min", "std")):   # (the LAST owned member's statistics stay in the block)
        logs[f"bellman_weights/{nm}"] = slot[L_BW + j]
    return out


def compute_backup_weights(logs, replay_dict, agent, target_agent, weight_type, weight_temp, batch_size,
                           discrete=False, _slot=None):
    """learning_utils.py:357-398.  "sunrise": sigmoid(-std_k(Qbar_k(s, a)) T) + 0.5 over the members' TARGET critics
    (discrete: the taken action's column); "softmax": B softmax_b(-std_k(Q_k(s', a'_k)) T) with a'_k sampled from
    member k's ONLINE actor and scored by member k's ONLINE critics on the target encoder's s'.  Q_k = min over the
    member's nets (agent.Critic.forward defaults, agent.py:22)."""
    if weight_type is None or weight_temp is None or agent.ensemble_size == 1:
        return 1.0
    assert weight_type in ("sunrise", "softmax"), f"unknown weight_type {weight_type!r}"
    o, a, _, o1, _ = replay_dict["primary_batch"]
    dev = a.device
    ws = agent_ws(agent, dev)
    slot = _slot if _slot is not None else log_block(dev)
    st = engine.stream()
    E = agent.ensemble_size
    bt = replay_dict.get("_ssac")
    if weight_type == "sunrise":
        s_rep = encode(target_agent.encoder, o)
        B, S = s_rep.shape
        if discrete:
            x, ldx = s_rep, _row_stride(s_rep)
        elif bt is not None and bt.xsa is not None and s_rep.data_ptr() == bt.xsa.data_ptr():
            x, ldx = bt.xsa, bt.xsa.stride(0)
        else:
            x = _concat_buffer(ws, "bw.x", s_rep, a.shape[1])
            x[:, S:].copy_(a)
            ldx = x.stride(0)
        qmin = ws.get("bw.qmin", (E, B))
        for k in range(E):
            ar = target_agent.critics[k].arena(dev)
            _, _, q = engine.mlp_forward(ar, x, ldx, 0, B, ws, f"bw.c{k}", save=False)
            check(lib.ssac_ensemble_min_select(q.data_ptr(), ar.n_nets, B, ar.out_dim,
                                               a.data_ptr() if discrete else 0, a.stride(0) if discrete else 0,
                                               qmin[k].data_ptr(), st))
        w = torch.empty(B, 1, device=dev)
        check(lib.ssac_sunrise_weights(qmin.data_ptr(), E, B, float(weight_temp), w.data_ptr(),
                                       slot[L_BW:].data_ptr(), st))
    else:
        s1_rep = encode(target_agent.encoder, o1)
        B, S = s1_rep.shape
        lds = _row_stride(s1_rep)
        qmin = ws.get("bw.qmin", (E, B))
        for k, (actor, critic) in enumerate(agent.ensemble):
            a_arena = engine.bind_arena(actor, "self", [actor], dev)
            c_arena = critic.arena(dev)
            _, _, aout = engine.mlp_forward(a_arena, s1_rep, lds, 0, B, ws, f"bw.a{k}", save=False)
            if discrete:
                a1 = rng.draw_categorical(aout[0]).to(torch.float32).view(B, 1)  # actor(s1_rep).sample()
                _, _, q = engine.mlp_forward(c_arena, s1_rep, lds, 0, B, ws, f"bw.c{k}", save=False)
                check(lib.ssac_ensemble_min_select(q.data_ptr(), c_arena.n_nets, B, c_arena.out_dim, a1.data_ptr(), 1,
                                                   qmin[k].data_ptr(), st))
            else:
                A = actor.action_size
                assert actor_kind(actor) == "stochastic", "softmax backup weights sample from a stochastic policy"
                x1 = _concat_buffer(ws, f"bw.x1.{k}", s1_rep, A)
                eps = rng.draw_normal((B, A), dev)  # actor(s1_rep).sample()
                check(lib.ssac_tanh_normal_fwd(aout.data_ptr(), 2 * A, eps.data_ptr(), B, A, float(actor.log_std_low),
                                               float(actor.log_std_high), x1.data_ptr(), S + A, S, 0, st))
                _, _, q = engine.mlp_forward(c_arena, x1, S + A, 0, B, ws, f"bw.c{k}", save=False)
                check(lib.ssac_ensemble_min_select(q.data_ptr(), c_arena.n_nets, B, 1, 0, 0, qmin[k].data_ptr(), st))
        w = torch.empty(B, 1, device=dev)
        check(lib.ssac_softmax_weights(qmin.data_ptr(), E, B, float(weight_temp), w.data_ptr(),
                                       slot[L_BW:].data_ptr(), st))
    for j, nm in enumerate(("mean", "max", "min", "std")):
        logs[f"bellman_weights/{nm}"] = slot[L_BW + j]
    return w


def _min_over_nets(q, n, B, out):
    """out[b] = min_j q[j][b] using the TD-target kernel in its plain-min configuration
    (gamma=1, r=0, d=0, no entropy, no PopArt): td = 0 + 1*(1-0)*min."""
    dev = q.device
    z = _zeros(dev, B)
    check(lib.ssac_td_target(q.data_ptr(), n, B, 1, 0, z.data_ptr(), z.data_ptr(), z.data_ptr(), 0, 1.0,
                             0, 0, out.data_ptr(), 0, engine.stream()))


_zero_cache = {}


def _zeros(dev, n):
    z = _zero_cache.get((dev, n))
    if z is None:
        z = _zero_cache[(dev, n)] = torch.zeros(n, device=dev)
    return z


# ------------------------------------------------------------------------------------------
# AFBC helpers (learning_utils.py:217-240, 287-295)
# ------------------------------------------------------------------------------------------
def adjust_priorities(logs, replay_dict, agent, buffer):
    """new PER priorities relu(A(s,a)) + 1e-4 from a random ensemble member (fresh policy samples)."""
    o, a = replay_dict["primary_batch"][0], replay_dict["primary_batch"][1]
    member = rng.choice(range(agent.ensemble_size))
    res = agent.adv_estimator.evaluate(o, a, member, want=("prio",))
    # (the reference blocks here, .cpu() in learning_utils.py:293; with the trees in HBM the new priorities and the
    # indices they belong to never leave the device)
    new_priorities = res["prio"].reshape(-1)
    if not getattr(buffer, "per_on_device", False):
        new_priorities = new_priorities.cpu().numpy()
    buffer.update_priorities(replay_dict["priority_idxs"], new_priorities)


def compute_filter_stats(buffer, agent, augmenter, batch_size):
    """percentage of a uniform batch the binary advantage filter accepts (learning_utils.py:217-238)."""
    rd = sample_move_and_augment(buffer=buffer, batch_size=batch_size, augmenter=augmenter, aug_mix=0.0,
                                 per=False)
    o, a = rd["primary_batch"][0], rd["primary_batch"][1]
    member = rng.choice(range(agent.ensemble_size))
    res = agent.adv_estimator.evaluate(o, a, member, want=("mask",))
    return float(res["mask"].sum() / res["mask"].numel() * 100.0)
