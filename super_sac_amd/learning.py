"""The three update functions of the reference's training loop, same names and keyword
arguments (super_sac/learning.py:18-141 critic_update, :344-421 online_actor_update,
:222-263 alpha_update; call sites main.py:380-405, :492-510, :529-542).

What changes underneath: the per-critic Python loop + autograd + per-parameter Adam of the
reference become ensemble-batched HIP launches -- forward of all N critics in three launches,
hand-derived backward, and weight-gradient GEMMs whose epilogue is the Adam step -- and the
logs stay on the device until somebody reads them.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import engine, parallel, rng
from . import learning_utils as lu
from . import _lib
from ._lib import check, lib


def _critic_input(bt, ws, tag, s_rep, a, discrete):
    """(X, ldx, in_dim) for the online critics: [s | a] (continuous) or s (discrete)."""
    B, S = s_rep.shape
    if bt is not None and bt.xsa is not None and s_rep.data_ptr() == bt.xsa.data_ptr():
        return bt.xsa, bt.xsa.stride(0)
    if discrete:
        return s_rep, lu._row_stride(s_rep)
    x = lu._concat_buffer(ws, tag, s_rep, a.shape[1])
    x[:, S:].copy_(a)
    return x, x.stride(0)


# Run the online critics' forward as a parallel branch beside the actor -> target critics -> TD-target chain.
# Off by default: on MI355X a cross-stream join costs ~10 us of signalling (measured, also inside a HIP graph),
# more than the overlap wins at the metric shape (6.6k vs 7.15k updates/s); tests exercise both settings.
# how a repeated update is re-issued: "list" = the library's recorded launch list (ssac_replay), "graph" = a
# hipGraph captured through torch.cuda.graph (leaves a ~13 us idle tail per launch on MI355X)
LAUNCH_MODE = os.environ.get("SSAC_LAUNCH_MODE", "list")


class _LaunchList:
    """recorded launch segments with host callables (collectives of the sharded update) between them"""

    def __init__(self):
        self.parts = []  # list handles (int) and callables, in issue order

    def add_list(self, handle):
        if not handle:
            raise RuntimeError("libssac_hip: " + lib.ssac_last_error().decode())
        if lib.ssac_launch_list_size(handle) > 0:
            self.parts.append(handle)
        else:
            lib.ssac_launch_list_free(handle)

    def replay(self):
        st = engine.stream()
        for part in self.parts:
            if callable(part):
                part()
            else:
                check(lib.ssac_replay(part, st))

    def __del__(self):
        for part in self.parts:
            if not callable(part):
                try:
                    lib.ssac_launch_list_free(part)
                except Exception:
                    pass


FUSED_ACTOR = True  # the online actor update in four fused launches
# ... of which the first three are ONE (ssac_actor_chain_fused: actor forward / critics' forward + dQ/da / actor backward as
# producer and consumer workgroups of a single launch).  Module constant; tests/test_hip_cases.py flips it for the A/B.
ACTOR_CHAIN = True
FEED_SLOTS = 32  # pinned input ring of a captured update: how far the host may run ahead of the GPU
# evaluate the TD target inside the critic launch instead of a launch of its own (continuous, no PopArt)
SHARDED_LISTS = True  # recorded launch lists on critic-sharded ranks
FOLD_BEGIN = True  # fold ssac_begin_update into the replay gather
# Log finalisation inside the weight-gradient launch: the last workgroup to ARRIVE (a device-scope ticket drawn after
# its write-through partial stores -- no fence, so none of the 17 MB of freshly written Adam state is flushed) sums the
# partials and publishes the log block; the separate 1-workgroup logs launch (~5 us per update) disappears.
FOLD_LOGS = True
LAZY_TD = True
SPLIT_FORWARD = False


# the TD-independent half of the critics' backward pass inside the target-critic launch (rank-1 loss gradient); a module
# knob, not an environment switch: tests turn it off to compare against the launch forms other configurations take
RANK1_BWD = True
SKIP_DZ2 = True  # dz2u stays inside the chained launch (rebuilt from h2 by the weight gradient)
EVENT_EVERY = 8  # must divide FEED_SLOTS
FOLD_LOSS = True  # rank-1 backward: dL/dq evaluated inside the weight-gradient launch
DUAL_LAUNCH = True  # critic forward inside the actor-sample launch


DUAL_MAX_WG = 320


def _dual_fits(arena, n_rows):
    """worth merging while the actor's tiles (16 rows) and the critic forward's (32 rows at this size) are about one
    round of workgroups (one per CU; a few more still pay for themselves because the merged path also carries the
    replay gather, the rank-1 backward and the folded loss gradient: Humanoid N 16 at B 512 = 288 workgroups, +4 %)"""
    return (n_rows + 15) // 16 + arena.n_nets * ((n_rows + 31) // 32) <= DUAL_MAX_WG


def _split_forward(n_nets, n_rows):
    """the branch pays off while the critic forward leaves CUs free for the small actor / target launches
    it runs beside: one 32-row workgroup per CU, at most 192 of the 256 CUs."""
    return SPLIT_FORWARD and n_nets * ((n_rows + 31) // 32) <= 192


def _clip_and_step(adam, members, clip, slot_norm, member_shard=None):
    """clip_grad_norm_ over ALL listed arenas jointly, then Adam from the stored gradients
    (learning.py:122-130 / :413-416).  members: list of (arena, key, grads, sumsq).  member_shard: the joint norm runs
    over the members of EVERY rank (one SUM all-reduce of this rank's squared norm)."""
    st = engine.stream()
    allss = members[0][3] if len(members) == 1 else torch.cat([m[3] for m in members])
    if member_shard is not None:
        allss = allss.sum().reshape(1)   # (device plumbing: one scalar to exchange)
        parallel.all_reduce_sum(allss)
    check(lib.ssac_clip_coef(adam.ctl.ptr, allss.data_ptr(), allss.numel(), float(clip), 0, st))
    for arena, key, grads, _ in members:
        m, v = adam.moments_for(key, arena.params)
        check(lib.ssac_adam_step(arena.params.data_ptr(), m.data_ptr(), v.data_ptr(), grads.data_ptr(),
                                 arena.params.numel(), adam.ctl.ptr, st))
        if arena.shadow is not None:
            arena.sync_shadow()   # bf16 mode: the operand copies follow the freshly stepped masters


def _encoder_step(encoder, encoder_optimizer, encoder_clip, dX, emb, ws, slot, dev, inv=None, accumulate=False,
                  step=True):
    """encoder backward from the critics' input gradients + clip + encoder_optimizer.step()
    (learning.py:121,127-129); logs the (clipped) encoder gradient norm (learning.py:137).
    inv = (as_rep, os_rep, encoder_lambda, stacked): the encoder invariance constraint (learning.py:114-117) adds
    lambda * d||as_rep - os_rep||_F / d as_rep -- to the critics' rows when the augmented batch IS the critic batch,
    to rows [B, 2B) of a stacked pass otherwise."""
    from . import conv_encoder
    eng = conv_encoder.conv_engine(encoder, dev)
    B = dX.shape[1]
    stacked = inv is not None and inv[3]
    d_rep = ws.get("cu.drep2" if stacked else "cu.drep", (2 * B if stacked else B, emb))
    torch.sum(dX[:, :, :emb], dim=0, out=d_rep[:B])  # device plumbing: sum of N small slices
    if inv is not None:
        as_rep, os_rep, lam, _ = inv
        tgt = d_rep[B:] if stacked else d_rep[:B]
        check(lib.ssac_frobenius_diff_bwd(as_rep.data_ptr(), lu._row_stride(as_rep), os_rep.data_ptr(),
                                          lu._row_stride(os_rep), B, emb, float(lam), tgt.data_ptr(), emb,
                                          0 if stacked else 1, slot[lu.L_ENC_INV:].data_ptr(),
                                          slot[lu.L_CRITIC_LOSS:].data_ptr(), engine.stream()))
    # ensemble members share the encoder: member i > 0 adds its gradient to the arena, the clip + optimizer step
    # follows the last member (learning.py:47-130: one backward over the summed loss, one encoder_optimizer.step())
    eng.backward(d_rep, accumulate=accumulate)
    if step:
        eng.optimizer_step(encoder_optimizer, encoder_clip, norm_out=slot[lu.L_ENC_GN:])


USE_GRAPHS = True   # replay the critic update's launch sequence as one HIP graph when it is static
GRAPH_WARMUP = 3    # eager calls before capture (workspace allocation, arena binding, kernel attributes)


class _Graphed:
    def __init__(self):
        self.calls = 0
        self.graph = None
        self.fast = None
        self.path = "slow"
        self.deferred = None   # ssac_deferred_logs of the recording (deferred log finalisation), else None
        self.pending = None    # log-ring slot of the newest update whose block has not been written yet

    def views(self, ring, slot_i):
        """log values of ring slot slot_i (built once per slot)"""
        v = self.log_views.get(slot_i)
        if v is None:
            if self.deferred is not None:
                v = lu.lazy_views(self, ring, slot_i, self.log_index)
            else:
                blk = ring.buf[slot_i]
                v = {k_: blk[i] for k_, i in self.log_index.items()}
            self.log_views[slot_i] = v
        return v


def critic_update(buffer, agent, target_agent, critic_optimizer, encoder_optimizer, log_alphas,
                  batch_size, gamma, critic_clip, encoder_clip, target_critic_ensemble_n,
                  weighted_bellman_temp, weight_type, pop, augmenter, encoder_lambda, random_process,
                  noise_clip, aug_mix=0.75, discrete=False, per=False, update_priorities=False,
                  dr3_coeff=0.0):
    """learning.py:18-141.  When the launch sequence is static (one ensemble member, vector
    observations, uniform sampling, single rank) it is captured once into a HIP graph and replayed:
    the host then only draws the indices / REDQ subset / noise, uploads them into fixed-address
    buffers and issues one graph launch, instead of ~15 kernel launches."""
    # ---- fast path: this exact call has been recorded already (ssac_step: one C call re-issues the update)
    fast = agent.__dict__.get("_ssac_fast")
    if fast is not None and USE_GRAPHS and engine.CAPTURE is None:
        fs = fast.get((id(buffer), id(target_agent), id(critic_optimizer), id(log_alphas[0]), id(augmenter), batch_size,
                       gamma, critic_clip, encoder_clip, target_critic_ensemble_n, weighted_bellman_temp, weight_type,
                       pop, encoder_lambda, id(random_process), noise_clip, discrete, per, update_priorities, dr3_coeff,
                       engine.USE_FUSED))
        if fs is not None and fs.still_valid():
            return fs.run()
    kw = dict(buffer=buffer, agent=agent, target_agent=target_agent, critic_optimizer=critic_optimizer,
              encoder_optimizer=encoder_optimizer, log_alphas=log_alphas, batch_size=batch_size, gamma=gamma,
              critic_clip=critic_clip, encoder_clip=encoder_clip,
              target_critic_ensemble_n=target_critic_ensemble_n, weighted_bellman_temp=weighted_bellman_temp,
              weight_type=weight_type, pop=pop, augmenter=augmenter, encoder_lambda=encoder_lambda,
              random_process=random_process, noise_clip=noise_clip, aug_mix=aug_mix, discrete=discrete, per=per,
              update_priorities=update_priorities, dr3_coeff=dr3_coeff)
    lu.ensure_adopted(agent, buffer)
    lu.ensure_adopted(target_agent, buffer)
    shard = parallel.shard_of(agent)
    graphable = (USE_GRAPHS and engine.CAPTURE is None and agent.ensemble_size == 1 and not per
                 and parallel.member_shard_of(agent) is None
                 and not update_priorities and not dr3_coeff and lu.is_identity(agent.encoder)
                 and random_process is None and torch.cuda.is_available()
                 # critic-sharded ranks: recorded launch lists only (the collective sits between two segments),
                 # continuous actions on the fused kernels (empty subset slots, in-launch TD target)
                 and (shard is None or (LAUNCH_MODE == "list" and SHARDED_LISTS and not discrete
                                        and not any(agent.popart)
                                        and agent.critics[0].arena(log_alphas[0].device).fused)))
    if not graphable:
        return _critic_update_eager(**kw)
    key = (id(buffer), id(target_agent), id(critic_optimizer), batch_size, float(gamma), critic_clip,
           target_critic_ensemble_n, bool(pop), bool(discrete), id(log_alphas[0]), engine.USE_FUSED)
    cache = agent.__dict__.setdefault("_ssac_graphs", {})  # lives and dies with the agent
    gs = cache.get(key)
    if gs is None:
        gs = cache[key] = _Graphed()
        gs.refs = (buffer, target_agent, critic_optimizer, log_alphas[0])  # pin the ids used in the key
    gs.calls += 1
    if gs.calls <= GRAPH_WARMUP or len(buffer) < batch_size:
        return _critic_update_eager(**kw)
    out = _critic_update_graphed(gs, kw)
    if (gs.graph is not None and LAUNCH_MODE == "list" and getattr(gs, "fast", None) is None
            and all(not callable(p_) for p_ in gs.graph.parts)):
        gs.fast = _FastStep(gs, kw)
        agent.__dict__.setdefault("_ssac_fast", {})[
            (id(buffer), id(target_agent), id(critic_optimizer), id(log_alphas[0]), id(augmenter), batch_size,
             gamma, critic_clip, encoder_clip, target_critic_ensemble_n, weighted_bellman_temp, weight_type,
             pop, encoder_lambda, id(random_process), noise_clip, discrete, per, update_priorities, dr3_coeff,
             engine.USE_FUSED)] = gs.fast
    return out


def _flush_other_recordings(agent, keep=None):
    """Deferred log finalisation keeps update k's partials in name-keyed workspace tensors of the AGENT until the next
    update of the same recording writes the block.  Any other critic update on the agent (an eager call, a second
    recording, a re-recording) would overwrite them first: finalise the pending block now."""
    for g in agent.__dict__.get("_ssac_graphs", {}).values():
        if g is not keep and g.pending is not None:
            lu.flush_pending_logs(g)


class _FastStep:
    """the recorded critic update behind ONE C call per update (ssac_step_run, include/ssac_hip.h): what stays in
    Python is the reference's host-RNG draws, in its order (indices -> injected noise -> REDQ subset -> logged-net
    choice), and handing out the log views of this update's ring slot"""

    def __init__(self, gs, kw):
        self.gs, self.kw = gs, kw
        self.buffer, self.agent = kw["buffer"], kw["agent"]
        self.B, self.n_sub = kw["batch_size"], kw["target_critic_ensemble_n"]
        self.dev = kw["log_alphas"][0].device
        self.ring = lu.ring_for(self.dev)
        B, n_pad = self.B, gs.n_pad
        draw_off = 8 * B + 4 * n_pad + 8
        h = lib.ssac_step_create(gs.ring.ptr, FEED_SLOTS, gs.slot_bytes, B, self.n_sub, 8 * B, 8 * B + 4 * n_pad,
                                 draw_off, EVENT_EVERY)
        if not h:
            raise RuntimeError("libssac_hip: " + lib.ssac_last_error().decode())
        self.handle = h
        for part in gs.graph.parts:
            check(lib.ssac_step_add_list(h, part))
        self.ids_c = (C.c_int32 * max(self.n_sub, 1))()
        self.shard = parallel.shard_of(self.agent)
        self.n_critics = self.agent.num_critics if self.shard is None else self.shard.num_critics  # GLOBAL ensemble
        self.in_kernel_noise = gs.in_kernel_noise
        self.late_arenas = getattr(gs, "late_arenas", None)   # (target arena, online arena) of the late-bound Polyak
        self.calls = 0
        # a few parameter addresses checked on every call (cheap), all of them every 256 calls
        actor = self.agent.actors[0]
        tgt = kw["target_agent"]
        self.arenas = [(self.agent.critics[0].arena(self.dev), list(self.agent.critics[0].nets)),
                       (tgt.critics[0].arena(self.dev), list(tgt.critics[0].nets)),
                       (engine.bind_arena(actor, "self", [actor], self.dev), [actor])]
        self.probes = []
        for arena, mods in self.arenas:
            lins = engine.MlpArena.linear_triples(mods[0]) + engine.MlpArena.linear_triples(mods[-1])
            for lin in (lins[0], lins[-1]):
                self.probes.append((lin.weight, lin.weight.data_ptr()))
        # the log views of every ring slot, built now (~7 ms, once): built on a slot's first visit they cost the first
        # 512 replayed updates ~11 us of host time each -- 30 us per step instead of 17, enough to starve the device in
        # short bursts right after recording (tools/burst_parts.py)
        for slot_i in range(self.ring.buf.shape[0]):
            gs.views(self.ring, slot_i)

    def still_valid(self):
        gs = self.gs
        if gs.fast is not self or LAUNCH_MODE != "list" or len(self.buffer) < self.B:
            return False
        if gs.eps_dev is not None and rng.normal_is_stock() != self.in_kernel_noise:
            return False  # a noise hook was installed / removed: the slow path records the update again
        self.calls += 1
        if self.calls & 255 == 0:
            if not all(arena.is_bound(mods) for arena, mods in self.arenas):
                gs.fast = None
                return False
        else:
            for p, ptr in self.probes:
                if p.data_ptr() != ptr:
                    gs.fast = None
                    return False
        return True

    def run(self):
        gs, buffer, agent = self.gs, self.buffer, self.agent
        if gs.path != "fast":
            # the Python path issued updates of this recording since: line the step up with the ring's device counter
            # (its slot-reuse events know nothing about those updates, so let them finish first)
            torch.cuda.synchronize()
            check(lib.ssac_step_seek(self.handle, gs.k))
            gs.path = "fast"
        if len(agent.__dict__["_ssac_graphs"]) > 1:
            _flush_other_recordings(agent, keep=gs)
        buffer.total_sample_calls += 1
        idx_cpu = rng.draw_indices(len(buffer), self.B)
        if gs.eps_dev is not None and not self.in_kernel_noise:
            rng.draw_normal_into(gs.eps_dev)  # injected noise (parity tests)
        ids = rng.draw_subset(self.n_critics, self.n_sub)
        ida, sh = self.ids_c, self.shard
        for j, v in enumerate(ids):
            # sharded: the LOCAL index of a subset member this rank owns, -(owner rank + 1) for one that lives elsewhere
            ida[j] = v if sh is None else sh.slot_code(v)
        slot_i = self.ring.advance()
        if sh is not None and gs.k % EVENT_EVERY == 0:
            parallel.check_exchange()
        draw = 0
        if self.in_kernel_noise:
            ns = lu.noise_stream(agent, self.dev)
            draw = ns[1]
            ns[1] += 1
        check(lib.ssac_step_run(self.handle, idx_cpu.data_ptr(), ida, slot_i, draw, engine.stream()))
        gs.k += 1
        if self.late_arenas is not None:
            agent.critics[0].__dict__["_ssac_last_step"] = self   # a soft_update that follows needs no launch
        if gs.deferred is not None:
            gs.pending = slot_i   # (the previous update's block is written by the launch just issued)
        rng.choice(agent.critics)  # keep the Python RNG stream in step with learning.py:135
        logs = gs.views(self.ring, slot_i)
        rd = gs.dicts[0]
        rd["priority_idxs"] = idx_cpu.numpy()
        rd["_subset"] = ids
        return dict(logs), gs.dicts

    def __del__(self):
        try:
            lib.ssac_step_destroy(self.handle)
        except Exception:
            pass


class _FeedRing:
    """the input ring of a recorded update (ssac_feed_ring_alloc): device-resident on large-BAR systems"""

    def __init__(self, nbytes):
        p, d = C.c_void_p(), C.c_int()
        check(lib.ssac_feed_ring_alloc(nbytes, C.byref(p), C.byref(d)))
        self.ptr, self.device_resident = p.value, bool(d.value)

    def __del__(self):
        try:
            lib.ssac_feed_ring_free(self.ptr, int(self.device_resident))
        except Exception:
            pass


def _critic_update_graphed(gs, kw):
    buffer, agent, B = kw["buffer"], kw["agent"], kw["batch_size"]
    agent.critics[0].__dict__.pop("_ssac_last_step", None)
    if gs.path == "fast":  # (the C step's slot-reuse events are not visible from here)
        torch.cuda.synchronize()
        gs.path = "slow"
    dev = kw["log_alphas"][0].device
    actor = agent.actors[0]
    kind = lu.actor_kind(actor)
    n_sub = kw["target_critic_ensemble_n"]
    ring = lu.ring_for(dev)
    if gs.graph is None and getattr(gs, "feed", None) is None:
        # one fixed device block holds the per-update inputs:
        #   [B int64 indices | n int32 subset ids (padded to 8 bytes) | int32 log-ring slot, int32 pad]
        # The host writes them into slot k % FEED_SLOTS of a pinned ring; the first captured launch pulls the
        # slot over PCIe (ssac_feed in include/ssac_hip.h), so an update is ONE graph launch and no copy node.
        n_pad = (n_sub + 1) // 2 * 2
        # ... | int64 draw number of the agent's noise stream]; slots are whole 16-byte words (ssac_feed contract)
        nbytes = (8 * B + 4 * n_pad + 16 + 15) // 16 * 16
        gs.inbuf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        gs.idx_dev = gs.inbuf[:8 * B].view(torch.int64)
        gs.ids_dev = gs.inbuf[8 * B:8 * B + 4 * n_pad].view(torch.int32)[:n_sub]
        # the slots are composed in ordinary host memory (numpy views: a fraction of the cost of torch indexing) and
        # copied into the input ring with ONE ssac_feed_write each; the ring itself is uncached device memory the
        # host stores into over the PCIe BAR when the system allows it, pinned host memory otherwise
        gs.stage = np.zeros((FEED_SLOTS, nbytes), np.uint8)
        gs.ring = _FeedRing(FEED_SLOTS * nbytes)
        gs.slot_bytes = nbytes
        hnp = gs.stage
        gs.np_idx = hnp[:, :8 * B].view(np.int64)                           # (slots, B)
        gs.np_i32 = hnp[:, 8 * B:8 * B + 4 * n_pad + 8].view(np.int32)      # ids..., then the log slot at [n_pad]
        gs.np_draw = hnp[:, 8 * B + 4 * n_pad + 8:8 * B + 4 * n_pad + 16].view(np.int64)  # (slots, 1)
        gs.n_pad = n_pad
        gs.log_views = {}
        gs.events = [None] * FEED_SLOTS
        gs.k = 0
        gs.eps_dev = torch.empty(B, actor.action_size, device=dev) if kind == "stochastic" else None
        gs.logblk = torch.zeros(lu.LOG_WIDTH, device=dev)
        # late-bound Polyak (include/ssac_hip.h): the decision word the update's first launch writes
        gs.late_word = torch.zeros(4, dtype=torch.int32, device=dev) if lu.LATE_POLYAK else None
        gs.feed = engine.DeviceStruct(_lib.Feed(gs.ring.ptr, gs.inbuf.data_ptr(), ring.buf.data_ptr(), 0,
                                                FEED_SLOTS, nbytes // 4, (8 * B + 4 * n_pad) // 4,
                                                lu.LOG_WIDTH,
                                                gs.late_word.data_ptr() if gs.late_word is not None else 0), dev)
    # ---- host draws, in the reference's order: indices -> (augmentation: none here) -> noise -> subset
    buffer.total_sample_calls += 1
    idx_cpu = rng.draw_indices(len(buffer), B)
    # (the in-kernel stream is used by the fused actor-sample launch only: wide action heads take the per-layer path)
    in_kernel_noise = (kind == "stochastic" and lu.IN_KERNEL_NOISE and rng.normal_is_stock()
                       and engine.bind_arena(actor, "self", [actor], dev).fused)
    if gs.graph is not None and gs.in_kernel_noise != in_kernel_noise:
        lu.flush_pending_logs(gs)  # (while the old recording's device structs are alive)
        gs.graph = None  # a noise hook was installed / removed since the recording: record the update again
        gs.feed = None
    _flush_other_recordings(agent, keep=gs if gs.graph is not None else None)
    if kind == "stochastic" and not in_kernel_noise:
        # injected noise (parity tests) goes straight into the captured update's input buffer
        rng.draw_normal_into(gs.eps_dev)
    shard = parallel.shard_of(agent)
    ids = rng.draw_subset(agent.num_critics if shard is None else shard.num_critics, n_sub)  # GLOBAL ensemble
    # ---- per-update inputs into this update's pinned slot
    k = gs.k % FEED_SLOTS
    # slot reuse: the replay that read this slot FEED_SLOTS updates ago must have finished.  One event per group of
    # EVENT_EVERY updates (recorded after the group's last update; an event costs a barrier packet on the queue)
    if gs.k % EVENT_EVERY == 0 and gs.k >= FEED_SLOTS:
        gs.events[((gs.k - FEED_SLOTS) // EVENT_EVERY) % (FEED_SLOTS // EVENT_EVERY)].synchronize()
    if shard is not None and gs.k % EVENT_EVERY == 0:
        parallel.check_exchange()
    slot_i = ring.advance()
    gs.np_idx[k] = idx_cpu.numpy()
    row = gs.np_i32[k]
    for j, v in enumerate(ids):
        # sharded: the LOCAL index of a subset member this rank owns, -(owner rank + 1) for one that lives elsewhere
        row[j] = v if shard is None else shard.slot_code(v)
    row[gs.n_pad] = slot_i
    if in_kernel_noise:
        gs.np_draw[k, 0] = lu.noise_stream(agent, dev)[1]
    check(lib.ssac_feed_write(gs.ring.ptr + k * gs.slot_bytes, gs.stage.ctypes.data + k * gs.slot_bytes, gs.slot_bytes))
    if gs.graph is None:
        gs.in_kernel_noise = in_kernel_noise
        ctx = engine.CaptureCtx(idx_cpu, gs.idx_dev, ids, gs.ids_dev,
                                [gs.eps_dev] if (gs.eps_dev is not None and not in_kernel_noise) else [],
                                gs.logblk, feed=gs.feed.ptr)
        if in_kernel_noise:
            # the draw number travels in the input slot, so recorded and eager updates of an agent may interleave
            ctx.tick_ptr = gs.inbuf.data_ptr() + 8 * B + 4 * gs.n_pad + 8
            ctx.noise_offset = 0
        st_ = buffer._storage
        keys_ = list(st_.s_stack.keys())
        # vector observations in one array: the whole-transition gather is the update's first launch and takes
        # over ssac_begin_update's work (ensemble_size 1: a single gather per update)
        ctx.defer_begin = (FOLD_BEGIN and agent.ensemble_size == 1 and len(keys_) == 1
                           and st_.s_stack[keys_[0]].dim() == 2)
        c_arena_ = agent.critics[0].arena(dev)
        if (lu.DEFERRED_LOGS and LAUNCH_MODE == "list" and FOLD_LOGS and FOLD_LOSS and LAZY_TD and ctx.defer_begin
                and not kw["critic_clip"] and kind == "stochastic" and not any(agent.popart) and c_arena_.out_dim == 1
                and B <= 4096):
            # deferred log finalisation (csrc/ssac_critic_logs.h): the buffers the weight-gradient launch will leave
            # its partials in are workspace tensors with fixed names, so the struct can be built before the body runs
            ws_ = lu.agent_ws(agent, dev)
            N_ = c_arena_.n_nets
            ttot_ = (engine.bf16_tiles_total(c_arena_) if c_arena_.shadow is not None
                     else engine.wgrad_tiles_total(c_arena_))
            n_glob_ = N_ if shard is None else shard.num_critics
            gs.td_stats = ws_.get("cu.tdstats", (4,), zero=True)
            ctx.deferred = _lib.DeferredLogs(
                ws_.get("cu.c0.fparts", (N_ * 2,)).data_ptr(), N_, N_ * ttot_, ws_.get("cu.ss0", (N_ * ttot_,)).data_ptr(),
                gs.td_stats.data_ptr(), lu.L_TD0, B, float(n_glob_), 0, gs.feed.ptr)
            t_arena_ = kw["target_agent"].critics[0].arena(dev)
            if (gs.late_word is not None and (c_arena_.shadow is None) == (t_arena_.shadow is None)
                    and t_arena_.params.numel() == c_arena_.params.numel()):
                # late-bound Polyak: the weight-gradient launch carries the target arena and waits for the decision
                ctx.late = gs.late_word.data_ptr()
                ctx.late_target = t_arena_

        def body():
            logs_, dicts_ = _critic_update_eager(**kw)
            assert not ctx.pending_begin, "deferred ssac_begin_update was never issued"
            assert ctx.deferred_used == ctx.deferred_chain, "deferred log finalisation: chain / weight-gradient mismatch"
            gs.deferred = ctx.deferred if ctx.deferred_used else None
            gs.late_arenas = ((ctx.late_target, agent.critics[0].arena(dev))
                              if (ctx.deferred_used and ctx.late_used) else None)
            if not ctx.published:
                check(lib.ssac_publish_logs(gs.logblk.data_ptr(), gs.feed.ptr, engine.stream()))
            return logs_, dicts_
        engine.CAPTURE = ctx
        try:
            if LAUNCH_MODE == "graph":
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    logs, dicts = body()
                graph.replay()
            else:
                # launch list: this call's launches run normally AND are recorded for the later calls
                graph = _LaunchList()

                def collective(fn):
                    graph.add_list(lib.ssac_record_end())
                    fn()
                    graph.parts.append(fn)
                    check(lib.ssac_record_begin())
                ctx.collective = collective
                check(lib.ssac_record_begin())
                try:
                    logs, dicts = body()
                finally:
                    handle = lib.ssac_record_end()
                graph.add_list(handle)
        finally:
            engine.CAPTURE = None
        gs.graph, gs.dicts = graph, dicts
        base = gs.logblk.data_ptr()
        gs.log_index = {k_: (v.data_ptr() - base) // 4 for k_, v in logs.items()}
        gs.log_views = {}   # the 0-dim views handed out as log values, built once per ring slot (gs.views)
    else:
        gs.graph.replay()  # ONE host call re-issues the whole update
    if gs.k % EVENT_EVERY == EVENT_EVERY - 1:
        gi = (gs.k // EVENT_EVERY) % (FEED_SLOTS // EVENT_EVERY)
        ev = gs.events[gi]
        if ev is None:
            ev = gs.events[gi] = torch.cuda.Event()
        ev.record()
    gs.k += 1
    if in_kernel_noise:
        lu.noise_stream(agent, dev)[1] += 1  # one draw of the agent's noise stream per update, as in eager launches
    rng.choice(agent.critics)  # keep the Python RNG stream in step with learning.py:135
    if gs.deferred is not None:
        gs.pending = slot_i
    logs = dict(gs.views(ring, slot_i))
    rd = gs.dicts[0]
    rd["priority_idxs"] = idx_cpu.numpy()
    rd["_subset"] = ids
    return logs, gs.dicts


def _critic_update_eager(buffer, agent, target_agent, critic_optimizer, encoder_optimizer, log_alphas,
                         batch_size, gamma, critic_clip, encoder_clip, target_critic_ensemble_n,
                         weighted_bellman_temp, weight_type, pop, augmenter, encoder_lambda, random_process,
                         noise_clip, aug_mix=0.75, discrete=False, per=False, update_priorities=False,
                         dr3_coeff=0.0):
    engine.require_gpu()
    if engine.CAPTURE is None:
        agent.critics[0].__dict__.pop("_ssac_last_step", None)
        _flush_other_recordings(agent)
    E = agent.ensemble_size
    assert E <= lu.MAX_MEMBERS
    # member-sharded rank (parallel.MemberShard, SURVEY 8(e) "SUNRISE variant"): `agent` holds the members [ms.lo, ms.hi)
    # of an ensemble of E_glob; the loss is averaged over the GLOBAL ensemble, every member's host draws are made here
    ms = parallel.member_shard_of(agent)
    E_glob = E if ms is None else ms.ensemble_size
    if ms is not None:
        assert lu.is_identity(agent.encoder), "member sharding: a trainable encoder is shared by all members (not sharded)"
        assert not per and not update_priorities and not dr3_coeff, "member sharding covers the online critic update"
        assert weight_type in (None, "sunrise", "softmax"), f"unknown weight_type {weight_type!r}"
    dev = log_alphas[0].device
    ws = lu.agent_ws(agent, dev)
    adam = engine.adam_group(critic_optimizer, dev)
    slot = lu.log_block(dev, adam)  # one launch: clear the log block + advance the Adam step
    logs = {}
    st = engine.stream()
    clip_members = []
    replay_dicts = []
    member_ss = []
    fused_logs = []
    logs_done_in_wgrad = False
    # Two passes over the ensemble members, as the reference's single backward at the end implies (learning.py:45-130):
    # every member's batch, TD target and backup weights are computed BEFORE any critic is updated -- the "softmax"
    # weights of member i look at the ONLINE critics of ALL members (learning_utils.py:383-393).
    preps = []
    all_rd = []   # member-sharded ranks: every global member's batch, in member order
    # member-sharded "softmax" weights (round 5): every member's policy samples on EVERY member's batch, drawn in the
    # reference's order -- right behind that batch's TD draws -- by every rank; the table is completed after the loop
    ms_softmax = (ms is not None and weight_type == "softmax" and weighted_bellman_temp is not None and E_glob > 1)
    sm_table = None
    if ms_softmax:
        sm_table = ws.get("bw.table", (E_glob, E_glob, batch_size))   # [batch of member i][member k][row]
        sm_table.zero_()
    for ig in range(E_glob):
        i = ig if ms is None else ms.local(ig)
        if i is None:
            # a member another rank owns: its batch (this rank's target critics score it for the sunrise weights) and
            # its host draws, in the reference's order -- sample, action noise, REDQ subset
            rd = lu.sample_move_and_augment(buffer=buffer, batch_size=batch_size, augmenter=augmenter,
                                            aug_mix=aug_mix, per=per, _invariance=bool(encoder_lambda))
            lu.skip_td_draws(agent, agent.actors[0], batch_size, dev, target_agent.critics[0].arena(dev).n_nets,
                             target_critic_ensemble_n, random_process)
            if ms_softmax:
                lu.member_sharded_softmax_scores(rd, agent, target_agent, ms, sm_table[ig])
            all_rd.append(rd)
            continue
        arena = agent.critics[i].arena(dev)
        N, qd = arena.n_nets, arena.out_dim
        H = arena.hidden
        tag = f"cu.c{i}"
        train_enc = not lu.is_identity(agent.encoder)
        # when the merged actor / critic-forward launch will run, its workgroups fetch their own replay rows
        # (ssac_gather) and the update has no gather launch
        dual_ok = (DUAL_LAUNCH and arena.fused_dbuf and not train_enc and not dr3_coeff and not discrete
                   and _dual_fits(arena, batch_size) and not _split_forward(N, batch_size))
        rd = lu.sample_move_and_augment(buffer=buffer, batch_size=batch_size, augmenter=augmenter,
                                        aug_mix=aug_mix, per=per, _defer_gather=dual_ok,
                                        _invariance=bool(encoder_lambda))
        o, a, r, o1, d = rd["primary_batch"]
        B = r.shape[0]
        # The online critics' FORWARD does not depend on the TD target: on a second stream (a parallel graph
        # branch) it overlaps the actor -> target critics -> TD-target chain, which occupies few CUs.
        branch = None
        if arena.fused and not train_enc and not dr3_coeff and _split_forward(N, B):
            s_rep = lu.encode(agent.encoder, o)
            X, ldx = _critic_input(rd.get("_ssac"), ws, f"cu.x{i}", s_rep, a, discrete)
            with engine.side_stream(dev, defer_join=True) as branch:
                with engine._timed("critic_fwd"):
                    h1, h2, q = engine.mlp_forward(arena, X, ldx, 0, B, ws, tag)
        co = cob = None
        if dual_ok and branch is None:
            # the critics' forward rides in the actor's launch (when compute_td_targets uses the fused sample
            # launch); the critic launch below is then only the backward half
            s_rep = lu.encode(agent.encoder, o)
            X, ldx = _critic_input(rd.get("_ssac"), ws, f"cu.x{i}", s_rep, a, discrete)
            co = (arena, X, ldx, ws.get(tag + ".h1", (N, B, H)), ws.get(tag + ".h2", (N, B, H)),
                  ws.get(tag + ".y", (N, B, qd)))
            if RANK1_BWD and qd == 1:
                # and the TD-independent half (rank-1 loss gradient) of the backward pass rides in the target critics' launch
                cob = (arena, co[3], co[4], a, a.stride(0), ws.get(tag + ".dz2", (N, B, H)),
                       ws.get(tag + ".dz1", (N, B, H)))
                # dz2u = W3 (.) [h2 > 0] need not leave the chained launch when the weight-gradient launch that follows
                # is the loss-fold form: its fc2 tiles rebuild it from the saved h2 while staging (same values, bit for
                # bit; 5 MB less written per update at the metric shape).  The chain launch reports whether it skipped.
                rd["_dz2_optional"] = (SKIP_DZ2 and FOLD_LOSS and B <= 4096 and arena.shadow is None
                                       and parallel.shard_of(agent) is None
                                       and H % 4 == 0)
        td, _ = lu.compute_td_targets(logs=logs, replay_dict=rd, agent=agent, target_agent=target_agent,
                                      ensemble_idx=i, ensemble_n=target_critic_ensemble_n,
                                      log_alphas=log_alphas, pop=pop, gamma=gamma,
                                      random_process=random_process, noise_clip=noise_clip,
                                      discrete=discrete, _slot=slot, _defer=arena.fused and LAZY_TD and not dr3_coeff,
                                      _co_forward=co, _co_backward=cob, _log_idx=ig)
        lu.ensure_gathered(rd.get("_ssac"))  # (no-op when the merged launch took the gather)
        co_done = bool(rd.pop("_co_fwd", False))
        bwd_done = bool(rd.pop("_co_bwd", False))
        if co_done:
            h1, h2, q = co[3], co[4], co[5]
        all_rd.append(rd)
        if ms_softmax:
            lu.member_sharded_softmax_scores(rd, agent, target_agent, ms, sm_table[ig])
        bw = 1.0 if ms is not None else \
            lu.compute_backup_weights(logs=logs, replay_dict=rd, agent=agent, target_agent=target_agent,
                                      weight_type=weight_type, weight_temp=weighted_bellman_temp,
                                      batch_size=batch_size, discrete=discrete, _slot=slot)
        preps.append(dict(arena=arena, rd=rd, branch=branch, co=co, td=td, co_done=co_done, bwd_done=bwd_done, bw=bw,
                          fwd=(h1, h2, q) if (co_done or branch is not None) else None,
                          xin=(s_rep, X, ldx) if (branch is not None or co is not None) else None))
    if ms is not None and weight_type is not None and weighted_bellman_temp is not None and E_glob > 1:
        if ms_softmax:
            wts = lu.member_sharded_softmax_finish(logs, sm_table, ms, weighted_bellman_temp, slot)
        else:
            wts = lu.member_sharded_sunrise_weights(logs, all_rd, agent, target_agent, ms, weighted_bellman_temp, discrete, slot)
        for i, P in enumerate(preps):
            P["bw"] = wts[ms.lo + i]
    for i, P in enumerate(preps):
        arena, rd, branch, co, td, co_done, bwd_done, bw = (P[k_] for k_ in ("arena", "rd", "branch", "co", "td",
                                                                             "co_done", "bwd_done", "bw"))
        N, qd, H = arena.n_nets, arena.out_dim, arena.hidden
        tag = f"cu.c{i}"
        train_enc = not lu.is_identity(agent.encoder)
        o, a, r, o1, d = rd["primary_batch"]
        B = r.shape[0]
        if P["fwd"] is not None:
            h1, h2, q = P["fwd"]
        if P["xin"] is not None:
            s_rep, X, ldx = P["xin"]
        if train_enc:
            # online encoder WITH gradient (learning.py:83): embedding goes straight into the critic input
            xin = ws.get(f"cu.x{i}", (B, arena.in_dim))
            inv = None
            okey = getattr(agent.encoder, "ssac_obs_key", "obs")
            # (the invariance constraint looks at the LAST member's batch only: learning.py:114-117 read the replay
            # dict the member loop left behind)
            lam_i = encoder_lambda if i == E - 1 else 0
            if lam_i and rd["augmented_obs"][0][okey] is not o[okey]:
                # encoder invariance on a partly augmented batch: the fully augmented observations need an encoder
                # pass of their own WITH gradient -- one stacked 2B-row pass [o ; ao], whose backward then receives
                # the critics' gradient in rows [0, B) and the constraint's in rows [B, 2B)
                from . import conv_encoder
                eng = conv_encoder.conv_engine(agent.encoder, dev)
                img2 = ws.get("cu.img2", (2 * B,) + tuple(o[okey].shape[1:]))
                img2[:B].copy_(o[okey])
                img2[B:].copy_(rd["augmented_obs"][0][okey])
                sall = ws.get("cu.sall", (2 * B, eng.emb))
                eng.forward(img2, sall, eng.emb, True)
                xin[:, :eng.emb].copy_(sall[:B])
                s_rep, as_rep, stacked = xin[:, :eng.emb], sall[B:], True
            else:
                s_rep = lu.encode(agent.encoder, o, dst=xin, save=True)
                as_rep, stacked = s_rep, False
            if lam_i:
                oo = rd["original_obs"][0]
                if oo[okey] is o[okey]:
                    os_rep = ws.get("cu.osrep", (B, s_rep.shape[1]))
                    os_rep.copy_(s_rep)   # un-augmented batch: the target of the constraint is the embedding itself
                else:
                    os_rep = lu.encode(agent.encoder, oo, dst=ws.get("cu.osrep", (B, s_rep.shape[1])), save=False)
                inv = (as_rep, os_rep, encoder_lambda, stacked)
                logs["encoder_constraint_loss"] = slot[lu.L_ENC_INV]
            if not discrete:
                xin[:, s_rep.shape[1]:].copy_(a)
            X, ldx = xin, xin.stride(0)
        elif branch is None and co is None:
            s_rep = lu.encode(agent.encoder, o)
            X, ldx = _critic_input(rd.get("_ssac"), ws, f"cu.x{i}", s_rep, a, discrete)
        if encoder_lambda and not train_enc:
            # identity encoder: augmented == original observations (only the identity augmentation applies to vectors),
            # the constraint is exactly zero and has no parameter to reach (learning_utils.py:401-409)
            logs["encoder_constraint_loss"] = slot[lu.L_ENC_INV]
        shard = parallel.shard_of(agent)
        n_glob = N if shard is None else shard.num_critics  # loss is averaged over the GLOBAL ensemble
        popart = agent.popart[i]
        weight_ptr = 0
        if not isinstance(bw, float):
            weight_ptr = bw.data_ptr()  # imp_weights is ones(1) on the uniform path
        if per:
            # learning.py:96-98 multiplies (B,1) errors by the (B,) importance weights: a (B,B) outer product
            # whose mean is mean(w) * mean(bw * err^2) -- i.e. every row's weight is scaled by mean(w)
            wrow = ws.get(f"cu.w{i}", (B, 1))
            scale = rd["imp_weights"].mean().to(torch.float32)
            if isinstance(bw, float):
                wrow.fill_(1.0)
            else:
                wrow.copy_(bw.view(B, 1))
            wrow.mul_(scale)
            weight_ptr = wrow.data_ptr()
        pp, dopop = (popart.ptr if popart else 0), (1 if (popart and pop) else 0)
        ttot = engine.bf16_tiles_total(arena) if arena.shadow is not None else engine.wgrad_tiles_total(arena)
        ss = ws.get(f"cu.ss{i}", (N * ttot,))
        dq = ws.get(f"cu.dq{i}", (N, B, qd))
        grads = ws.get(f"cu.g{i}", (arena.params.numel(),), zero=True) if critic_clip else None
        if dr3_coeff:
            # DR3 (learning.py:100-108): the critics also run on (s', a'); both batches go through the per-layer
            # kernels as ONE stacked 2B-row batch, the co-adaptation gradient enters at the fc2 pre-activations
            assert not train_enc and shard is None, "DR3 is supported for identity encoders on a single rank"
            x1 = rd.get("_x1")
            X1 = x1 if x1 is not None else lu.encode(target_agent.encoder, o1)
            Xc = ws.get(f"cu.xcat{i}", (2 * B, arena.in_dim))
            Xc[:B].copy_(torch.as_strided(X, (B, arena.in_dim), (ldx, 1)))
            Xc[B:].copy_(X1[:, :arena.in_dim])
            ch1, ch2, cq = engine.mlp_forward(arena, Xc, arena.in_dim, 0, 2 * B, ws, tag + ".dr3", force_layers=True)
            qc = ws.get(tag + ".dr3.qc", (N, B, qd))
            qc.copy_(cq[:, :B])
            dqc = ws.get(tag + ".dr3.dqc", (N, B, qd))
            check(lib.ssac_critic_loss_bwd(qc.data_ptr(), N, B, qd, a.data_ptr(), a.stride(0), td.data_ptr(),
                                           weight_ptr, pp, dopop, float(E_glob * n_glob), dqc.data_ptr(),
                                           slot.data_ptr(), st))
            dq2 = ws.get(tag + ".dr3.dq", (N, 2 * B, qd), zero=True)
            dq2[:, :B].copy_(dqc)
            nblk = int(lib.ssac_dr3_blocks())
            dparts = ws.get(tag + ".dr3.parts", (nblk,))
            coef = float(dr3_coeff) / (E_glob * n_glob) / (N * B)

            def dr3_hook(dz2_, _h2=ch2, _parts=dparts):
                check(lib.ssac_dr3_add(dz2_.data_ptr(), _h2.data_ptr(), N, B, H, coef, _parts.data_ptr(), st))
            engine.mlp_backward(arena, dq2, Xc, arena.in_dim, 0, ch1, ch2, 2 * B, ws, tag + ".dr3", adam=adam,
                                adam_key=("critic", i), grads=grads, sumsq=ss, after_dz2=dr3_hook)
            fca = dparts.sum() / (N * B)
            logs[f"dr3_dotproduct_{i}"] = fca
            # the logged overall loss includes the regulariser (learning.py:108, 133)
            slot[lu.L_CRITIC_LOSS:lu.L_CRITIC_LOSS + 1].add_(fca * (float(dr3_coeff) / (E_glob * n_glob)))
            fused_logs.append(None)
        elif arena.fused:
            dz2 = ws.get(tag + ".dz2", (N, B, H))
            dz1 = ws.get(tag + ".dz1", (N, B, H))
            tiles = int(lib.ssac_fused_row_tiles(C.byref(arena.desc()), B, N))
            parts = ws.get(tag + ".parts", (N * tiles * 2,))
            spec = getattr(td, "_ssac_spec", None)  # the TD target is evaluated inside the critic launch
            spec_ptr = C.addressof(spec) if spec is not None else 0
            lossfold = None
            w3_snapshot = rd.pop("_dz2_skipped", None)
            dz2_skipped = w3_snapshot is not None
            if dz2_skipped and not (bwd_done and FOLD_LOSS and B <= 4096):
                raise RuntimeError("internal: the chained launch skipped dz2u but no loss-fold weight-gradient launch follows")
            if bwd_done and FOLD_LOSS and B <= 4096:
                # dz2u / dz1u exist already; what depends on the TD target is one scalar per (net, row), dL/dq, and the
                # weight-gradient launch evaluates it itself (per workgroup, in LDS): no loss launch at all
                fparts = ws.get(tag + ".fparts", (N * 2,))
                lossfold = dict(q=q, td_ptr=0 if spec is not None else td.data_ptr(), spec_ptr=spec_ptr,
                                weight_ptr=weight_ptr, popart_ptr=pp, pop=dopop, denom=float(E_glob * n_glob),
                                partials=fparts, dz2_from_h2=dz2_skipped, w3_snapshot=w3_snapshot)
                if arena.shadow is not None:
                    lossfold["bf"] = arena.bf_buffers(ws, "cu", B)
                cap = engine.CAPTURE
                if cap is not None and cap.deferred is not None and cap.deferred_chain and spec is not None:
                    cap.deferred_used = True
                    # recorded update: the launch leaves partials + TD statistics behind and advances the input ring;
                    # the next update's first launch (or a flush) writes the ring slot
                    late_ptr = 0
                    if cap.late is not None:
                        late_ptr = cap.late
                        cap.late_used = True
                        lossfold["late_target"] = cap.late_target.params
                        if arena.shadow is not None:
                            lossfold["target_shadow"] = cap.late_target.shadow
                    lossfold["logfold"] = _lib.LogFold(0, slot.data_ptr(), 0, cap.feed, cap.deferred.td_stats, late_ptr)
                elif FOLD_LOGS and E_glob == 1 and not critic_clip:
                    # the log finalisation rides in the weight-gradient launch (its last workgroup to arrive): no logs launch
                    lossfold["logfold"] = _lib.LogFold(
                        ws.get("cu.done", (1,), dtype=torch.int32, zero=True).data_ptr(), slot.data_ptr(),
                        td._ssac_logs.data_ptr() if spec is not None else 0,
                        cap.feed if (cap is not None and cap.feed) else 0, 0, 0)
            elif bwd_done:
                # ... or a single-workgroup launch writes the N x B scalars for the weight-gradient launch to read
                if spec is not None:
                    check(lib.ssac_critic_loss_bwd_lazy(q.data_ptr(), N, B, qd, a.data_ptr(), a.stride(0), spec_ptr,
                                                        weight_ptr, pp, dopop, float(E_glob * n_glob), dq.data_ptr(),
                                                        slot.data_ptr(), st))
                else:
                    check(lib.ssac_critic_loss_bwd(q.data_ptr(), N, B, qd, a.data_ptr(), a.stride(0), td.data_ptr(),
                                                   weight_ptr, pp, dopop, float(E_glob * n_glob), dq.data_ptr(),
                                                   slot.data_ptr(), st))
            elif branch is not None or co_done:
                # loss gradient + head backward + backward-data on the saved forward: ONE launch
                if branch is not None:
                    branch.join()
                with engine._timed("critic_bwd") as tm:
                    for _ in range(tm.reps):  # 1, except under bench.py's live kernel timing (idempotent launch)
                        check(lib.ssac_critic_bwd_fused(
                            C.byref(arena.desc()), B, td.data_ptr(), weight_ptr, a.data_ptr(), a.stride(0), pp,
                            dopop, float(E_glob * n_glob), h1.data_ptr(), h2.data_ptr(), q.data_ptr(), dq.data_ptr(),
                            dz2.data_ptr(), dz1.data_ptr(), parts.data_ptr(), spec_ptr, st))
            else:
                # forward of all N critics + loss gradient + backward-data: ONE launch
                h1 = ws.get(tag + ".h1", (N, B, H))
                h2 = ws.get(tag + ".h2", (N, B, H))
                q = ws.get(tag + ".y", (N, B, qd))
                with engine._timed("critic_fused") as tm:
                    for _ in range(tm.reps):  # 1, except under bench.py's live kernel timing (idempotent launch)
                        check(lib.ssac_critic_fwd_bwd_fused(
                            C.byref(arena.desc()), X.data_ptr(), ldx, B, td.data_ptr(), weight_ptr, a.data_ptr(),
                            a.stride(0), pp, dopop, float(E_glob * n_glob), h1.data_ptr(), h2.data_ptr(), q.data_ptr(),
                            dq.data_ptr(), dz2.data_ptr(), dz1.data_ptr(), parts.data_ptr(), spec_ptr, st))
            if train_enc:  # dL/d(embedding) = sum over critics of dz1 W1[:, :emb], BEFORE W1 is updated
                dX = ws.get(tag + ".dx", (N, B, arena.in_dim))
                check(lib.ssac_mlp_layer_dgrad(C.byref(arena.desc()), 0, 0, N, dz1.data_ptr(), H, B * H, 0, 0, 0,
                                               B, dX.data_ptr(), arena.in_dim, B * arena.in_dim, st))
                _encoder_step(agent.encoder, encoder_optimizer, encoder_clip, dX, s_rep.shape[1], ws, slot, dev, inv,
                              accumulate=i > 0, step=i == E - 1)
            if arena.shadow is not None and lossfold is None:
                raise NotImplementedError(
                    "bf16 mode covers the chained critic update (continuous single-output critics, stochastic actor, "
                    "identity encoder, with or without PopArt / gradient clipping; no DR3, no discrete critics)")
            folded = engine.weight_grads(arena, X, ldx, 0, h1, h2, dq, dz2, dz1, B, adam=adam,
                                         adam_key=("critic", i), grads=grads, sumsq=ss,
                                         rowscale=dq if (bwd_done and lossfold is None) else None, lossfold=lossfold,
                                         target=lossfold.get("late_target") if lossfold is not None else None)
            if folded:
                logs_done_in_wgrad = True
                if lossfold["logfold"].feed:
                    engine.CAPTURE.published = True
            if lossfold is not None:
                fused_logs.append((lossfold["partials"], N, 1, B, n_glob, td))
            else:
                fused_logs.append((parts, 0 if bwd_done else N, tiles, B, n_glob, td))
        else:
            if arena.shadow is not None:
                raise NotImplementedError("bf16 mode needs the fused kernel family (hidden % 32 == 0, <= 256)")
            h1, h2, q = engine.mlp_forward(arena, X, ldx, 0, B, ws, tag)
            check(lib.ssac_critic_loss_bwd(q.data_ptr(), N, B, qd, a.data_ptr(), a.stride(0),
                                           td.data_ptr(), weight_ptr, pp, dopop, float(E_glob * n_glob),
                                           dq.data_ptr(), slot.data_ptr(), st))
            if train_enc:
                dX = engine.mlp_backward(arena, dq, X, ldx, 0, h1, h2, B, ws, tag, need_dx=True, update=False)
                _encoder_step(agent.encoder, encoder_optimizer, encoder_clip, dX, s_rep.shape[1], ws, slot, dev, inv,
                              accumulate=i > 0, step=i == E - 1)
                dz2, dz1 = ws.get(tag + ".dz2", (N, B, arena.hidden)), ws.get(tag + ".dz1", (N, B, arena.hidden))
                engine.weight_grads(arena, X, ldx, 0, h1, h2, dq, dz2, dz1, B, adam=adam,
                                    adam_key=("critic", i), grads=grads, sumsq=ss)
            else:
                engine.mlp_backward(arena, dq, X, ldx, 0, h1, h2, B, ws, tag, adam=adam,
                                    adam_key=("critic", i), grads=grads, sumsq=ss)
            fused_logs.append(None)
        if critic_clip:
            clip_members.append((arena, ("critic", i), grads, ss))
        member_ss.append(ss)
        rd["td_target"] = td
        replay_dicts.append(rd)
    if critic_clip:
        _clip_and_step(adam, clip_members, critic_clip, None, member_shard=ms)
    # encoder: identity encoders carry no trainable tensor on this path (their dummy Linear(1,1)
    # never receives a gradient, nets/__init__.py:24), so encoder_optimizer.step() is a no-op.
    logs["losses/last_member_critic_td_error"] = slot[lu.L_TD_ERR]
    logs["losses/critic_overall_loss"] = slot[lu.L_CRITIC_LOSS]
    if ms is not None:
        # learning.py:135 picks over the GLOBAL ensemble (the same Python draw on every rank); a rank that does not hold
        # the picked member logs the gradient norm of its first one
        k = ms.local(rng.choice(range(E_glob))) or 0
    else:
        pick = agent.critics[0] if engine.CAPTURE is not None else rng.choice(agent.critics)  # learning.py:135
        k = next(j for j, c in enumerate(agent.critics) if c is pick)
    clip_ctl = adam.ctl.ptr if critic_clip else 0
    done_norm = False
    for j, fl in enumerate(fused_logs):
        if fl is None or logs_done_in_wgrad:
            done_norm = done_norm or logs_done_in_wgrad
            continue
        parts, n_, tiles_, b_, ng_, td_ = fl
        spec_ = getattr(td_, "_ssac_spec", None)
        want = j == k
        cap = engine.CAPTURE
        last = cap is not None and cap.feed and j == len(fused_logs) - 1 and (done_norm or want)
        check(lib.ssac_critic_logs(parts.data_ptr(), n_, tiles_, b_, float(E_glob * ng_),
                                   member_ss[k].data_ptr() if want else 0, member_ss[k].numel() if want else 0,
                                   clip_ctl, slot.data_ptr(), C.addressof(spec_) if spec_ is not None else 0,
                                   td_._ssac_logs.data_ptr() if spec_ is not None else 0,
                                   cap.feed if last else 0, st))
        if last:
            cap.published = True
        done_norm = done_norm or want
    if not done_norm:
        check(lib.ssac_group_norms(member_ss[k].data_ptr(), 1, member_ss[k].numel(), clip_ctl,
                                   slot[lu.L_CRITIC_GN:].data_ptr(), st))
    logs["gradients/critic_random_grad"] = slot[lu.L_CRITIC_GN]
    logs["gradients/encoder_criticloss_grad_norm"] = slot[lu.L_ENC_GN]
    if update_priorities:  # learning.py:139-140: advantage-based priorities on the LAST member's batch
        lu.adjust_priorities(logs, replay_dicts[-1], agent, buffer)
    return logs, replay_dicts


def _actor_chain_form(a_arena, A, B, c_arena):
    """does this member's online actor update take the chained launch (ssac_actor_chain_fused)?  ONE predicate for the
    launch selection in _online_actor_update and for the recording's choice of noise source in online_actor_update: a
    member on the three-launch form reads its noise from a buffer, which a recording must own and refill before every replay.
    (B <= 2048 = SSAC_ACTOR_CHAIN_MAX_ROWS: the chained launch's actor workgroups wait for critic tiles dispatched behind
    them and must leave them CUs to run on.)
    Round 6: an actor whose carve holds only ONE weight-staging buffer (Humanoid's 376 -> 256 -> 34) can take the chained
    launch too, its two passes single-buffered -- `fused`, not `fused_dbuf` -- and the form is taken only while the launch
    is ONE RESIDENT ROUND (actor tiles + critic tiles <= 256 workgroups).  Measured: Humanoid with N = 16 critics is 32 + 512
    workgroups; the actor workgroups then sit on 32 CUs through two and a half rounds of critic tiles and the chained launch
    is SLOWER than the three launches, 177.2 against 162.8 us per actor update (profiles/r6_kernel_stats.md)."""
    if not (ACTOR_CHAIN and a_arena.fused and A <= 32 and a_arena.hidden * A <= 512 * 9 and B <= 2048 and c_arena.fused_dbuf
            and c_arena.out_dim == 1):
        return False
    tiles_c = int(lib.ssac_fused_row_tiles(C.byref(c_arena.desc()), B, c_arena.n_nets)) * c_arena.n_nets
    return (B + 15) // 16 + tiles_c <= 256


def _actor_fused_member(kind, random_process, use_baseline, clip, a_arena, c_arena):
    """does this member's online actor update take the fused launches at all (the sharded and the unsharded branch of
    _online_actor_update share it; which of the two is decided by parallel.shard_of at the call site)?"""
    return bool(FUSED_ACTOR and kind == "stochastic" and random_process is None and not use_baseline and not clip
                and a_arena.fused and c_arena.fused_dbuf and c_arena.out_dim == 1)


def _member_noise_in_kernel(chain, rec_eps):
    """the chained launch with the stock generator draws the member's noise inside the kernel (no buffer, no draw from
    the device generator) -- unless a recording owns noise buffers for it"""
    return bool(chain and lu.IN_KERNEL_NOISE and rng.normal_is_stock() and rec_eps is None)


def _actor_noise_in_kernel(agent, batch_size, dev):
    """recorded actor update: the noise is drawn inside the launches only if EVERY member takes the chained launch.
    (the arenas' shapes cannot change under a recording, so the per-form answer is cached on the agent; what CAN change
    between calls -- a noise hook, the IN_KERNEL_NOISE / ACTOR_CHAIN switches -- is re-read every time)"""
    if not (lu.IN_KERNEL_NOISE and rng.normal_is_stock()):
        return False
    key = (batch_size, str(dev), ACTOR_CHAIN, tuple(id(a_) for a_ in agent.actors))
    memo = agent.__dict__.setdefault("_ssac_chain_form_memo", {})
    if key not in memo:
        if len(memo) > 16:
            memo.clear()
        memo[key] = all(_actor_chain_form(engine.bind_arena(a_, "self", [a_], dev), a_.action_size, batch_size, c_.arena(dev))
                        for a_, c_ in zip(agent.actors, agent.critics))
    return memo[key]


class _RecordedActor:
    def __init__(self):
        self.calls, self.list, self.blk, self.eps, self.index = 0, None, None, None, None
        self.in_kernel = False
        self.published = False

    def __del__(self):
        if self.list:
            try:
                lib.ssac_launch_list_free(self.list)
            except Exception:
                pass


def online_actor_update(buffer, agent, pop, actor_optimizer, log_alphas, batch_size, clip,
                        random_process, noise_clip, augmenter, aug_mix, premade_replay_dicts=None,
                        per=False, discrete=False, use_baseline=False):
    """learning.py:344-421.  On the fused path (continuous stochastic actor, no clipping / exploration process /
    baseline) the update is four launches + a log launch; called repeatedly on the batch buffers of a recorded critic
    update (premade_replay_dicts at fixed addresses) it is recorded after two calls and re-issued from ONE C call."""
    engine.require_gpu()
    lu.ensure_adopted(agent, buffer)
    kw = dict(buffer=buffer, agent=agent, pop=pop, actor_optimizer=actor_optimizer, log_alphas=log_alphas,
              batch_size=batch_size, clip=clip, random_process=random_process, noise_clip=noise_clip,
              augmenter=augmenter, aug_mix=aug_mix, premade_replay_dicts=premade_replay_dicts, per=per,
              discrete=discrete, use_baseline=use_baseline)
    dev = log_alphas[0].device
    recordable = (USE_GRAPHS and LAUNCH_MODE == "list" and FUSED_ACTOR and engine.CAPTURE is None
                  and premade_replay_dicts is not None and not discrete and not clip and random_process is None
                  and not use_baseline and parallel.shard_of(agent) is None
                  and parallel.member_shard_of(agent) is None and lu.is_identity(agent.encoder)
                  and all(lu.actor_kind(a_) == "stochastic" and engine.bind_arena(a_, "self", [a_], dev).fused
                          for a_ in agent.actors)
                  and all(c_.arena(dev).fused_dbuf and c_.arena(dev).out_dim == 1 for c_ in agent.critics))
    if not recordable:
        return _online_actor_update(**kw)
    ptrs = tuple(lu.encode(agent.encoder, rd_["primary_batch"][0]).data_ptr() for rd_ in premade_replay_dicts)
    key = (id(actor_optimizer), ptrs, batch_size, bool(pop), tuple(id(la_) for la_ in log_alphas), engine.USE_FUSED)
    cache = agent.__dict__.setdefault("_ssac_actor_rec", {})
    rec = cache.get(key)
    if rec is None:
        if len(cache) > 8:
            cache.clear()
        rec = cache[key] = _RecordedActor()
    rec.calls += 1
    if rec.calls <= 2:
        return _online_actor_update(**kw)
    A = agent.actors[0].action_size
    ring = lu.ring_for(dev)
    if rec.list is None:
        rec.blk = torch.zeros(lu.LOG_WIDTH, device=dev)
        # the chained launch (ACTOR_CHAIN) with the stock generator draws its noise in the kernel: no buffers, no launches
        # (... if every member takes it: a member on the three-launch form reads a buffer this recording must own)
        rec.in_kernel = _actor_noise_in_kernel(agent, batch_size, dev)
        rec.eps = None if rec.in_kernel else [torch.empty(batch_size, A, device=dev) for _ in agent.actors]
    elif rec.in_kernel != _actor_noise_in_kernel(agent, batch_size, dev):
        # a noise hook was installed / removed (or the form switched) since the recording: record again
        del cache[key]
        return online_actor_update(**kw)
    if rec.eps is not None:
        for e_ in rec.eps:
            rng.draw_normal_into(e_)  # a_dist.rsample() of every member, in member order (learning.py:392)
    slot_i = ring.advance()   # this call's slot of the log ring
    if rec.list is None:
        pub = {"buf": ring.buf, "slot": slot_i, "published": False}
        check(lib.ssac_record_begin())
        try:
            logs = _online_actor_update(**kw, _rec_blk=rec.blk, _rec_eps=rec.eps, _rec_ring=pub)
        finally:
            rec.list = lib.ssac_record_end()
        rec.published = pub["published"]   # (the log launch writes the ring slot itself: single-member agents)
        base = rec.blk.data_ptr()
        rec.index = {k_: (v.data_ptr() - base) // 4 for k_, v in logs.items()}
    else:
        ns_upd = lu.noise_stream(agent, dev)
        # (this update's number: tags, noise draws -- and, when the log launch publishes, the ring slot it writes)
        if rec.published:
            check(lib.ssac_replay_value2(rec.list, engine.stream(), ns_upd[2], slot_i))
        else:
            check(lib.ssac_replay_value(rec.list, engine.stream(), ns_upd[2]))
        ns_upd[2] += 1
        rng.choice(agent.actors)  # learning.py:417-419 (keeps the Python RNG stream in step)
    if not rec.published:
        ring.buf[slot_i].copy_(rec.blk)   # this call's log block -> its own ring slot (device-to-device, no sync)
    blk = ring.buf[slot_i]
    return {k_: blk[i] for k_, i in rec.index.items()}


def _online_actor_update(buffer, agent, pop, actor_optimizer, log_alphas, batch_size, clip,
                         random_process, noise_clip, augmenter, aug_mix, premade_replay_dicts=None,
                         per=False, discrete=False, use_baseline=False, _rec_blk=None, _rec_eps=None, _rec_ring=None):
    E = agent.ensemble_size
    dev = log_alphas[0].device
    ws = lu.agent_ws(agent, dev)
    adam = engine.adam_group(actor_optimizer, dev)
    # a RECORDED update clears its (fixed) log block and advances the Adam step by a recorded launch: the chained launch of
    # the first member does it (ACTOR_CHAIN), or a begin launch in front
    begin_folded = _rec_blk is not None and ACTOR_CHAIN
    if _rec_blk is not None:
        slot = _rec_blk
        if not begin_folded:
            check(lib.ssac_begin_update(slot.data_ptr(), lu.LOG_WIDTH, adam.ctl.ptr, 0, engine.stream()))
    else:
        slot = lu.log_block(dev, adam)
    # this update's number: hand-off tags and in-kernel noise draws of the chained launches (checkpointed with the agent's
    # noise stream, so a resumed run continues the sequence)
    ns_upd = lu.noise_stream(agent, dev)
    first_fused = True
    logs = {}
    st = engine.stream()
    # member-sharded rank (parallel.MemberShard): the loss is averaged over the GLOBAL ensemble (learning.py:409), and the
    # members other ranks own still take their host draws here, in member order
    ms = parallel.member_shard_of(agent)
    E_glob = len(agent.actors) if ms is None else ms.ensemble_size
    inv_e = 1.0 / E_glob
    clip_members, member_ss = [], []
    members = list(zip(agent.ensemble, agent.popart, log_alphas))
    for ig in range(E_glob):
        i = ig if ms is None else ms.local(ig)
        if i is None:
            assert not use_baseline, "use_baseline is not supported on member-sharded ranks"
            if premade_replay_dicts is None:
                lu.sample_move_and_augment(buffer=buffer, batch_size=batch_size, augmenter=augmenter, aug_mix=aug_mix, per=per)
            # (what the member's owner consumes from the device generator: nothing when its chained launch draws the noise
            # in the kernel -- members share their shapes, so this rank's first member answers for the absent one)
            a0_, c0_ = agent.actors[0], agent.critics[0]
            ar0_, cr0_ = engine.bind_arena(a0_, "self", [a0_], dev), c0_.arena(dev)
            owner_in_kernel = (_actor_fused_member(lu.actor_kind(a0_), random_process, use_baseline, clip, ar0_, cr0_)
                               and parallel.shard_of(agent) is None
                               and _member_noise_in_kernel(_actor_chain_form(ar0_, a0_.action_size, batch_size, cr0_), _rec_eps))
            if not owner_in_kernel:
                lu.skip_actor_draws(agent.actors[0], batch_size, dev, random_process)
            continue
        (actor, critic), popart, log_alpha = members[i]
        if premade_replay_dicts is not None:
            rd = premade_replay_dicts[i]
        else:
            rd = lu.sample_move_and_augment(buffer=buffer, batch_size=batch_size, augmenter=augmenter,
                                            aug_mix=aug_mix, per=per)
        o = rd["primary_batch"][0]
        s_rep = lu.encode(agent.encoder, o)  # no gradient to the encoder (learning.py:378-380)
        B, S = s_rep.shape
        lds = lu._row_stride(s_rep)
        a_arena = engine.bind_arena(actor, "self", [actor], dev)
        c_arena = critic.arena(dev)
        N = c_arena.n_nets
        kind = lu.actor_kind(actor)
        pp, dopop = (popart.ptr if popart else 0), (1 if (popart and pop) else 0)
        shard = parallel.shard_of(agent)
        fused_member = _actor_fused_member(kind, random_process, use_baseline, clip, a_arena, c_arena)
        if fused_member and shard is not None:
            # ---- critic-sharded rank: the same fused launches, cut at the two exchange steps of SURVEY 8(e) -- sample +
            #      [s|a] rows; the LOCAL critics' forward + dQ/da; local arg-min, MIN over the ranks, the global arg-min's
            #      owner keeps its dQ/da row, SUM over the ranks; actor backward on (global min Q, routed dQ/da)
            A, H = actor.action_size, a_arena.hidden
            xpi = ws.get(f"au.x{i}", (B, S + A))
            logp = ws.get(f"au.logp{i}", (B,))
            ah1, ah2 = ws.get(f"au.a{i}.h1", (1, B, H)), ws.get(f"au.a{i}.h2", (1, B, H))
            aout = ws.get(f"au.a{i}.y", (1, B, 2 * A))
            eps = rng.draw_normal((B, A), dev)  # a_dist.rsample() (learning.py:392): every rank makes the same draw
            check(lib.ssac_actor_sample_concat_fused(C.byref(a_arena.desc()), s_rep.data_ptr(), lds, B, eps.data_ptr(),
                                                     float(actor.log_std_low), float(actor.log_std_high),
                                                     xpi.data_ptr(), S + A, logp.data_ptr(), ah1.data_ptr(),
                                                     ah2.data_ptr(), aout.data_ptr(), 0, st))
            q = ws.get(f"au.c{i}.y", (N, B, 1))
            dxu = ws.get(f"au.dxu{i}", (N, B, A))
            check(lib.ssac_critic_fwd_dx_fused(C.byref(c_arena.desc()), xpi.data_ptr(), S + A, B, S, A, q.data_ptr(),
                                               dxu.data_ptr(), st))
            qloc, qglob = ws.get(f"au.qloc{i}", (B,)), ws.get(f"au.qmin{i}", (B,))
            dsel = ws.get(f"au.da{i}", (1, B, A))
            check(lib.ssac_actor_route_local(q.data_ptr(), dxu.data_ptr(), N, B, A, qloc.data_ptr(), qglob.data_ptr(),
                                             dsel.data_ptr(), st))
            parallel.all_reduce_min(qglob)
            # (bit-equal minima on two ranks: the lowest rank keeps the row, as torch.min keeps the first index)
            claim = ws.get(f"au.claim{i}", (B,))
            check(lib.ssac_actor_route_claim(qloc.data_ptr(), qglob.data_ptr(), B, shard.rank, claim.data_ptr(), st))
            parallel.all_reduce_min(claim)
            check(lib.ssac_actor_route_mask(claim.data_ptr(), shard.rank, B, A, dsel.data_ptr(), st))
            parallel.all_reduce_sum(dsel)
            tiles = int(lib.ssac_fused_row_tiles(C.byref(a_arena.desc()), B, 1))
            parts = ws.get(f"au.parts{i}", (tiles,))
            d_out = ws.get(f"au.dout{i}", (1, B, 2 * A))
            dz2, dz1 = ws.get(f"au.a{i}.dz2", (1, B, H)), ws.get(f"au.a{i}.dz1", (1, B, H))
            check(lib.ssac_actor_bwd_fused(C.byref(a_arena.desc()), ah1.data_ptr(), ah2.data_ptr(), B, qglob.data_ptr(), 1,
                                           dsel.data_ptr(), aout.data_ptr(), eps.data_ptr(), logp.data_ptr(),
                                           log_alpha.data_ptr(), 1, float(actor.log_std_low),
                                           float(actor.log_std_high), inv_e, pp, dopop, d_out.data_ptr(),
                                           dz2.data_ptr(), dz1.data_ptr(), parts.data_ptr(), st))
            ttot = engine.wgrad_tiles_total(a_arena)
            ss = ws.get(f"au.ss{i}", (ttot,))
            one = E_glob == 1   # (then the logged actor is this one: both logs in one launch)
            # (a recorded update of a single-member agent: the log step also writes the finished block to its slot of the log
            #  ring -- the replay names the slot, ssac_replay_value2 -- instead of a copy launch behind every replay)
            pub = _rec_ring if (one and _rec_ring is not None and _rec_blk is not None) else None
            folded = False
            if one and engine.actor_fold_applies(a_arena, None, ss, None, None, None):
                # ... and the log step itself rides in the weight-gradient launch: its last workgroup to finish sums the
                # tiles' loss terms and the gradient-norm partials (ssac_actor_logs' arithmetic), no launch of its own
                done = ws.get(f"au.logdone{i}", (1,), dtype=torch.int32, zero=True)
                af = _lib.ActorLogFold(done.data_ptr(), parts.data_ptr(), tiles, B, inv_e, lu.LOG_WIDTH,
                                       slot[lu.L_ACTOR_LOSS:].data_ptr(), slot[lu.L_ACTOR_GN:].data_ptr(), slot.data_ptr(),
                                       pub["buf"].data_ptr() if pub else 0, pub["slot"] if pub else 0)
                folded = bool(engine.weight_grads(a_arena, s_rep, lds, 0, ah1, ah2, d_out, dz2, dz1, B, adam=adam,
                                                  adam_key=("actor", i), sumsq=ss, actor_fold=af))
            if not folded:
                engine.weight_grads(a_arena, s_rep, lds, 0, ah1, ah2, d_out, dz2, dz1, B, adam=adam,
                                    adam_key=("actor", i), sumsq=ss)
                check(lib.ssac_actor_logs(parts.data_ptr(), tiles, B, inv_e, ss.data_ptr() if one else 0, ss.numel(),
                                          slot[lu.L_ACTOR_LOSS:].data_ptr(), slot[lu.L_ACTOR_GN:].data_ptr() if one else 0,
                                          slot.data_ptr(), lu.LOG_WIDTH, pub["buf"].data_ptr() if pub else 0,
                                          pub["slot"] if pub else 0, st))
            if pub is not None:
                pub["published"] = True
            member_ss.append(None if one else ss)
            continue
        if fused_member and shard is None:
            # ---- four launches instead of ~15 (include/ssac_hip.h, "the online actor update"): sample + [s|a] rows,
            #      critics' forward + unscaled dQ/da, arg-min routing + tanh-normal backward + actor backward-data,
            #      weight gradients + Adam; then one launch for the two log values
            A, H = actor.action_size, a_arena.hidden
            xpi = ws.get(f"au.x{i}", (B, S + A))
            logp = ws.get(f"au.logp{i}", (B,))
            ah1, ah2 = ws.get(f"au.a{i}.h1", (1, B, H)), ws.get(f"au.a{i}.h2", (1, B, H))
            aout = ws.get(f"au.a{i}.y", (1, B, 2 * A))
            # the noise: with the stock generator the chained launch takes it from the engine's Philox stream INSIDE the
            # kernel (draw number = this update's number: no generator launch, no buffer); otherwise a draw -- into the
            # recording's fixed buffer, or a fresh one -- as a_dist.rsample() makes it (learning.py:392)
            chain = _actor_chain_form(a_arena, A, B, c_arena)
            in_kernel = _member_noise_in_kernel(chain, _rec_eps)
            eps = None   # (kept alive to the end of the member's launches: the kernels read it asynchronously)
            if in_kernel:
                eps_ptr = 0
            elif _rec_eps is not None:
                eps = _rec_eps[i]   # recorded update: the draw was written into a fixed buffer by the caller
                eps_ptr = eps.data_ptr()
            else:
                # (never inside a recording: a replay would re-read a buffer long returned to the allocator)
                assert _rec_blk is None, "recorded actor update: the noise must come from the kernel or from the recording's buffers"
                eps = rng.draw_normal((B, A), dev)
                eps_ptr = eps.data_ptr()
            q = ws.get(f"au.c{i}.y", (N, B, 1))
            dxu = ws.get(f"au.dxu{i}", (N, B, A))
            tiles = int(lib.ssac_fused_row_tiles(C.byref(a_arena.desc()), B, 1))
            parts = ws.get(f"au.parts{i}", (tiles,))
            d_out = ws.get(f"au.dout{i}", (1, B, 2 * A))
            dz2, dz1 = ws.get(f"au.a{i}.dz2", (1, B, H)), ws.get(f"au.a{i}.dz1", (1, B, H))
            if chain:
                ho = ws.get(f"au.handoff{i}", (int(lib.ssac_actor_chain_handoff_words(B, N, A)),), dtype=torch.int64,
                            zero=True)
                rs = None
                if in_kernel:
                    # (its own stream: the critic updates number their draws from the same seed)
                    # (the GLOBAL member index: a member-sharded rank draws its members' noise as the unsharded run does)
                    rs = _lib.Rng((ns_upd[0] ^ 0x5DEECE66D1CEB00C) & (2 ** 64 - 1), 0, (ig << 40) + ns_upd[2])
                fold = begin_folded and first_fused
                check(lib.ssac_actor_chain_fused(
                    C.byref(a_arena.desc()), s_rep.data_ptr(), lds, B, eps_ptr, C.byref(rs) if rs is not None else None,
                    float(actor.log_std_low), float(actor.log_std_high), xpi.data_ptr(), S + A, logp.data_ptr(),
                    ah1.data_ptr(), ah2.data_ptr(), aout.data_ptr(), C.byref(c_arena.desc()), q.data_ptr(),
                    dxu.data_ptr(), log_alpha.data_ptr(), 1, inv_e, pp, dopop, d_out.data_ptr(), dz2.data_ptr(),
                    dz1.data_ptr(), parts.data_ptr(), ho.data_ptr(), ns_upd[2],
                    slot.data_ptr() if fold else 0, lu.LOG_WIDTH if fold else 0, adam.ctl.ptr if fold else 0, st))
            else:
                if begin_folded and first_fused:   # (a member the chained launch does not take: the begin launch after all)
                    check(lib.ssac_begin_update(slot.data_ptr(), lu.LOG_WIDTH, adam.ctl.ptr, 0, st))
                check(lib.ssac_actor_sample_concat_fused(C.byref(a_arena.desc()), s_rep.data_ptr(), lds, B, eps_ptr,
                                                         float(actor.log_std_low), float(actor.log_std_high),
                                                         xpi.data_ptr(), S + A, logp.data_ptr(), ah1.data_ptr(),
                                                         ah2.data_ptr(), aout.data_ptr(), 0, st))
                check(lib.ssac_critic_fwd_dx_fused(C.byref(c_arena.desc()), xpi.data_ptr(), S + A, B, S, A, q.data_ptr(),
                                                   dxu.data_ptr(), st))
                check(lib.ssac_actor_bwd_fused(C.byref(a_arena.desc()), ah1.data_ptr(), ah2.data_ptr(), B, q.data_ptr(), N,
                                               dxu.data_ptr(), aout.data_ptr(), eps_ptr, logp.data_ptr(),
                                               log_alpha.data_ptr(), 1, float(actor.log_std_low),
                                               float(actor.log_std_high), inv_e, pp, dopop, d_out.data_ptr(),
                                               dz2.data_ptr(), dz1.data_ptr(), parts.data_ptr(), st))
            first_fused = False
            ttot = engine.wgrad_tiles_total(a_arena)
            ss = ws.get(f"au.ss{i}", (ttot,))
            one = E_glob == 1   # (then the logged actor is this one: both logs in one launch)
            # (a recorded update of a single-member agent: the log step also writes the finished block to its slot of the log
            #  ring -- the replay names the slot, ssac_replay_value2 -- instead of a copy launch behind every replay)
            pub = _rec_ring if (one and _rec_ring is not None and _rec_blk is not None) else None
            folded = False
            if one and engine.actor_fold_applies(a_arena, None, ss, None, None, None):
                # ... and the log step itself rides in the weight-gradient launch: its last workgroup to finish sums the
                # tiles' loss terms and the gradient-norm partials (ssac_actor_logs' arithmetic), no launch of its own
                done = ws.get(f"au.logdone{i}", (1,), dtype=torch.int32, zero=True)
                af = _lib.ActorLogFold(done.data_ptr(), parts.data_ptr(), tiles, B, inv_e, lu.LOG_WIDTH,
                                       slot[lu.L_ACTOR_LOSS:].data_ptr(), slot[lu.L_ACTOR_GN:].data_ptr(), slot.data_ptr(),
                                       pub["buf"].data_ptr() if pub else 0, pub["slot"] if pub else 0)
                folded = bool(engine.weight_grads(a_arena, s_rep, lds, 0, ah1, ah2, d_out, dz2, dz1, B, adam=adam,
                                                  adam_key=("actor", i), sumsq=ss, actor_fold=af))
            if not folded:
                engine.weight_grads(a_arena, s_rep, lds, 0, ah1, ah2, d_out, dz2, dz1, B, adam=adam,
                                    adam_key=("actor", i), sumsq=ss)
                check(lib.ssac_actor_logs(parts.data_ptr(), tiles, B, inv_e, ss.data_ptr() if one else 0, ss.numel(),
                                          slot[lu.L_ACTOR_LOSS:].data_ptr(), slot[lu.L_ACTOR_GN:].data_ptr() if one else 0,
                                          slot.data_ptr(), lu.LOG_WIDTH, pub["buf"].data_ptr() if pub else 0,
                                          pub["slot"] if pub else 0, st))
            if pub is not None:
                pub["published"] = True
            member_ss.append(None if one else ss)
            continue
        ah1, ah2, aout = engine.mlp_forward(a_arena, s_rep, lds, 0, B, ws, f"au.a{i}")
        if kind == "discrete":
            A = a_arena.out_dim
            _, _, q = engine.mlp_forward(c_arena, s_rep, lds, 0, B, ws, f"au.c{i}", save=False)
            if shard is not None:  # elementwise min over the local critics, then over the ranks
                qm = ws.get(f"au.qmin{i}", (1, B, A))
                lu._min_over_nets(q, N, B * A, qm)
                parallel.all_reduce_min(qm)
                q, N = qm, 1
            d_out = ws.get(f"au.dout{i}", (1, B, A))
            check(lib.ssac_discrete_actor_loss_bwd(aout.data_ptr(), q.data_ptr(), N, B, A,
                                                   log_alpha.data_ptr(), pp, dopop, inv_e,
                                                   d_out.data_ptr(), slot[lu.L_ACTOR_LOSS:].data_ptr(), st))
        else:
            A = actor.action_size
            xpi = lu._concat_buffer(ws, f"au.x{i}", s_rep, A)
            logp = ws.get(f"au.logp{i}", (B,))
            eps = rng.draw_normal((B, A), dev)  # a_dist.rsample() (learning.py:392)
            if kind == "stochastic":
                check(lib.ssac_tanh_normal_fwd(aout.data_ptr(), 2 * A, eps.data_ptr(), B, A,
                                               float(actor.log_std_low), float(actor.log_std_high),
                                               xpi.data_ptr(), S + A, S, logp.data_ptr(), st))
                use_entropy = 1
                if random_process is not None:
                    # exploration noise on the sampled action, no entropy term (learning.py:393-395); the gradient
                    # passes through the clamp unchanged (learning_utils.py:56-59)
                    noise = rng.draw_normal((B, A), dev)
                    check(lib.ssac_exploration_noise(xpi.data_ptr(), S + A, S, noise.data_ptr(),
                                                     float(random_process.current_scale),
                                                     float(noise_clip) if noise_clip is not None else 0.0, B, A, st))
                    use_entropy = 0
            elif random_process is not None:
                noise = rng.draw_normal((B, A), dev)
                check(lib.ssac_det_action_fwd(aout.data_ptr(), A, eps.data_ptr(), 1e-4, noise.data_ptr(),
                                              float(random_process.current_scale),
                                              float(noise_clip) if noise_clip is not None else 0.0, B, A,
                                              xpi.data_ptr(), S + A, S, st))
                use_entropy = 0
            else:
                # deterministic actor without a process: a = loc + 1e-4 eps, entropy term = its Normal(loc, 1e-4)
                # log-density (no gradient: a - loc does not depend on the actor; learning.py:396-399)
                check(lib.ssac_det_action_fwd(aout.data_ptr(), A, eps.data_ptr(), 1e-4, 0, 0.0, 0.0, B, A,
                                              xpi.data_ptr(), S + A, S, st))
                check(lib.ssac_det_logprob(eps.data_ptr(), B, A, logp.data_ptr(), st))
                use_entropy = 1
            ch1, ch2, q = engine.mlp_forward(c_arena, xpi, S + A, 0, B, ws, f"au.c{i}")
            dq = ws.get(f"au.dq{i}", (N, B, 1))
            qmin_ptr = 0
            if shard is not None:  # min over ALL critics: local min, then MIN all-reduce over ranks
                qmin = ws.get(f"au.qmin{i}", (B,))
                lu._min_over_nets(q, N, B, qmin)
                parallel.all_reduce_min(qmin)
                qmin_ptr = qmin.data_ptr()
            if use_baseline:
                # learning.py:401: vals = A(s, a_theta) = Q'(s, a_theta) - V(s) from the advantage estimator (four fresh
                # policy samples, drawn after the rsample; adv_estimator.py:31-36 applies the PopArt layer whenever
                # the member has one).  V has no gradient: the routing of dL/dq is the plain path's.
                assert shard is None, "use_baseline is not supported on critic-sharded ranks"
                res = agent.adv_estimator.evaluate(o, xpi[:, S:], i, want=("adv",))
                check(lib.ssac_actor_loss_bwd_adv(q.data_ptr(), N, B, logp.data_ptr(), log_alpha.data_ptr(),
                                                  use_entropy, pp, 1 if popart else 0, inv_e,
                                                  res["adv"].data_ptr(), dq.data_ptr(),
                                                  slot[lu.L_ACTOR_LOSS:].data_ptr(), st))
            else:
                check(lib.ssac_actor_loss_bwd(q.data_ptr(), N, B, logp.data_ptr(), log_alpha.data_ptr(),
                                              use_entropy, pp, dopop, inv_e, qmin_ptr, dq.data_ptr(),
                                              slot[lu.L_ACTOR_LOSS:].data_ptr(), st))
            # dQ/da through the arg-min critic of every row; critic weights are NOT updated here
            dX = engine.mlp_backward(c_arena, dq, xpi, S + A, 0, ch1, ch2, B, ws, f"au.c{i}",
                                     need_dx=True, update=False)
            n_dx, ld_dx, s_dx, col_dx = N, S + A, B * (S + A), S
            if shard is not None:  # SUM all-reduce of the (B x A) action gradient
                da = ws.get(f"au.da{i}", (1, B, A))
                torch.sum(dX[:, :, S:], dim=0, out=da[0])
                parallel.all_reduce_sum(da)
                dX, n_dx, ld_dx, s_dx, col_dx = da, 1, A, B * A, 0
            d_out = ws.get(f"au.dout{i}", (1, B, a_arena.out_dim))
            if kind == "stochastic":
                check(lib.ssac_tanh_normal_bwd(dX.data_ptr(), n_dx, ld_dx, s_dx, col_dx, aout.data_ptr(),
                                               2 * A, eps.data_ptr(), B, A, float(actor.log_std_low),
                                               float(actor.log_std_high), log_alpha.data_ptr(), use_entropy, inv_e,
                                               d_out.data_ptr(), 2 * A, st))
            else:
                check(lib.ssac_det_action_bwd(dX.data_ptr(), n_dx, ld_dx, s_dx, col_dx, aout.data_ptr(), A,
                                              B, A, d_out.data_ptr(), A, st))
        ttot = engine.wgrad_tiles_total(a_arena)
        ss = ws.get(f"au.ss{i}", (ttot,))
        if clip:
            grads = ws.get(f"au.g{i}", (a_arena.params.numel(),), zero=True)
            engine.mlp_backward(a_arena, d_out, s_rep, lds, 0, ah1, ah2, B, ws, f"au.a{i}", grads=grads,
                                sumsq=ss)
            clip_members.append((a_arena, ("actor", i), grads, ss))
        else:
            engine.mlp_backward(a_arena, d_out, s_rep, lds, 0, ah1, ah2, B, ws, f"au.a{i}", adam=adam,
                                adam_key=("actor", i), sumsq=ss)
        member_ss.append(ss)
    ns_upd[2] += 1
    if clip:
        _clip_and_step(adam, clip_members, clip, None, member_shard=ms)
    for actor in agent.actors:  # bf16 mode: the actor step ran on the fp32 masters; refresh the shadows
        for ar in actor.__dict__.get("_ssac_arenas", {}).values():
            ar.sync_shadow()
    if ms is not None:   # the pick is over the GLOBAL ensemble; a rank that does not hold it logs its first member's norm
        k = ms.local(rng.choice(range(E_glob))) or 0
    else:
        pick = rng.choice(agent.actors)  # learning.py:417-419
        k = next(j for j, a_ in enumerate(agent.actors) if a_ is pick)
    if member_ss[k] is not None:  # (None: the fused path's log launch wrote the norm already)
        check(lib.ssac_group_norms(member_ss[k].data_ptr(), 1, member_ss[k].numel(),
                                   adam.ctl.ptr if clip else 0, slot[lu.L_ACTOR_GN:].data_ptr(), st))
    logs["gradients/random_actor_online_grad"] = slot[lu.L_ACTOR_GN]
    logs["losses/actor_pg_loss"] = slot[lu.L_ACTOR_LOSS]
    return logs


def offline_actor_update(buffer, agent, actor_optimizer, encoder_optimizer, batch_size, actor_clip,
                         update_encoder, encoder_clip, augmenter, actor_lambda, aug_mix,
                         premade_replay_dicts=None, per=True, discrete=False, filter_=True):
    """learning.py:144-219: advantage-filtered behavioural cloning (AWAC / AFBC actor update), optionally on a
    prioritised batch whose priorities are refreshed from the advantage afterwards."""
    engine.require_gpu()
    lu.ensure_adopted(agent, buffer)
    E = agent.ensemble_size
    # the BC warm-up (main.py:292-312) trains a pixel encoder THROUGH the BC loss: the actor's input gradient goes
    # back through the conv engine, then clip + encoder_optimizer.step()
    pixel = not lu.is_identity(agent.encoder)
    train_enc = bool(update_encoder) and pixel
    # the action invariance constraint (learning_utils.py:272-285) reaches the encoder through the AUGMENTED
    # observations whether or not update_encoder is set (only encoder_optimizer.step() depends on it)
    enc_grad = train_enc or (bool(actor_lambda) and pixel)
    # (ensemble members share the encoder: member i > 0 adds its gradient, the clip / step follows the last member)
    dev = next(agent.actors[0].parameters()).device
    ws = lu.agent_ws(agent, dev)
    adam = engine.adam_group(actor_optimizer, dev)
    slot = lu.log_block(dev, adam)
    logs = {}
    st = engine.stream()
    inv_e = 1.0 / E
    clip_members, member_ss = [], []
    rd = None
    for i in range(E):
        if premade_replay_dicts is not None:
            rd = premade_replay_dicts[i]
        else:
            rd = lu.sample_move_and_augment(buffer=buffer, batch_size=batch_size, augmenter=augmenter,
                                            aug_mix=aug_mix, per=per, _invariance=bool(actor_lambda))
        o, a = rd["primary_batch"][0], rd["primary_batch"][1]
        actor = agent.actors[i]
        mask_ptr = 0
        if filter_:
            res = agent.adv_estimator.evaluate(o, a, i, want=("mask",), log_ptr=slot[lu.L_ADVW:].data_ptr())
            mask_ptr = res["mask"].data_ptr()
            logs["losses/adv_weights_mean"] = slot[lu.L_ADVW]
        a_arena = engine.bind_arena(actor, "self", [actor], dev)
        A = actor.action_size
        det = not discrete and lu.actor_kind(actor) == "deterministic"   # Normal(tanh(out), 1e-4), distributions.py:107-114
        O = A if (discrete or det) else 2 * A
        rows = batch_size
        if actor_lambda:
            # ---- action invariance (learning_utils.py:272-285).  (1) at the ORIGINAL observations, without gradient:
            #      sample an action from the actor's distribution and keep its log-probability there
            if rd.get("augmented_obs") is None:
                raise NotImplementedError("actor_lambda needs replay dicts made with the invariance observations")
            oo, ao = rd["original_obs"][0], rd["augmented_obs"][0]
            B = batch_size
            os_rep = lu.encode(agent.encoder, oo, dst=ws.get("bc.osrep", (B, agent.encoder.embedding_dim))
                               if pixel else None)
            _, _, out_o = engine.mlp_forward(a_arena, os_rep, lu._row_stride(os_rep), 0, B, ws, f"bc.o{i}", save=False)
            olp = ws.get(f"bc.olp{i}", (B,))
            if discrete:
                a_inv = ws.get(f"bc.ainv{i}", (B,))
                a_inv.copy_(rng.draw_categorical(out_o[0]))   # o_dist.sample() (device generator, as the reference)
            elif det:
                a_inv = None   # o_dist.sample() is tanh(out_o): no draw; the constraint kernel takes out_o itself
            else:
                a_inv = ws.get(f"bc.ainv{i}", (B, A))
                eps = rng.draw_normal((B, A), dev)
                check(lib.ssac_tanh_normal_fwd(out_o.data_ptr(), 2 * A, eps.data_ptr(), B, A,
                                               float(actor.log_std_low), float(actor.log_std_high), a_inv.data_ptr(), A,
                                               0, olp.data_ptr(), st))
            # (2) the BC rows and the AUGMENTED rows go through the encoder / actor as ONE stacked 2B-row pass: their
            #     weight gradients add up inside one backward
            if pixel:
                from . import conv_encoder
                okey = getattr(agent.encoder, "ssac_obs_key", "obs")
                eng = conv_encoder.conv_engine(agent.encoder, dev)
                img2 = ws.get("bc.img2", (2 * B,) + tuple(o[okey].shape[1:]))
                img2[:B].copy_(o[okey])
                img2[B:].copy_(ao[okey])
                X2 = ws.get("bc.sall", (2 * B, eng.emb))
                eng.forward(img2, X2, eng.emb, True)
            else:
                s_rep, as_rep = lu.encode(agent.encoder, o), lu.encode(agent.encoder, ao)
                X2 = ws.get("bc.x2", (2 * B, s_rep.shape[1]))
                X2[:B].copy_(s_rep)
                X2[B:].copy_(as_rep)
            s_rep, rows = X2, 2 * B
        else:
            s_rep = lu.encode(agent.encoder, o, save=train_enc)
        B, S = batch_size, s_rep.shape[1]
        lds = lu._row_stride(s_rep)
        ah1, ah2, aout = engine.mlp_forward(a_arena, s_rep, lds, 0, rows, ws, f"bc.a{i}")
        d_out = ws.get(f"bc.dout{i}", (1, rows, O))
        if discrete:
            check(lib.ssac_bc_discrete_bwd(aout.data_ptr(), a.data_ptr(), a.stride(0), mask_ptr, B, A, inv_e,
                                           d_out.data_ptr(), slot[lu.L_BC0 + i:].data_ptr(),
                                           slot[lu.L_BC_TOTAL:].data_ptr(), st))
            if actor_lambda:
                check(lib.ssac_action_invariance_discrete_bwd(
                    out_o.data_ptr(), aout[0, B:].data_ptr(), a_inv.data_ptr(), B, A, float(actor_lambda) * inv_e,
                    d_out[0, B:].data_ptr(), slot[lu.L_ACT_INV:].data_ptr(), slot[lu.L_BC_TOTAL:].data_ptr(), st))
        elif det:
            check(lib.ssac_bc_det_logprob_bwd(aout.data_ptr(), A, a.data_ptr(), a.stride(0), mask_ptr, B, A, inv_e,
                                              d_out.data_ptr(), A, slot[lu.L_BC0 + i:].data_ptr(),
                                              slot[lu.L_BC_TOTAL:].data_ptr(), st))
            if actor_lambda:
                check(lib.ssac_action_invariance_det_bwd(
                    out_o.data_ptr(), A, aout[0, B:].data_ptr(), A, B, A, float(actor_lambda) * inv_e,
                    d_out[0, B:].data_ptr(), A, slot[lu.L_ACT_INV:].data_ptr(), slot[lu.L_BC_TOTAL:].data_ptr(), st))
        else:
            check(lib.ssac_bc_logprob_bwd(aout.data_ptr(), 2 * A, a.data_ptr(), a.stride(0), mask_ptr, B, A,
                                          float(actor.log_std_low), float(actor.log_std_high), inv_e,
                                          d_out.data_ptr(), 2 * A, slot[lu.L_BC0 + i:].data_ptr(),
                                          slot[lu.L_BC_TOTAL:].data_ptr(), st))
            if actor_lambda:
                check(lib.ssac_action_invariance_bwd(
                    aout[0, B:].data_ptr(), 2 * A, a_inv.data_ptr(), A, olp.data_ptr(), B, A,
                    float(actor.log_std_low), float(actor.log_std_high), float(actor_lambda) * inv_e,
                    d_out[0, B:].data_ptr(), 2 * A, slot[lu.L_ACT_INV:].data_ptr(), slot[lu.L_BC_TOTAL:].data_ptr(), st))
        logs[f"losses/filterd_bc_loss_{i}"] = slot[lu.L_BC0 + i]
        ttot = engine.wgrad_tiles_total(a_arena)
        ss = ws.get(f"bc.ss{i}", (ttot,))
        if actor_clip:
            grads = ws.get(f"bc.g{i}", (a_arena.params.numel(),), zero=True)
            dX = engine.mlp_backward(a_arena, d_out, s_rep, lds, 0, ah1, ah2, rows, ws, f"bc.a{i}", grads=grads,
                                     sumsq=ss, need_dx=enc_grad)
            clip_members.append((a_arena, ("actor", i), grads, ss))
        else:
            dX = engine.mlp_backward(a_arena, d_out, s_rep, lds, 0, ah1, ah2, rows, ws, f"bc.a{i}", adam=adam,
                                     adam_key=("actor", i), sumsq=ss, need_dx=enc_grad)
        if enc_grad and actor_lambda:
            # stacked encoder pass: the BC rows carry a gradient only when the encoder is trained through the BC loss
            # (filtered_bc_loss computes s_rep without gradient otherwise), the augmented rows always do; the encoder
            # is clipped and logged either way and stepped only with update_encoder (learning.py:203-208)
            d_rep = ws.get("bc.drep2", (2 * B, S))
            d_rep.copy_(dX[0])
            if not train_enc:
                d_rep[:B].zero_()
            eng.backward(d_rep, accumulate=i > 0)
            if i == E - 1:
                eng.optimizer_step(encoder_optimizer, encoder_clip, norm_out=slot[lu.L_ENC_GN:], step=train_enc)
        elif train_enc:  # (dX was taken before the epilogue of the weight-gradient launch touched W1)
            _encoder_step(agent.encoder, encoder_optimizer, encoder_clip, dX, S, ws, slot, dev, accumulate=i > 0,
                          step=i == E - 1)
        member_ss.append(ss)
    if actor_clip:
        _clip_and_step(adam, clip_members, actor_clip, None)
    for actor in agent.actors:
        for ar in actor.__dict__.get("_ssac_arenas", {}).values():
            ar.sync_shadow()
    logs["losses/filtered_bc_overall_loss"] = slot[lu.L_BC_TOTAL]
    pick = rng.choice(agent.actors)  # learning.py:210-212
    k = next(j for j, a_ in enumerate(agent.actors) if a_ is pick)
    check(lib.ssac_group_norms(member_ss[k].data_ptr(), 1, member_ss[k].numel(),
                               adam.ctl.ptr if actor_clip else 0, slot[lu.L_BC_GN:].data_ptr(), st))
    logs["gradients/actor_offline_grad_norm"] = slot[lu.L_BC_GN]
    logs["gradients/encoder_offline_actorloss_grad_norm"] = slot[lu.L_ENC_GN]  # identity encoders: no gradient
    if per:
        lu.adjust_priorities(logs, rd, agent, buffer)
    return logs


def markov_state_abstraction_update(buffer, agent, optimizer, batch_size, augmenter, aug_mix, discrete,
                                    inverse_coeff, contrastive_coeff, smoothness_coeff, smoothness_max_dist,
                                    grad_clip):
    """learning.py:266-341: the self-supervised abstraction loss of "Learning Markov State Abstractions for Deep RL" --
    inverse model (which action led from s to s'), contrastive model (is (s, s') a real transition; negatives by
    shuffling s' over the batch) and a smoothness hinge on ||s' - s|| -- trained through ONE optimizer over
    chain(encoder, inverse_model, contrastive_model) (main.py:218-224).

    The two MLPs run on the engine's forward / backward kernels; a pixel encoder takes s and s' as ONE stacked 2B-row
    pass (the weight gradients of the two uses add up inside its backward).  Loss heads: ssac_bc_logprob_bwd /
    ssac_bc_discrete_bwd (log-probability of the data action), ssac_bce_sigmoid_bwd, ssac_markov_smoothness_bwd."""
    engine.require_gpu()
    lu.ensure_adopted(agent, buffer)
    inv, con = agent.inverse_model, agent.contrastive_model
    dev = next(inv.parameters()).device
    ws = lu.agent_ws(agent, dev)
    adam = engine.adam_group(optimizer, dev)
    slot = lu.log_block(dev, adam)  # a log block of its own; advances the optimizer step
    st = engine.stream()
    rd = lu.sample_move_and_augment(buffer=buffer, batch_size=batch_size, augmenter=augmenter, aug_mix=aug_mix,
                                    per=False)
    o, a, _, o1, _ = rd["primary_batch"]
    B = batch_size
    ident = lu.is_identity(agent.encoder)
    eng = None
    if ident:
        s_rep, s1_rep = lu.encode(agent.encoder, o), lu.encode(agent.encoder, o1)
        D = s_rep.shape[1]
    else:
        from . import conv_encoder
        key = getattr(agent.encoder, "ssac_obs_key", "obs")
        eng = conv_encoder.conv_engine(agent.encoder, dev)
        if eng is None:
            raise NotImplementedError(f"{type(agent.encoder).__name__}: this encoder has no HIP path")
        D = eng.emb
        img = ws.get("mk.img", (2 * B,) + tuple(o[key].shape[1:]))
        img[:B].copy_(o[key])
        img[B:].copy_(o1[key])   # (device plumbing: the two observation batches behind one another)
        s_all = ws.get("mk.sall", (2 * B, D))
        eng.forward(img, s_all, D, True)
        s_rep, s1_rep = s_all[:B], s_all[B:]
    # ---- inputs: [s | s'] for the inverse model; [s | s'] over [s | s'[perm]] for the contrastive model
    perm = rng.draw_permutation(B)
    perm_dev = perm.to(dev)
    x_inv = ws.get("mk.xinv", (B, 2 * D))
    x_inv[:, :D].copy_(s_rep)
    x_inv[:, D:].copy_(s1_rep)
    x_con = ws.get("mk.xcon", (2 * B, 2 * D))
    x_con[:B].copy_(x_inv)
    x_con[B:, :D].copy_(s_rep)
    torch.index_select(s1_rep, 0, perm_dev, out=x_con[B:, D:])
    i_arena = engine.bind_arena(inv, "self", [inv], dev)
    c_arena = engine.bind_arena(con, "self", [con], dev)
    ih1, ih2, iout = engine.mlp_forward(i_arena, x_inv, 2 * D, 0, B, ws, "mk.inv")
    ch1, ch2, cout = engine.mlp_forward(c_arena, x_con, 2 * D, 0, 2 * B, ws, "mk.con")
    # ---- loss heads (each also leaves its share of d markov_loss / d output)
    A = inv.action_size
    if discrete:
        d_iout = ws.get("mk.dinv", (1, B, A))
        check(lib.ssac_bc_discrete_bwd(iout.data_ptr(), a.data_ptr(), a.stride(0), 0, B, A, float(inverse_coeff),
                                       d_iout.data_ptr(), slot[lu.L_MK_RAW:].data_ptr(),
                                       slot[lu.L_MK_TMP:].data_ptr(), st))
        inv_scale = 1.0
    else:
        # -a_dist.log_prob(a).mean(): the mean runs over the B x A per-dimension log-probabilities (learning.py:298)
        d_iout = ws.get("mk.dinv", (1, B, 2 * A))
        check(lib.ssac_bc_logprob_bwd(iout.data_ptr(), 2 * A, a.data_ptr(), a.stride(0), 0, B, A,
                                      float(inv.log_std_low), float(inv.log_std_high), float(inverse_coeff) / A,
                                      d_iout.data_ptr(), 2 * A, slot[lu.L_MK_RAW:].data_ptr(),
                                      slot[lu.L_MK_TMP:].data_ptr(), st))
        inv_scale = 1.0 / A
    d_cout = ws.get("mk.dcon", (1, 2 * B, 1))
    check(lib.ssac_bce_sigmoid_bwd(cout.data_ptr(), B, 2 * B, float(contrastive_coeff), d_cout.data_ptr(),
                                   slot[lu.L_MK_CON:].data_ptr(), st))
    d_all = None if ident else ws.get("mk.dall", (2 * B, D))
    check(lib.ssac_markov_smoothness_bwd(s_rep.data_ptr(), lu._row_stride(s_rep), s1_rep.data_ptr(),
                                         lu._row_stride(s1_rep), B, D, float(smoothness_max_dist),
                                         float(smoothness_coeff), engine._ptr(None if ident else d_all[:B]), D,
                                         engine._ptr(None if ident else d_all[B:]), D, 0,
                                         slot[lu.L_MK_SMOOTH:].data_ptr(), st))
    check(lib.ssac_markov_logs(slot[lu.L_MK_RAW:].data_ptr(), inv_scale, slot[lu.L_MK_CON:].data_ptr(),
                               slot[lu.L_MK_SMOOTH:].data_ptr(), float(inverse_coeff), float(contrastive_coeff),
                               float(smoothness_coeff), slot[lu.L_MK_LOSS:].data_ptr(), st))
    # ---- backward: gradients stored (the clip needs the joint norm first), input gradients only for a real encoder
    members = []
    for tag, arena, d_out, x, h1, h2, rows in (("mk.inv", i_arena, d_iout, x_inv, ih1, ih2, B),
                                               ("mk.con", c_arena, d_cout, x_con, ch1, ch2, 2 * B)):
        grads = ws.get(tag + ".g", (arena.params.numel(),), zero=True)
        ss = ws.get(tag + ".ss", (engine.wgrad_tiles_total(arena),))
        dx = engine.mlp_backward(arena, d_out, x, 2 * D, 0, h1, h2, rows, ws, tag, grads=grads, sumsq=ss,
                                 need_dx=not ident)
        members.append((arena, ("markov", tag), grads, ss, dx))
    allss = [members[0][3], members[1][3]]
    if not ident:
        dxi, dxc = members[0][4][0], members[1][4][0]
        d_all[:B] += dxi[:, :D]
        d_all[:B] += dxc[:B, :D]
        d_all[:B] += dxc[B:, :D]
        d_all[B:] += dxi[:, D:]
        d_all[B:] += dxc[:B, D:]
        d_all[B:].index_add_(0, perm_dev, dxc[B:, D:])  # (a permutation: every row receives exactly one addend)
        eng.backward(d_all)
        ss_enc = ws.get("mk.enc.ss", (int(lib.ssac_sumsq_blocks()),))
        check(lib.ssac_sumsq(eng.grads.data_ptr(), eng.numel, ss_enc.data_ptr(), st))
        allss.append(ss_enc)
    # ---- clip_grad_norm_ over chain(encoder, inverse, contrastive) jointly, then optimizer.step() (learning.py:321-330)
    cat = torch.cat(allss)
    check(lib.ssac_clip_coef(adam.ctl.ptr, cat.data_ptr(), cat.numel(), float(grad_clip) if grad_clip else 0.0, 0, st))
    for arena, key_, grads, _, _ in members:
        m, v = adam.moments_for(key_, arena.params)
        check(lib.ssac_adam_step(arena.params.data_ptr(), m.data_ptr(), v.data_ptr(), grads.data_ptr(),
                                 arena.params.numel(), adam.ctl.ptr, st))
        arena.sync_shadow()
    if not ident:
        m, v = adam.moments_for("conv_encoder", eng.flat)
        check(lib.ssac_adam_step(eng.flat.data_ptr(), m.data_ptr(), v.data_ptr(), eng.grads.data_ptr(), eng.numel,
                                 adam.ctl.ptr, st))
    # ---- logs (learning.py:332-340): the norms are read after the clip rescaled the gradients
    scale = adam.ctl.ptr if grad_clip else 0
    for (arena, _, _, ss, _), off in zip(members, (lu.L_MK_GN_INV, lu.L_MK_GN_CON)):
        check(lib.ssac_group_norms(ss.data_ptr(), 1, ss.numel(), scale, slot[off:].data_ptr(), st))
    if not ident:
        check(lib.ssac_group_norms(allss[2].data_ptr(), 1, allss[2].numel(), scale, slot[lu.L_MK_GN_ENC:].data_ptr(), st))
    return {"gradients/contrastive_model_grad_norm": slot[lu.L_MK_GN_CON],
            "gradients/inverse_model_grad_norm": slot[lu.L_MK_GN_INV],
            "gradients/encoder_markovloss_grad_norm": slot[lu.L_MK_GN_ENC],
            "losses/markov_loss": slot[lu.L_MK_LOSS + 3],
            "losses/inverse_model_loss": slot[lu.L_MK_LOSS],
            "losses/contrastive_model_loss": slot[lu.L_MK_LOSS + 1],
            "losses/smoothness_loss": slot[lu.L_MK_LOSS + 2]}


def alpha_update(buffer, agent, optimizers, batch_size, log_alphas, augmenter, aug_mix, target_entropy,
                 premade_replay_dicts, discrete):
    engine.require_gpu()
    lu.ensure_adopted(agent, buffer)
    dev = log_alphas[0].device
    ws = lu.agent_ws(agent, dev)
    if engine.CAPTURE is None:
        # this update's block of the log ring as it stands: ssac_alpha_update ASSIGNS its two entries per member and nothing
        # else of the block is handed out, so no launch is spent on clearing it (one of the update's three launches, round 6)
        ring = lu.ring_for(dev)
        slot = ring.buf[ring.advance()]
    else:
        slot = lu.log_block(dev)
    logs = {}
    st = engine.stream()
    ms = parallel.member_shard_of(agent)   # member-sharded rank: log_alphas / optimizers are the LOCAL members'
    for ig in range(agent.ensemble_size if ms is None else ms.ensemble_size):
        i = ig if ms is None else ms.local(ig)
        if i is None:   # a member another rank owns: its host draws, in member order
            if premade_replay_dicts is None:
                lu.sample_move_and_augment(buffer=buffer, batch_size=batch_size, augmenter=augmenter, per=False,
                                           aug_mix=aug_mix)
            lu.skip_alpha_draws(agent, agent.actors[0], batch_size, dev)
            continue
        if premade_replay_dicts is not None:
            rd = premade_replay_dicts[i]
        else:
            rd = lu.sample_move_and_augment(buffer=buffer, batch_size=batch_size, augmenter=augmenter,
                                            per=False, aug_mix=aug_mix)
        o = rd["primary_batch"][0]
        s_rep = lu.encode(agent.encoder, o)
        B, S = s_rep.shape
        actor = agent.actors[i]
        a_arena = engine.bind_arena(actor, "self", [actor], dev)
        kind = lu.actor_kind(actor)
        fused_sample = kind == "stochastic" and a_arena.fused
        if not fused_sample:
            _, _, aout = engine.mlp_forward(a_arena, s_rep, lu._row_stride(s_rep), 0, B, ws, f"al.a{i}")
        if kind == "discrete":
            lp_ptr, n_act = aout.data_ptr(), a_arena.out_dim
        else:
            A = actor.action_size
            logp = ws.get(f"al.logp{i}", (B,))
            if fused_sample:
                # actor forward + a_dist.sample() + log pi (learning.py:255) in ONE launch; the noise comes from the
                # agent's Philox stream unless a noise hook is installed (then it is drawn like everywhere else)
                scratch = ws.get(f"al.act{i}", (B, A))
                if lu.IN_KERNEL_NOISE and rng.normal_is_stock():
                    ns = lu.noise_stream(agent, dev)
                    rs = _lib.Rng(ns[0], 0, ns[1])
                    ns[1] += 1
                    eps_ptr, rng_ptr = 0, C.addressof(rs)
                else:
                    eps = rng.draw_normal((B, A), dev)
                    eps_ptr, rng_ptr = eps.data_ptr(), 0
                check(lib.ssac_actor_sample_fused(C.byref(a_arena.desc()), s_rep.data_ptr(), lu._row_stride(s_rep), B,
                                                  eps_ptr, float(actor.log_std_low), float(actor.log_std_high),
                                                  scratch.data_ptr(), A, 0, logp.data_ptr(), 0, 0, 0, rng_ptr, st))
            elif kind == "stochastic":
                eps = rng.draw_normal((B, A), dev)  # a_dist.sample() (learning.py:255)
                scratch = ws.get(f"al.act{i}", (B, A))
                check(lib.ssac_tanh_normal_fwd(aout.data_ptr(), 2 * A, eps.data_ptr(), B, A,
                                               float(actor.log_std_low), float(actor.log_std_high),
                                               scratch.data_ptr(), A, 0, logp.data_ptr(), st))
            else:
                import math
                logp.fill_(A * (-math.log(1e-4) - 0.5 * math.log(2 * math.pi)))
            lp_ptr, n_act = logp.data_ptr(), 1
        adam = engine.adam_group(optimizers[i], dev)
        la = log_alphas[i]
        m, v = adam.moments_for("log_alpha", la.data)
        check(lib.ssac_alpha_update(la.data_ptr(), m.data_ptr(), v.data_ptr(), adam.ctl.ptr, lp_ptr, B,
                                    n_act, float(target_entropy), slot[lu.L_ALPHA0 + 2 * i:].data_ptr(), st))
        logs[f"losses/alpha_loss_{ig}"] = slot[lu.L_ALPHA0 + 2 * i]
        logs[f"alphas/alpha_{ig}"] = slot[lu.L_ALPHA0 + 2 * i + 1]
    return logs
