"""Replays the multi-step parity cases of tests/golden with either backend:

  * ``run_oracle(name)``  -- the CPU oracle (oracle/ssac_oracle.py); CPU test, no GPU needed
  * ``run_engine(name)``  -- the HIP engine through super_sac_amd's reference-shaped API

Both consume the host-RNG draws recorded in the fixture (replay indices, REDQ subsets,
normal draws) and return the same record layout as the fixture, so one comparison routine
serves oracle-vs-reference, engine-vs-reference and engine-vs-oracle.
"""
import copy
import math
import os
import random
import re
import sys
from itertools import chain

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import ssac_oracle as orc  # noqa: E402
import synth  # noqa: E402

GOLDEN = os.path.join(HERE, "golden")


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN, f"{name}.npz")))


def _buffers(cfg):
    px = cfg.get("pixels")
    if px:
        return synth.synth_pixel_transitions(cfg["rows"], px["channels"], px["hw"],
                                             n_actions=cfg["act"] if cfg["discrete"] else None,
                                             act_dim=cfg["act"], seed=cfg["seed"] + 100)
    return synth.synth_transitions(cfg["rows"], cfg["obs"], cfg["act"], cfg["discrete"],
                                   seed=cfg["seed"] + 100, n_actions=cfg["act"])


ENC_KEYS = {"big": ["c1w", "c1b", "c2w", "c2b", "c3w", "c3b", "c4w", "c4b", "fcw", "fcb", "lnw", "lnb"],
            "small": ["c1w", "c1b", "c2w", "c2b", "c3w", "c3b", "fcw", "fcb"]}


def _oracle_encoder(cfg):
    px = cfg.get("pixels")
    if not px:
        return None
    p = orc.make_conv_encoder(np.random.RandomState(cfg["seed"] + 7), px["kind"], px["channels"], px["emb"])
    return {"kind": px["kind"], "key": "obs", "p": p}


def _oracle_agent(cfg):
    oa = orc.AgentOracle(state_dim=cfg["obs"], act_dim=cfg["act"], hidden=cfg["hidden"],
                         num_critics=cfg["N"], ensemble_size=cfg["E"], discrete=cfg["discrete"],
                         actor_kind=cfg["actor"], log_std_low=cfg["lo"], log_std_high=cfg["hi"],
                         popart=cfg["popart"], encoder=_oracle_encoder(cfg), seed=cfg["seed"])
    if cfg["popart"]:
        for p in oa.popart:
            p.min_steps = cfg.get("popart_min_steps", 1000)
    return oa


def _target_entropy(cfg):
    return (-math.log(1.0 / cfg["act"]) * 0.98) if cfg["discrete"] else -float(cfg["act"])


def _flat(params):
    return np.concatenate([p.detach().cpu().numpy().ravel() for p in params])


def _fingerprint(params):
    vals = []
    for p in params:
        flat = p.detach().cpu().numpy().ravel()
        vals.append(flat[synth.fingerprint_indices(flat.size)])
    return np.concatenate(vals)


# ------------------------------------------------------------------------------------------
# resume cases (cfg["resume"]): the reference's checkpoint travels in the fixture as arrays "ckpt|<file>|<key>"
CKPT_PREFIX = "ckpt|"


def checkpoint_arrays(fx):
    """{file name: {state-dict key: array}} of the reference checkpoint held by a resume fixture"""
    out = {}
    for k, v in fx.items():
        if k.startswith(CKPT_PREFIX):
            _, fname, key = k.split("|", 2)
            out.setdefault(fname, {})
            if key:   # (an empty state dict -- the reference's popart{i}.pt, popart.py:11-16 -- is the bare file entry)
                out[fname][key] = v
    return out


def write_checkpoint_dir(fx, path):
    """the fixture's arrays back into the directory layout Agent.save writes (agent.py:172-195): one torch-saved
    state dict per file"""
    os.makedirs(path, exist_ok=True)
    for fname, sd in checkpoint_arrays(fx).items():
        torch.save({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, os.path.join(path, fname))
    return path


def oracle_load_checkpoint(oa, cfg, files):
    """Agent.load (agent.py:196-202) for the oracle's parameter dicts; `files` = checkpoint_arrays(...).  PopArt: the
    reference's layer keeps its statistics in plain attributes, its state dict is empty -- a loaded agent starts from
    fresh statistics, and so does the oracle's."""
    head = {"stochastic": "fc3", "deterministic": "out", "discrete": "act_p"}[cfg["actor"]]

    def put(dst, sd, prefix, names):
        with torch.no_grad():
            for (wk, bk), nm in zip((("w1", "b1"), ("w2", "b2"), ("w3", "b3")), names):
                dst[wk].copy_(torch.from_numpy(np.array(sd[f"{prefix}{nm}.weight"])))
                dst[bk].copy_(torch.from_numpy(np.array(sd[f"{prefix}{nm}.bias"])))
    for i in range(cfg["E"]):
        put(oa.actors[i], files[f"actor{i}.pt"], "", ("fc1", "fc2", head))
        for j in range(cfg["N"]):
            put(oa.critics[i][j], files[f"critic{i}.pt"], f"nets.{j}.", ("fc1", "fc2", "out"))
    px = cfg.get("pixels")
    if px:
        sd, p = files["encoder.pt"], oa.encoder["p"]
        names = ["conv1", "conv2", "conv3", "conv4"] if px["kind"] == "big" else ["conv1", "conv2", "conv3"]
        pairs = [(f"c{i_}", nm) for i_, nm in enumerate(names, 1)] + [("fc", "fc")] + ([("ln", "ln")] if px["kind"] == "big" else [])
        with torch.no_grad():
            for short, nm in pairs:
                p[f"{short}w"].copy_(torch.from_numpy(np.array(sd[f"conv_block.{nm}.weight"])))
                p[f"{short}b"].copy_(torch.from_numpy(np.array(sd[f"conv_block.{nm}.bias"])))


# ------------------------------------------------------------------------------------------
def run_oracle(name):
    cfg = synth.CASES[name]
    fx = load_fixture(name)
    B, E = cfg["B"], cfg["E"]
    obuf = orc.ReplayOracle(cfg["cap"])
    obuf.load_experience(*_buffers(cfg))
    oa = _oracle_agent(cfg)
    if cfg.get("resume"):
        oracle_load_checkpoint(oa, cfg, checkpoint_arrays(fx))
    oa.requires_grad_(True)
    ot = oa.clone()
    copt = orc.AdamOracle(oa.critic_params(), lr=cfg["lr"])
    aopt = orc.AdamOracle(oa.actor_params(), lr=cfg["lr"])
    px = cfg.get("pixels")
    eopt = orc.AdamOracle(oa.encoder_params(), lr=px["enc_lr"] if px else 1e-4)
    init_alpha = max(cfg["init_alpha"], 1e-15)
    las = [torch.tensor([math.log(init_alpha)], requires_grad=True) for _ in range(E)]
    lopts = [orc.AdamOracle([la], lr=cfg["alpha_lr"], betas=(0.5, 0.999)) for la in las]
    aug = orc.AugOracle(px["aug"] if px else "identity", B)
    aug_mix = px["aug_mix"] if px else 0.0
    nscale = cfg["noise"]["scale"] if cfg["noise"] else None
    nclip = cfg["noise"]["clip"] if cfg["noise"] else None
    stochastic = cfg["actor"] == "stochastic"
    T = lambda key: torch.from_numpy(fx[key])
    rec, upd = {}, 0
    dicts = None
    for cyc in range(cfg["cycles"]):
        for k in range(cfg["utd"]):
            if px:
                aug.forced = [T(f"u{upd}_shift{i}") for i in range(E)]
            logs, dicts = orc.critic_update(
                obuf, oa, ot, copt, eopt, las, B, cfg["gamma"], cfg["clip"], cfg["clip"], cfg["n"],
                cfg["temp"], cfg["weight_type"], cfg["pop"], aug, aug_mix=aug_mix, noise_scale=nscale,
                noise_clip=nclip, idx_list=[fx[f"u{upd}_idx{i}"] for i in range(E)],
                eps_list=[T(f"u{upd}_eps{i}") for i in range(E)] if stochastic else None,
                noise_list=[T(f"u{upd}_noise{i}") for i in range(E)] if cfg["noise"] else None,
                subset_list=[list(fx[f"u{upd}_subset{i}"]) for i in range(E)],
                bw_eps_list=([[T(f"u{upd}_bweps{i}_{k_}") for k_ in range(E)] for i in range(E)]
                             if f"u{upd}_bweps0_0" in fx else None),
                bw_cat_list=([[T(f"u{upd}_cat{i}_{k_}") for k_ in range(E)] for i in range(E)]
                             if f"u{upd}_cat0_0" in fx else None),
                grad_pick=int(fx[f"u{upd}_gpick"]), encoder_lambda=cfg.get("encoder_lambda", 0))
            for i in range(E):
                rec[f"u{upd}_td{i}"] = dicts[i]["td_target"].numpy()
                if cfg["popart"]:
                    s = oa.popart[i].state()
                    rec[f"u{upd}_popart{i}"] = np.array([s["mu"], s["nu"], s["w"], s["b"], s["sigma"], s["t"]])
            for key, val in logs.items():
                rec[f"u{upd}_log:{key}"] = np.float64(val)
            if int(fx[f"u{upd}_polyak"]):
                orc.soft_update(ot.critic_params(), oa.critic_params(), cfg["tau"])
                if px:
                    orc.soft_update(ot.encoder_params(), oa.encoder_params(), px["enc_tau"])
            upd += 1
        have_eps = not cfg["discrete"]
        alog = orc.online_actor_update(
            oa, aopt, las, dicts, cfg["pop"], cfg["clip"],
            eps_list=[T(f"a{cyc}_eps{i}") for i in range(E)] if have_eps else None,
            noise_scale=nscale, noise_clip=nclip,
            noise_list=[T(f"a{cyc}_noise{i}") for i in range(E)] if cfg["noise"] else None,
            use_baseline=bool(cfg.get("use_baseline", False)),
            base_eps_lists=([[torch.from_numpy(e) for e in fx[f"a{cyc}_base{i}"]] for i in range(E)]
                            if cfg.get("use_baseline") else None),
            grad_pick=int(fx[f"a{cyc}_gpick"]))
        for key, val in alog.items():
            rec[f"a{cyc}_log:{key}"] = np.float64(val)
        if cfg["init_alpha"] > 0 and cfg["alpha_lr"] > 0:
            llog = orc.alpha_update(oa, lopts, las, dicts, _target_entropy(cfg),
                                    eps_list=[T(f"l{cyc}_eps{i}") for i in range(E)] if stochastic else None)
            for key, val in llog.items():
                rec[f"l{cyc}_log:{key}"] = np.float64(val)
    _finalise(rec, fx, oa.critic_params(), oa.actor_params(), ot.critic_params(),
              copt.m, copt.v, las)
    if px:
        rec["finalfp_encoder"] = _fingerprint(oa.encoder_params())
        rec["finalfp_target_encoder"] = _fingerprint(ot.encoder_params())
    return rec


def _finalise(rec, fx, crit, act, tcrit, m_list, v_list, las):
    for tag, plist in (("critic", crit), ("actor", act), ("target_critic", tcrit)):
        if f"final_{tag}" in fx:
            rec[f"final_{tag}"] = _flat(plist)
        else:
            rec[f"finalfp_{tag}"] = _fingerprint(plist)
    rec["finalfp_critic_m"] = _fingerprint(m_list)
    rec["finalfp_critic_v"] = _fingerprint(v_list)
    rec["final_log_alpha"] = np.array([float(x) for x in las], np.float64)


# ------------------------------------------------------------------------------------------
class DrawPlayer:
    """Feeds recorded host-RNG draws to super_sac_amd.rng in the order the engine asks."""

    def __init__(self, device):
        self.device = device
        self.idx, self.sub, self.normal, self.shift = [], [], [], []
        self.picks, self.cats = [], []

    def install(self, rng_mod):
        self._saved = (rng_mod.draw_indices, rng_mod.draw_subset, rng_mod.draw_normal)
        self._saved_choice, self._saved_cat = rng_mod.choice, rng_mod.draw_categorical
        # random.choice(agent.critics / agent.actors) of the gradient-norm logs: the recorded pick when one is queued
        rng_mod.choice = lambda seq: seq[self.picks.pop(0)] if self.picks else random.choice(seq)
        rng_mod.draw_categorical = lambda logits: torch.from_numpy(self.cats.pop(0)).to(self.device)
        self._saved_shift = rng_mod.draw_drqv2_shift
        rng_mod.draw_drqv2_shift = lambda b, pad: torch.from_numpy(self.shift.pop(0))
        self._saved_into = rng_mod.draw_normal_into
        rng_mod.draw_normal_into = lambda dst: dst.copy_(torch.from_numpy(self.normal.pop(0)))
        rng_mod.draw_indices = lambda n, b: torch.from_numpy(self.idx.pop(0).astype(np.int64))
        rng_mod.draw_subset = lambda n, k: [int(v) for v in self.sub.pop(0)]
        rng_mod.draw_normal = lambda shape, device: torch.from_numpy(self.normal.pop(0)).to(self.device)
        self._mod = rng_mod

    def restore(self):
        m = self._mod
        m.draw_indices, m.draw_subset, m.draw_normal = self._saved
        m.draw_drqv2_shift = self._saved_shift
        m.draw_normal_into = self._saved_into
        m.choice, m.draw_categorical = self._saved_choice, self._saved_cat


def build_engine_agent(cfg, device, shard=None, foreign=False, members=None):
    """super_sac_amd.Agent holding the same seeded weights as the oracle agent (with `shard`: only
    the critics [shard.lo, shard.hi) of the global ensemble; with `members` (parallel.MemberShard): only the ensemble
    members [members.lo, members.hi)).  foreign: an agent made of stand-ins that carry only
    the REFERENCE classes' attributes (tests/foreign_agent.py), to be adopted by the update functions."""
    import super_sac_amd as ssa
    oa = _oracle_agent(cfg)
    lo = 0 if shard is None else shard.lo
    n_loc = cfg["N"] if shard is None else shard.n_local
    m_lo, m_n = (0, cfg["E"]) if members is None else (members.lo, members.n_local)
    if foreign:
        import foreign_agent
        assert not cfg.get("pixels")
        ag = foreign_agent.ForeignAgent(cfg, n_loc)
        head = {"stochastic": "fc3", "deterministic": "out", "discrete": "act_p"}[cfg["actor"]]
        with torch.no_grad():
            for i in range(cfg["E"]):
                for mod, p, names in [(ag.actors[i], oa.actors[i], ("fc1", "fc2", head))] + \
                        [(ag.critics[i].nets[j], oa.critics[i][lo + j], ("fc1", "fc2", "out")) for j in range(n_loc)]:
                    for (wk, bk), nm in zip((("w1", "b1"), ("w2", "b2"), ("w3", "b3")), names):
                        getattr(mod, nm).weight.copy_(p[wk])
                        getattr(mod, nm).bias.copy_(p[bk])
        ag.to(device)
        if cfg["popart"]:
            for p in ag.popart:
                p.min_steps = cfg.get("popart_min_steps", 1000)
        return ag
    actor_cls = {"stochastic": ssa.nets.ContinuousStochasticActor,
                 "deterministic": ssa.nets.ContinuousDeterministicActor,
                 "discrete": ssa.nets.DiscreteActor}[cfg["actor"]]
    critic_cls = ssa.nets.DiscreteCritic if cfg["discrete"] else ssa.nets.ContinuousCritic
    px = cfg.get("pixels")
    if px:
        cls = ssa.nets.BigPixelEncoder if px["kind"] == "big" else ssa.nets.SmallPixelEncoder
        conv = cls((px["channels"], px["hw"], px["hw"]), px["emb"])
        ep = _oracle_encoder(cfg)["p"]
        names = ["conv1", "conv2", "conv3", "conv4"] if px["kind"] == "big" else ["conv1", "conv2", "conv3"]
        with torch.no_grad():
            for i_, nm in enumerate(names, 1):
                getattr(conv, nm).weight.copy_(ep[f"c{i_}w"]); getattr(conv, nm).bias.copy_(ep[f"c{i_}b"])
            conv.fc.weight.copy_(ep["fcw"]); conv.fc.bias.copy_(ep["fcb"])
            if px["kind"] == "big":
                conv.ln.weight.copy_(ep["lnw"]); conv.ln.bias.copy_(ep["lnb"])
        enc = ssa.nets.PixelEncoder(conv)
    else:
        enc = ssa.nets.IdentityEncoder(cfg["obs"])
    ag = ssa.Agent(act_space_size=cfg["act"], encoder=enc,
                   actor_network_cls=actor_cls, critic_network_cls=critic_cls, discrete=cfg["discrete"],
                   ensemble_size=m_n, num_critics=n_loc, ucb_bonus=0.0,
                   hidden_size=cfg["hidden"], auto_rescale_targets=cfg["popart"],
                   log_std_low=cfg["lo"], log_std_high=cfg["hi"])
    head = {"stochastic": "fc3", "deterministic": "out", "discrete": "act_p"}[cfg["actor"]]

    def load(mod, p, names):
        with torch.no_grad():
            for (wk, bk), nm in zip((("w1", "b1"), ("w2", "b2"), ("w3", "b3")), names):
                getattr(mod, nm).weight.copy_(p[wk])
                getattr(mod, nm).bias.copy_(p[bk])
    for i in range(m_n):
        load(ag.actors[i], oa.actors[m_lo + i], ("fc1", "fc2", head))
        for j in range(n_loc):
            load(ag.critics[i].nets[j], oa.critics[m_lo + i][lo + j], ("fc1", "fc2", "out"))
    ag.to(device)
    if cfg["popart"]:
        for p in ag.popart:
            p.min_steps = cfg.get("popart_min_steps", 1000)
    ag.train()
    return ag


def run_engine(name, device="cuda", shard=None, foreign=False, precision="fp32", foreign_io=False, members=None):
    """`shard` (super_sac_amd.parallel.Shard): run as one rank of a critic-sharded job; the record
    then holds this rank's critics only (see slice_fixture).  foreign_io: the replay buffer and the augmenter are
    stand-ins carrying only the REFERENCE classes' attributes (tests/foreign_agent.py), adopted by the update functions."""
    import super_sac_amd as ssa
    cfg = synth.CASES[name]
    fx = load_fixture(name)
    B, E = cfg["B"], cfg["E"]
    device = torch.device(device)
    if foreign_io:
        import foreign_agent
        buf = foreign_agent.ForeignReplayBuffer(cfg["cap"], *_buffers(cfg))
    else:
        buf = ssa.replay.ReplayBuffer(cfg["cap"], device=device)
        buf.load_experience(*_buffers(cfg))
    agent = build_engine_agent(cfg, device, shard, foreign=foreign, members=members)
    if cfg.get("resume"):
        # the checkpoint the REFERENCE wrote (fixture arrays -> the files Agent.save writes) through the agent's own load();
        # a reference-class stand-in (foreign) has no such method to exercise, a shard holds a slice of the files' nets
        assert shard is None and members is None and not foreign, "resume cases run unsharded on this package's Agent"
        import tempfile
        with tempfile.TemporaryDirectory() as ckpt_dir:
            agent.load(write_checkpoint_dir(fx, ckpt_dir))
    if precision != "fp32":
        ssa.set_precision(agent, precision)  # (the deepcopy below inherits it)
    target = copy.deepcopy(agent)
    if shard is not None:
        ssa.parallel.install(agent, target, shard)
        fx = slice_fixture(fx, cfg, shard)
    if members is not None:
        # member-sharded rank (parallel.MemberShard): the agent holds EL of the E members; every member's recorded
        # draws are still fed (the rank consumes the draws of the members it does not hold), records carry GLOBAL indices
        ssa.parallel.install_members(agent, target, members)
    EL = agent.ensemble_size
    glob = (lambda i: i) if members is None else (lambda i: members.lo + i)
    NL = agent.num_critics
    copt = torch.optim.Adam(chain(*(c.parameters() for c in agent.critics)), lr=cfg["lr"],
                            weight_decay=0, betas=(0.9, 0.999))
    aopt = torch.optim.Adam(chain(*(a.parameters() for a in agent.actors)), lr=cfg["lr"],
                            weight_decay=0, betas=(0.9, 0.999))
    px = cfg.get("pixels")
    eopt = torch.optim.Adam(agent.encoder.parameters(), lr=px["enc_lr"] if px else 1e-4, betas=(0.9, 0.999))
    init_alpha = max(cfg["init_alpha"], 1e-15)
    las, lopts = [], []
    for _ in range(EL):
        la = torch.Tensor([math.log(init_alpha)]).to(device)
        la.requires_grad = True
        las.append(la)
        lopts.append(torch.optim.Adam([la], lr=cfg["alpha_lr"], betas=(0.5, 0.999)))
    if foreign_io:
        aug = foreign_agent.ForeignAugmentationSequence(
            [foreign_agent.Drqv2Aug(B) if (px and px["aug"] == "drqv2") else foreign_agent.IdentityAug(B)])
    elif px and px["aug"] == "drqv2":
        aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.Drqv2Aug(B)])
    else:
        aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.IdentityAug(B)])
    aug_mix = px["aug_mix"] if px else 0.0
    rproc = None
    nclip = None
    if cfg["noise"]:
        import types
        space = types.SimpleNamespace(low=-np.ones(cfg["act"], np.float32), high=np.ones(cfg["act"], np.float32))
        rproc = ssa.learning_utils.GaussianExplorationNoise(space, start_scale=cfg["noise"]["scale"],
                                                            final_scale=cfg["noise"]["scale"] * 0.1,
                                                            steps_annealed=1000)
        nclip = cfg["noise"]["clip"]
    stochastic = cfg["actor"] == "stochastic"
    player = DrawPlayer(device)
    player.install(ssa.rng)
    rec, upd = {}, 0
    try:
        for cyc in range(cfg["cycles"]):
            for k in range(cfg["utd"]):
                for i in range(E):
                    player.idx.append(fx[f"u{upd}_idx{i}"])
                    if px:
                        player.shift.append(fx[f"u{upd}_shift{i}"])
                    if stochastic:
                        player.normal.append(fx[f"u{upd}_eps{i}"])
                    if cfg["noise"]:
                        player.normal.append(fx[f"u{upd}_noise{i}"])
                    player.sub.append(fx[f"u{upd}_subset{i}"])
                    for k_ in range(E):  # "softmax" backup weights: every member's actor samples a' (lu:383-393)
                        if f"u{upd}_bweps{i}_{k_}" in fx:
                            player.normal.append(fx[f"u{upd}_bweps{i}_{k_}"])
                        if f"u{upd}_cat{i}_{k_}" in fx:
                            player.cats.append(fx[f"u{upd}_cat{i}_{k_}"])
                if shard is None:
                    player.picks.append(int(fx[f"u{upd}_gpick"]))
                logs, dicts = ssa.learning.critic_update(
                    buffer=buf, agent=agent, target_agent=target, critic_optimizer=copt,
                    encoder_optimizer=eopt, log_alphas=las, batch_size=B, gamma=cfg["gamma"],
                    critic_clip=cfg["clip"], encoder_clip=cfg["clip"],
                    target_critic_ensemble_n=cfg["n"], weighted_bellman_temp=cfg["temp"],
                    weight_type=cfg["weight_type"], pop=cfg["pop"], augmenter=aug,
                    encoder_lambda=cfg.get("encoder_lambda", 0),
                    aug_mix=aug_mix, discrete=cfg["discrete"], random_process=rproc, noise_clip=nclip,
                    per=False, update_priorities=False, dr3_coeff=0.0)
                for i in range(EL):
                    assert np.array_equal(dicts[i]["priority_idxs"], fx[f"u{upd}_idx{glob(i)}"])
                    rec[f"u{upd}_td{glob(i)}"] = dicts[i]["td_target"].cpu().numpy()
                    if cfg["popart"]:
                        p = agent.popart[i]._read()
                        rec[f"u{upd}_popart{glob(i)}"] = np.array([p.mu, p.nu, p.w, p.b, agent.popart[i].sigma, p.t])
                for key, val in logs.items():
                    rec[f"u{upd}_log:{key}"] = np.float64(float(val))
                if int(fx[f"u{upd}_polyak"]):
                    for ac, tc in zip(agent.critics, target.critics):
                        ssa.learning_utils.soft_update(tc, ac, cfg["tau"])
                    if px:
                        ssa.learning_utils.soft_update(target.encoder, agent.encoder, px["enc_tau"])
                upd += 1
            for i in range(E):
                if not cfg["discrete"]:
                    player.normal.append(fx[f"a{cyc}_eps{i}"])
                if cfg["noise"]:
                    player.normal.append(fx[f"a{cyc}_noise{i}"])
                if cfg.get("use_baseline"):
                    player.normal.extend(list(fx[f"a{cyc}_base{i}"]))
            player.picks.append(int(fx[f"a{cyc}_gpick"]))
            alog = ssa.learning.online_actor_update(
                buffer=buf, agent=agent, pop=cfg["pop"], actor_optimizer=aopt, log_alphas=las,
                batch_size=B, aug_mix=aug_mix, clip=cfg["clip"], augmenter=aug, per=False,
                discrete=cfg["discrete"], random_process=rproc, noise_clip=nclip,
                premade_replay_dicts=dicts, use_baseline=bool(cfg.get("use_baseline", False)))
            for key, val in alog.items():
                rec[f"a{cyc}_log:{key}"] = np.float64(float(val))
            if cfg["init_alpha"] > 0 and cfg["alpha_lr"] > 0:
                for i in range(E):
                    if stochastic:
                        player.normal.append(fx[f"l{cyc}_eps{i}"])
                llog = ssa.learning.alpha_update(
                    buffer=buf, agent=agent, optimizers=lopts, batch_size=B, log_alphas=las,
                    augmenter=aug, aug_mix=aug_mix, target_entropy=_target_entropy(cfg),
                    premade_replay_dicts=dicts, discrete=cfg["discrete"])
                for key, val in llog.items():
                    rec[f"l{cyc}_log:{key}"] = np.float64(float(val))
        assert not player.idx and not player.sub and not player.normal and not player.shift and \
            not player.cats and not player.picks, "unconsumed recorded draws"
    finally:
        player.restore()
    crit = [p for i in range(EL) for j in range(NL) for p in agent.critics[i].nets[j].parameters()]
    tcrit = [p for i in range(EL) for j in range(NL) for p in target.critics[i].nets[j].parameters()]
    act = [p for i in range(EL) for p in agent.actors[i].parameters()]
    grp = copt._ssac_adam
    m_list, v_list = [], []
    for i in range(EL):
        ar = agent.critics[i].arena(device)
        m, v = grp.moments_for(("critic", i), ar.params)
        for j in range(NL):
            for seg in ("w1", "b1", "w2", "b2", "w3", "b3"):
                m_list.append(ar.view(j, seg, m))
                v_list.append(ar.view(j, seg, v))
    _finalise(rec, fx, crit, act, tcrit, m_list, v_list, las)
    if px:
        def enc_params(enc):
            conv = enc.conv_block
            names = ["conv1", "conv2", "conv3", "conv4"] if px["kind"] == "big" else ["conv1", "conv2", "conv3"]
            out = []
            for nm in names:
                out += [getattr(conv, nm).weight, getattr(conv, nm).bias]
            out += [conv.fc.weight, conv.fc.bias]
            if px["kind"] == "big":
                out += [conv.ln.weight, conv.ln.bias]
            return out
        rec["finalfp_encoder"] = _fingerprint(enc_params(agent.encoder))
        rec["finalfp_target_encoder"] = _fingerprint(enc_params(target.encoder))
    return rec


def run_engine_stock(name, device="cuda", members=None, cycles=3):
    """The update sequence of a case with the STOCK generators -- no injected draws: replay indices from torch's CPU
    generator, REDQ subsets / logged-net picks from Python's `random`, the policy noise from the engine's Philox stream
    (in-kernel where the launches draw it) or the device generator.  Returns {global member index: flat parameters} for the
    actors and critics of the members this process holds, and their temperatures.  (What the fixtures cannot pin: they
    replay recorded draws through hooks, so the stock noise path of a member-sharded rank never ran under them.)"""
    import random
    import super_sac_amd as ssa
    cfg = synth.CASES[name]
    B, E = cfg["B"], cfg["E"]
    device = torch.device(device)
    torch.manual_seed(cfg["seed"]); np.random.seed(cfg["seed"]); random.seed(cfg["seed"])
    buf = ssa.replay.ReplayBuffer(cfg["cap"], device=device)
    buf.load_experience(*_buffers(cfg))
    agent = build_engine_agent(cfg, device, None, members=members)
    target = copy.deepcopy(agent)
    if members is not None:
        ssa.parallel.install_members(agent, target, members)
    EL = agent.ensemble_size
    glob = (lambda i: i) if members is None else (lambda i: members.lo + i)
    copt = torch.optim.Adam(chain(*(c.parameters() for c in agent.critics)), lr=cfg["lr"], betas=(0.9, 0.999))
    aopt = torch.optim.Adam(chain(*(a.parameters() for a in agent.actors)), lr=cfg["lr"], betas=(0.9, 0.999))
    eopt = torch.optim.Adam(agent.encoder.parameters(), lr=1e-4, betas=(0.9, 0.999))
    las, lopts = [], []
    for _ in range(EL):
        la = torch.Tensor([math.log(max(cfg["init_alpha"], 1e-15))]).to(device)
        la.requires_grad = True
        las.append(la)
        lopts.append(torch.optim.Adam([la], lr=cfg["alpha_lr"], betas=(0.5, 0.999)))
    aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.IdentityAug(B)])
    # every process re-seeds behind the construction (a rank that built fewer members consumed fewer draws)
    torch.manual_seed(cfg["seed"] + 1); np.random.seed(cfg["seed"] + 1); random.seed(cfg["seed"] + 1)
    torch.cuda.manual_seed(cfg["seed"] + 1)
    upd = 0
    for cyc in range(cycles):
        for k in range(cfg["utd"]):
            _, dicts = ssa.learning.critic_update(
                buffer=buf, agent=agent, target_agent=target, critic_optimizer=copt, encoder_optimizer=eopt,
                log_alphas=las, batch_size=B, gamma=cfg["gamma"], critic_clip=cfg["clip"], encoder_clip=cfg["clip"],
                target_critic_ensemble_n=cfg["n"], weighted_bellman_temp=cfg["temp"], weight_type=cfg["weight_type"],
                pop=cfg["pop"], augmenter=aug, encoder_lambda=0, aug_mix=0.0, discrete=cfg["discrete"],
                random_process=None, noise_clip=None, per=False, update_priorities=False, dr3_coeff=0.0)
            if upd % cfg["target_delay"] == 0:
                for ac, tc in zip(agent.critics, target.critics):
                    ssa.learning_utils.soft_update(tc, ac, cfg["tau"])
            upd += 1
        ssa.learning.online_actor_update(
            buffer=buf, agent=agent, pop=cfg["pop"], actor_optimizer=aopt, log_alphas=las, batch_size=B, aug_mix=0.0,
            clip=cfg["clip"], augmenter=aug, per=False, discrete=cfg["discrete"], random_process=None, noise_clip=None,
            premade_replay_dicts=dicts)
        ssa.learning.alpha_update(
            buffer=buf, agent=agent, optimizers=lopts, batch_size=B, log_alphas=las, augmenter=aug, aug_mix=0.0,
            target_entropy=_target_entropy(cfg), premade_replay_dicts=dicts, discrete=cfg["discrete"])
    torch.cuda.synchronize()
    out = {}
    for i in range(EL):
        out[f"actor{glob(i)}"] = _flat(agent.actors[i].parameters())
        out[f"critic{glob(i)}"] = _flat(chain(*(n_.parameters() for n_ in agent.critics[i].nets)))
        out[f"log_alpha{glob(i)}"] = np.float64(float(las[i].detach()))
    return out


def slice_fixture(fx, cfg, shard):
    """the part of a (single-member, full-dump) fixture that rank `shard` can reproduce: its own
    critics' parameters/moments; TD targets, actor parameters and temperature are global."""
    assert cfg["E"] == 1, "sharded replays use single-member fixtures"
    out = dict(fx)
    in_dim = cfg["obs"] if cfg["discrete"] else cfg["obs"] + cfg["act"]
    out_dim = cfg["act"] if cfg["discrete"] else 1
    H = cfg["hidden"]
    per = H * in_dim + H + H * H + H + out_dim * H + out_dim
    for key in ("final_critic", "final_target_critic"):
        if key in fx:
            out[key] = fx[key][shard.lo * per: shard.hi * per]
    sizes = [H * in_dim, H, H * H, H, out_dim * H, out_dim]
    fpn = [min(48, n) for n in sizes]
    perfp = sum(fpn)
    for key in ("finalfp_critic", "finalfp_target_critic", "finalfp_critic_m", "finalfp_critic_v"):
        if key in fx:
            out[key] = fx[key][shard.lo * perfp: shard.hi * perfp]
    # the critic-loss / td-error / grad-norm logs are per-rank partial sums in a sharded run
    for key in list(out):
        if "_log:losses/critic" in key or "_log:losses/last_member" in key or "_log:gradients/" in key:
            del out[key]
    return out


def slice_fixture_members(fx, cfg, ms):
    """the part of a full-dump fixture that a member-sharded rank (parallel.MemberShard) reproduces: TD targets, TD and
    temperature logs, parameters, moments and temperatures of ITS members -- under their GLOBAL indices.  The overall
    critic loss / TD error / actor loss are per-rank partial sums; the bellman-weight statistics are the last OWNED
    member's (the fixture's when this rank holds the last member); the gradient-norm logs are the picked member's when
    this rank holds it."""
    E, N = cfg["E"], cfg["N"]
    in_dim = cfg["obs"] if cfg["discrete"] else cfg["obs"] + cfg["act"]
    out_dim = cfg["act"] if cfg["discrete"] else 1
    H = cfg["hidden"]
    per_c = (H * in_dim + H + H * H + H + out_dim * H + out_dim) * N
    a_out = cfg["act"] if cfg["actor"] != "stochastic" else 2 * cfg["act"]
    per_a = H * cfg["obs"] + H + H * H + H + a_out * H + a_out
    perfp = sum(min(48, n) for n in [H * in_dim, H, H * H, H, out_dim * H, out_dim]) * N
    out = {}
    owned = range(ms.lo, ms.hi)
    for key, val in fx.items():
        m = re.fullmatch(r"([ual]\d+)_(td|popart)(\d+)", key)
        if m:
            if int(m.group(3)) in owned:
                out[key] = val
            continue
        m = re.fullmatch(r"[ual]\d+_log:(td_targets/\w+?|losses/alpha_loss|alphas/alpha)_(\d+)", key)
        if m:
            if int(m.group(2)) in owned:
                out[key] = val
            continue
        if "_log:bellman_weights/" in key:
            if ms.owns(E - 1):
                out[key] = val
            continue
        if "_log:gradients/critic_random_grad" in key or "_log:gradients/random_actor_online_grad" in key:
            pick = int(fx[key.split("_log:")[0] + "_gpick"])
            if ms.owns(pick):
                out[key] = val
            continue
        if "_log:losses/" in key:
            continue   # (sums over members: partial on a rank)
        if key in ("final_critic", "final_target_critic"):
            out[key] = val[ms.lo * per_c: ms.hi * per_c]
        elif key == "final_actor":
            out[key] = val[ms.lo * per_a: ms.hi * per_a]
        elif key in ("finalfp_critic_m", "finalfp_critic_v"):
            out[key] = val[ms.lo * perfp: ms.hi * perfp]
        elif key == "final_log_alpha":
            out[key] = val[ms.lo: ms.hi]
        else:
            out[key] = val
    return out


# ------------------------------------------------------------------------------------------
# fixture keys that are INPUTS of a replay (the host draws the reference consumed, step counts) -- everything else in a
# fixture is an OUTPUT of the reference and a backend's record must carry it
_INPUT_KEY = re.compile(r"(n_updates|n_steps|ckpt\|.*|[ualsm]\d+_(eps\d*|ceps|prio_eps|bweps\d+_\d+|noise\d+|idx\d*|subset\d*|shift\d*|"
                        r"cat\d+(_\d+)?|gpick|polyak|base\d+|perm|per|filter|critic))")   # (ckpt|...: the reference checkpoint a resume case starts from)


def straggler_check(err, tol, max_step, who, key):
    """Parameters after k Adam steps: every element within `tol`, except SIGN-FLIP STRAGGLERS -- weights whose gradient
    is pure rounding noise (|g| ~ 1e-9 behind a dead ReLU): Adam moves such a weight by ~lr per step whatever |g| is, so
    another summation order can send it the other way.  They are COUNTED (at most 1 in 10^4 elements, at least one
    allowed per tensor), bounded by the distance opposite steps can open (`max_step`), never waved through by a wide
    worst-element tolerance."""
    err = np.asarray(err)
    bad = err > tol
    allowed = max(1, -(-err.size // 10000))
    assert int(bad.sum()) <= allowed, f"{who}: {key}: {int(bad.sum())} of {err.size} elements over {tol} (allowed {allowed})"
    assert float(err.max(initial=0.0)) <= max_step, f"{who}: {key}: worst element {err.max():.3e} > {max_step:.3e}"
    return int(bad.sum())


def compare(rec, fx, *, td_tol=2e-4, log_rtol=5e-4, par_tol=3e-5, enc_max_step=3e-3, who="backend", mlp_max_step=0.0):
    """Assert rec (a backend's record) matches the reference fixture within the stated
    fp32 tolerances.  Returns the worst deviations for reporting.  EVERY output key of the fixture must be in the
    record: a dropped key fails (round 3 skipped silently anything that was not a TD / log key)."""
    worst = {"td": 0.0, "log": 0.0, "param": 0.0, "stragglers": 0}
    for key, ref in fx.items():
        if _INPUT_KEY.fullmatch(key):
            continue
        assert key in rec, f"{who}: the record lacks the fixture's output {key}"
        got = np.asarray(rec[key], dtype=np.float64)
        ref = np.asarray(ref, dtype=np.float64)
        if "_td" in key:
            dv = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
            worst["td"] = max(worst["td"], dv)
            assert dv <= td_tol, f"{who}: {key} deviates {dv:.3e} > {td_tol}"
        elif "_log:" in key:
            # (gradient norms are compared RELATIVE to their size: they range from 1e-3 to 1e2)
            dv = float(abs(got - ref) / (max(1.0, abs(ref)) if "gradients/" not in key else max(1e-6, abs(ref))))
            if "gradients/" in key and abs(ref) < 1e-12:
                dv = float(abs(got))
            worst["log"] = max(worst["log"], dv)
            assert dv <= log_rtol, f"{who}: {key} = {got} vs reference {ref}"
        elif "encoder" in key and key.startswith("final"):
            err = np.abs(got - ref)
            worst["stragglers"] += straggler_check(err, par_tol, enc_max_step, who, key)
            assert float(np.median(err)) <= 1e-6, f"{who}: {key} median {np.median(err):.3e}"
        elif key.startswith("final") and key != "final_log_alpha" and mlp_max_step > 0.0 and not key.endswith("_v") \
                and not key.endswith("_m"):
            # (full-size pixel cases, synth.FULL_SIZE: counted sign-flip stragglers in the MLP parameters as well, bounded
            # by the distance opposite Adam steps can open)
            err = np.abs(got - ref)
            worst["stragglers"] += straggler_check(err, par_tol, mlp_max_step, who, key)
        elif key.startswith("final") and key != "final_log_alpha":
            scale = 1.0 if not key.endswith("_v") else max(1e-12, float(np.max(np.abs(ref))))
            dv = float(np.max(np.abs(got - ref)) / scale)
            tol = par_tol if not key.endswith("_v") else 1e-3
            worst["param"] = max(worst["param"], dv)
            assert dv <= tol, f"{who}: {key} deviates {dv:.3e} > {tol}"
        elif key == "final_log_alpha":
            assert np.max(np.abs(got - ref)) <= 1e-6, f"{who}: log_alpha {got} vs {ref}"
        elif "_popart" in key:
            assert np.allclose(got, ref, rtol=1e-4, atol=1e-5), f"{who}: {key} {got} vs {ref}"
        else:
            raise AssertionError(f"{who}: fixture key {key} is neither a declared input nor an output compare() knows")
    return worst


def compare_full_size(rec, fx, cfg, who):
    """The pixel configurations at BASELINE.json's full sizes (synth.FULL_SIZE) against the reference's fixture.
    * Update 0 -- everything that does not sit behind an optimizer step: TD targets 2e-4, logs 5e-4 (measured on the GPU:
      TD 7.5e-7, losses exact, gradient norms <= 3.5e-4 relative): the full-size forward, loss and backward.
    * Parameters / Polyak targets after the sequence: median error <= 2e-6, at least 99 % of the elements within 3e-5, EVERY
      element within the distance opposite Adam steps can open (2.2 lr per step).  Adam's first steps move a weight by
      ~lr * sign(g) whatever |g| is; at these sizes a fraction of a percent of the weights (fc columns behind mostly-dead
      features, 860 k-term convolution sums) have gradients at rounding level, which another summation order sends the
      other way -- the reference's own CPU arithmetic against the oracle's flips 6 of 499 220 MLP weights, the GPU's
      implicit-GEMM order ~0.5 % of the sampled encoder weights.  Counted and bounded, not waved through.
    * Quantities BEHIND an optimizer step (update >= 1, the actor / temperature step) inherit that: TD targets 3e-3, logs 1e-2
      (measured: <= 8.6e-4 / 5.3e-3)."""
    lr_max = max(cfg["lr"], (cfg.get("pixels") or {}).get("enc_lr", 0.0))
    max_step = 2.2 * lr_max * cfg["cycles"] * cfg["utd"]
    worst = {"td0": 0.0, "log0": 0.0, "td": 0.0, "log": 0.0, "param_max": 0.0, "param_bad_frac": 0.0}
    for key, ref in fx.items():
        if _INPUT_KEY.fullmatch(key):
            continue
        assert key in rec, f"{who}: the record lacks the fixture's output {key}"
        got, ref = np.asarray(rec[key], np.float64), np.asarray(ref, np.float64)
        first = key.startswith("u0_")
        if "_td" in key:
            dv = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
            worst["td0" if first else "td"] = max(worst["td0" if first else "td"], dv)
            assert dv <= (2e-4 if first else 3e-3), f"{who}: {key} deviates {dv:.3e}"
        elif "_log:" in key:
            dv = float(abs(got - ref) / (max(1.0, abs(ref)) if "gradients/" not in key else max(1e-6, abs(ref))))
            worst["log0" if first else "log"] = max(worst["log0" if first else "log"], dv)
            assert dv <= (5e-4 if first else 1e-2), f"{who}: {key} = {got} vs reference {ref}"
        elif key == "final_log_alpha":
            assert np.max(np.abs(got - ref)) <= 1e-6, f"{who}: log_alpha {got} vs {ref}"
        elif key.endswith("_v") or key.endswith("_m"):
            scale = max(1e-12, float(np.max(np.abs(ref))))
            assert float(np.max(np.abs(got - ref))) / scale <= (1e-3 if key.endswith("_v") else 2e-2), f"{who}: {key}"
        elif key.startswith("final"):
            err = np.abs(got - ref)
            frac = float((err > 3e-5).mean())
            worst["param_max"], worst["param_bad_frac"] = max(worst["param_max"], float(err.max())), max(worst["param_bad_frac"], frac)
            assert float(np.median(err)) <= 2e-6 and frac <= 0.01 and float(err.max()) <= max_step, \
                f"{who}: {key}: median {np.median(err):.3e}, {frac:.2%} over 3e-5, max {err.max():.3e} (bound {max_step:.2e})"
        else:
            raise AssertionError(f"{who}: fixture key {key} is neither a declared input nor an output compare_full_size knows")
    return worst


# ------------------------------------------------------------------------------------------
# AFBC / PER cases (synth.AFBC_CASES, fixtures written by oracle/gen_golden.py::run_afbc_case)
# ------------------------------------------------------------------------------------------
def run_afbc_oracle(name):
    """the CPU oracle on the recorded draws; returns a record shaped like the fixture."""
    cfg = synth.AFBC_CASES[name]
    fx = load_fixture(name)
    B = cfg["B"]
    torch.manual_seed(cfg["seed"]); np.random.seed(cfg["seed"]); random.seed(cfg["seed"])
    obuf = orc.ReplayOracle(cfg["cap"])
    obuf.load_experience(*_buffers(cfg))
    tree = orc.PerOracle(cfg["cap"], 0.6, 1.0)
    tree.push_rows(np.arange(cfg["rows"]))
    oa = _oracle_agent(cfg)
    oa.requires_grad_(True)
    aopt = orc.AdamOracle(oa.actor_params(), lr=cfg["lr"])
    aug = orc.AugOracle("identity", B)
    rec = {}
    if any(st_ == "critic" for st_ in cfg["steps"]):
        ot = oa.clone()
        copt = orc.AdamOracle(oa.critic_params(), lr=cfg["lr"])
        eopt = orc.AdamOracle(oa.encoder_params(), lr=1e-4)
        ola = torch.tensor([math.log(cfg["init_alpha"])], requires_grad=True)
    for k, step in enumerate(cfg["steps"]):
        if step == "critic":
            idx = fx[f"s{k}_idx"]
            torch.randint(len(obuf), (B,))
            logs, odicts = orc.critic_update(
                obuf, oa, ot, copt, eopt, [ola], B, cfg["gamma"], cfg["clip"], cfg["clip"], cfg["n"], None, None,
                False, aug, aug_mix=0.0, idx_list=[idx], eps_list=[torch.from_numpy(fx[f"s{k}_ceps"])],
                subset_list=[[int(v) for v in fx[f"s{k}_subset"]]], dr3_coeff=cfg.get("dr3", 0.0))
            rd = odicts[-1]
            adv = orc.advantage(oa, rd["primary_batch"][0], rd["primary_batch"][1], 0,
                                [torch.from_numpy(e) for e in fx[f"s{k}_prio_eps"]])
            prio = (torch.relu(adv) + 1e-4).squeeze(1).numpy()
            tree.update(idx, prio)
            orc.soft_update(ot.critic_params(), oa.critic_params(), cfg["tau"])
            rec[f"s{k}_idx"], rec[f"s{k}_prio"] = idx, prio
            rec[f"s{k}_leaves"] = tree.sum[tree.cap + idx].copy()
            for key, v in logs.items():
                rec[f"s{k}_log:{key}"] = np.float64(v)
            continue
        per, filt = step
        disc = bool(cfg["discrete"])
        eps = [torch.from_numpy(e) for e in fx[f"s{k}_eps"]] if (filt and not disc) else None
        peps = [torch.from_numpy(e) for e in fx[f"s{k}_prio_eps"]] if (per and not disc) else None
        if per:  # the oracle's own prioritised draw must reproduce the recorded one
            idx, w = tree.sample(len(obuf), B)
            rec[f"s{k}_idx"], rec[f"s{k}_weights"] = idx, w
        else:
            idx = fx[f"s{k}_idx"]
            torch.randint(len(obuf), (B,))  # keep the torch CPU stream in step
            rec[f"s{k}_idx"] = idx
        logs, rd, prio, _ = orc.offline_actor_update(
            obuf, tree if per else None, oa, aopt, B, cfg["clip"], aug, 0.0, per=per, filter_=filt,
            idx_list=[idx], eps_lists=[eps] if (filt and not disc) else None, prio_member=0 if per else None,
            prio_eps=peps)
        if per:
            rec[f"s{k}_prio"] = prio
            rec[f"s{k}_leaves"] = tree.sum[tree.cap + idx].copy()
        for key, v in logs.items():
            rec[f"s{k}_log:{key}"] = np.float64(v)
    rec["final_actor"] = _flat(oa.actor_params())
    if "final_critic" in fx:
        rec["final_critic"] = _flat(oa.critic_params())
    rec["final_max_priority"] = np.float64(tree.max_priority)
    rec["final_tree_total"] = np.float64(tree.sum[1])
    return rec


def run_afbc_engine(name, device="cuda"):
    """super_sac_amd.learning.offline_actor_update on the HIP path, fed the recorded normal draws; the PER index
    draw is the engine's own (numpy global generator seeded like the reference run)."""
    import super_sac_amd as ssa
    cfg = synth.AFBC_CASES[name]
    fx = load_fixture(name)
    B = cfg["B"]
    device = torch.device(device)
    torch.manual_seed(cfg["seed"]); np.random.seed(cfg["seed"]); random.seed(cfg["seed"])
    buf = ssa.replay.ReplayBuffer(cfg["cap"], device=device)
    buf.load_experience(*_buffers(cfg))
    agent = build_engine_agent(cfg, device)
    aopt = torch.optim.Adam(chain(*(a.parameters() for a in agent.actors)), lr=cfg["lr"], betas=(0.9, 0.999))
    eopt = torch.optim.Adam(agent.encoder.parameters(), lr=1e-4)
    aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.IdentityAug(B)])
    player = DrawPlayer(device)
    player.install(ssa.rng)
    rec = {}
    if any(st_ == "critic" for st_ in cfg["steps"]):
        target = copy.deepcopy(agent)
        copt = torch.optim.Adam(chain(*(c.parameters() for c in agent.critics)), lr=cfg["lr"], betas=(0.9, 0.999))
        la = torch.Tensor([math.log(cfg["init_alpha"])]).to(device)
        la.requires_grad = True
    try:
        for k, step in enumerate(cfg["steps"]):
            seen = {}
            orig_upd = buf.update_priorities

            def spy(idxes, prios, _seen=seen, _orig=orig_upd):
                if torch.is_tensor(prios):   # (trees in HBM: indices and priorities arrive as device data)
                    prios_h = prios.detach().cpu().numpy().astype(np.float64)
                else:
                    prios_h = np.asarray(prios, np.float64).copy()
                _seen["idx"], _seen["prio"] = np.asarray(idxes).copy(), prios_h
                return _orig(idxes, prios)
            if step == "critic":
                player.idx.append(fx[f"s{k}_idx"])
                player.normal.append(fx[f"s{k}_ceps"])
                player.sub.append(fx[f"s{k}_subset"])
                player.normal.extend(list(fx[f"s{k}_prio_eps"]))
                buf.update_priorities = spy
                logs, _ = ssa.learning.critic_update(
                    buffer=buf, agent=agent, target_agent=target, critic_optimizer=copt, encoder_optimizer=eopt,
                    log_alphas=[la], batch_size=B, gamma=cfg["gamma"], critic_clip=cfg["clip"],
                    encoder_clip=cfg["clip"], target_critic_ensemble_n=cfg["n"], weighted_bellman_temp=None,
                    weight_type=None, pop=False, augmenter=aug, encoder_lambda=0, aug_mix=0.0, discrete=False,
                    random_process=None, noise_clip=None, per=False, update_priorities=True,
                    dr3_coeff=cfg.get("dr3", 0.0))
                buf.update_priorities = orig_upd
                for ac, tc in zip(agent.critics, target.critics):
                    ssa.learning_utils.soft_update(tc, ac, cfg["tau"])
                rec[f"s{k}_idx"], rec[f"s{k}_prio"] = seen["idx"], seen["prio"]
                rec[f"s{k}_leaves"] = buf._per.sum_tree[buf._per.cap + seen["idx"]].copy()
                for key, v in logs.items():
                    rec[f"s{k}_log:{key}"] = np.float64(float(v))
                assert not player.normal and not player.idx and not player.sub
                continue
            per, filt = step
            if not per:
                player.idx.append(fx[f"s{k}_idx"])
            disc = bool(cfg["discrete"])
            if filt and not disc:
                player.normal.extend(list(fx[f"s{k}_eps"]))
            if per and not disc:
                player.normal.extend(list(fx[f"s{k}_prio_eps"]))
            buf.update_priorities = spy
            logs = ssa.learning.offline_actor_update(
                buffer=buf, agent=agent, actor_optimizer=aopt, encoder_optimizer=eopt, batch_size=B,
                actor_clip=cfg["clip"], update_encoder=False, encoder_clip=cfg["clip"], augmenter=aug,
                actor_lambda=0.0, aug_mix=0.0, premade_replay_dicts=None, per=per, discrete=disc, filter_=filt)
            buf.update_priorities = orig_upd
            if per:
                rec[f"s{k}_idx"], rec[f"s{k}_prio"] = seen["idx"], seen["prio"]
                rec[f"s{k}_leaves"] = buf._per.sum_tree[buf._per.cap + seen["idx"]].copy()
            for key, v in logs.items():
                rec[f"s{k}_log:{key}"] = np.float64(float(v))
            assert not player.normal and not player.idx, "the engine consumed a different number of draws"
    finally:
        player.restore()
    rec["final_actor"] = _flat([p for a in agent.actors for p in a.parameters()])
    if "final_critic" in fx:
        rec["final_critic"] = _flat([p for c in agent.critics for p in c.parameters()])
    rec["final_max_priority"] = np.float64(buf._per._max_priority)
    rec["final_tree_total"] = np.float64(buf._per.sum_tree[1])
    return rec


def compare_afbc(rec, fx, log_rtol=5e-4, par_tol=3e-5, prio_tol=2e-4):
    n = int(fx["n_steps"])
    for k in range(n):
        if f"s{k}_critic" in fx:
            np.testing.assert_allclose(rec[f"s{k}_prio"], fx[f"s{k}_prio"], rtol=prio_tol, atol=2e-6,
                                       err_msg=f"step {k}: priorities after the critic update")
            np.testing.assert_allclose(rec[f"s{k}_leaves"], fx[f"s{k}_leaves"], rtol=prio_tol, atol=2e-6)
        elif int(fx[f"s{k}_per"]):
            assert np.array_equal(rec[f"s{k}_idx"], fx[f"s{k}_idx"]), f"step {k}: prioritised index draw differs"
            if f"s{k}_weights" in rec:
                np.testing.assert_allclose(rec[f"s{k}_weights"], fx[f"s{k}_weights"], rtol=1e-9)
            np.testing.assert_allclose(rec[f"s{k}_prio"], fx[f"s{k}_prio"], rtol=prio_tol, atol=2e-6,
                                       err_msg=f"step {k}: new priorities")
            np.testing.assert_allclose(rec[f"s{k}_leaves"], fx[f"s{k}_leaves"], rtol=prio_tol, atol=2e-6)
        for key in fx:
            if key.startswith(f"s{k}_log:"):
                v, r = float(rec[key]), float(fx[key])
                scale = max(1.0, abs(r)) if "gradients/" not in key else max(1e-6, abs(r))  # norms: relative
                assert abs(v - r) <= log_rtol * scale, (key, v, r)
    d = float(np.abs(rec["final_actor"] - fx["final_actor"]).max())
    assert d <= par_tol, f"final actor parameters differ by {d:.3e}"
    if "final_critic" in fx:
        d = float(np.abs(rec["final_critic"] - fx["final_critic"]).max())
        assert d <= max(par_tol, 3e-5), f"final critic parameters differ by {d:.3e}"
    assert abs(float(rec["final_max_priority"]) - float(fx["final_max_priority"])) < 1e-6
    assert abs(float(rec["final_tree_total"]) - float(fx["final_tree_total"])) <= 1e-5 * float(fx["final_tree_total"])


# ------------------------------------------------------------------------------------------
# Markov state-abstraction update (synth.MARKOV_CASES, fixtures written by oracle/gen_golden.py::run_markov_case)
# ------------------------------------------------------------------------------------------
def markov_models(cfg):
    """seeded inverse / contrastive model weights (the draw oracle/gen_golden.py::synth_markov_models makes)"""
    px = cfg.get("pixels")
    emb = px["emb"] if px else cfg["obs"]
    rng_ = np.random.RandomState(cfg["seed"] + 11)
    inv_out = cfg["act"] if cfg["discrete"] else 2 * cfg["act"]
    return orc.make_mlp(rng_, 2 * emb, cfg["hidden"], inv_out), orc.make_mlp(rng_, 2 * emb, cfg["hidden"], 1)


def run_markov_oracle(name):
    cfg = synth.MARKOV_CASES[name]
    fx = load_fixture(name)
    B, mk, px = cfg["B"], cfg["markov"], cfg.get("pixels")
    obuf = orc.ReplayOracle(cfg["cap"])
    obuf.load_experience(*_buffers(cfg))
    oa = _oracle_agent(cfg).requires_grad_(True)
    inv_p, con_p = markov_models(cfg)
    for t in list(inv_p.values()) + list(con_p.values()):
        t.requires_grad_(True)
    opt = orc.AdamOracle(oa.encoder_params() + [inv_p[k] for k in orc.MLP_KEYS] + [con_p[k] for k in orc.MLP_KEYS],
                         lr=cfg["lr"])
    aug = orc.AugOracle(px["aug"] if px else "identity", B)
    ic, cc, sc = mk["coeffs"]
    rec = {}
    for k in range(mk["steps"]):
        if f"m{k}_shift" in fx:
            aug.forced = [torch.from_numpy(fx[f"m{k}_shift"])]
        logs, _ = orc.markov_state_abstraction_update(
            obuf, oa, inv_p, con_p, opt, B, aug, px["aug_mix"] if px else 0.0, ic, cc, sc, mk["max_dist"],
            mk["clip"][k], idx=fx[f"m{k}_idx"], perm=torch.from_numpy(fx[f"m{k}_perm"]))
        for key, val in logs.items():
            rec[f"m{k}_log:{key}"] = np.float64(val)
    rec["final_inverse"] = _flat([inv_p[k] for k in orc.MLP_KEYS])
    rec["final_contrastive"] = _flat([con_p[k] for k in orc.MLP_KEYS])
    if px:
        rec["finalfp_encoder"] = _fingerprint(oa.encoder_params())
    return rec


def run_markov_engine(name, device="cuda"):
    import super_sac_amd as ssa
    cfg = synth.MARKOV_CASES[name]
    fx = load_fixture(name)
    B, mk, px = cfg["B"], cfg["markov"], cfg.get("pixels")
    device = torch.device(device)
    buf = ssa.replay.ReplayBuffer(cfg["cap"], device=device)
    buf.load_experience(*_buffers(cfg))
    agent = build_engine_agent(cfg, device)
    inv_p, con_p = markov_models(cfg)
    with torch.no_grad():
        for mod, p, names in ((agent.inverse_model, inv_p, ("fc1", "fc2", "act_p" if cfg["discrete"] else "fc3")),
                              (agent.contrastive_model, con_p, ("fc1", "fc2", "out"))):
            for (wk, bk), nm in zip((("w1", "b1"), ("w2", "b2"), ("w3", "b3")), names):
                getattr(mod, nm).weight.copy_(p[wk])
                getattr(mod, nm).bias.copy_(p[bk])
    opt = torch.optim.Adam(chain(agent.encoder.parameters(), agent.inverse_model.parameters(),
                                 agent.contrastive_model.parameters()), lr=cfg["lr"], weight_decay=0, betas=(0.9, 0.999))
    if px and px["aug"] == "drqv2":
        aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.Drqv2Aug(B)])
    else:
        aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.IdentityAug(B)])
    ic, cc, sc = mk["coeffs"]
    player = DrawPlayer(device)
    player.install(ssa.rng)
    saved_perm = ssa.rng.draw_permutation
    perms = []
    ssa.rng.draw_permutation = lambda n: torch.from_numpy(perms.pop(0))
    rec = {}
    try:
        for k in range(mk["steps"]):
            player.idx.append(fx[f"m{k}_idx"])
            if f"m{k}_shift" in fx:
                player.shift.append(fx[f"m{k}_shift"])
            perms.append(fx[f"m{k}_perm"])
            logs = ssa.learning.markov_state_abstraction_update(
                buffer=buf, agent=agent, optimizer=opt, batch_size=B, augmenter=aug,
                aug_mix=px["aug_mix"] if px else 0.0, discrete=cfg["discrete"], inverse_coeff=ic, contrastive_coeff=cc,
                smoothness_coeff=sc, smoothness_max_dist=mk["max_dist"], grad_clip=mk["clip"][k])
            for key, val in logs.items():
                rec[f"m{k}_log:{key}"] = np.float64(float(val))
        assert not player.idx and not player.shift and not perms
    finally:
        player.restore()
        ssa.rng.draw_permutation = saved_perm
    rec["final_inverse"] = _flat(list(agent.inverse_model.parameters()))
    rec["final_contrastive"] = _flat(list(agent.contrastive_model.parameters()))
    if px:
        conv = agent.encoder.conv_block if hasattr(agent.encoder, "conv_block") else ssa.conv_encoder.find_conv_module(agent.encoder)
        names = ["conv1", "conv2", "conv3", "conv4"] if px["kind"] == "big" else ["conv1", "conv2", "conv3"]
        plist = []
        for nm in names:
            plist += [getattr(conv, nm).weight, getattr(conv, nm).bias]
        plist += [conv.fc.weight, conv.fc.bias]
        if px["kind"] == "big":
            plist += [conv.ln.weight, conv.ln.bias]
        rec["finalfp_encoder"] = _fingerprint(plist)
    return rec


def compare_markov(rec, fx, who, log_tol=5e-4, gn_tol=2e-3, par_tol=3e-5, max_step=0.0):
    """logs relative (gradient norms a little wider: norms of sums in another order), parameters absolute"""
    worst = {"log": 0.0, "par": 0.0}
    for key, want in fx.items():
        if "_log:" in key:
            got, want = float(rec[key]), float(want)
            tol = gn_tol if "gradients/" in key else log_tol
            err = abs(got - want) / max(1.0, abs(want))
            assert err <= tol, f"{who}: {key}: {got} vs {want}"
            worst["log"] = max(worst["log"], err)
        elif key.startswith("final"):
            d = np.abs(rec[key] - want)
            nbad = int((d > par_tol).sum())
            # Adam's first steps move a weight by ~lr whatever the size of its gradient: a weight whose gradient is
            # rounding noise (behind a dead ReLU at these tiny batches) can step the other way under another summation
            # order (oracle/gen_golden.py::run_markov_case).  So: all but a handful within par_tol, the handful within
            # the distance two opposite steps per update can open, the median at rounding level.
            few = max(2, d.size // 1000) if max_step else 0
            assert nbad <= few and (nbad == 0 or d.max() <= max_step), \
                f"{who}: {key}: max|diff| {d.max()}, {nbad} of {d.size} over tolerance"
            assert np.median(d) <= 1e-6, f"{who}: {key}: median |diff| {np.median(d)}"
            worst["par"] = max(worst["par"], float(np.median(d)))
    return worst


# ------------------------------------------------------------------------------------------
# BC warm-up on pixels (synth.BC_PIXEL_CASES, fixtures written by oracle/gen_golden.py::run_bc_pixels_case)
# ------------------------------------------------------------------------------------------
def _encoder_param_list(conv, kind):
    names = ["conv1", "conv2", "conv3", "conv4"] if kind == "big" else ["conv1", "conv2", "conv3"]
    plist = []
    for nm in names:
        plist += [getattr(conv, nm).weight, getattr(conv, nm).bias]
    plist += [conv.fc.weight, conv.fc.bias]
    if kind == "big":
        plist += [conv.ln.weight, conv.ln.bias]
    return plist


def _per_member(idx, shift, E):
    """the recorded draws of a step as per-member lists (single-member fixtures store them without the member axis)"""
    if E == 1:
        return [idx], [shift]
    return [idx[i] for i in range(E)], [shift[i] for i in range(E)]


def run_bc_pixels_oracle(name):
    cfg = synth.BC_PIXEL_CASES[name]
    fx = load_fixture(name)
    B, px = cfg["B"], cfg["pixels"]
    obuf = orc.ReplayOracle(cfg["cap"])
    obuf.load_experience(*_buffers(cfg))
    oa = _oracle_agent(cfg).requires_grad_(True)
    aopt = orc.AdamOracle(oa.actor_params(), lr=cfg["lr"])
    eopt = orc.AdamOracle(oa.encoder_params(), lr=px["enc_lr"])
    aug = orc.AugOracle("drqv2", B)
    rec = {}
    for k in range(len(cfg["steps"])):
        idxs, shifts = _per_member(fx[f"s{k}_idx"], fx[f"s{k}_shift"], cfg["E"])
        aug.forced = [torch.from_numpy(sh) for sh in shifts]
        logs, _, _, _ = orc.offline_actor_update(
            obuf, None, oa, aopt, B, cfg["clip"], aug, px["aug_mix"], per=False, filter_=False,
            idx_list=idxs, update_encoder=True, encoder_opt=eopt, encoder_clip=cfg["enc_clip"][k],
            grad_pick=int(fx[f"s{k}_gpick"]) if f"s{k}_gpick" in fx else 0)
        for key, val in logs.items():
            rec[f"s{k}_log:{key}"] = np.float64(val)
    rec["final_actor"] = _flat(oa.actor_params())
    rec["finalfp_encoder"] = _fingerprint(oa.encoder_params())
    return rec


def run_bc_pixels_engine(name, device="cuda"):
    import super_sac_amd as ssa
    cfg = synth.BC_PIXEL_CASES[name]
    fx = load_fixture(name)
    B, px = cfg["B"], cfg["pixels"]
    device = torch.device(device)
    buf = ssa.replay.ReplayBuffer(cfg["cap"], device=device)
    buf.load_experience(*_buffers(cfg))
    agent = build_engine_agent(cfg, device)
    aopt = torch.optim.Adam(chain(*(a.parameters() for a in agent.actors)), lr=cfg["lr"], betas=(0.9, 0.999))
    eopt = torch.optim.Adam(agent.encoder.parameters(), lr=px["enc_lr"], betas=(0.9, 0.999))
    aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.Drqv2Aug(B)])
    player = DrawPlayer(device)
    player.install(ssa.rng)
    rec = {}
    try:
        for k in range(len(cfg["steps"])):
            for idx_i, shift_i in zip(*_per_member(fx[f"s{k}_idx"], fx[f"s{k}_shift"], cfg["E"])):
                player.idx.append(idx_i)
                player.shift.append(shift_i)
            if f"s{k}_gpick" in fx:
                player.picks.append(int(fx[f"s{k}_gpick"]))
            logs = ssa.learning.offline_actor_update(
                buffer=buf, agent=agent, actor_optimizer=aopt, encoder_optimizer=eopt, batch_size=B,
                actor_clip=cfg["clip"], update_encoder=True, encoder_clip=cfg["enc_clip"][k], augmenter=aug,
                actor_lambda=0.0, aug_mix=px["aug_mix"], premade_replay_dicts=None, per=False,
                discrete=cfg["discrete"], filter_=False)
            for key, val in logs.items():
                rec[f"s{k}_log:{key}"] = np.float64(float(val))
        assert not player.idx and not player.shift
    finally:
        player.restore()
    rec["final_actor"] = _flat([p for a in agent.actors for p in a.parameters()])
    rec["finalfp_encoder"] = _fingerprint(_encoder_param_list(ssa.conv_encoder.find_conv_module(agent.encoder), px["kind"]))
    return rec


# ------------------------------------------------------------------------------------------
# action invariance constraint (synth.ACTOR_INV_CASES, fixtures written by oracle/gen_golden.py::run_actor_inv_case)
# ------------------------------------------------------------------------------------------
def run_actor_inv_oracle(name):
    cfg = synth.ACTOR_INV_CASES[name]
    fx = load_fixture(name)
    B, E, px, disc = cfg["B"], cfg["E"], cfg.get("pixels"), bool(cfg["discrete"])
    obuf = orc.ReplayOracle(cfg["cap"])
    obuf.load_experience(*_buffers(cfg))
    oa = _oracle_agent(cfg).requires_grad_(True)
    aopt = orc.AdamOracle(oa.actor_params(), lr=cfg["lr"])
    eopt = orc.AdamOracle(oa.encoder_params(), lr=px["enc_lr"] if px else 1e-4)
    aug = orc.AugOracle("drqv2" if px else "identity", B)
    rec = {}
    for k, stp in enumerate(cfg["steps"]):
        if px:
            aug.forced = [torch.from_numpy(fx[f"s{k}_shift{i}"]) for i in range(E)]
        logs, _, _, _ = orc.offline_actor_update(
            obuf, None, oa, aopt, B, stp["clip"], aug, px["aug_mix"] if px else 0.0, per=False, filter_=False,
            idx_list=[fx[f"s{k}_idx{i}"] for i in range(E)], update_encoder=stp["update_encoder"], encoder_opt=eopt,
            encoder_clip=stp.get("enc_clip"), actor_lambda=cfg["actor_lambda"],
            inv_eps_list=None if (disc or f"s{k}_eps0" not in fx) else [torch.from_numpy(fx[f"s{k}_eps{i}"]) for i in range(E)],
            inv_cat_list=[torch.from_numpy(fx[f"s{k}_cat{i}"]) for i in range(E)] if disc else None,
            grad_pick=int(fx[f"s{k}_gpick"]))
        for key, val in logs.items():
            rec[f"s{k}_log:{key}"] = np.float64(val)
    rec["final_actor"] = _flat(oa.actor_params())
    if px:
        rec["finalfp_encoder"] = _fingerprint(oa.encoder_params())
    return rec


def run_actor_inv_engine(name, device="cuda"):
    import super_sac_amd as ssa
    cfg = synth.ACTOR_INV_CASES[name]
    fx = load_fixture(name)
    B, E, px, disc = cfg["B"], cfg["E"], cfg.get("pixels"), bool(cfg["discrete"])
    device = torch.device(device)
    buf = ssa.replay.ReplayBuffer(cfg["cap"], device=device)
    buf.load_experience(*_buffers(cfg))
    agent = build_engine_agent(cfg, device)
    aopt = torch.optim.Adam(chain(*(a.parameters() for a in agent.actors)), lr=cfg["lr"], betas=(0.9, 0.999))
    eopt = torch.optim.Adam(agent.encoder.parameters(), lr=px["enc_lr"] if px else 1e-4, betas=(0.9, 0.999))
    aug = ssa.augmentations.AugmentationSequence(
        [ssa.augmentations.Drqv2Aug(B) if px else ssa.augmentations.IdentityAug(B)])
    player = DrawPlayer(device)
    player.install(ssa.rng)
    rec = {}
    try:
        for k, stp in enumerate(cfg["steps"]):
            for i in range(E):
                player.idx.append(fx[f"s{k}_idx{i}"])
                if px:
                    player.shift.append(fx[f"s{k}_shift{i}"])
                if disc:
                    player.cats.append(fx[f"s{k}_cat{i}"])
                elif f"s{k}_eps{i}" in fx:   # (a deterministic actor's sample() is its loc: no draw)
                    player.normal.append(fx[f"s{k}_eps{i}"])
            player.picks.append(int(fx[f"s{k}_gpick"]))
            logs = ssa.learning.offline_actor_update(
                buffer=buf, agent=agent, actor_optimizer=aopt, encoder_optimizer=eopt, batch_size=B,
                actor_clip=stp["clip"], update_encoder=stp["update_encoder"], encoder_clip=stp.get("enc_clip"),
                augmenter=aug, actor_lambda=cfg["actor_lambda"], aug_mix=px["aug_mix"] if px else 0.0,
                premade_replay_dicts=None, per=False, discrete=disc, filter_=False)
            for key, val in logs.items():
                rec[f"s{k}_log:{key}"] = np.float64(float(val))
        assert not player.idx and not player.shift and not player.normal and not player.cats and not player.picks
    finally:
        player.restore()
    rec["final_actor"] = _flat([p for a in agent.actors for p in a.parameters()])
    if px:
        rec["finalfp_encoder"] = _fingerprint(_encoder_param_list(ssa.conv_encoder.find_conv_module(agent.encoder), px["kind"]))
    return rec
