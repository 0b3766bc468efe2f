"""Generate tests/golden/*.npz by driving the UNMODIFIED reference (dev container only).

    python oracle/gen_golden.py            # writes tests/golden/*.npz

The reference has no tests or known-answer vectors of its own, so
these fixtures are the pin for the oracle (and, through it, for the HIP kernels).  Each
fixture stores the host-RNG draws the reference consumed (replay indices, REDQ subsets,
normal draws) and the reference's outputs; the big inputs (buffers, weights) are rebuilt
from numpy seeds by ``tests/synth.py`` + ``oracle/ssac_oracle.py::make_mlp``.

While generating, the oracle is run on the same inputs and the maximum deviation from the
reference is printed (and asserted), i.e. this script is also the oracle-vs-reference check.
"""
import copy
import math
import os
import random
import sys
import types
from itertools import chain

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_harness  # noqa: E402
import ssac_oracle as orc  # noqa: E402
import synth  # noqa: E402
import case_runner  # noqa: E402  (tests/: the checkpoint <-> fixture-array helpers shared with the parity tests)

ref = ref_harness.import_reference()
rl = ref.learning
rlu = ref.learning_utils
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(4)


class RefIdentityEncoder(ref.nets.Encoder):
    """Same role as experiments/gym/train_gym.py:18-28 (built here; scripts need gym)."""

    def __init__(self, dim):
        super().__init__()
        self._dim = dim

    @property
    def embedding_dim(self):
        return self._dim

    def forward(self, obs_dict):
        return obs_dict["obs"]


class RefPixelEncoder(ref.nets.Encoder):
    """Same role as DMCPixelEncoder / AtariEncoder (experiments/dmc/train_dmc_from_pixels.py:15-27,
    experiments/atari/train_atari.py:9-20); those scripts need gym/dmc2gym, so it is rebuilt here."""

    def __init__(self, conv_block, dim):
        super().__init__()
        self.conv_block = conv_block
        self._dim = dim

    @property
    def embedding_dim(self):
        return self._dim

    def forward(self, obs_dict):
        return self.conv_block(obs_dict["obs"])


def build_encoders(cfg):
    """(reference encoder module, oracle encoder dict) for a case."""
    px = cfg.get("pixels")
    if not px:
        return RefIdentityEncoder(cfg["obs"]), None
    p = orc.make_conv_encoder(np.random.RandomState(cfg["seed"] + 7), px["kind"], px["channels"], px["emb"])
    cls = ref.nets.cnns.BigPixelEncoder if px["kind"] == "big" else ref.nets.cnns.SmallPixelEncoder
    m = cls((px["channels"], px["hw"], px["hw"]), px["emb"])
    names = ["conv1", "conv2", "conv3", "conv4"] if px["kind"] == "big" else ["conv1", "conv2", "conv3"]
    for i, nm in enumerate(names, 1):
        load_linear(getattr(m, nm), p[f"c{i}w"], p[f"c{i}b"])
    load_linear(m.fc, p["fcw"], p["fcb"])
    if px["kind"] == "big":
        load_linear(m.ln, p["lnw"], p["lnb"])
    return RefPixelEncoder(m, px["emb"]), {"kind": px["kind"], "key": "obs", "p": p}


def ref_encoder_params(enc, cfg):
    """reference encoder tensors in the oracle's dict order (c1w, c1b, ..., fcw, fcb[, lnw, lnb])."""
    m = enc.conv_block
    names = ["conv1", "conv2", "conv3", "conv4"] if cfg["pixels"]["kind"] == "big" else ["conv1", "conv2", "conv3"]
    out = []
    for nm in names:
        out += [getattr(m, nm).weight, getattr(m, nm).bias]
    out += [m.fc.weight, m.fc.bias]
    if cfg["pixels"]["kind"] == "big":
        out += [m.ln.weight, m.ln.bias]
    return out


def load_linear(lin, w, b):
    lin.weight.data.copy_(w)
    lin.bias.data.copy_(b)


def load_mlp(mod, p, names):
    for (wk, bk), name in zip((("w1", "b1"), ("w2", "b2"), ("w3", "b3")), names):
        load_linear(getattr(mod, name), p[wk], p[bk])


def build_pair(cfg):
    """(reference agent, oracle agent) holding identical seeded weights."""
    r_enc, o_enc = build_encoders(cfg)
    oa = orc.AgentOracle(state_dim=cfg["obs"], act_dim=cfg["act"], hidden=cfg["hidden"],
                         num_critics=cfg["N"], ensemble_size=cfg["E"], discrete=cfg["discrete"],
                         actor_kind=cfg["actor"], log_std_low=cfg["lo"], log_std_high=cfg["hi"],
                         popart=cfg["popart"], encoder=o_enc, seed=cfg["seed"])
    actor_cls = {"stochastic": ref.nets.mlps.ContinuousStochasticActor,
                 "deterministic": ref.nets.mlps.ContinuousDeterministicActor,
                 "discrete": ref.nets.mlps.DiscreteActor}[cfg["actor"]]
    critic_cls = ref.nets.mlps.DiscreteCritic if cfg["discrete"] else ref.nets.mlps.ContinuousCritic
    ra = ref.Agent(act_space_size=cfg["act"], encoder=r_enc,
                   actor_network_cls=actor_cls, critic_network_cls=critic_cls,
                   discrete=cfg["discrete"], ensemble_size=cfg["E"], num_critics=cfg["N"],
                   ucb_bonus=0.0, hidden_size=cfg["hidden"], auto_rescale_targets=cfg["popart"],
                   log_std_low=cfg["lo"], log_std_high=cfg["hi"])
    last = {"stochastic": "fc3", "deterministic": "out", "discrete": "act_p"}[cfg["actor"]]
    for i in range(cfg["E"]):
        load_mlp(ra.actors[i], oa.actors[i], ("fc1", "fc2", last))
        for j in range(cfg["N"]):
            load_mlp(ra.critics[i].nets[j], oa.critics[i][j], ("fc1", "fc2", "out"))
        if cfg["popart"]:
            ms = cfg.get("popart_min_steps", 1000)
            ra.popart[i].min_steps = ms
            oa.popart[i].min_steps = ms
    ra.to(ref.device)
    ra.train()
    return ra, oa


def ref_params(ra, cfg):
    """flat list in the oracle's order (critics, then actors)."""
    crit = [p for i in range(cfg["E"]) for j in range(cfg["N"])
            for p in ra.critics[i].nets[j].parameters()]
    act = [p for i in range(cfg["E"]) for p in ra.actors[i].parameters()]
    return crit, act


def maxdiff(a_list, b_list):
    return max(float((a.detach() - b.detach()).abs().max()) for a, b in zip(a_list, b_list))


def assert_encoder_close(ref_list, orc_list):
    """Adam's first steps move a weight by lr * sign-like(g): a weight whose gradient is rounding noise (fc columns
    behind a dead ReLU at these tiny batches) can step the other way under a different summation order, so the pin is
    "all but a handful of the ~2M encoder weights agree", not the maximum."""
    denc = maxdiff(ref_list, orc_list)
    nbad = sum(int(((a_.detach() - b_.detach()).abs() > 5e-5).sum()) for a_, b_ in zip(ref_list, orc_list))
    ntot = sum(a_.numel() for a_ in ref_list)
    print(f"   encoder params max|diff| {denc:.3e}; {nbad} of {ntot} differ by more than 5e-5")
    assert nbad <= max(2, ntot // 100000)


_CKPT_DIRS = {}   # base case -> directory the REFERENCE agent saved itself to at the end of that case


def reference_checkpoint(base):
    """Agent.save (agent.py:172-195) of the reference agent at the end of case `base`'s update sequence"""
    if base not in _CKPT_DIRS:
        import tempfile
        ra = run_case(base, synth.CASES[base], write=False)
        d = tempfile.mkdtemp(prefix=f"ssac_ckpt_{base}_")
        ra.save(d)
        _CKPT_DIRS[base] = d
    return _CKPT_DIRS[base]


def run_case(name, cfg, write=True):
    print(f"== {name}" + ("" if write else "  (for its checkpoint only)"))
    ckpt_rec = {}
    ckpt_dir = reference_checkpoint(cfg["resume"]) if cfg.get("resume") else None
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    random.seed(cfg["seed"])
    B, E, N = cfg["B"], cfg["E"], cfg["N"]
    px = cfg.get("pixels")
    if px:
        s, a, r, s1, d = synth.synth_pixel_transitions(cfg["rows"], px["channels"], px["hw"],
                                                      n_actions=cfg["act"] if cfg["discrete"] else None,
                                                      act_dim=cfg["act"], seed=cfg["seed"] + 100)
    else:
        s, a, r, s1, d = synth.synth_transitions(cfg["rows"], cfg["obs"], cfg["act"], cfg["discrete"],
                                                seed=cfg["seed"] + 100, n_actions=cfg["act"])
    rbuf = ref.replay.ReplayBuffer(cfg["cap"])
    rbuf.load_experience(s, a, r, s1, d)
    obuf = orc.ReplayOracle(cfg["cap"])
    obuf.load_experience(s, a, r, s1, d)

    ra, oa = build_pair(cfg)
    if ckpt_dir:
        # a reference user's resume: Agent.load into a freshly built agent (other seed), target = deepcopy, new optimizers
        before = torch.cat([p.detach().flatten() for p in ra.critics[0].parameters()]).clone()
        ra.load(ckpt_dir)
        assert not torch.equal(before, torch.cat([p.detach().flatten() for p in ra.critics[0].parameters()]))
        for fname in sorted(os.listdir(ckpt_dir)):
            sd = torch.load(os.path.join(ckpt_dir, fname), map_location="cpu")
            ckpt_rec[f"{case_runner.CKPT_PREFIX}{fname}|"] = np.zeros(0, np.float32)   # (the file exists, even when empty)
            for key, val in sd.items():
                ckpt_rec[f"{case_runner.CKPT_PREFIX}{fname}|{key}"] = val.detach().cpu().numpy()
        case_runner.oracle_load_checkpoint(oa, cfg, case_runner.checkpoint_arrays(ckpt_rec))
        ra.train()
    rt = copy.deepcopy(ra)
    oa.requires_grad_(True)
    ot = oa.clone()

    # optimizers exactly as main.py:188-239
    r_copt = torch.optim.Adam(chain(*(c.parameters() for c in ra.critics)), lr=cfg["lr"],
                              weight_decay=0, betas=(0.9, 0.999))
    r_aopt = torch.optim.Adam(chain(*(ac.parameters() for ac in ra.actors)), lr=cfg["lr"],
                              weight_decay=0, betas=(0.9, 0.999))
    enc_lr = px["enc_lr"] if px else 1e-4
    r_eopt = torch.optim.Adam(ra.encoder.parameters(), lr=enc_lr, betas=(0.9, 0.999))
    init_alpha = max(cfg["init_alpha"], 1e-15)
    r_las, r_lopts, o_las, o_lopts = [], [], [], []
    for _ in range(E):
        la = torch.Tensor([math.log(init_alpha)]).to(ref.device)
        la.requires_grad = True
        r_las.append(la)
        r_lopts.append(torch.optim.Adam([la], lr=cfg["alpha_lr"], betas=(0.5, 0.999)))
        ola = torch.tensor([math.log(init_alpha)], requires_grad=True)
        o_las.append(ola)
        o_lopts.append(orc.AdamOracle([ola], lr=cfg["alpha_lr"], betas=(0.5, 0.999)))
    o_copt = orc.AdamOracle(oa.critic_params(), lr=cfg["lr"])
    o_aopt = orc.AdamOracle(oa.actor_params(), lr=cfg["lr"])
    o_eopt = orc.AdamOracle(oa.encoder_params(), lr=enc_lr)
    target_entropy = (-math.log(1.0 / cfg["act"]) * 0.98) if cfg["discrete"] else -float(cfg["act"])

    if px and px["aug"] == "drqv2":
        r_aug = ref.augmentations.AugmentationSequence([ref.augmentations.Drqv2Aug(B)])
        o_aug = orc.AugOracle("drqv2", B)
    else:
        r_aug = ref.augmentations.AugmentationSequence([ref.augmentations.IdentityAug(B)])
        o_aug = orc.AugOracle("identity", B)
    aug_mix = px["aug_mix"] if px else 0.0
    if cfg["noise"]:
        space = types.SimpleNamespace(low=-np.ones(cfg["act"], np.float32),
                                      high=np.ones(cfg["act"], np.float32))
        rproc = rlu.GaussianExplorationNoise(space, start_scale=cfg["noise"]["scale"],
                                             final_scale=cfg["noise"]["scale"] * 0.1,
                                             steps_annealed=1000)
        nscale, nclip = cfg["noise"]["scale"], cfg["noise"]["clip"]
    else:
        rproc, nscale, nclip = None, None, None

    # capture the reference's TD targets without touching its source
    captured = []
    orig_td = rlu.compute_td_targets

    def td_spy(*args, **kw):
        out = orig_td(*args, **kw)
        captured.append(out[0].detach().clone())
        return out
    rlu.compute_td_targets = td_spy

    stochastic = (cfg["actor"] == "stochastic")
    rec = {}
    worst = 0.0
    upd = 0
    for cyc in range(cfg["cycles"]):
        for k in range(cfg["utd"]):
            # replicate the host draws the reference is about to make, then rewind
            st, pst = torch.get_rng_state(), random.getstate()
            idxs, epss, noises, subsets, shifts = [], [], [], [], []
            bw_eps, bw_cat = [], []
            softmax_w = cfg["weight_type"] == "softmax" and E > 1
            for i in range(E):
                idxs.append(torch.randint(len(rbuf), (B,)).numpy())
                if px and px["aug"] == "drqv2":
                    shifts.append(orc.drqv2_draw_shift(B))
                if stochastic:
                    epss.append(torch.randn(B, cfg["act"]))
                if cfg["noise"]:
                    noises.append(torch.randn(B, cfg["act"]))
                subsets.append(random.sample(range(N), k=cfg["n"]))
                if softmax_w:
                    # learning_utils.py:383-393: every member's actor samples a' on THIS member's s' batch
                    assert not px, "softmax cases use vector observations"
                    if cfg["discrete"]:
                        with torch.no_grad():
                            s1_rep = rt.encoder({"obs": torch.from_numpy(s1["obs"][idxs[-1]]).float()})
                            bw_cat.append([ra.actors[k_](s1_rep).sample() for k_ in range(E)])
                    else:
                        bw_eps.append([torch.randn(B, cfg["act"]) for _ in range(E)])
            gpick = random.choice(range(E))  # random.choice(agent.critics), learning.py:135
            torch.set_rng_state(st)
            random.setstate(pst)

            captured.clear()
            rlogs, rdicts = rl.critic_update(
                buffer=rbuf, agent=ra, target_agent=rt, critic_optimizer=r_copt,
                encoder_optimizer=r_eopt, log_alphas=r_las, batch_size=B, gamma=cfg["gamma"],
                critic_clip=cfg["clip"], encoder_clip=cfg["clip"],
                target_critic_ensemble_n=cfg["n"], weighted_bellman_temp=cfg["temp"],
                weight_type=cfg["weight_type"], pop=cfg["pop"], augmenter=r_aug,
                encoder_lambda=cfg.get("encoder_lambda", 0),
                aug_mix=aug_mix, discrete=cfg["discrete"], random_process=rproc, noise_clip=nclip,
                per=False, update_priorities=False, dr3_coeff=0.0)
            if shifts:
                assert torch.equal(r_aug.aug_list[0].shift, shifts[-1]), "shift stream mismatch"
                o_aug.forced = [sh.clone() for sh in shifts]
            ologs, odicts = orc.critic_update(
                obuf, oa, ot, o_copt, o_eopt, o_las, B, cfg["gamma"], cfg["clip"], cfg["clip"],
                cfg["n"], cfg["temp"], cfg["weight_type"], cfg["pop"], o_aug, aug_mix=aug_mix,
                noise_scale=nscale, noise_clip=nclip, idx_list=idxs,
                eps_list=epss if stochastic else None,
                noise_list=noises if cfg["noise"] else None, subset_list=subsets,
                bw_eps_list=bw_eps or None, bw_cat_list=bw_cat or None, grad_pick=gpick,
                encoder_lambda=cfg.get("encoder_lambda", 0))
            rec[f"u{upd}_gpick"] = np.int64(gpick)
            for i in range(E):
                for k_ in range(E if softmax_w else 0):
                    if cfg["discrete"]:
                        rec[f"u{upd}_cat{i}_{k_}"] = bw_cat[i][k_].numpy().astype(np.int64)
                    else:
                        rec[f"u{upd}_bweps{i}_{k_}"] = bw_eps[i][k_].numpy()
            for i in range(E):
                assert np.array_equal(rdicts[i]["priority_idxs"], idxs[i]), "index stream mismatch"
                dtd = float((captured[i] - odicts[i]["td_target"]).abs().max())
                worst = max(worst, dtd)
                rec[f"u{upd}_idx{i}"] = idxs[i]
                rec[f"u{upd}_subset{i}"] = np.array(subsets[i], np.int64)
                if shifts:
                    rec[f"u{upd}_shift{i}"] = shifts[i].numpy()
                if stochastic:
                    rec[f"u{upd}_eps{i}"] = epss[i].numpy()
                if cfg["noise"]:
                    rec[f"u{upd}_noise{i}"] = noises[i].numpy()
                rec[f"u{upd}_td{i}"] = captured[i].numpy()
            for key, val in rlogs.items():
                v = float(val)
                rec[f"u{upd}_log:{key}"] = np.float64(v)
                dv = abs(v - float(ologs[key]))
                assert dv <= 2e-4 * max(1.0, abs(v)), (key, v, ologs[key])
            if cfg["popart"]:
                for i in range(E):
                    pr, po = ra.popart[i], oa.popart[i]
                    stt = np.array([float(pr.mu), float(pr.nu), float(pr.w), float(pr.b),
                                    float(pr.sigma), pr._t], np.float64)
                    rec[f"u{upd}_popart{i}"] = stt
                    ost = po.state()
                    assert abs(ost["mu"] - stt[0]) < 1e-6 and abs(ost["w"] - stt[2]) < 1e-5
            # polyak (main.py:409-414)
            if (k + cyc) % cfg["target_delay"] == 0:
                for ac, tc in zip(ra.critics, rt.critics):
                    rlu.soft_update(tc, ac, cfg["tau"])
                orc.soft_update(ot.critic_params(), oa.critic_params(), cfg["tau"])
                if px:
                    rlu.soft_update(rt.encoder, ra.encoder, px["enc_tau"])
                    orc.soft_update(ot.encoder_params(), oa.encoder_params(), px["enc_tau"])
                rec[f"u{upd}_polyak"] = np.int64(1)
            else:
                rec[f"u{upd}_polyak"] = np.int64(0)
            upd += 1

        # online actor update on the last critic batch (main.py:489-511)
        st, pst = torch.get_rng_state(), random.getstate()
        aeps, anoise, abase = [], [], []
        use_baseline = bool(cfg.get("use_baseline", False))
        for i in range(E):
            if not cfg["discrete"]:
                aeps.append(torch.randn(B, cfg["act"]))
            if cfg["noise"]:
                anoise.append(torch.randn(B, cfg["act"]))
            if use_baseline:  # adv_estimator.py:58-64: 4 policy samples for V(s), drawn after the rsample
                abase.append([torch.randn(B, cfg["act"]) for _ in range(4)])
        apick = random.choice(range(E))  # random.choice(agent.actors), learning.py:417-419
        torch.set_rng_state(st)
        random.setstate(pst)
        ralogs = rl.online_actor_update(
            buffer=rbuf, agent=ra, pop=cfg["pop"], actor_optimizer=r_aopt, log_alphas=r_las,
            batch_size=B, aug_mix=0.0, clip=cfg["clip"], augmenter=r_aug, per=False,
            discrete=cfg["discrete"], random_process=rproc, noise_clip=nclip,
            premade_replay_dicts=rdicts, use_baseline=use_baseline)
        oalogs = orc.online_actor_update(oa, o_aopt, o_las, odicts, cfg["pop"], cfg["clip"],
                                         eps_list=aeps if aeps else None, noise_scale=nscale,
                                         noise_clip=nclip, noise_list=anoise if anoise else None,
                                         use_baseline=use_baseline, base_eps_lists=abase or None,
                                         grad_pick=apick)
        rec[f"a{cyc}_gpick"] = np.int64(apick)
        for i in range(E):
            if aeps:
                rec[f"a{cyc}_eps{i}"] = aeps[i].numpy()
            if anoise:
                rec[f"a{cyc}_noise{i}"] = anoise[i].numpy()
            if use_baseline:
                rec[f"a{cyc}_base{i}"] = torch.stack(abase[i]).numpy()
        for key, val in ralogs.items():
            v = float(val)
            rec[f"a{cyc}_log:{key}"] = np.float64(v)
            assert abs(v - oalogs[key]) <= 2e-4 * max(1.0, abs(v)), (key, v, oalogs)

        if cfg["init_alpha"] > 0 and cfg["alpha_lr"] > 0:
            st = torch.get_rng_state()
            leps = []
            for i in range(E):
                if stochastic:
                    leps.append(torch.randn(B, cfg["act"]))
            torch.set_rng_state(st)
            rllogs = rl.alpha_update(buffer=rbuf, agent=ra, optimizers=r_lopts, batch_size=B,
                                     log_alphas=r_las, augmenter=r_aug, aug_mix=0.0,
                                     target_entropy=target_entropy, premade_replay_dicts=rdicts,
                                     discrete=cfg["discrete"])
            ollogs = orc.alpha_update(oa, o_lopts, o_las, odicts, target_entropy,
                                      eps_list=leps if leps else None)
            for i in range(E):
                if leps:
                    rec[f"l{cyc}_eps{i}"] = leps[i].numpy()
            for key, val in rllogs.items():
                rec[f"l{cyc}_log:{key}"] = np.float64(val)
                assert abs(val - ollogs[key]) <= 2e-4 * max(1.0, abs(val)), (key, val, ollogs[key])

    rlu.compute_td_targets = orig_td
    rc, rac = ref_params(ra, cfg)
    rtc, _ = ref_params(rt, cfg)
    dpar = max(maxdiff(rc, oa.critic_params()), maxdiff(rac, oa.actor_params()),
               maxdiff(rtc, ot.critic_params()))
    dla = max(abs(float(x) - float(y)) for x, y in zip(r_las, o_las))
    print(f"   td max|diff| {worst:.3e}   params max|diff| {dpar:.3e}   log_alpha diff {dla:.3e}")
    if name in synth.FULL_SIZE:
        # BASELINE's full pixel sizes (hidden-1024 MLPs behind a 39 200-wide encoder output, B 512 / 1024): the counted
        # sign-flip rule of assert_encoder_close for the MLPs too -- a weight whose gradient is rounding noise takes Adam's
        # first steps of ~lr in either direction under another summation order
        pairs = list(zip(rc, oa.critic_params())) + list(zip(rac, oa.actor_params())) + list(zip(rtc, ot.critic_params()))
        nbad = sum(int(((a_.detach() - b_.detach()).abs() > 5e-5).sum()) for a_, b_ in pairs)
        ntot = sum(a_.numel() for a_, _ in pairs)
        print(f"   {nbad} of {ntot} MLP parameters differ by more than 5e-5")
        assert worst < 5e-4 and dla < 1e-6 and nbad <= max(2, ntot // 10000) and dpar <= 2.2 * cfg["lr"] * max(upd, 1)
    else:
        assert worst < 5e-4 and dpar < 5e-5 and dla < 1e-6

    if px:
        re_, rte_ = ref_encoder_params(ra.encoder, cfg), ref_encoder_params(rt.encoder, cfg)
        assert_encoder_close(re_, oa.encoder_params())
        assert_encoder_close(rte_, ot.encoder_params())
        for tag, plist in (("encoder", re_), ("target_encoder", rte_)):
            vals = []
            for p in plist:
                flat = p.detach().numpy().ravel()
                vals.append(flat[synth.fingerprint_indices(flat.size)])
            rec[f"finalfp_{tag}"] = np.concatenate(vals)
    small = sum(p.numel() for p in rc) < 40000
    for tag, plist in (("critic", rc), ("actor", rac), ("target_critic", rtc)):
        if small:
            rec[f"final_{tag}"] = np.concatenate([p.detach().numpy().ravel() for p in plist])
        else:
            vals = []
            for p in plist:
                flat = p.detach().numpy().ravel()
                vals.append(flat[synth.fingerprint_indices(flat.size)])
            rec[f"finalfp_{tag}"] = np.concatenate(vals)
    # Adam moments of the critic optimizer (exp_avg / exp_avg_sq), fingerprinted
    ms, vs = [], []
    for p in rc:
        stt = r_copt.state[p]
        fi = synth.fingerprint_indices(p.numel())
        ms.append(stt["exp_avg"].numpy().ravel()[fi])
        vs.append(stt["exp_avg_sq"].numpy().ravel()[fi])
    rec["finalfp_critic_m"] = np.concatenate(ms)
    rec["finalfp_critic_v"] = np.concatenate(vs)
    rec["final_log_alpha"] = np.array([float(x) for x in r_las], np.float64)
    rec["n_updates"] = np.int64(upd)
    rec.update(ckpt_rec)
    if write:
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **rec)
    return ra


def gen_indices():
    print("== replay_indices")
    rec = {}
    for seed in (0, 7, 123):
        for n in (1000, 100_000, 1_000_000):
            torch.manual_seed(seed)
            draws = np.stack([torch.randint(n, (512,)).numpy() for _ in range(3)])
            rec[f"s{seed}_n{n}"] = draws
            mine = orc.randint_from_raw32(orc.MT19937(seed).raw32(3 * 512), 0, n).reshape(3, 512)
            assert np.array_equal(draws, mine), (seed, n)
    # the buffer gather itself: reference storage vs oracle storage on wrapped pushes
    s, a, r, s1, d = synth.synth_transitions(700, 5, 2, seed=3)
    rb = ref.replay.ReplayBuffer(512)
    ob = orc.ReplayOracle(512)
    for lo in range(0, 700, 100):
        sl = slice(lo, lo + 100)
        rb.push({"obs": s["obs"][sl]}, a[sl], r[sl, None], {"obs": s1["obs"][sl]}, d[sl, None])
        ob.push({"obs": s["obs"][sl]}, a[sl], r[sl, None], {"obs": s1["obs"][sl]}, d[sl, None])
    torch.manual_seed(5)
    (ro, ra_, rr, ro1, rd), ridx = rb.sample_uniform(64)
    oo, oa_, or_, oo1, od = ob.gather(ridx)
    assert torch.equal(ro["obs"].float(), oo["obs"]) and torch.equal(ra_, oa_)
    assert torch.equal(rr, or_) and torch.equal(rd.float(), od) and len(rb) == len(ob) == 512
    rec["wrap_idx"] = ridx
    rec["wrap_obs"] = ro["obs"].numpy()
    rec["wrap_next_obs"] = ro1["obs"].numpy()
    rec["wrap_act"] = ra_.numpy()
    rec["wrap_rew"] = rr.numpy()
    rec["wrap_done"] = rd.numpy()
    np.savez_compressed(os.path.join(OUT, "replay_indices.npz"), **rec)


def gen_popart():
    print("== popart")
    rec = {}
    rng = np.random.RandomState(5)
    rp = ref.popart.PopArtLayer(beta=1e-2, min_steps=3)
    op = orc.PopArtOracle(beta=1e-2, min_steps=3)
    states, outs = [], []
    for t in range(10):
        v = torch.from_numpy((rng.standard_normal((64, 1)) * (1.0 + t) + 0.5 * t).astype(np.float32))
        rec[f"v{t}"] = v.numpy()
        rp.update_stats(v)
        op.update_stats(v)
        states.append([float(rp.mu), float(rp.nu), float(rp.w), float(rp.b), float(rp.sigma),
                       rp._t, float(rp._stable)])
        x = torch.linspace(-2, 2, 9).unsqueeze(1)
        outs.append(torch.cat([rp(x), rp(x, normalized=False), rp.normalize_values(x)], 1).numpy())
        o = op.state()
        assert abs(o["mu"] - states[-1][0]) < 1e-6 and abs(o["b"] - states[-1][3]) < 1e-5
        assert float(op.stable) == states[-1][6]
    rec["states"] = np.array(states, np.float64)
    rec["outs"] = np.stack(outs)
    np.savez_compressed(os.path.join(OUT, "popart.npz"), **rec)


def gen_aug():
    print("== augmentations")
    rec = {}
    rng = np.random.RandomState(9)
    B, C, H = 6, 3, 24
    x0 = torch.from_numpy(rng.randint(0, 256, (B, C, H, H)).astype(np.float32))
    x1 = torch.from_numpy(rng.randint(0, 256, (B, C, H, H)).astype(np.float32))
    rec["x0"], rec["x1"] = x0.numpy(), x1.numpy()
    # Drqv2: one shift draw shared by s and s'
    torch.manual_seed(21)
    seq = ref.augmentations.AugmentationSequence([ref.augmentations.Drqv2Aug(B)])
    y0, y1 = seq({"obs": x0}, {"obs": x1})
    shift = seq.aug_list[0].shift.clone()
    rec["v2_shift"] = shift.numpy()
    rec["v2_y0"], rec["v2_y1"] = y0["obs"].numpy(), y1["obs"].numpy()
    torch.manual_seed(21)
    oaug = orc.AugOracle("drqv2", B)
    z0, z1 = oaug({"obs": x0}, {"obs": x1})
    assert torch.equal(oaug.last, shift)
    dv2 = max(float((z0["obs"] - y0["obs"]).abs().max()), float((z1["obs"] - y1["obs"]).abs().max()))
    print(f"   drqv2 oracle vs reference max|diff| = {dv2:.3e} (0-255 scale)")
    assert dv2 < 2e-3
    # production image size (84 -> padded 92): pins the fp32 grid arithmetic at that size
    bx = torch.from_numpy(np.random.RandomState(10).randint(0, 256, (3, 2, 84, 84)).astype(np.float32))
    torch.manual_seed(24)
    seqb = ref.augmentations.AugmentationSequence([ref.augmentations.Drqv2Aug(3)])
    by = seqb({"obs": bx})
    rec["big_x"], rec["big_shift"], rec["big_y"] = bx.numpy(), seqb.aug_list[0].shift.numpy(), by["obs"].numpy()
    bz = orc.drqv2_shift(bx, seqb.aug_list[0].shift)
    d84 = float((bz - by["obs"]).abs().max())
    print(f"   drqv2 84x84 oracle vs reference max|diff| = {d84:.3e}")
    assert d84 < 1e-3  # fp32 summation order of the 4 bilinear taps differs (0-255 scale)
    # DrQ v1 without noise: exact integer crop of the reflection-padded image
    torch.manual_seed(22)
    seq = ref.augmentations.AugmentationSequence([ref.augmentations.DrqNoNoiseAug(B)])
    y0, y1 = seq({"obs": x0}, {"obs": x1})
    rec["v1_w1"], rec["v1_h1"] = seq.aug_list[0].w1.numpy(), seq.aug_list[0].h1.numpy()
    rec["v1_y0"], rec["v1_y1"] = y0["obs"].numpy(), y1["obs"].numpy()
    torch.manual_seed(22)
    oaug = orc.AugOracle("drq_nonoise", B)
    z0, z1 = oaug({"obs": x0}, {"obs": x1})
    assert torch.equal(z0["obs"], y0["obs"]) and torch.equal(z1["obs"], y1["obs"])
    # DrQ v1 with N(0,1) noise (noise drawn after the crop, s first then s')
    torch.manual_seed(23)
    seq = ref.augmentations.AugmentationSequence([ref.augmentations.DrqAug(B)])
    st_after_ctor = None
    y0, y1 = seq({"obs": x0}, {"obs": x1})
    torch.manual_seed(23)
    oaug = orc.AugOracle("drq", B)
    z0, z1 = oaug({"obs": x0}, {"obs": x1})
    dn = max(float((z0["obs"] - y0["obs"]).abs().max()), float((z1["obs"] - y1["obs"]).abs().max()))
    print(f"   drq(noise) oracle vs reference max|diff| = {dn:.3e}")
    assert dn < 1e-5
    rec["v1n_y0"], rec["v1n_y1"] = y0["obs"].numpy(), y1["obs"].numpy()
    rec["v1n_w1"], rec["v1n_h1"] = seq.aug_list[0].w1.numpy(), seq.aug_list[0].h1.numpy()
    # the noise tensors themselves (so a device path can be checked with explicit noise)
    torch.manual_seed(23)
    orc.drq_draw_offsets(B)
    orc.drq_draw_offsets(B)
    rec["v1n_noise0"] = torch.randn(B, C, H, H).numpy()
    rec["v1n_noise1"] = torch.randn(B, C, H, H).numpy()
    # aug_mix row mixing through sample_move_and_augment (learning_utils.py:200-206)
    np.savez_compressed(os.path.join(OUT, "augmentations.npz"), **rec)


def gen_nets():
    print("== nets")
    rec = {}
    rng = np.random.RandomState(31)
    # ensemble-Q known answers at the metric shape
    N, B, obs, act, Hd = 10, 512, 17, 6, 256
    crit = [orc.make_mlp(rng, obs + act, Hd, 1) for _ in range(N)]
    s = torch.from_numpy(rng.standard_normal((B, obs)).astype(np.float32))
    a = torch.from_numpy(rng.uniform(-1, 1, (B, act)).astype(np.float32))
    mods = []
    for p in crit:
        m = ref.nets.mlps.ContinuousCritic(obs, act, Hd)
        load_mlp(m, p, ("fc1", "fc2", "out"))
        mods.append(m)
    with torch.no_grad():
        q_ref = torch.stack([m(s, a) for m in mods], 0).squeeze(-1)
        q_orc = torch.stack([orc.critic_q(p, s, a) for p in crit], 0).squeeze(-1)
    assert float((q_ref - q_orc).abs().max()) < 1e-5
    rec["ensq_q"] = q_ref.numpy()
    rec["ensq_seed"] = np.int64(31)
    # tanh-normal sample / log-prob
    for tag, lo, hi in (("redq", -5.0, 2.0), ("default", -10.0, 2.0)):
        out = torch.from_numpy((rng.standard_normal((64, 12)) * 1.5).astype(np.float32))
        eps = torch.from_numpy(rng.standard_normal((64, 6)).astype(np.float32))
        dist = ref.nets.distributions.create_tanh_normal(out, lo, hi)
        u = dist.loc + dist.scale * eps
        a_ref = dist.transforms[0](u)  # fills the transform cache like sample()/rsample()
        lp_ref = dist.log_prob(a_ref).sum(-1, keepdim=True)
        a_o, lp_o = orc.tanh_normal_sample(out, lo, hi, eps)
        assert float((a_ref - a_o).abs().max()) < 1e-6 and float((lp_ref - lp_o).abs().max()) < 2e-4
        rec[f"tn_{tag}_out"], rec[f"tn_{tag}_eps"] = out.numpy(), eps.numpy()
        rec[f"tn_{tag}_a"], rec[f"tn_{tag}_logp"] = a_ref.numpy(), lp_ref.numpy()
    # pixel encoders (forward)
    for kind, ch, emb in (("big", 9, 50), ("small", 4, 128)):
        p = orc.make_conv_encoder(np.random.RandomState(40 + ch), kind, ch, emb)
        cls = ref.nets.cnns.BigPixelEncoder if kind == "big" else ref.nets.cnns.SmallPixelEncoder
        m = cls((ch, 84, 84), emb)
        names = ["conv1", "conv2", "conv3", "conv4"] if kind == "big" else ["conv1", "conv2", "conv3"]
        for i, nm in enumerate(names, 1):
            load_linear(getattr(m, nm), p[f"c{i}w"], p[f"c{i}b"])
        load_linear(m.fc, p["fcw"], p["fcb"])
        if kind == "big":
            load_linear(m.ln, p["lnw"], p["lnb"])
        x = torch.from_numpy(np.random.RandomState(50 + ch).randint(0, 256, (3, ch, 84, 84)).astype(np.float32))
        with torch.no_grad():
            y_ref = m(x)
            y_orc = orc.encode({"kind": kind, "key": "obs", "p": p}, {"obs": x})
        assert float((y_ref - y_orc).abs().max()) < 1e-5
        rec[f"enc_{kind}_y"] = y_ref.numpy()
    np.savez_compressed(os.path.join(OUT, "nets.npz"), **rec)


def gen_per():
    print("== per (prioritised replay sample path, SURVEY 8(f) rank 1)")
    rec = {}
    s, a, r, s1, d = synth.synth_transitions(300, 4, 2, seed=8)
    rb = ref.replay.ReplayBuffer(400, alpha=0.6, beta=1.0)
    rb.load_experience(s, a, r, s1, d)
    per = orc.PerOracle(400, 0.6, 1.0)
    per.push_rows(np.arange(300))
    np.random.seed(77)
    st = np.random.get_state()
    _, w0, i0 = rb.sample(32)
    np.random.set_state(st)
    oi0, ow0 = per.sample(300, 32)
    assert np.array_equal(i0, oi0) and np.allclose(w0.numpy(), ow0)
    pr = np.random.RandomState(3).uniform(0.1, 4.0, 32)
    rb.update_priorities(i0, pr)
    per.update(i0, pr)
    st = np.random.get_state()
    _, w1, i1 = rb.sample(32)
    np.random.set_state(st)
    oi1, ow1 = per.sample(300, 32)
    assert np.array_equal(i1, oi1) and np.allclose(w1.numpy(), ow1, rtol=1e-12)
    rec.update(i0=i0, w0=w0.numpy(), prios=pr, i1=i1, w1=w1.numpy(), np_seed=np.int64(77))
    np.savez_compressed(os.path.join(OUT, "per.npz"), **rec)


def run_afbc_case(name, cfg):
    """learning.offline_actor_update (+ adv_estimator, PER sample / adjust_priorities) on the unmodified
    reference; the oracle follows on the recorded draws and every output is asserted against it."""
    print(f"== {name}")
    torch.manual_seed(cfg["seed"]); np.random.seed(cfg["seed"]); random.seed(cfg["seed"])
    B, E, A = cfg["B"], cfg["E"], cfg["act"]
    disc = bool(cfg["discrete"])
    s, a, r, s1, d = synth.synth_transitions(cfg["rows"], cfg["obs"], cfg["act"], disc, seed=cfg["seed"] + 100,
                                             n_actions=cfg["act"])
    rbuf = ref.replay.ReplayBuffer(cfg["cap"])
    rbuf.load_experience(s, a, r, s1, d)
    obuf = orc.ReplayOracle(cfg["cap"])
    obuf.load_experience(s, a, r, s1, d)
    tree = orc.PerOracle(cfg["cap"], 0.6, 1.0)
    tree.push_rows(np.arange(cfg["rows"]))
    ra, oa = build_pair(cfg)
    oa.requires_grad_(True)
    r_aopt = torch.optim.Adam(chain(*(ac.parameters() for ac in ra.actors)), lr=cfg["lr"], betas=(0.9, 0.999))
    r_eopt = torch.optim.Adam(ra.encoder.parameters(), lr=1e-4)
    o_aopt = orc.AdamOracle(oa.actor_params(), lr=cfg["lr"])
    r_aug = ref.augmentations.AugmentationSequence([ref.augmentations.IdentityAug(B)])
    o_aug = orc.AugOracle("identity", B)

    advs = []
    orig_fwd = type(ra.adv_estimator).forward

    def adv_spy(self, *args, **kw):
        out = orig_fwd(self, *args, **kw)
        advs.append(out.detach().clone())
        return out
    type(ra.adv_estimator).forward = adv_spy

    has_critic = any(st_ == "critic" for st_ in cfg["steps"])
    if has_critic:
        rt, ot = copy.deepcopy(ra), oa.clone()
        r_copt = torch.optim.Adam(chain(*(c.parameters() for c in ra.critics)), lr=cfg["lr"], betas=(0.9, 0.999))
        o_copt = orc.AdamOracle(oa.critic_params(), lr=cfg["lr"])
        o_eopt = orc.AdamOracle(oa.encoder_params(), lr=1e-4)
        rla = torch.Tensor([math.log(cfg["init_alpha"])]).to(ref.device); rla.requires_grad = True
        ola = torch.tensor([math.log(cfg["init_alpha"])], requires_grad=True)

    rec = {"n_steps": np.int64(len(cfg["steps"]))}
    for k, step in enumerate(cfg["steps"]):
        if step == "critic":
            # learning.critic_update(per=False, update_priorities=True) as in the offline phase of main.py
            st, pst = torch.get_rng_state(), random.getstate()
            idx = torch.randint(len(rbuf), (B,)).numpy()
            ceps = torch.randn(B, A)
            sub = random.sample(range(cfg["N"]), k=cfg["n"])
            random.choice(list(range(cfg["N"])))  # random.choice(agent.critics) for the grad-norm log
            pm = random.choice(range(E))
            peps = [torch.randn(B, A) for _ in range(4)]
            torch.set_rng_state(st); random.setstate(pst)
            advs.clear()
            rlogs, _ = rl.critic_update(
                buffer=rbuf, agent=ra, target_agent=rt, critic_optimizer=r_copt, encoder_optimizer=r_eopt,
                log_alphas=[rla], batch_size=B, gamma=cfg["gamma"], critic_clip=cfg["clip"], encoder_clip=cfg["clip"],
                target_critic_ensemble_n=cfg["n"], weighted_bellman_temp=None, weight_type=None, pop=False,
                augmenter=r_aug, encoder_lambda=0, aug_mix=0.0, discrete=False, random_process=None,
                noise_clip=None, per=False, update_priorities=True, dr3_coeff=cfg.get("dr3", 0.0))
            ologs, odicts = orc.critic_update(
                obuf, oa, ot, o_copt, o_eopt, [ola], B, cfg["gamma"], cfg["clip"], cfg["clip"], cfg["n"], None, None,
                False, o_aug, aug_mix=0.0, idx_list=[idx], eps_list=[ceps], subset_list=[sub],
                dr3_coeff=cfg.get("dr3", 0.0))
            ord_ = odicts[-1]
            oadv = orc.advantage(oa, ord_["primary_batch"][0], ord_["primary_batch"][1], pm, peps)
            oprio = (torch.relu(oadv) + 1e-4).squeeze(1).numpy()
            tree.update(idx, oprio)
            leaves = np.array([rbuf._it_sum[int(j)] for j in idx], np.float64)
            assert np.allclose(leaves, tree.sum[tree.cap + idx], rtol=1e-4, atol=1e-6), "critic priority refresh"
            for ac, tc in zip(ra.critics, rt.critics):
                rlu.soft_update(tc, ac, cfg["tau"])
            orc.soft_update(ot.critic_params(), oa.critic_params(), cfg["tau"])
            rec[f"s{k}_critic"] = np.int64(1)
            rec[f"s{k}_idx"], rec[f"s{k}_ceps"] = np.asarray(idx, np.int64), ceps.numpy()
            rec[f"s{k}_subset"] = np.array(sub, np.int64)
            rec[f"s{k}_prio_eps"], rec[f"s{k}_prio"], rec[f"s{k}_leaves"] = torch.stack(peps).numpy(), oprio, leaves
            for key, val in rlogs.items():
                rec[f"s{k}_log:{key}"] = np.float64(float(val))
                assert abs(float(val) - float(ologs[key])) <= 2e-4 * max(1.0, abs(float(val))), key
            continue
        per, filt = step
        # replicate the host draws the reference is about to make, then rewind
        st, nst, pst = torch.get_rng_state(), np.random.get_state(), random.getstate()
        if per:
            idx, w = tree.sample(len(obuf), B)
        else:
            idx, w = torch.randint(len(rbuf), (B,)).numpy(), None
        eps = [torch.randn(B, A) for _ in range(4)] if (filt and not disc) else None
        random.choice(list(range(E)))              # random.choice(agent.actors) for the grad-norm log
        pm = random.choice(range(E)) if per else None
        peps = [torch.randn(B, A) for _ in range(4)] if (per and not disc) else None
        torch.set_rng_state(st); np.random.set_state(nst); random.setstate(pst)

        advs.clear()
        rlogs = rl.offline_actor_update(
            buffer=rbuf, agent=ra, actor_optimizer=r_aopt, encoder_optimizer=r_eopt, batch_size=B,
            actor_clip=cfg["clip"], update_encoder=False, encoder_clip=cfg["clip"], augmenter=r_aug,
            actor_lambda=0.0, aug_mix=0.0, premade_replay_dicts=None, per=per, discrete=disc, filter_=filt)
        ologs, ord_, oprio, _ = orc.offline_actor_update(
            obuf, tree if per else None, oa, o_aopt, B, cfg["clip"], o_aug, 0.0, per=per, filter_=filt,
            idx_list=[idx], eps_lists=[eps] if (filt and not disc) else None, prio_member=pm, prio_eps=peps)
        rec[f"s{k}_per"], rec[f"s{k}_filter"] = np.int64(per), np.int64(filt)
        rec[f"s{k}_idx"] = np.asarray(idx, np.int64)
        if per:
            rec[f"s{k}_weights"] = np.asarray(w, np.float64)
            if not disc:
                rec[f"s{k}_prio_eps"] = torch.stack(peps).numpy()
            rec[f"s{k}_prio"] = np.asarray(oprio, np.float64)
            # the reference's trees after its own adjust_priorities
            leaves = np.array([rbuf._it_sum[int(j)] for j in idx], np.float64)
            rprio = (torch.relu(advs[-1]) + 1e-4).squeeze(1).numpy()
            print("   prio adv diff", float(np.abs(rprio - oprio).max()), "filter adv n", len(advs))
            assert np.allclose(leaves, tree.sum[tree.cap + idx], rtol=1e-4, atol=1e-6), "priority update mismatch"
            rec[f"s{k}_leaves"] = leaves
        if filt:
            if not disc:
                rec[f"s{k}_eps"] = torch.stack(eps).numpy()
            rec[f"s{k}_adv"] = advs[0].numpy()
        for key, val in rlogs.items():
            v = float(val)
            rec[f"s{k}_log:{key}"] = np.float64(v)
            assert abs(v - float(ologs[key])) <= 2e-4 * max(1.0, abs(v)), (key, v, ologs[key])
    type(ra.adv_estimator).forward = orig_fwd
    rcc, rac = ref_params(ra, cfg)
    if has_critic:
        assert maxdiff(rcc, oa.critic_params()) < 5e-5
        rec["final_critic"] = np.concatenate([p.detach().numpy().ravel() for p in rcc])
    dpar = maxdiff(rac, oa.actor_params())
    print(f"   actor params max|diff| {dpar:.3e}; max priority {rbuf._max_priority:.4f} / {tree.max_priority:.4f}")
    assert dpar < 5e-5 and abs(rbuf._max_priority - tree.max_priority) < 1e-6
    rec["final_actor"] = np.concatenate([p.detach().numpy().ravel() for p in rac])
    rec["final_max_priority"] = np.float64(rbuf._max_priority)
    rec["final_tree_total"] = np.float64(rbuf._it_sum.sum())
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **rec)


def run_markov_case(name, cfg):
    """learning.markov_state_abstraction_update on the unmodified reference (main.py:218-224 builds its optimizer over
    chain(encoder, inverse_model, contrastive_model)); the oracle follows on the recorded draws."""
    print(f"== {name}")
    torch.manual_seed(cfg["seed"]); np.random.seed(cfg["seed"]); random.seed(cfg["seed"])
    B, A, mk, px = cfg["B"], cfg["act"], cfg["markov"], cfg.get("pixels")
    disc = bool(cfg["discrete"])
    if px:
        s, a, r, s1, d = synth.synth_pixel_transitions(cfg["rows"], px["channels"], px["hw"],
                                                      n_actions=A if disc else None, act_dim=A, seed=cfg["seed"] + 100)
    else:
        s, a, r, s1, d = synth.synth_transitions(cfg["rows"], cfg["obs"], A, disc, seed=cfg["seed"] + 100, n_actions=A)
    rbuf = ref.replay.ReplayBuffer(cfg["cap"])
    rbuf.load_experience(s, a, r, s1, d)
    obuf = orc.ReplayOracle(cfg["cap"])
    obuf.load_experience(s, a, r, s1, d)
    ra, oa = build_pair(cfg)
    oa.requires_grad_(True)
    emb = px["emb"] if px else cfg["obs"]
    inv_p, con_p = synth_markov_models(cfg)
    load_mlp(ra.inverse_model, inv_p, ("fc1", "fc2", "act_p" if disc else "fc3"))
    load_mlp(ra.contrastive_model, con_p, ("fc1", "fc2", "out"))
    for t in list(inv_p.values()) + list(con_p.values()):
        t.requires_grad_(True)
    r_opt = torch.optim.Adam(chain(ra.encoder.parameters(), ra.inverse_model.parameters(),
                                   ra.contrastive_model.parameters()), lr=cfg["lr"], weight_decay=0, betas=(0.9, 0.999))
    o_opt = orc.AdamOracle(oa.encoder_params() + [inv_p[k] for k in orc.MLP_KEYS] + [con_p[k] for k in orc.MLP_KEYS],
                           lr=cfg["lr"])
    if px and px["aug"] == "drqv2":
        r_aug = ref.augmentations.AugmentationSequence([ref.augmentations.Drqv2Aug(B)])
        o_aug = orc.AugOracle("drqv2", B)
    else:
        r_aug = ref.augmentations.AugmentationSequence([ref.augmentations.IdentityAug(B)])
        o_aug = orc.AugOracle("identity", B)
    aug_mix = px["aug_mix"] if px else 0.0
    ic, cc, sc = mk["coeffs"]
    rec = {"n_steps": np.int64(mk["steps"])}
    for k in range(mk["steps"]):
        clip = mk["clip"][k]
        st = torch.get_rng_state()
        idx = torch.randint(len(rbuf), (B,)).numpy()
        shift = orc.drqv2_draw_shift(B) if (px and px["aug"] == "drqv2") else None
        perm = torch.randperm(B)
        torch.set_rng_state(st)
        rlogs = rl.markov_state_abstraction_update(
            buffer=rbuf, agent=ra, optimizer=r_opt, batch_size=B, augmenter=r_aug, aug_mix=aug_mix, discrete=disc,
            inverse_coeff=ic, contrastive_coeff=cc, smoothness_coeff=sc, smoothness_max_dist=mk["max_dist"],
            grad_clip=clip)
        if shift is not None:
            assert torch.equal(r_aug.aug_list[0].shift, shift), "shift stream mismatch"
            o_aug.forced = [shift.clone()]
        ologs, _ = orc.markov_state_abstraction_update(
            obuf, oa, inv_p, con_p, o_opt, B, o_aug, aug_mix, ic, cc, sc, mk["max_dist"], clip, idx=idx, perm=perm,
            inv_lo=ra.inverse_model.log_std_low if not disc else -10.0,
            inv_hi=ra.inverse_model.log_std_high if not disc else 2.0)
        rec[f"m{k}_idx"], rec[f"m{k}_perm"] = np.asarray(idx, np.int64), perm.numpy().astype(np.int64)
        if shift is not None:
            rec[f"m{k}_shift"] = shift.numpy()
        for key, val in rlogs.items():
            v = float(val)
            rec[f"m{k}_log:{key}"] = np.float64(v)
            assert abs(v - float(ologs[key])) <= 2e-4 * max(1.0, abs(v)), (key, v, ologs[key])
    r_inv = [p for p in ra.inverse_model.parameters()]
    r_con = [p for p in ra.contrastive_model.parameters()]
    dpar = max(maxdiff(r_inv, [inv_p[k] for k in orc.MLP_KEYS]), maxdiff(r_con, [con_p[k] for k in orc.MLP_KEYS]))
    print(f"   inverse / contrastive params max|diff| {dpar:.3e}")
    assert dpar < 5e-5
    rec["final_inverse"] = np.concatenate([p.detach().numpy().ravel() for p in r_inv])
    rec["final_contrastive"] = np.concatenate([p.detach().numpy().ravel() for p in r_con])
    if px:
        re_ = ref_encoder_params(ra.encoder, cfg)
        assert_encoder_close(re_, oa.encoder_params())
        vals = []
        for p in re_:
            flat = p.detach().numpy().ravel()
            vals.append(flat[synth.fingerprint_indices(flat.size)])
        rec["finalfp_encoder"] = np.concatenate(vals)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **rec)


def run_bc_pixels_case(name, cfg):
    """the BC warm-up of main.py:292-312 on the unmodified reference: learning.offline_actor_update(update_encoder=True,
    filter_=False, per=False) with a pixel encoder trained through the BC loss."""
    print(f"== {name}")
    torch.manual_seed(cfg["seed"]); np.random.seed(cfg["seed"]); random.seed(cfg["seed"])
    B, A, px, disc = cfg["B"], cfg["act"], cfg["pixels"], bool(cfg["discrete"])
    s, a, r, s1, d = synth.synth_pixel_transitions(cfg["rows"], px["channels"], px["hw"], n_actions=A if disc else None,
                                                  act_dim=A, seed=cfg["seed"] + 100)
    rbuf = ref.replay.ReplayBuffer(cfg["cap"])
    rbuf.load_experience(s, a, r, s1, d)
    obuf = orc.ReplayOracle(cfg["cap"])
    obuf.load_experience(s, a, r, s1, d)
    ra, oa = build_pair(cfg)
    oa.requires_grad_(True)
    r_aopt = torch.optim.Adam(chain(*(ac.parameters() for ac in ra.actors)), lr=cfg["lr"], betas=(0.9, 0.999))
    r_eopt = torch.optim.Adam(ra.encoder.parameters(), lr=px["enc_lr"], betas=(0.9, 0.999))
    o_aopt = orc.AdamOracle(oa.actor_params(), lr=cfg["lr"])
    o_eopt = orc.AdamOracle(oa.encoder_params(), lr=px["enc_lr"])
    r_aug = ref.augmentations.AugmentationSequence([ref.augmentations.Drqv2Aug(B)])
    o_aug = orc.AugOracle("drqv2", B)
    rec = {"n_steps": np.int64(len(cfg["steps"]))}
    for k in range(len(cfg["steps"])):
        st, pst = torch.get_rng_state(), random.getstate()
        E = cfg["E"]
        idxs, shifts = [], []
        for _ in range(E):   # per member: index draw, then the augmenter's shift (learning_utils.py:174-201)
            idxs.append(torch.randint(len(rbuf), (B,)).numpy())
            shifts.append(orc.drqv2_draw_shift(B))
        idx, shift = idxs[0], shifts[-1]
        gpick = random.choice(range(E))   # random.choice(agent.actors), learning.py:210-212
        torch.set_rng_state(st); random.setstate(pst)
        rlogs = rl.offline_actor_update(
            buffer=rbuf, agent=ra, actor_optimizer=r_aopt, encoder_optimizer=r_eopt, batch_size=B,
            actor_clip=cfg["clip"], update_encoder=True, encoder_clip=cfg["enc_clip"][k], augmenter=r_aug,
            actor_lambda=0.0, aug_mix=px["aug_mix"], premade_replay_dicts=None, per=False, discrete=disc,
            filter_=False)
        assert torch.equal(r_aug.aug_list[0].shift, shift), "shift stream mismatch"
        o_aug.forced = [sh.clone() for sh in shifts]
        ologs, _, _, _ = orc.offline_actor_update(
            obuf, None, oa, o_aopt, B, cfg["clip"], o_aug, px["aug_mix"], per=False, filter_=False, idx_list=idxs,
            update_encoder=True, encoder_opt=o_eopt, encoder_clip=cfg["enc_clip"][k], grad_pick=gpick)
        if E > 1:
            rec[f"s{k}_gpick"] = np.int64(gpick)
        if E == 1:
            rec[f"s{k}_idx"], rec[f"s{k}_shift"] = np.asarray(idx, np.int64), shift.numpy()
        else:   # (E, B) / (E, B, 1, 1, 2)
            rec[f"s{k}_idx"] = np.stack([np.asarray(v, np.int64) for v in idxs])
            rec[f"s{k}_shift"] = np.stack([sh.numpy() for sh in shifts])
        for key, val in rlogs.items():
            v = float(val)
            rec[f"s{k}_log:{key}"] = np.float64(v)
            assert abs(v - float(ologs[key])) <= 2e-4 * max(1.0, abs(v)), (key, v, ologs[key])
    _, rac = ref_params(ra, cfg)
    dpar = maxdiff(rac, oa.actor_params())
    print(f"   actor params max|diff| {dpar:.3e}")
    assert dpar < 5e-5
    rec["final_actor"] = np.concatenate([p.detach().numpy().ravel() for p in rac])
    re_ = ref_encoder_params(ra.encoder, cfg)
    assert_encoder_close(re_, oa.encoder_params())
    vals = []
    for p in re_:
        flat = p.detach().numpy().ravel()
        vals.append(flat[synth.fingerprint_indices(flat.size)])
    rec["finalfp_encoder"] = np.concatenate(vals)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **rec)


def run_actor_inv_case(name, cfg):
    """learning.offline_actor_update(actor_lambda > 0, filter_=False, per=False) on the unmodified reference: the BC
    loss plus the action invariance constraint (learning_utils.py:272-285), with and without update_encoder."""
    print(f"== {name}")
    torch.manual_seed(cfg["seed"]); np.random.seed(cfg["seed"]); random.seed(cfg["seed"])
    B, A, E, px, disc = cfg["B"], cfg["act"], cfg["E"], cfg.get("pixels"), bool(cfg["discrete"])
    if px:
        s, a, r, s1, d = synth.synth_pixel_transitions(cfg["rows"], px["channels"], px["hw"], n_actions=A if disc else None,
                                                      act_dim=A, seed=cfg["seed"] + 100)
    else:
        s, a, r, s1, d = synth.synth_transitions(cfg["rows"], cfg["obs"], A, disc, seed=cfg["seed"] + 100, n_actions=A)
    rbuf = ref.replay.ReplayBuffer(cfg["cap"])
    rbuf.load_experience(s, a, r, s1, d)
    obuf = orc.ReplayOracle(cfg["cap"])
    obuf.load_experience(s, a, r, s1, d)
    ra, oa = build_pair(cfg)
    oa.requires_grad_(True)
    enc_lr = px["enc_lr"] if px else 1e-4
    r_aopt = torch.optim.Adam(chain(*(ac.parameters() for ac in ra.actors)), lr=cfg["lr"], betas=(0.9, 0.999))
    r_eopt = torch.optim.Adam(ra.encoder.parameters(), lr=enc_lr, betas=(0.9, 0.999))
    o_aopt = orc.AdamOracle(oa.actor_params(), lr=cfg["lr"])
    o_eopt = orc.AdamOracle(oa.encoder_params(), lr=enc_lr)
    if px:
        r_aug = ref.augmentations.AugmentationSequence([ref.augmentations.Drqv2Aug(B)])
        o_aug = orc.AugOracle("drqv2", B)
    else:
        r_aug = ref.augmentations.AugmentationSequence([ref.augmentations.IdentityAug(B)])
        o_aug = orc.AugOracle("identity", B)
    aug_mix = px["aug_mix"] if px else 0.0
    lam = cfg["actor_lambda"]
    rec = {"n_steps": np.int64(len(cfg["steps"]))}
    for k, stp in enumerate(cfg["steps"]):
        st, pst = torch.get_rng_state(), random.getstate()
        idxs, shifts, epss, cats = [], [], [], []
        for i in range(E):
            idx = torch.randint(len(rbuf), (B,)).numpy()
            idxs.append(idx)
            if px:
                shifts.append(orc.drqv2_draw_shift(B))
            if disc:
                with torch.no_grad():   # o_dist.sample(): the reference's own draw at this point of the stream
                    oo = {"obs": torch.from_numpy(s["obs"][idx]).float()}
                    cats.append(ra.actors[i](ra.encoder(oo)).sample())
            elif cfg["actor"] != "deterministic":   # (ContinuousDeterministic.sample() draws nothing)
                epss.append(torch.randn(B, A))
        gpick = random.choice(range(E))   # random.choice(agent.actors), learning.py:210-212
        torch.set_rng_state(st); random.setstate(pst)
        rlogs = rl.offline_actor_update(
            buffer=rbuf, agent=ra, actor_optimizer=r_aopt, encoder_optimizer=r_eopt, batch_size=B,
            actor_clip=stp["clip"], update_encoder=stp["update_encoder"], encoder_clip=stp.get("enc_clip"),
            augmenter=r_aug, actor_lambda=lam, aug_mix=aug_mix, premade_replay_dicts=None, per=False, discrete=disc,
            filter_=False)
        if px:
            o_aug.forced = [sh.clone() for sh in shifts]
        ologs, _, _, _ = orc.offline_actor_update(
            obuf, None, oa, o_aopt, B, stp["clip"], o_aug, aug_mix, per=False, filter_=False, idx_list=idxs,
            update_encoder=stp["update_encoder"], encoder_opt=o_eopt, encoder_clip=stp.get("enc_clip"),
            actor_lambda=lam, inv_eps_list=epss or None, inv_cat_list=cats or None, grad_pick=gpick)
        rec[f"s{k}_gpick"] = np.int64(gpick)
        for i in range(E):
            rec[f"s{k}_idx{i}"] = np.asarray(idxs[i], np.int64)
            if px:
                rec[f"s{k}_shift{i}"] = shifts[i].numpy()
            if disc:
                rec[f"s{k}_cat{i}"] = cats[i].numpy().astype(np.int64)
            elif epss:
                rec[f"s{k}_eps{i}"] = epss[i].numpy()
        for key, val in rlogs.items():
            v = float(val)
            rec[f"s{k}_log:{key}"] = np.float64(v)
            assert abs(v - float(ologs[key])) <= 2e-4 * max(1.0, abs(v)), (key, v, ologs[key])
    _, rac = ref_params(ra, cfg)
    dpar = maxdiff(rac, oa.actor_params())
    print(f"   actor params max|diff| {dpar:.3e}")
    assert dpar < 5e-5
    rec["final_actor"] = np.concatenate([p.detach().numpy().ravel() for p in rac])
    if px:
        re_ = ref_encoder_params(ra.encoder, cfg)
        assert_encoder_close(re_, oa.encoder_params())
        vals = []
        for p in re_:
            flat = p.detach().numpy().ravel()
            vals.append(flat[synth.fingerprint_indices(flat.size)])
        rec["finalfp_encoder"] = np.concatenate(vals)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **rec)


def synth_markov_models(cfg):
    """seeded inverse / contrastive model weights of a Markov case (the same draw everywhere: generator, tests)"""
    px = cfg.get("pixels")
    emb = px["emb"] if px else cfg["obs"]
    rng_ = np.random.RandomState(cfg["seed"] + 11)
    inv_out = cfg["act"] if cfg["discrete"] else 2 * cfg["act"]
    return orc.make_mlp(rng_, 2 * emb, cfg["hidden"], inv_out), orc.make_mlp(rng_, 2 * emb, cfg["hidden"], 1)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    units = {"indices": gen_indices, "popart": gen_popart, "aug": gen_aug, "nets": gen_nets,
             "per": gen_per}
    for nm, fn in units.items():
        if not only or nm in only:
            fn()
    for name, cfg in synth.CASES.items():
        if not only or name in only:
            run_case(name, cfg)
    for name, cfg in synth.AFBC_CASES.items():
        if not only or name in only:
            run_afbc_case(name, cfg)
    for name, cfg in synth.MARKOV_CASES.items():
        if not only or name in only:
            run_markov_case(name, cfg)
    for name, cfg in synth.BC_PIXEL_CASES.items():
        if not only or name in only:
            run_bc_pixels_case(name, cfg)
    for name, cfg in synth.ACTOR_INV_CASES.items():
        if not only or name in only:
            run_actor_inv_case(name, cfg)
    print("golden fixtures written to", OUT)
