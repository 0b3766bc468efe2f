"""CPU: the oracle (oracle/ssac_oracle.py) against the reference's outputs stored in tests/golden.

The fixtures were produced by oracle/gen_golden.py, which imports the unmodified reference in the
development container; this file needs neither the reference nor a GPU.
"""
import numpy as np
import pytest
import torch

import case_runner
import ssac_oracle as orc
import synth


def fx(name):
    return case_runner.load_fixture(name)


# ---------------------------------------------------------------- a2: replay index stream (bit exact)
@pytest.mark.parametrize("seed", [0, 7, 123])
@pytest.mark.parametrize("n", [1000, 100_000, 1_000_000])
def test_index_stream_bit_exact(seed, n):
    ref = fx("replay_indices")[f"s{seed}_n{n}"]
    mine = orc.randint_from_raw32(orc.MT19937(seed).raw32(3 * 512), 0, n).reshape(3, 512)
    assert np.array_equal(ref, mine)
    # and torch's own generator still behaves like the fixture (guards a torch upgrade)
    torch.manual_seed(seed)
    again = np.stack([torch.randint(n, (512,)).numpy() for _ in range(3)])
    assert np.array_equal(ref, again)


def test_ring_storage_wraparound_gather():
    f = fx("replay_indices")
    s, a, r, s1, d = synth.synth_transitions(700, 5, 2, seed=3)
    ob = orc.ReplayOracle(512)
    for lo in range(0, 700, 100):
        sl = slice(lo, lo + 100)
        ob.push({"obs": s["obs"][sl]}, a[sl], r[sl, None], {"obs": s1["obs"][sl]}, d[sl, None])
    assert len(ob) == 512
    o, act, rew, o1, done = ob.gather(f["wrap_idx"])
    assert np.array_equal(o["obs"].numpy(), f["wrap_obs"])
    assert np.array_equal(o1["obs"].numpy(), f["wrap_next_obs"])
    assert np.array_equal(act.numpy(), f["wrap_act"])
    assert np.array_equal(rew.numpy(), f["wrap_rew"])
    assert np.array_equal(done.numpy(), f["wrap_done"].astype(np.float32))


def test_ragged_and_empty_edges():
    ob = orc.ReplayOracle(8)
    assert len(ob) == 0
    ob.push({"obs": np.zeros((3, 2), np.float32)}, np.zeros((3, 1), np.float32), np.zeros((3, 1)),
            {"obs": np.ones((3, 2), np.float32)}, np.zeros((3, 1)))
    assert len(ob) == 3
    with pytest.raises(AssertionError):
        orc.sample_move_and_augment(ob, 4, orc.AugOracle("identity", 4), 0.0)  # len < batch (lu:175)
    # single un-batched transition
    ob.push({"obs": np.zeros(2, np.float32)}, np.zeros(1, np.float32), 1.0, {"obs": np.ones(2, np.float32)}, False)
    assert len(ob) == 4


# ---------------------------------------------------------------- a3: PER sample path
def test_per_sample_and_priority_update():
    f = fx("per")
    per = orc.PerOracle(400, 0.6, 1.0)
    per.push_rows(np.arange(300))
    np.random.seed(int(f["np_seed"]))
    i0, w0 = per.sample(300, 32)
    assert np.array_equal(i0, f["i0"]) and np.allclose(w0, f["w0"], rtol=1e-12)
    per.update(i0, f["prios"])
    i1, w1 = per.sample(300, 32)
    assert np.array_equal(i1, f["i1"]) and np.allclose(w1, f["w1"], rtol=1e-12)


# ---------------------------------------------------------------- a12: PopArt
def test_popart_sequence():
    f = fx("popart")
    op = orc.PopArtOracle(beta=1e-2, min_steps=3)
    x = torch.linspace(-2, 2, 9).unsqueeze(1)
    for t in range(10):
        op.update_stats(torch.from_numpy(f[f"v{t}"]))
        s = op.state()
        ref = f["states"][t]
        assert np.allclose([s["mu"], s["nu"], s["w"], s["b"], s["sigma"]], ref[:5], rtol=1e-5, atol=1e-6)
        assert s["t"] == int(ref[5]) and float(op.stable) == ref[6]
        out = torch.cat([op(x), op(x, normalized=False), op.normalize(x)], 1).numpy()
        assert np.allclose(out, f["outs"][t], rtol=1e-5, atol=1e-5)
    assert f["states"][:, 6].max() == 1.0, "fixture must exercise the stable (rescale) branch"


# ---------------------------------------------------------------- a5: augmentations
def test_drqv2_shift_exact():
    f = fx("augmentations")
    x0, x1 = torch.from_numpy(f["x0"]), torch.from_numpy(f["x1"])
    shift = torch.from_numpy(f["v2_shift"])
    assert float((orc.drqv2_shift(x0, shift) - torch.from_numpy(f["v2_y0"])).abs().max()) <= 1e-4
    assert float((orc.drqv2_shift(x1, shift) - torch.from_numpy(f["v2_y1"])).abs().max()) <= 1e-4


def test_drqv2_shift_84():
    f = fx("augmentations")
    x = torch.from_numpy(f["big_x"])
    y = orc.drqv2_shift(x, torch.from_numpy(f["big_shift"]))
    assert float((y - torch.from_numpy(f["big_y"])).abs().max()) <= 1e-3  # 0-255 scale


def test_drq_crop_exact_and_noise():
    f = fx("augmentations")
    x0 = torch.from_numpy(f["x0"])
    y = orc.drq_crop(x0, f["v1_w1"], f["v1_h1"])
    assert np.array_equal(y.numpy(), f["v1_y0"])
    yn = orc.drq_crop(x0, f["v1n_w1"], f["v1n_h1"], noise=torch.from_numpy(f["v1n_noise0"]))
    assert np.allclose(yn.numpy(), f["v1n_y0"], atol=1e-5)


def test_drq_draw_order_matches_reference_stream():
    f = fx("augmentations")
    torch.manual_seed(21)
    aug = orc.AugOracle("drqv2", 6)
    aug({"obs": torch.from_numpy(f["x0"])}, {"obs": torch.from_numpy(f["x1"])})
    assert np.array_equal(aug.last.numpy(), f["v2_shift"])
    torch.manual_seed(22)
    aug = orc.AugOracle("drq_nonoise", 6)
    aug({"obs": torch.from_numpy(f["x0"])})
    assert np.array_equal(aug.last[0].numpy(), f["v1_w1"]) and np.array_equal(aug.last[1].numpy(), f["v1_h1"])


# ---------------------------------------------------------------- a6-a9: networks
def test_ensemble_q_known_answers():
    f = fx("nets")
    rng = np.random.RandomState(int(f["ensq_seed"]))
    crit = [orc.make_mlp(rng, 23, 256, 1) for _ in range(10)]
    s = torch.from_numpy(rng.standard_normal((512, 17)).astype(np.float32))
    a = torch.from_numpy(rng.uniform(-1, 1, (512, 6)).astype(np.float32))
    q = torch.stack([orc.critic_q(p, s, a) for p in crit], 0).squeeze(-1)
    assert np.allclose(q.numpy(), f["ensq_q"], atol=2e-5)
    # REDQ min over a subset == elementwise min of the stored rows
    mn = orc.ensemble_q(crit, s, a, subset_ids=[3, 7]).squeeze(-1)
    assert np.allclose(mn.numpy(), np.minimum(f["ensq_q"][3], f["ensq_q"][7]), atol=2e-5)


@pytest.mark.parametrize("tag,lo,hi", [("redq", -5.0, 2.0), ("default", -10.0, 2.0)])
def test_tanh_normal(tag, lo, hi):
    f = fx("nets")
    a, lp = orc.tanh_normal_sample(torch.from_numpy(f[f"tn_{tag}_out"]), lo, hi,
                                   torch.from_numpy(f[f"tn_{tag}_eps"]))
    assert np.allclose(a.numpy(), f[f"tn_{tag}_a"], atol=1e-6)
    assert np.allclose(lp.numpy(), f[f"tn_{tag}_logp"], atol=2e-4)


@pytest.mark.parametrize("kind,ch,emb", [("big", 9, 50), ("small", 4, 128)])
def test_pixel_encoders(kind, ch, emb):
    f = fx("nets")
    p = orc.make_conv_encoder(np.random.RandomState(40 + ch), kind, ch, emb)
    x = torch.from_numpy(np.random.RandomState(50 + ch).randint(0, 256, (3, ch, 84, 84)).astype(np.float32))
    y = orc.encode({"kind": kind, "key": "obs", "p": p}, {"obs": x})
    assert np.allclose(y.numpy(), f[f"enc_{kind}_y"], atol=2e-5)


# ---------------------------------------------------------------- a10-a17: multi-step update cases
@pytest.mark.parametrize("name", list(synth.CASES))
def test_update_cases_match_reference(name):
    rec = case_runner.run_oracle(name)
    cfg = synth.CASES[name]
    # (the full-size pixel cases: counted sign-flip stragglers in the hidden-1024 / B 1024 MLPs too, see compare())
    step = 2.2 * cfg["lr"] * cfg["cycles"] * cfg["utd"] if name in synth.FULL_SIZE else 0.0
    worst = case_runner.compare(rec, case_runner.load_fixture(name), who=f"oracle[{name}]", mlp_max_step=step)
    assert worst["param"] < 3e-5


def test_product_priority_sampler_matches_reference_fixture():
    """super_sac_amd.replay.PrioritySampler (host trees of the product) against the reference's draws."""
    from super_sac_amd.replay import PrioritySampler
    f = fx("per")
    ps = PrioritySampler(400, 0.6, 1.0)
    ps.push_rows(np.arange(300))
    np.random.seed(int(f["np_seed"]))
    i0, w0 = ps.sample(300, 32)
    assert np.array_equal(i0, f["i0"]) and np.allclose(w0, f["w0"], rtol=1e-12)
    ps.update_priorities(i0, f["prios"], 300)
    i1, w1 = ps.sample(300, 32)
    assert np.array_equal(i1, f["i1"]) and np.allclose(w1, f["w1"], rtol=1e-12)
    with pytest.raises(AssertionError):
        ps.update_priorities(np.array([0, 1]), np.array([1.0, 0.0]), 300)   # priorities must be > 0
    with pytest.raises(AssertionError):
        ps.update_priorities(np.array([300]), np.array([1.0]), 300)          # index range (replay.py:187)


@pytest.mark.parametrize("name", sorted(synth.AFBC_CASES))
def test_oracle_reproduces_afbc_fixture(name):
    """offline_actor_update + advantage filter + PER sample / priority refresh: the
    oracle, drawing its own prioritised indices from the seeded numpy stream, lands on the reference's
    recorded indices, weights, priorities, logs and final actor."""
    rec = case_runner.run_afbc_oracle(name)
    case_runner.compare_afbc(rec, case_runner.load_fixture(name), par_tol=1e-6)


@pytest.mark.parametrize("name", sorted(synth.MARKOV_CASES))
def test_oracle_reproduces_markov_fixture(name):
    """markov_state_abstraction_update (learning.py:266-341): inverse model, contrastive model with
    the recorded shuffle, smoothness hinge, joint clip and the single optimizer over encoder + both models -- on vector
    observations (continuous / discrete actions) and through both pixel encoders."""
    rec = case_runner.run_markov_oracle(name)
    case_runner.compare_markov(rec, case_runner.load_fixture(name), f"oracle[{name}]", log_tol=2e-4, gn_tol=2e-4,
                               par_tol=1e-6 if "pixels" not in name else 3e-5,
                               max_step=2.2 * synth.MARKOV_CASES[name]["lr"] * 3 if "pixels" in name else 0.0)


@pytest.mark.parametrize("name", sorted(synth.BC_PIXEL_CASES))
def test_oracle_reproduces_bc_warmup_on_pixels(name):
    """main.py:292-312: offline_actor_update(update_encoder=True, filter_=False) -- the pixel encoder trained through
    the BC loss, its own clip and optimizer."""
    rec = case_runner.run_bc_pixels_oracle(name)
    case_runner.compare_markov(rec, case_runner.load_fixture(name), f"oracle[{name}]", log_tol=2e-4, gn_tol=2e-4,
                               par_tol=3e-5, max_step=2.2 * synth.BC_PIXEL_CASES[name]["lr"] * 3)


@pytest.mark.parametrize("name", sorted(synth.ACTOR_INV_CASES))
def test_oracle_reproduces_action_invariance_fixture(name):
    """offline_actor_update(actor_lambda > 0): the BC loss + the action invariance constraint (learning_utils.py:272-285)
    on vector observations (two members; categorical actor with the reference's summed log-probabilities) and through
    a pixel encoder that is clipped / logged always and stepped only with update_encoder."""
    rec = case_runner.run_actor_inv_oracle(name)
    cfg = synth.ACTOR_INV_CASES[name]
    case_runner.compare_markov(rec, case_runner.load_fixture(name), f"oracle[{name}]", log_tol=2e-4, gn_tol=2e-4,
                               par_tol=1e-6 if "pixels" not in name else 3e-5,
                               max_step=2.2 * cfg["lr"] * len(cfg["steps"]) if "pixels" in name else 0.0)
