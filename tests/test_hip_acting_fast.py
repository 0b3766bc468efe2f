"""GPU: the ONE-CALL acting path (super_sac_amd/acting.py + csrc/ssac_act.hip: a recorded launch list per rule, the
observation written straight into device-visible memory, the action arriving in pinned host memory) against the CPU oracle's
arithmetic (agent.py:204-315 restated) -- the in-kernel Philox noise regenerated through ssac_philox_normal (the stream's
definition) -- and against the general eager path of agent.py on the same noise.  tests/test_hip_acting.py (injected noise ->
the general path) stays as it was."""
import ctypes as C
import random

import numpy as np
import pytest
import torch

import case_runner
import ssac_oracle as orc
import synth
from test_hip_bench_bridge import _philox4x32_10

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pair(name, ucb=0.0):
    cfg = synth.CASES[name]
    agent = case_runner.build_engine_agent(cfg, torch.device(DEV))
    agent.ucb_bonus = ucb
    return cfg, agent, case_runner._oracle_agent(cfg)


def _plan(agent, rule, n, bonus=0.0):
    from super_sac_amd import acting
    return acting._PLANS[agent][(rule, n, bonus)]


def _eps_of_call(agent, plan, member, call_no, n, A):
    """the standard normals actor `member` drew in the plan's call number `call_no` (0-based count of publish steps before it)"""
    from super_sac_amd import _lib
    from super_sac_amd._lib import check, lib
    r = plan.rng_for(agent, member)
    r = _lib.Rng(r.seed, None, r.offset + call_no)
    out = torch.empty(n, A, device=DEV)
    check(lib.ssac_philox_normal(out.data_ptr(), n, A, C.byref(r), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out.cpu()


def _calls(plan):
    from super_sac_amd._lib import lib
    return int(lib.ssac_act_calls(plan.handle))


@pytest.mark.parametrize("n", [1, 5])
def test_fast_forward_and_sample_match_the_oracle(n):
    from super_sac_amd import acting
    cfg, agent, oa = _pair("sunrise")   # 3 members, continuous, hidden 64
    rs = np.random.RandomState(10 + n)
    A = cfg["act"]
    for rep in range(3):   # (the first call records, the others replay)
        obs = rs.standard_normal((n, cfg["obs"]) if n > 1 else (cfg["obs"],)).astype(np.float32)
        s = torch.from_numpy(obs.reshape(n, -1))
        act = agent.forward({"obs": obs}, num_envs=n)
        want = torch.stack([torch.tanh(orc.mlp3(a, s)[0][:, :A]) for a in oa.actors], 0).mean(0).clamp(-1, 1).numpy()
        assert act.dtype == np.float32 and act.shape == ((n, A) if n > 1 else (A,))
        np.testing.assert_allclose(act.reshape(n, A), want, atol=2e-6)
        random.seed(100 + rep)
        k = random.choice(range(cfg["E"]))   # (the draw agent.sample_action is about to make: random.choice(self.actors))
        random.seed(100 + rep)
        act, dist = agent.sample_action({"obs": obs}, num_envs=n, return_dist=True)
        plan = _plan(agent, "sample", n)
        # the draw number of a call = the publish steps before it (recording passes included) = the count behind it - 1
        eps = _eps_of_call(agent, plan, k, _calls(plan) - 1, n, A)
        out = orc.mlp3(oa.actors[k], s)[0]
        want = orc.tanh_normal_sample(out, oa.lo, oa.hi, eps)[0].clamp(-1, 1).numpy()
        np.testing.assert_allclose(act.reshape(n, A), want, atol=3e-6)
        np.testing.assert_allclose(dist.cpu().numpy(), out.numpy(), atol=3e-6)   # the chosen actor's head output
    assert ("forward", n, 0.0) in acting._PLANS[agent] and ("sample", n, 0.0) in acting._PLANS[agent]


def test_fast_ucb_picks_the_argmax_candidate_and_equals_the_general_path():
    import super_sac_amd as ssa
    from super_sac_amd import acting
    cfg, agent, oa = _pair("sunrise", ucb=0.7)
    rs = np.random.RandomState(3)
    n, E, A = 6, cfg["E"], cfg["act"]
    for rep in range(3):
        obs = rs.standard_normal((n, cfg["obs"])).astype(np.float32)
        act = agent.sample_action({"obs": obs}, num_envs=n)
        plan = _plan(agent, "ucb", n, 0.7)
        call_no = _calls(plan) - 1
        eps = [_eps_of_call(agent, plan, a, call_no, n, A) for a in range(E)]
        s = torch.from_numpy(obs)
        cands = torch.stack([orc.tanh_normal_sample(orc.mlp3(oa.actors[a], s)[0], oa.lo, oa.hi, eps[a])[0] for a in range(E)], 0)
        q = torch.stack([torch.stack([orc.ensemble_q(oa.critics[c], s, cands[a]).squeeze(-1) for a in range(E)], 0)
                         for c in range(E)], 0)                         # (members, candidates, envs)
        ucb = q.mean(0) + 0.7 * q.std(0)
        want = cands[ucb.argmax(0), torch.arange(n)].clamp(-1, 1)
        top2 = ucb.topk(2, dim=0).values
        clear = ((top2[0] - top2[1]) > 1e-4).numpy()   # (a near-tie may flip under fp32 reordering)
        assert clear.sum() >= n - 1
        np.testing.assert_allclose(act[clear], want.numpy()[clear], atol=3e-6)
        # the general path of agent.py on the same noise (the hook switches the fast path off)
        queue = [e.clone() for e in eps]
        saved = ssa.rng.draw_normal
        ssa.rng.draw_normal = lambda shape, device: queue.pop(0).to(device)
        try:
            general = agent.sample_action({"obs": obs}, num_envs=n)
        finally:
            ssa.rng.draw_normal = saved
        assert not queue and _calls(plan) == call_no + 1, "the hooked call must have taken the general path"
        np.testing.assert_allclose(act[clear], general[clear], atol=3e-6)


def test_fast_discrete_rules():
    """greedy = arg-max of the mean probabilities; sample = inversion of the cumulative distribution with the Philox uniform
    of (row, draw): recomputed here in numpy, index for index, and checked against the softmax frequencies"""
    torch.manual_seed(1234)   # (the acting stream's seed is drawn from torch's device generator: a fixed run)
    cfg, agent, oa = _pair("sac_discrete")
    A, n = cfg["act"], 16
    rs = np.random.RandomState(5)
    obs = rs.standard_normal((n, cfg["obs"])).astype(np.float32)
    s = torch.from_numpy(obs)
    probs = torch.softmax(orc.mlp3(oa.actors[0], s)[0], -1).numpy().astype(np.float64)
    act = agent.forward({"obs": obs}, num_envs=n)
    assert act.shape == (n, 1) and act.dtype == np.int64 and np.array_equal(act[:, 0], probs.argmax(-1))
    one = agent.forward({"obs": obs[0]}, num_envs=1)
    assert one.shape == (1,) and one[0] == probs[0].argmax()
    counts = np.zeros((n, A))
    mismatches = 0
    calls = 400
    for it in range(calls):
        a = agent.sample_action({"obs": obs}, num_envs=n)
        assert a.shape == (n, 1) and a.dtype == np.int64 and a.min() >= 0 and a.max() < A
        plan = _plan(agent, "sample", n)
        draw = plan.rng_for(agent, 0).offset + _calls(plan) - 1
        seed = plan.rng_for(agent, 0).seed
        cnt = np.stack([np.arange(n), np.zeros(n, np.int64), np.full(n, draw & 0xFFFFFFFF), np.full(n, draw >> 32)], 1)
        u = _philox4x32_10(cnt, (seed & 0xFFFFFFFF, seed >> 32))[:, 0].astype(np.float32) * np.float32(2.3283064365386963e-10)
        want = np.array([min(int(np.searchsorted(np.cumsum(probs[b]), float(u[b]) * probs[b].sum(), side="right")), A - 1)
                         for b in range(n)])
        mismatches += int(np.sum(want != a[:, 0]))
        counts[np.arange(n), a[:, 0]] += 1
    assert mismatches <= 3, f"{mismatches} of {calls * n} draws differ from the numpy restatement (rounding at a boundary only)"
    # (400 draws per row: sigma of a p = 0.5 frequency is 0.025; the index-for-index check above is the sharp one, this one
    #  only says the draws are not degenerate -- 6 sigma, so that 64 cells never trip it by chance)
    assert np.max(np.abs(counts / calls - probs)) < 0.15


def test_fast_path_reads_the_weights_of_the_moment_and_survives_a_reload(tmp_path):
    """the recorded launches point INTO the arenas: an optimizer step / load_state_dict (in place) is seen by the next call;
    parameters that moved (a re-packed arena) invalidate the plan"""
    cfg, agent, oa = _pair("redq_small")
    obs = np.random.RandomState(1).standard_normal(cfg["obs"]).astype(np.float32)
    a0 = agent.forward({"obs": obs})
    with torch.no_grad():
        agent.actors[0].fc3.bias.add_(0.25)
    a1 = agent.forward({"obs": obs})
    with torch.no_grad():
        oa.actors[0]["b3"].add_(0.25)
    want = torch.tanh(orc.mlp3(oa.actors[0], torch.from_numpy(obs)[None])[0][:, :cfg["act"]])[0].numpy()
    assert not np.allclose(a0, a1) and np.allclose(a1, want, atol=2e-6)
    agent.save(str(tmp_path))
    with torch.no_grad():
        agent.actors[0].fc3.bias.sub_(0.5)
    assert not np.allclose(agent.forward({"obs": obs}), a1)
    agent.load(str(tmp_path))
    np.testing.assert_allclose(agent.forward({"obs": obs}), a1, atol=1e-7)


def test_pixel_agents_take_the_one_call_path_and_equal_the_general_path():
    """a uint8 image observation of a pixel-encoder agent: the encoder's forward is recorded in front of the rule's launches
    (the cast and the /255 normalisation happen in the first layer's patch gather, straight from the observation buffer);
    against agent.py's general path on the same frames: the logits (return_dist) and the greedy actions"""
    from super_sac_amd import acting
    cfg, agent, _ = _pair("atari_pixels")
    rs = np.random.RandomState(2)
    for n in (1, 3):
        for rep in range(2):
            obs = rs.randint(0, 256, (n, 4, 84, 84) if n > 1 else (4, 84, 84)).astype(np.uint8)
            act, logits = agent.sample_action({"obs": obs}, num_envs=n, return_dist=True)
            greedy = agent.forward({"obs": obs}, num_envs=n)
            assert ("sample", n, 0.0) in acting._PLANS[agent] and ("forward", n, 0.0) in acting._PLANS[agent]
            assert act.shape == ((n, 1) if n > 1 else (1,)) and act.dtype == np.int64
            acting.ENABLED = False
            try:
                _, want_logits = agent.sample_action({"obs": obs}, num_envs=n, return_dist=True)
                want_greedy = agent.forward({"obs": obs}, num_envs=n)
            finally:
                acting.ENABLED = True
            np.testing.assert_allclose(logits.cpu().numpy(), want_logits.cpu().numpy(), atol=2e-5)
            assert np.array_equal(greedy, want_greedy)


def test_networks_outside_the_fused_shapes_take_the_per_layer_launches():
    """DrQv2's actor is 50 -> 1024 -> 1024 -> A: outside the fused MLP kernel (hidden <= 256).  The plan then records the
    per-layer GEMM launches (three per network); a deterministic actor's sample is its mean action.  Against the general path."""
    import super_sac_amd as ssa
    from super_sac_amd import acting
    torch.manual_seed(5)
    agent = ssa.Agent(act_space_size=4, encoder=ssa.nets.IdentityEncoder(50), actor_network_cls=ssa.nets.ContinuousDeterministicActor,
                      critic_network_cls=ssa.nets.ContinuousCritic, discrete=False, ensemble_size=2, num_critics=2,
                      hidden_size=288, auto_rescale_targets=False)
    agent.to(torch.device(DEV))
    assert not ssa.engine.bind_arena(agent.actors[0], "self", [agent.actors[0]], torch.device(DEV)).fused
    rs = np.random.RandomState(4)
    for n in (1, 7):
        obs = rs.standard_normal((n, 50) if n > 1 else (50,)).astype(np.float32)
        random.seed(3)
        fast_s = agent.sample_action({"obs": obs}, num_envs=n)
        fast_f = agent.forward({"obs": obs}, num_envs=n)
        assert ("sample", n, 0.0) in acting._PLANS[agent] and ("forward", n, 0.0) in acting._PLANS[agent]
        acting.ENABLED = False
        try:
            random.seed(3)
            want_s = agent.sample_action({"obs": obs}, num_envs=n)
            want_f = agent.forward({"obs": obs}, num_envs=n)
        finally:
            acting.ENABLED = True
        np.testing.assert_allclose(fast_s, want_s, atol=3e-6)
        np.testing.assert_allclose(fast_f, want_f, atol=3e-6)


def test_ineligible_calls_take_the_general_path():
    from super_sac_amd import acting
    cfg, agent, _ = _pair("atari_pixels")
    obs = np.random.RandomState(2).randint(0, 255, (4, 84, 84)).astype(np.float32)   # float frames: not the uint8 contract
    act = agent.sample_action({"obs": obs})
    assert act.shape == (1,) and agent not in acting._PLANS
    cfg, agent, _ = _pair("sunrise")
    s = torch.randn(2, cfg["obs"], device=DEV)
    act = agent.sample_action({"obs": s}, from_cpu=False, num_envs=2)   # device tensors in, device tensor out
    assert torch.is_tensor(act) and agent not in acting._PLANS
