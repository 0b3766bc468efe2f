"""GPU: the acting path on the engine's weights (Agent.forward / Agent.sample_action, agent.py:204-315,
SURVEY 8(f) rank 2) against the CPU oracle's arithmetic on the same seeded weights and injected noise."""
import random

import numpy as np
import pytest
import torch

import case_runner
import ssac_oracle as orc
import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pair(name, ucb=0.0):
    cfg = synth.CASES[name]
    agent = case_runner.build_engine_agent(cfg, torch.device(DEV))
    agent.ucb_bonus = ucb
    return cfg, agent, case_runner._oracle_agent(cfg)


def test_forward_is_the_mean_of_the_actors_mean_actions():
    cfg, agent, oa = _pair("sunrise")  # 3 members, continuous
    obs = np.random.RandomState(0).standard_normal((5, cfg["obs"])).astype(np.float32)
    act = agent.forward({"obs": obs}, from_cpu=True, num_envs=5)
    s = torch.from_numpy(obs)
    want = torch.stack([torch.tanh(orc.mlp3(a, s)[0][:, :cfg["act"]]) for a in oa.actors], 0).mean(0).clamp(-1, 1)
    assert act.shape == (5, cfg["act"])
    np.testing.assert_allclose(act, want.numpy(), atol=2e-6)
    one = agent.forward({"obs": obs[0]}, from_cpu=True, num_envs=1)  # single env: batch dim squeezed away
    np.testing.assert_allclose(one, want.numpy()[0], atol=2e-6)


def test_discrete_forward_is_argmax_of_mean_probabilities():
    cfg, agent, oa = _pair("sac_discrete")
    obs = np.random.RandomState(1).standard_normal((7, cfg["obs"])).astype(np.float32)
    act = agent.forward({"obs": obs}, num_envs=7)
    probs = torch.stack([torch.softmax(orc.mlp3(a, torch.from_numpy(obs))[0], -1) for a in oa.actors], 0).mean(0)
    assert np.array_equal(act[:, 0], probs.argmax(-1).numpy())


def test_sample_action_uses_one_random_actor_and_the_drawn_noise():
    import super_sac_amd as ssa
    cfg, agent, oa = _pair("sunrise")
    rs = np.random.RandomState(2)
    obs = rs.standard_normal((4, cfg["obs"])).astype(np.float32)
    eps = rs.standard_normal((4, cfg["act"])).astype(np.float32)
    saved = ssa.rng.draw_normal
    ssa.rng.draw_normal = lambda shape, device: torch.from_numpy(eps).to(device)
    try:
        random.seed(9)
        act = agent.sample_action({"obs": obs}, num_envs=4)
        random.seed(9)
        k = random.choice(range(len(oa.actors)))  # same consumption as random.choice(self.actors)
    finally:
        ssa.rng.draw_normal = saved
    out = orc.mlp3(oa.actors[k], torch.from_numpy(obs))[0]
    want = orc.tanh_normal_sample(out, oa.lo, oa.hi, torch.from_numpy(eps))[0].clamp(-1, 1)
    np.testing.assert_allclose(act, want.numpy(), atol=3e-6)


def test_ucb_sample_action_picks_the_argmax_candidate():
    """SUNRISE: candidates from every actor, value = mean over members of min-Q + bonus * std (unbiased)."""
    import super_sac_amd as ssa
    cfg, agent, oa = _pair("sunrise", ucb=0.7)
    rs = np.random.RandomState(3)
    n, E, A = 6, cfg["E"], cfg["act"]
    obs = rs.standard_normal((n, cfg["obs"])).astype(np.float32)
    eps = [rs.standard_normal((n, A)).astype(np.float32) for _ in range(E)]
    queue = [torch.from_numpy(e) for e in eps]
    saved = ssa.rng.draw_normal
    ssa.rng.draw_normal = lambda shape, device: queue.pop(0).to(device)
    try:
        act = agent.sample_action({"obs": obs}, num_envs=n)
    finally:
        ssa.rng.draw_normal = saved
    s = torch.from_numpy(obs)
    cands = torch.stack([orc.tanh_normal_sample(orc.mlp3(oa.actors[a], s)[0], oa.lo, oa.hi,
                                                torch.from_numpy(eps[a]))[0] for a in range(E)], 0)
    q = torch.stack([torch.stack([orc.ensemble_q(oa.critics[c], s, cands[a]).squeeze(-1) for a in range(E)], 0)
                     for c in range(E)], 0)                         # (members, candidates, envs)
    ucb = q.mean(0) + 0.7 * q.std(0)
    best = ucb.argmax(0)
    want = cands[best, torch.arange(n)].clamp(-1, 1)
    # a near-tie between two candidates may flip under fp32 reordering: require agreement where the margin is clear
    top2 = ucb.topk(2, dim=0).values
    clear = (top2[0] - top2[1]) > 1e-4
    assert clear.sum() >= n - 1
    np.testing.assert_allclose(act[clear.numpy()], want.numpy()[clear.numpy()], atol=3e-6)
