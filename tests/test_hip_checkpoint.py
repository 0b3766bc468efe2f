"""GPU: checkpoint + true resume: a run that is saved, torn down, rebuilt from scratch,
loaded and continued lands on exactly the parameters, optimizer state, priorities and logs of the run that was
never interrupted."""
import copy
import math
import random
from itertools import chain

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _build(ssa, dev, popart):
    agent = ssa.Agent(act_space_size=4, encoder=ssa.nets.IdentityEncoder(9),
                      actor_network_cls=ssa.nets.ContinuousStochasticActor,
                      critic_network_cls=ssa.nets.ContinuousCritic, ensemble_size=1, num_critics=3,
                      hidden_size=64, auto_rescale_targets=popart, log_std_low=-5.0, log_std_high=2.0)
    agent.to(dev)
    target = copy.deepcopy(agent)
    copt = torch.optim.Adam(chain(*(c.parameters() for c in agent.critics)), lr=3e-4)
    aopt = torch.optim.Adam(chain(*(a.parameters() for a in agent.actors)), lr=3e-4)
    eopt = torch.optim.Adam(agent.encoder.parameters(), lr=1e-4)
    la = torch.Tensor([math.log(0.1)]).to(dev)
    la.requires_grad = True
    lopt = torch.optim.Adam([la], lr=1e-4, betas=(0.5, 0.999))
    buf = ssa.replay.ReplayBuffer(2048, device=dev)
    return agent, target, copt, aopt, eopt, la, lopt, buf


def _steps(ssa, objs, n, popart):
    agent, target, copt, aopt, eopt, la, lopt, buf = objs
    aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.IdentityAug(64)])
    last = None
    for k in range(n):
        logs, dicts = ssa.learning.critic_update(
            buffer=buf, agent=agent, target_agent=target, critic_optimizer=copt, encoder_optimizer=eopt,
            log_alphas=[la], batch_size=64, gamma=0.99, critic_clip=10.0 if popart else None, encoder_clip=None,
            target_critic_ensemble_n=2, weighted_bellman_temp=None, weight_type=None, pop=popart, augmenter=aug,
            encoder_lambda=0, aug_mix=0.0, discrete=False, random_process=None, noise_clip=None, per=False,
            update_priorities=(k % 3 == 0), dr3_coeff=0.0)
        if k % 2 == 0:
            ssa.learning_utils.soft_update(target.critics[0], agent.critics[0], 0.01)
        ssa.learning.online_actor_update(
            buffer=buf, agent=agent, pop=popart, actor_optimizer=aopt, log_alphas=[la], batch_size=64, clip=None,
            random_process=None, noise_clip=None, augmenter=aug, aug_mix=0.0, premade_replay_dicts=dicts)
        ssa.learning.alpha_update(buffer=buf, agent=agent, optimizers=[lopt], batch_size=64, log_alphas=[la],
                                  augmenter=aug, aug_mix=0.0, target_entropy=-4.0, premade_replay_dicts=dicts,
                                  discrete=False)
        ssa.learning.offline_actor_update(
            buffer=buf, agent=agent, actor_optimizer=aopt, encoder_optimizer=eopt, batch_size=64, actor_clip=None,
            update_encoder=False, encoder_clip=None, augmenter=aug, actor_lambda=0.0, aug_mix=0.0, per=True,
            discrete=False, filter_=True)
        last = float(logs["losses/critic_overall_loss"])
    flat = lambda mods: torch.cat([p.detach().flatten() for m in mods for p in m.parameters()]).cpu().numpy()
    return (flat(agent.critics), flat(agent.actors), flat(target.critics), float(la), last,
            buf._per.sum_tree.copy())


@pytest.mark.parametrize("popart", [False, True], ids=["plain", "popart+clip"])
def test_resume_is_bit_identical(tmp_path, popart):
    import super_sac_amd as ssa
    dev = torch.device("cuda")

    def fresh():
        torch.manual_seed(4); np.random.seed(4); random.seed(4)
        objs = _build(ssa, dev, popart)
        objs[-1].load_experience(*synth.synth_transitions(1500, 9, 4, seed=6))
        return objs

    # uninterrupted: 6 + 5 rounds (the first 6 include the recorded-launch-list switch-over at call 4)
    objs = fresh()
    _steps(ssa, objs, 6, popart)
    ref = _steps(ssa, objs, 5, popart)

    # interrupted: 6 rounds, save, rebuild everything from different seeds, load, 5 more
    objs = fresh()
    _steps(ssa, objs, 6, popart)
    agent, target, copt, aopt, eopt, la, lopt, buf = objs
    ssa.checkpoint.save_training_state(str(tmp_path), agent, target, {"critic": copt, "actor": aopt, "alpha": [lopt]},
                                       [la], buf)
    del objs, agent, target, copt, aopt, eopt, la, lopt, buf
    torch.manual_seed(99); np.random.seed(99); random.seed(99)
    objs = _build(ssa, dev, popart)
    agent, target, copt, aopt, eopt, la, lopt, buf = objs
    ssa.checkpoint.load_training_state(str(tmp_path), agent, target, {"critic": copt, "actor": aopt, "alpha": [lopt]},
                                       [la], buf)
    got = _steps(ssa, objs, 5, popart)
    for a, b, what in zip(ref, got, ("critics", "actors", "target critics", "log_alpha", "last critic loss",
                                     "priority tree")):
        assert np.array_equal(np.asarray(a), np.asarray(b)), f"resumed run differs in {what}"


RESUME_CASES = [n_ for n_, c_ in synth.CASES.items() if c_.get("resume")]


@pytest.mark.parametrize("name", RESUME_CASES)
def test_reference_checkpoint_loads_and_resumes_on_the_gpu(tmp_path, name):
    """SURVEY 8(f) rank 3 against the REFERENCE, not against this package: the fixture holds the state dicts the reference's
    Agent.save wrote after the base case's update sequence (agent.py:172-195; oracle/gen_golden.py::reference_checkpoint) and
    the outputs of the reference continuing from Agent.load of them (fresh agent of another seed, target = deepcopy, new
    optimizers: main.py:188-239, 321).  Here: the arrays go back into the files, this package's Agent.load reads them on the
    GPU -- every parameter bit-equal to what the reference saved -- the base case's update schedule runs on the HIP path
    and must land on the reference's continuation at the tolerances of every other fixture; then Agent.save of the RESUMED
    agent is read back file by file: the reference's file set, its keys, its shapes (what its load_state_dict expects)."""
    import os
    import case_runner
    import super_sac_amd as ssa
    cfg, fx = synth.CASES[name], case_runner.load_fixture(name)
    files = case_runner.checkpoint_arrays(fx)
    assert {"encoder.pt", "actor0.pt", "critic0.pt", "inverse.pt", "contrastive.pt"} <= set(files)
    dev = torch.device("cuda")
    # (1) load only: bit-equal parameters, arenas re-pointed at the loaded values
    agent = case_runner.build_engine_agent(cfg, dev)
    agent.load(case_runner.write_checkpoint_dir(fx, str(tmp_path / "from_reference")))
    for i, critic in enumerate(agent.critics):
        sd = critic.state_dict()
        assert set(sd) == set(files[f"critic{i}.pt"])
        for k, v in files[f"critic{i}.pt"].items():
            assert np.array_equal(sd[k].cpu().numpy(), v), (i, k)
        arena = critic.arena(dev)
        assert np.array_equal(arena.view(0, "w1").cpu().numpy(), files[f"critic{i}.pt"]["nets.0.fc1.weight"])
    for i, actor in enumerate(agent.actors):
        for k, v in files[f"actor{i}.pt"].items():
            assert np.array_equal(actor.state_dict()[k].cpu().numpy(), v), (i, k)
    for k, v in files["encoder.pt"].items():
        assert np.array_equal(agent.encoder.state_dict()[k].cpu().numpy(), v), k
    # (2) load + the update schedule on the HIP path vs the reference's continuation
    rec = case_runner.run_engine(name)
    worst = case_runner.compare(rec, fx, who=f"hip[{name}: resumed from the reference's checkpoint]")
    print(f"{name}: worst deviations vs the reference's continuation {worst}")
    # (3) the file set / keys / shapes this package writes for the same agent = the reference's
    out = tmp_path / "from_engine"
    out.mkdir()
    agent.save(str(out))
    written = sorted(f for f in os.listdir(out) if not f.endswith("_stats.pt"))
    assert written == sorted(files), (written, sorted(files))
    for fname in written:
        sd = torch.load(os.path.join(out, fname), map_location="cpu")
        assert set(sd) == set(files[fname]), fname
        for k, v in files[fname].items():
            assert tuple(sd[k].shape) == v.shape and np.array_equal(sd[k].numpy(), v), (fname, k)
