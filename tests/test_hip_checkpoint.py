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
