"""Timing of the pixel configurations (BASELINE.json configs 3 and 4) on the HIP path.
    python tools/bench_pixels.py [dmc|atari] [steps]
`build(which, dev)` returns the update closure (bench.py's secondary rows use it too)."""
import copy, math, os, sys, time, types
from itertools import chain
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np, torch


def build(which, dev):
    """one critic update (+ the Polyak updates of critics and encoder) of BASELINE config 3 ("dmc") or 4 ("atari") on
    synthetic uint8 observations; returns (step, batch_size)"""
    import super_sac_amd as ssa
    import synth
    if which == "dmc":   # DMC cheetah-run pixels 84x84x9, DrQv2 random shift + BigPixelEncoder, batch 512, hidden 1024
        C, B, act, emb, hid, discrete = 9, 512, 6, 50, 1024, False
        conv = ssa.nets.BigPixelEncoder((C, 84, 84), emb)
        actor_cls, critic_cls = ssa.nets.ContinuousDeterministicActor, ssa.nets.ContinuousCritic
        gamma, clip, mix = 0.99 ** 3, None, 1.0
    else:                # Atari 84x84x4, SAC-Discrete, SmallPixelEncoder, batch 1024
        C, B, act, emb, hid, discrete = 4, 1024, 4, 128, 256, True
        conv = ssa.nets.SmallPixelEncoder((C, 84, 84), emb)
        actor_cls, critic_cls = ssa.nets.DiscreteActor, ssa.nets.DiscreteCritic
        gamma, clip, mix = 0.99 ** 3, 40.0, 0.9
    agent = ssa.Agent(act_space_size=act, encoder=ssa.nets.PixelEncoder(conv), actor_network_cls=actor_cls,
                      critic_network_cls=critic_cls, discrete=discrete, ensemble_size=1, num_critics=2,
                      hidden_size=hid, auto_rescale_targets=False)
    agent.to(dev); agent.train()
    target = copy.deepcopy(agent)
    buf = ssa.replay.ReplayBuffer(20000, device=dev)
    buf.load_experience(*synth.synth_pixel_transitions(4096, C, 84, n_actions=act if discrete else None, act_dim=act,
                                                       seed=1))
    copt = torch.optim.Adam(chain(*(c.parameters() for c in agent.critics)), lr=1e-4)
    eopt = torch.optim.Adam(agent.encoder.parameters(), lr=1e-4)
    la = torch.Tensor([math.log(0.1)]).to(dev); la.requires_grad = True
    aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.Drqv2Aug(B)])
    space = types.SimpleNamespace(low=-np.ones(act, np.float32), high=np.ones(act, np.float32))
    rproc = None if discrete else ssa.learning_utils.GaussianExplorationNoise(space, 1.0, 0.1, 500000)

    def step():
        step.out = ssa.learning.critic_update(buffer=buf, agent=agent, target_agent=target, critic_optimizer=copt,
            encoder_optimizer=eopt, log_alphas=[la], batch_size=B, gamma=gamma, critic_clip=clip, encoder_clip=clip,
            target_critic_ensemble_n=2, weighted_bellman_temp=None, weight_type=None, pop=False, augmenter=aug,
            encoder_lambda=0, aug_mix=mix, discrete=discrete, random_process=rproc, noise_clip=0.3, per=False,
            update_priorities=False, dr3_coeff=0.0)
        for ac, tc in zip(agent.critics, target.critics):
            ssa.learning_utils.soft_update(tc, ac, 0.01)
        ssa.learning_utils.soft_update(target.encoder, agent.encoder, 0.01)
    step.objects = dict(agent=agent, target=target)   # (tests/test_hip_cases.py compares two engine settings through this)
    return step, B


if __name__ == "__main__":
    import super_sac_amd.conv_encoder as _ce
    for env, knob in (("PIX_MIN_ROWS", "IMPLICIT_MIN_ROWS"), ("PIX_FIRST_WG", "FIRST_WG_PER_CU"),
                      ("PIX_IMPL_WG", "IMPLICIT_WG_PER_CU"), ("PIX_FIRST_RPS", "FIRST_ROWS_PER_SLICE"),
                      ("PIX_IMPL_RPS", "IMPLICIT_ROWS_PER_SLICE"), ("PIX_FC_SLICES", "FC_SLICES")):   # knob sweeps
        if os.environ.get(env):
            setattr(_ce, knob, int(os.environ[env]))
    which = sys.argv[1] if len(sys.argv) > 1 else "dmc"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    step, B = build(which, torch.device("cuda"))
    for _ in range(3): step()
    import gc
    gc.collect(); gc.freeze()   # (as bench.py and INTEGRATION.md: no generation-2 collection pause inside the timed region)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{which}: {1e3*dt:.2f} ms per critic update (B={B}), {1/dt:.1f} updates/s, peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
