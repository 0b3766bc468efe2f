"""are eager critic updates bit-reproducible run to run?  python tools/dbg/determinism.py"""
import copy, math, os, random, sys
from itertools import chain
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p_ in ("tests", ""):
    sys.path.insert(0, os.path.join(ROOT, p_))
import numpy as np, torch
import super_sac_amd as ssa
import synth

def run(use_lists, n=6, hidden=64, N=4, B=128):
    ssa.learning.USE_GRAPHS = use_lists
    torch.manual_seed(4); np.random.seed(4); random.seed(4)
    dev = torch.device("cuda")
    agent = ssa.Agent(act_space_size=6, encoder=ssa.nets.IdentityEncoder(17), actor_network_cls=ssa.nets.ContinuousStochasticActor,
                      critic_network_cls=ssa.nets.ContinuousCritic, ensemble_size=1, num_critics=N, hidden_size=hidden,
                      auto_rescale_targets=False, log_std_low=-5.0, log_std_high=2.0)
    agent.to(dev)
    target = copy.deepcopy(agent)
    buf = ssa.replay.ReplayBuffer(4096, device=dev)
    buf.load_experience(*synth.synth_transitions(2000, 17, 6, seed=5))
    copt = torch.optim.Adam(chain(*(c.parameters() for c in agent.critics)), lr=3e-4)
    eopt = torch.optim.Adam(agent.encoder.parameters(), lr=1e-4)
    la = torch.Tensor([math.log(0.1)]).to(dev); la.requires_grad = True
    aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.IdentityAug(B)])
    out = []
    for k in range(n):
        logs, _ = ssa.learning.critic_update(buffer=buf, agent=agent, target_agent=target, critic_optimizer=copt, encoder_optimizer=eopt,
            log_alphas=[la], batch_size=B, gamma=0.99, critic_clip=None, encoder_clip=None, target_critic_ensemble_n=2,
            weighted_bellman_temp=None, weight_type=None, pop=False, augmenter=aug, encoder_lambda=0, aug_mix=0.0, discrete=False,
            random_process=None, noise_clip=None, per=False, update_priorities=False, dr3_coeff=0.0)
        vals = {k_: float(v) for k_, v in logs.items()}
        out.append((vals["gradients/critic_random_grad"], vals["losses/critic_overall_loss"],
                    agent.critics[0].arena(dev).params.double().sum().item()))
    return out

for trial in range(4):
    a = run(False)
    print("eager", trial, ["%.9g/%.9g/%.12g" % t for t in a[:3]])
b = run(True)
print("lists  ", ["%.9g/%.9g/%.12g" % t for t in b[:3]])
