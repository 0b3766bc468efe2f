"""The online actor update alone (learning.py:344-421 of the reference) at a bench_configs shape:
    python tools/actor_update_rows.py <obs> <act> <B> <N> [updates]      (GPU box)
us per actor update, recorded launch list, on the batch of one critic update (premade_replay_dicts, as redq.gin runs it).
Under rocprofv3 --kernel-trace it gives the per-launch breakdown (rocprofv3 --kernel-trace --stats -- python3 tools/actor_update_rows.py 17 6 512 10)."""
import os, sys, time
sys.argv, args = sys.argv[:1], sys.argv[1:]
sys.argv.append("__none__")
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_configs.py"))
bc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bc)
import math, torch
from itertools import chain
ssa = bc.ssa
if os.environ.get("SSAC_ACTOR_CHAIN"):   # A/B: the chained actor launch against the three-launch form (tools only)
    ssa.learning.ACTOR_CHAIN = os.environ["SSAC_ACTOR_CHAIN"] == "1"
obs, act, B, N = (int(v) for v in args[:4])
n_upd = int(args[4]) if len(args) > 4 else 600
critic, env_step = bc.build(obs, act, B, N, 2)
agent = critic.objects["agent"]
dev = bc.dev
for _ in range(5):
    dicts = critic()
aopt = torch.optim.Adam(chain(*(a.parameters() for a in agent.actors)), lr=3e-4)
la = torch.Tensor([math.log(0.1)]).to(dev); la.requires_grad = True
aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.IdentityAug(B)])


def actor():
    ssa.learning.online_actor_update(buffer=None, agent=agent, pop=False, actor_optimizer=aopt, log_alphas=[la],
                                     batch_size=B, clip=None, random_process=None, noise_clip=None, augmenter=aug,
                                     aug_mix=0.0, premade_replay_dicts=dicts)


t = bc.timed(actor, n_upd, 50)
print(f"actor update {args}: {t * 1e6:.1f} us")
