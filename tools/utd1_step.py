"""A UTD-1 environment step on the device path (BASELINE configs 1, 3, 4, 5 run UTD 1): acting + critic update (+ Polyak / 2)
+ online actor update + temperature update, wall clock per step, and the pieces one by one (each followed by a device
synchronisation, so the pieces add up to more than the step).      python tools/utd1_step.py [obs act N B]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

obs, act, N, B = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (3, 1, 2, 256)
dev = torch.device("cuda")
step, env_step, ssa = bench.build_engine(dev, N, None, batch=B, obs=obs, act=act, ncrit=N)
ob = step.objects
agent, la, buf = ob["agent"], ob["log_alpha"], ob["buffer"]
actor_step = ob["actor_step"]
lopt = torch.optim.Adam([la], lr=1e-4, betas=(0.5, 0.999))
aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.IdentityAug(B)])
o = np.random.RandomState(0).standard_normal(obs).astype(np.float32)


def alpha(dicts):
    ssa.learning.alpha_update(buffer=buf, agent=agent, optimizers=[lopt], batch_size=B, log_alphas=[la], augmenter=aug,
                              aug_mix=0.0, target_entropy=-float(act), premade_replay_dicts=dicts, discrete=False)


def one():
    agent.sample_action({"obs": o})
    dicts = step()
    actor_step(dicts)
    alpha(dicts)


for _ in range(50):
    one()
torch.cuda.synchronize()
import gc
gc.collect(); gc.freeze()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        one()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 300)
print(f"UTD-1 env step (obs {obs} act {act} N {N} B {B}): {statistics.median(ts) * 1e6:.1f} us per step "
      f"= {1 / statistics.median(ts):.0f} env steps/s")
dicts = step()
for name, fn in (("sample_action", lambda: agent.sample_action({"obs": o})), ("critic_update (+Polyak/2)", step),
                 ("online_actor_update", lambda: actor_step(dicts)), ("alpha_update", lambda: alpha(dicts))):
    host, tot = [], []
    for _ in range(200):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); tot.append(t2 - t0)
    print(f"  {name:28s} host {statistics.median(host) * 1e6:6.1f} us   host + device (synchronised) {statistics.median(tot) * 1e6:6.1f} us")
