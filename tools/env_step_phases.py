"""Where the time of one REDQ environment step goes (20 critic updates + actor + temperature update): host time of
every call (before the device has finished) and wall time with a device sync after the block.

    python tools/env_step_phases.py [fp32|bf16]      (GPU box)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv, precision = [sys.argv[0], "__none__"], (sys.argv[1] if len(sys.argv) > 1 else "fp32")
import torch
import tools.bench_configs as bc  # the filter above selects no row: only build() is used
import super_sac_amd as ssa

critic, env_step = bc.build(17, 6, 512, 10, 2, precision=precision)
for _ in range(8):
    env_step()
torch.cuda.synchronize()
src = env_step.__code__  # noqa: keep the closure's objects reachable through its cells
cells = dict(zip(env_step.__code__.co_freevars, (c.cell_contents for c in env_step.__closure__)))
buf, agent, aopt, lopt, la, aug, B, act = (cells[k] for k in ("buf", "agent", "aopt", "lopt", "la", "aug", "B", "act"))


def actor(d):
    return ssa.learning.online_actor_update(buffer=buf, agent=agent, pop=False, actor_optimizer=aopt, log_alphas=[la],
                                            batch_size=B, clip=None, random_process=None, noise_clip=None,
                                            augmenter=aug, aug_mix=0.0, premade_replay_dicts=d)


def alpha(d):
    return ssa.learning.alpha_update(buffer=buf, agent=agent, optimizers=[lopt], batch_size=B, log_alphas=[la],
                                     augmenter=aug, aug_mix=0.0, target_entropy=-float(act), premade_replay_dicts=d,
                                     discrete=False)


rows = {"critic x20": [], "actor": [], "alpha": [], "critic x20 (after actor)": []}
host = {k: [] for k in rows}
for it in range(30):
    for name, fn in (("critic x20", None), ("actor", actor), ("alpha", alpha)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if fn is None:
            for _ in range(20):
                d = critic()
        else:
            fn(d)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host[name].append(t1 - t0); rows[name].append(t2 - t0)
print(f"precision {precision}")
for k in ("critic x20", "actor", "alpha"):
    h, w = sorted(host[k])[len(host[k]) // 2], sorted(rows[k])[len(rows[k]) // 2]
    print(f"{k:12s} host {h * 1e6:8.1f} us   wall(sync) {w * 1e6:8.1f} us")
t = bc.timed(env_step, 60, 5)
print(f"env step back to back: {t * 1e6:.1f} us")
rec = agent.__dict__.get("_ssac_actor_rec", {})
print("recorded actor lists:", [(r.calls, bool(r.list)) for r in rec.values()])
