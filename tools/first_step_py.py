"""where the Python part of the FIRST step() after a device synchronisation goes (tools/first_step.py shows the C
calls are not it)           python tools/first_step_py.py       (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import super_sac_amd as ssa
from super_sac_amd import rng, learning, learning_utils as lu

dev = torch.device("cuda:0")
step, env_step, _ = bench.build_engine(dev, bench.NCRIT, None)
for _ in range(40):
    step()
torch.cuda.synchronize()
T = {}


def wrap(mod, name, key=None):
    f = getattr(mod, name)
    key = key or name

    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T.setdefault(key, []).append(time.perf_counter() - t0); return r
    setattr(mod, name, g)


wrap(rng, "draw_indices"); wrap(rng, "draw_subset"); wrap(rng, "choice")
wrap(ssa._lib.lib, "ssac_step_run"); wrap(ssa._lib.lib, "ssac_step_polyak")
wrap(lu, "soft_update"); wrap(learning, "critic_update")
wrap(learning._FastStep, "still_valid"); wrap(learning._FastStep, "run")
for trial in range(3):
    torch.cuda.synchronize()
    T.clear()
    tot = []
    for k in range(5):
        t0 = time.perf_counter(); step(); tot.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print("step():", " ".join(f"{1e6 * t:6.1f}" for t in tot))
    for k, v in T.items():
        print(f"  {k:18s}", " ".join(f"{1e6 * t:6.1f}" for t in v))
