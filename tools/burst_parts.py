"""which part of the host side of step() is slow right after engine construction?   python tools/burst_parts.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import super_sac_amd as ssa
from super_sac_amd import rng

dev = torch.device("cuda:0")
step, env_step, _ = bench.build_engine(dev, bench.NCRIT, None)
import gc
gc.collect(); gc.freeze()
T = {}


def wrap(mod, name):
    f = getattr(mod, name)

    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T.setdefault(name, []).append(time.perf_counter() - t0); return r
    setattr(mod, name, g)


wrap(rng, "draw_indices"); wrap(rng, "draw_subset"); wrap(ssa._lib.lib, "ssac_step_run"); wrap(ssa._lib.lib, "ssac_step_polyak")
for _ in range(5):
    step()
for rep in range(40):
    torch.cuda.synchronize()
    T.clear()
    t0 = time.perf_counter()
    for k in range(20):
        step()
    ti = time.perf_counter()
    torch.cuda.synchronize()
    if rep < 10 or rep % 10 == 0:
        parts = {k: 1e6 * np.sum(v) / 20 for k, v in T.items()}
        tot = 1e6 * (ti - t0) / 20
        print(f"region {rep:2d}: host {tot:5.1f} us per step: " + ", ".join(f"{k} {v:5.1f}" for k, v in parts.items()) + f", other python {tot - sum(parts.values()):5.1f}")
