"""host time of the pieces of a recorded critic update, for the first calls after a device synchronisation
(why a 20-step burst pays ~80 us up front)            python tools/first_step.py       (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import super_sac_amd as ssa

dev = torch.device("cuda:0")
step, env_step, _ = bench.build_engine(dev, bench.NCRIT, None)
for _ in range(40):
    step()
torch.cuda.synchronize()
lib = ssa._lib.lib
real_run, real_pol = lib.ssac_step_run, lib.ssac_step_polyak
acc = {"run": [], "pol": []}


def t_run(*a):
    t0 = time.perf_counter(); rc = real_run(*a); acc["run"].append(time.perf_counter() - t0); return rc


def t_pol(*a):
    t0 = time.perf_counter(); rc = real_pol(*a); acc["pol"].append(time.perf_counter() - t0); return rc


lib.ssac_step_run, lib.ssac_step_polyak = t_run, t_pol
for trial in range(3):
    time.sleep(0.05)
    torch.cuda.synchronize()
    acc["run"].clear(); acc["pol"].clear()
    tot = []
    for k in range(6):
        t0 = time.perf_counter(); step(); tot.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print("step() host us:", " ".join(f"{1e6 * t:6.1f}" for t in tot))
    print("  ssac_step_run:", " ".join(f"{1e6 * t:6.1f}" for t in acc["run"]))
    print("  ssac_step_polyak:", " ".join(f"{1e6 * t:6.1f}" for t in acc["pol"]))

# (hipDeviceScheduleSpin makes no difference here: the extra ~20-40 us of the first call is cold Python / cold caches
#  and the decided-wait of the first soft_update on an idle device, not the wake-up from the synchronisation)
