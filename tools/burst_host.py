"""host-side anatomy of a 20-step timed region (the driver's bench shape): per-call host time of step(), the closing
synchronisation, and the region against 20 x the steady per-update time.        python tools/burst_host.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

dev = torch.device("cuda:0")
step, env_step, _ = bench.build_engine(dev, bench.NCRIT, None)
for _ in range(300):
    step()
torch.cuda.synchronize()
# steady per-update time
t0 = time.perf_counter()
for _ in range(2000):
    step()
torch.cuda.synchronize()
steady = (time.perf_counter() - t0) / 2000 * 1e6
rows = []
for rep in range(12):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = []
    for k in range(20):
        step()
        ts.append(time.perf_counter())
    t_issue = ts[-1]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    d = np.diff([t0] + ts) * 1e6
    rows.append((1e6 * (t1 - t0), d[0], d[1], np.median(d[2:]), 1e6 * (t_issue - t0), 1e6 * (t1 - t_issue)))
rows = np.array(rows)
print(f"steady {steady:.2f} us per update; 20 x steady = {20 * steady:.0f} us")
print("region us | first step() | second | median of the rest | all 20 issued after | closing sync waits")
for r in rows:
    print("  ".join(f"{v:8.1f}" for v in r))
print("median region", np.median(rows[:, 0]), "-> per update", np.median(rows[:, 0]) / 20, "overhead vs steady", np.median(rows[:, 0]) - 20 * steady)
