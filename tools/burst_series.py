"""20-step regions back to back right after engine construction (the driver's bench shape): how long until a region
reaches the steady per-update time?  (device clock ramp vs host effects)          python tools/burst_series.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

dev = torch.device("cuda:0")
step, env_step, _ = bench.build_engine(dev, bench.NCRIT, None)
import gc
gc.collect(); gc.freeze()
if len(sys.argv) > 1 and sys.argv[1] == "prewarm":   # experiment: does a busy device before the warm-up steps remove the transient?
    a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    while time.perf_counter() - t < 0.05:
        c = a @ b
    torch.cuda.synchronize()
    print("prewarmed for", round(1e3 * (time.perf_counter() - t), 1), "ms")
if len(sys.argv) > 1 and sys.argv[1] == "tiny":   # experiment: 2000 tiny launches (runtime pools) before the warm-up steps
    z = torch.zeros(64, device=dev)
    for _ in range(2000):
        z.add_(1.0)
    torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == "sleep":   # experiment: is it time since construction rather than work?
    time.sleep(0.5)
if len(sys.argv) > 1 and sys.argv[1] == "settle_sleep":   # experiment: settle, idle 100 ms, then the series
    for _ in range(1500):
        step()
    torch.cuda.synchronize()
    time.sleep(0.1)
if len(sys.argv) > 1 and sys.argv[1] == "settle":
    for _ in range(1500):
        step()
    torch.cuda.synchronize()
for _ in range(5):
    step()
T0 = time.perf_counter()
out = []
for rep in range(400):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    tf = time.perf_counter()
    for k in range(19):
        step()
    ti = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out.append((1e3 * (t0 - T0), 1e6 * (t1 - t0) / 20, 1e6 * (tf - t0), 1e6 * (ti - t0), 1e6 * (t1 - ti)))
out = np.array(out)
for i in list(range(0, 12)) + list(range(12, 400, 20)):
    print(f"region {i:3d} at {out[i,0]:7.1f} ms: {out[i,1]:6.2f} us per update; first step() {out[i,2]:6.1f} us, 20 issued after {out[i,3]:6.1f}, closing sync {out[i,4]:6.1f}")
