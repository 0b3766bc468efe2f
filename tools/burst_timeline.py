"""Why a 20-step burst (the driver's `bench.py --steps 20 --warmup 5`) is slower per step than 2000 steps:
per-step device timeline of a burst after an idle gap -- HIP events after every step.

    python tools/burst_timeline.py        (GPU box)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda:0")
step, env_step, ssa = bench.build_engine(dev, bench.NCRIT, None)
for _ in range(30):
    step()
torch.cuda.synchronize()
st = torch.cuda.current_stream()
for idle_ms in (0, 1, 20, 200):
    time.sleep(idle_ms * 1e-3)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    host = []
    t0 = time.perf_counter()
    ev[0].record(st)
    for k in range(20):
        step()
        ev[k + 1].record(st)
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    gaps = [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(20)]
    print(f"idle {idle_ms:4d} ms: wall {1e6 * (t1 - t0) / 20:6.1f} us/step; device gaps us:",
          " ".join(f"{g:5.1f}" for g in gaps))
    print("                host submit times us:", " ".join(f"{1e6 * h:5.0f}" for h in host))
