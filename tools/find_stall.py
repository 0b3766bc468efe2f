import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
step, ssa = bench.build_engine(torch.device("cuda", 0), bench.NCRIT)
torch.cuda.synchronize()
slow = []
for i in range(400):
    t0 = time.perf_counter()
    step()
    dt = time.perf_counter() - t0
    if dt > 1e-3:
        slow.append((i, round(dt * 1e3, 2)))
torch.cuda.synchronize()
print("host-side slow steps (index, ms):", slow)
