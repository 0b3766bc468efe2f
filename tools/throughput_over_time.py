"""updates/s of consecutive 250-update windows from a cold start (clock ramp / one-time costs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
step, ssa = bench.build_engine(torch.device("cuda", 0), bench.NCRIT)
torch.cuda.synchronize()
out = []
for w in range(int(os.environ.get("WINDOWS", "24"))):
    t0 = time.perf_counter()
    for _ in range(int(os.environ.get("WIN", "250"))):
        step()
    torch.cuda.synchronize()
    out.append(int(os.environ.get("WIN", "250")) / (time.perf_counter() - t0))
print(" ".join(f"{v:.0f}" for v in out))
