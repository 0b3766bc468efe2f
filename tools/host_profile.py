"""Where does the host time of one critic_update go?  (run on the GPU box)"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

step, ssa = bench.build_engine(torch.device("cuda", 0), bench.NCRIT)
for _ in range(100):
    step()
torch.cuda.synchronize()
_wait = [0.0]
_orig_sync = torch.cuda.Event.synchronize
def _timed_sync(self):
    t = time.perf_counter()
    _orig_sync(self)
    _wait[0] += time.perf_counter() - t
torch.cuda.Event.synchronize = _timed_sync
t0 = time.perf_counter()
for _ in range(2000):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
torch.cuda.Event.synchronize = _orig_sync
print(f"host enqueue {1e6*t_host/2000:.1f} us/step (of which {1e6*_wait[0]/2000:.1f} us waiting for the GPU to free "
      f"an input slot), wall {1e6*t_all/2000:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
