"""host time against wall time per critic update of one configuration of tools/bench_configs.py:
    python tools/host_time_config.py <obs> <act> <B> <N> <n> <fp32|bf16>"""
import sys, os, time
sys.argv, args = sys.argv[:1], sys.argv[1:]
sys.argv.append("__none__")
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_configs.py"))
bc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bc)
import torch
obs, act, B, N, n = (int(v) for v in args[:5])
critic, env_step = bc.build(obs, act, B, N, n, precision=args[5])
for _ in range(300):
    critic()
torch.cuda.synchronize()
wait = [0.0]
orig = torch.cuda.Event.synchronize
def timed(self):
    t = time.perf_counter(); orig(self); wait[0] += time.perf_counter() - t
torch.cuda.Event.synchronize = timed
import gc; gc.collect(); gc.freeze()
K = 3000
t0 = time.perf_counter()
for _ in range(K):
    critic()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
torch.cuda.Event.synchronize = orig
print(f"{args}: host {1e6 * t_host / K:.1f} us per update, of which {1e6 * wait[0] / K:.1f} us waiting for the GPU (input-slot events); "
      f"wall {1e6 * t_all / K:.1f} us per update")
# pure host cost: short bursts right behind a synchronisation (the host runs ahead of the device until the queue fills)
for burst in (16, 64, 256):
    ts = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(burst):
            critic()
        ts.append((time.perf_counter() - t0) / burst)
        torch.cuda.synchronize()
    ts.sort()
    print(f"   bursts of {burst}: host {1e6 * ts[len(ts) // 2]:.1f} us per update (median of 20)")
