"""one-off soak of the one-call acting path interleaved with recorded updates (GPU box): 60 000 Agent.sample_action calls, a critic +
actor update every third step.  python tools/r6/soak_act.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
dev = torch.device("cuda")
st, _, ssa = bench.build_engine(dev, 2, None, batch=256, obs=3, act=1, ncrit=2)
ob = st.objects
agent, actor = ob["agent"], ob["actor_step"]
o = {"obs": np.random.RandomState(0).standard_normal(3).astype(np.float32)}
t0 = time.time()
acts = []
for k in range(60000):
    a = agent.sample_action(o)
    if k % 3 == 0:
        d = st()
        actor(d)
    if k % 1000 == 0:
        acts.append(float(a[0]))
assert all(np.isfinite(acts)), acts
print("60000 acting calls interleaved with 20000 critic+actor updates:", round(time.time() - t0, 1), "s; sampled actions", acts[:5])
