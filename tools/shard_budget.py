"""Inputs of the sharding latency budget (profiles/r5_sharding_budget.md): the critic update of ONE rank that holds n_local
of the N = 16 critics, measured ON A SHARDED RANK (round-4 review, weak #2): the agent is `parallel.install`-ed as rank 0 of a
one-rank job with the one-shot exchange switched on, so the recorded update is the rank's real launch list -- chained launch ->
`xchg_kernel` (one workgroup: payload into the receive slot, flag, poll, rank-ordered reduction) -> weight-gradient launch --
with both subset members owned locally (the rank's most expensive draw; the flag it polls is its own, i.e. the exchange
costs its launch boundary + one local round trip, not an xGMI hop).  Beside it the round-4 way: the UNSHARDED engine built
with n_local critics.            python tools/shard_budget.py      (GPU box)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "__none__"]
import torch
import torch.distributed as dist
import tools.bench_configs as bc
from super_sac_amd import parallel

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
dist.init_process_group("gloo", rank=0, world_size=1)
# (enable_one_shot() wants more than one rank; the exchange object itself is happy with a world of one)
parallel._exchange = parallel.Exchange(0, 1, parallel.ONE_SHOT_MAX_FLOATS, torch.device("cuda", 0))

print("| shape | obs / act | B | critics on the rank | sharded rank: us per critic update (+Polyak/2) | unsharded engine with that many critics |")
print("|---|---|---|---|---|---|")
for name, obs, act in (("M", 17, 6), ("S (Humanoid)", 376, 17)):
    for n_local in (16, 8, 4, 2):
        critic, _ = bc.build(obs, act, 512, n_local, 2, as_rank=True)
        t = bc.timed(critic, 1000, 200)
        assert not parallel.exchange_failed()
        x, parallel._exchange = parallel._exchange, None
        critic_u, _ = bc.build(obs, act, 512, n_local, 2)
        tu = bc.timed(critic_u, 1000, 200)
        parallel._exchange = x
        print(f"| {name} | {obs} / {act} | 512 | {n_local} | {t * 1e6:.1f} | {tu * 1e6:.1f} |", flush=True)
        del critic, critic_u
dist.destroy_process_group()
