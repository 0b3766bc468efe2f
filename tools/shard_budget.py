"""Inputs of the sharding latency budget (profiles/r2_sharding_budget.md): the critic update of ONE rank that owns
n_local of the N=16 critics, measured as the unsharded engine with n_local critics at the same batch (the per-rank
launches are the same kernels with fewer critic workgroups; the exchange kernel is measured separately by
`bench.py --gpus 2` under rocprofv3).            python tools/shard_budget.py      (GPU box)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "__none__"]
import tools.bench_configs as bc

print("| shape | obs / act | B | critics on the rank | us per critic update (+Polyak/2) |")
print("|---|---|---|---|---|")
for name, obs, act in (("M", 17, 6), ("S (Humanoid)", 376, 17)):
    for n_local in (16, 8, 4, 2):
        critic, _ = bc.build(obs, act, 512, n_local, 2)
        t = bc.timed(critic, 1000, 200)
        print(f"| {name} | {obs} / {act} | 512 | {n_local} | {t * 1e6:.1f} |", flush=True)
