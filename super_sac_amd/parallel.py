"""Critic-ensemble sharding across the GPUs of one node.

The update path shards BY CRITIC: each critic's loss term, backward pass, Adam state and Polyak
target are independent (learning.py:90-98, main.py:188-193, 410-413).  Rank g owns the critics
[lo, hi) of the global ensemble -- its ``agent.critics[i]`` holds only those -- while the actor,
temperature, PopArt state and the replay buffer are replicated and every rank reproduces the same
host draws (indices, REDQ subset, action noise) from identically seeded generators, so there is no
batch broadcast.  The real exchange steps are:

  critic update   MIN all-reduce of the (B x q_dim) partial min-Q over the target subset
  actor  update   MIN all-reduce of the (B,) min-Q over all critics, then SUM all-reduce of the
                  (B x A) action gradient coming back through the arg-min critics

2-35 KiB messages, latency bound; ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the
GPUs; "gloo" in the CPU tests) carries them.
"""
import torch
import torch.distributed as dist


class Shard:
    def __init__(self, rank, world, num_critics):
        assert 0 <= rank < world and num_critics >= world, "need at least one critic per rank"
        base, rem = divmod(num_critics, world)
        sizes = [base + (1 if r < rem else 0) for r in range(world)]
        self.rank, self.world, self.num_critics = rank, world, num_critics
        self.lo = sum(sizes[:rank])
        self.n_local = sizes[rank]
        self.hi = self.lo + self.n_local

    def local_subset(self, global_ids):
        """local indices of the globally drawn REDQ subset members this rank owns."""
        return [j - self.lo for j in global_ids if self.lo <= j < self.hi]

    def owns(self, j):
        return self.lo <= j < self.hi


def install(agent, target_agent, shard):
    """mark both agents as holding shard `shard` of the global critic ensemble."""
    for ag in (agent, target_agent):
        assert ag.num_critics == shard.n_local, "agent must be built with the LOCAL number of critics"
        ag.ssac_shard = shard


def shard_of(agent):
    return getattr(agent, "ssac_shard", None)


def all_reduce_min(t):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t


def all_reduce_sum(t):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
