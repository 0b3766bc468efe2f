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

2-35 KiB messages, latency bound.  On the GPUs they go through the one-shot exchange of csrc/ssac_xchg.hip (``Exchange``:
IPC-mapped receive buffers, one recordable launch per reduction, no host step between the launches of an update);
``torch.distributed`` (backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests) is the fallback and the set-up channel.
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

    def owner(self, j):
        """rank that holds global critic j"""
        base, rem = divmod(self.num_critics, self.world)
        return j // (base + 1) if j < rem * (base + 1) else rem + (j - rem * (base + 1)) // base

    def slot_code(self, j):
        """entry of a recorded update's id block for subset member j (include/ssac_hip.h, ssac_xchg_reduce_owned): the
        LOCAL index of a member this rank owns, else -(owner rank + 1) -- negative, which is all the kernels that skip a
        foreign slot look at, and it tells the exchange who will send"""
        return j - self.lo if self.owns(j) else -(self.owner(j) + 1)


def install(agent, target_agent, shard):
    """mark both agents as holding shard `shard` of the global critic ensemble."""
    for ag in (agent, target_agent):
        assert ag.num_critics == shard.n_local, "agent must be built with the LOCAL number of critics"
        ag.ssac_shard = shard


def shard_of(agent):
    return getattr(agent, "ssac_shard", None)


class MemberShard:
    """SUNRISE variant of SURVEY.md 8(e): with ``ensemble_size`` E > 1 the MEMBERS are independent -- own actor, own
    critics, own temperature, own replay batch (learning.py:47-117 loops over them; experiments/gym/sunrise.gin:6-9 has
    E 5 x 2 critics, so sharding the critics of a member could use 2 GPUs at most) -- and rank g holds the members
    [lo, hi).  The replay buffer is replicated and every rank makes EVERY member's host draws (replay indices, REDQ
    subset, action noise) in the reference's order, so the generators stay in step and every rank can gather every
    member's batch locally.  The one exchange step: ``compute_backup_weights("sunrise")`` needs the target Q of ALL
    members on a member's (s, a) (learning_utils.py:372-382) -- every rank evaluates ITS members' target critics on all E
    batches, one all-gather (a SUM of disjointly filled (E x E x B) blocks) completes the table, each rank forms the
    weights of its own members.  Actor and temperature updates are member-local: no exchange."""

    def __init__(self, rank, world, ensemble_size):
        assert 0 <= rank < world and ensemble_size >= world, "need at least one ensemble member per rank"
        base, rem = divmod(ensemble_size, world)
        sizes = [base + (1 if r < rem else 0) for r in range(world)]
        self.rank, self.world, self.ensemble_size = rank, world, ensemble_size
        self.lo = sum(sizes[:rank])
        self.n_local = sizes[rank]
        self.hi = self.lo + self.n_local

    def owns(self, i):
        return self.lo <= i < self.hi

    def local(self, i):
        """local index of global member i, None when another rank holds it"""
        return i - self.lo if self.owns(i) else None


def install_members(agent, target_agent, member_shard):
    """mark both agents as holding the members [lo, hi) of the global ensemble (they are built with the LOCAL
    ``ensemble_size``; the ``log_alphas`` / temperature optimizers handed to the update functions are the local members')"""
    for ag in (agent, target_agent):
        assert ag.ensemble_size == member_shard.n_local, "agent must be built with the LOCAL number of ensemble members"
        assert shard_of(ag) is None, "members OR the critics of one member are sharded, not both"
        ag.ssac_member_shard = member_shard


def member_shard_of(agent):
    return getattr(agent, "ssac_member_shard", None)


def all_gather_blocks(t):
    """every rank has filled ITS block of `t` and zeroed the rest: the SUM over ranks is the all-gather (one exchange
    launch / one collective, rank-ordered sums of one non-zero term each: identical bits everywhere)"""
    return all_reduce_sum(t)


class Exchange:
    """the one-shot exchange of csrc/ssac_xchg.hip: IPC-mapped receive buffers on every rank, one recordable launch
    per reduction.  Built once per process after ``torch.distributed`` is up (the 64-byte IPC handles travel through
    an all_gather of the process group -- a set-up step, not the data path)."""

    def __init__(self, rank, world, max_floats, device):
        import ctypes as C
        from ._lib import check, lib
        self.rank, self.world, self.max_floats = rank, world, int(max_floats)
        torch.cuda.set_device(device)
        # do all ranks sit on one device (the one-GPU test box)?  Only then may the receive buffer be ordinary cached
        # memory; across devices the library insists on uncached memory and otherwise fails -- on EVERY rank alike, so
        # the job falls back to the collective as a whole (csrc/ssac_xchg.hip)
        props = torch.cuda.get_device_properties(device)
        me = str(getattr(props, "uuid", None) or getattr(props, "pci_bus_id", None) or torch.device(device).index)
        ids = [None] * world
        dist.all_gather_object(ids, me)
        shared = len(set(ids)) == 1
        h = lib.ssac_xchg_create(rank, world, self.max_floats, 1 if shared else 0)
        made = [None] * world
        dist.all_gather_object(made, None if h else lib.ssac_last_error().decode())
        if any(m is not None for m in made):
            if h:
                lib.ssac_xchg_destroy(h)
            self.handle = None
            raise RuntimeError(f"one-shot exchange unavailable: {dict((r, m) for r, m in enumerate(made) if m is not None)}")
        self.handle = h
        nb = int(lib.ssac_xchg_handle_bytes())
        mine = C.create_string_buffer(nb)
        check(lib.ssac_xchg_handle(h, mine))
        gathered = [None] * world
        dist.all_gather_object(gathered, bytes(mine.raw))
        # every rank must reach the same verdict (a rank that raised here while its peers went on would leave them in
        # a collective nobody else joins): connect, then ONE exchange of known values, and the ranks vote
        err = None
        if lib.ssac_xchg_connect(h, b"".join(gathered)) != 0:
            err = "connect: " + lib.ssac_last_error().decode()
        dist.barrier()
        votes = [None] * world
        dist.all_gather_object(votes, err)
        if all(v is None for v in votes):
            probe = torch.full((16,), float(rank + 1), device=device)
            probe_sum = probe.clone()
            self.reduce(probe, 0)
            self.reduce(probe_sum, 1)
            torch.cuda.synchronize(device)
            if self.failed():
                err = "probe: a peer's flag did not arrive"
            elif not (bool((probe == 1.0).all()) and bool((probe_sum == world * (world + 1) / 2).all())):
                err = "probe: wrong reduction"
            dist.all_gather_object(votes, err)
        if any(v is not None for v in votes):
            bad = {r: v for r, v in enumerate(votes) if v is not None}
            lib.ssac_xchg_destroy(h)
            self.handle = None
            raise RuntimeError(f"one-shot exchange unavailable on ranks {bad}")

    def reduce(self, t, op):
        from . import engine
        from ._lib import check, lib
        check(lib.ssac_xchg_reduce(self.handle, t.data_ptr(), t.numel(), op, engine.stream()))

    def reduce_min_owned(self, t, ids_dev, n_slots, n_parts=1):
        from . import engine
        from ._lib import check, lib
        # (n_parts > 1: t holds every slot's values as n_parts partial sums -- the payload is their sum, n_slots x B)
        check(lib.ssac_xchg_reduce_owned(self.handle, t.data_ptr(), t.numel() // n_parts, ids_dev.data_ptr(), n_slots,
                                         n_parts, engine.stream()))

    def failed(self):
        from ._lib import lib
        return bool(lib.ssac_xchg_error(self.handle))


_exchange = None
# issue the collective even in a process group of ONE rank (bench.py's SSAC_BENCH_FORCE_DIST=1: the RCCL check a
# one-GPU box can run -- communicator set-up and ncclMin / ncclSum all-reduce through the fallback path)
FORCE_COLLECTIVE = False
ONE_SHOT_MAX_FLOATS = 1 << 15  # largest payload the receive slots hold (the actor step's (B x A) action gradient)


def enable_one_shot(device, max_floats=ONE_SHOT_MAX_FLOATS):
    """switch the exchange steps of this process to the IPC one-shot kernel (needs an initialised process group whose
    ranks can map each other's device memory: one node, HSA_ENABLE_IPC_MODE_LEGACY=0 on this platform)"""
    global _exchange
    if _exchange is None and dist.is_initialized() and dist.get_world_size() > 1:
        _exchange = Exchange(dist.get_rank(), dist.get_world_size(), max_floats, device)
    return _exchange


def disable_one_shot():
    """back to the collective (torch.distributed all_reduce) for the exchange steps of this process.  Every rank of the job
    must do the same, and update functions recorded with the one-shot kernel in their launch lists must be recorded again
    (new agent / new recording).  The exchange object itself is left to the end of the process: a peer may still have
    its buffers mapped."""
    global _exchange, _retired
    if _exchange is not None:
        _retired.append(_exchange)
        _exchange = None


_retired = []


def exchange_failed():
    """True when an exchange kernel of this process gave up waiting for a peer (bounded spin, csrc/ssac_xchg.hip): the
    reductions since then are not reductions (and were poisoned with NaN).  Reads and CLEARS a pinned host word; for the
    verdict of a particular exchange synchronise the stream first."""
    return _exchange is not None and _exchange.failed()


def check_exchange():
    """raise when an exchange of this process gave up on a peer.  The error word is pinned host memory (no device
    synchronisation), so the update functions call this every few updates: a run whose reductions stopped being
    reductions (the kernel poisons them with NaN) ends here instead of training on."""
    if _exchange is not None and _exchange.failed():
        raise RuntimeError("one-shot exchange: a peer rank's flag did not arrive within the spin bound (is a rank of "
                           "the job gone?); the reductions since then were poisoned with NaN")


def one_shot_ready(t):
    x = _exchange
    return (x is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
            and t.numel() <= x.max_floats)


def all_reduce_min(t):
    if one_shot_ready(t):
        check_exchange()  # (of the exchanges issued so far: a host load)
        _exchange.reduce(t, 0)
    elif dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVE):
        dist.all_reduce(t, op=dist.ReduceOp.MIN)  # fallback: RCCL on the GPUs, gloo in the CPU tests
    return t


def all_reduce_min_owned(t, ids_dev, n_slots, n_parts=1):
    """MIN all-reduce of the (n_slots x B) target-critic outputs of a recorded sharded update, where only the ranks
    that own a subset member send (the id block `ids_dev` says who: Shard.slot_code).  n_parts > 1: t is
    (n_slots x n_parts x B), a slot's value the sum of its parts (column-split target critics); the reduced value comes
    back in part 0, the other parts zeroed."""
    check_exchange()
    _exchange.reduce_min_owned(t, ids_dev, n_slots, n_parts)
    return t


def all_reduce_sum(t):
    if one_shot_ready(t):
        check_exchange()
        _exchange.reduce(t, 1)
    elif dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVE):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
