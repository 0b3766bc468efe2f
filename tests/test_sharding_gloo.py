"""CPU, world_size 2, gloo: the critic-sharded update (super_sac_amd.parallel) is the same
computation as the unsharded one.

No GPU here, so the per-shard arithmetic is done by the oracle; what is under test is the
product's host logic for N > 1 ranks -- the shard ownership map, the "draw the REDQ subset over
the GLOBAL ensemble, evaluate the locally owned members, MIN all-reduce the partial" protocol
(+inf when a rank owns none), the global loss denominator -- over a real collective.
The GPU counterpart (same protocol on the HIP kernels) is tests/test_hip_sharded.py.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _sharded_critic_sequence(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    import case_runner  # noqa: F401  (sets sys.path)
    import ssac_oracle as orc
    import synth
    from super_sac_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = synth.CASES["redq_small"]
    fx = case_runner.load_fixture("redq_small")
    shard = parallel.Shard(rank, world, cfg["N"])
    full = case_runner._oracle_agent(cfg).requires_grad_(True)
    target = full.clone()
    buf = orc.ReplayOracle(cfg["cap"])
    buf.load_experience(*case_runner._buffers(cfg))
    mine = [full.critics[0][j] for j in range(shard.lo, shard.hi)]
    tmine = [target.critics[0][j] for j in range(shard.lo, shard.hi)]
    opt = orc.AdamOracle([p[k] for p in mine for k in orc.MLP_KEYS], lr=cfg["lr"])
    la = torch.tensor([np.log(cfg["init_alpha"])], dtype=torch.float32)
    B = cfg["B"]
    tds = []
    for upd in range(int(fx["n_updates"])):
        o, a, r, o1, d = buf.gather(fx[f"u{upd}_idx0"])
        ids = [int(v) for v in fx[f"u{upd}_subset0"]]
        with torch.no_grad():
            a1, logp = orc.tanh_normal_sample(orc.mlp3(full.actors[0], o1["obs"])[0], cfg["lo"], cfg["hi"],
                                              torch.from_numpy(fx[f"u{upd}_eps0"]))
            local = shard.local_subset(ids)
            if local:
                part = torch.stack([orc.critic_q(tmine[j], o1["obs"], a1) for j in local], 0).min(0).values
            else:
                part = torch.full((B, 1), float("inf"))
            parallel.all_reduce_min(part)  # the exchange step
            td = r + cfg["gamma"] * (1.0 - d) * (part - la.exp() * logp)
        tds.append(td.numpy())
        loss = 0.0
        for p in mine:
            loss = loss + ((td - orc.critic_q(p, o["obs"], a)) ** 2).mean()
        loss = loss / cfg["N"]  # GLOBAL ensemble size in the denominator (learning.py:112)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if int(fx[f"u{upd}_polyak"]):
            orc.soft_update([p[k] for p in tmine for k in orc.MLP_KEYS],
                            [p[k] for p in mine for k in orc.MLP_KEYS], cfg["tau"])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), td=np.stack(tds),
             params=np.concatenate([p[k].detach().numpy().ravel() for p in mine for k in orc.MLP_KEYS]),
             lo=shard.lo, hi=shard.hi)
    dist.destroy_process_group()


def _unsharded_reference():
    import case_runner
    import ssac_oracle as orc
    import synth
    cfg = synth.CASES["redq_small"]
    fx = case_runner.load_fixture("redq_small")
    full = case_runner._oracle_agent(cfg).requires_grad_(True)
    target = full.clone()
    buf = orc.ReplayOracle(cfg["cap"])
    buf.load_experience(*case_runner._buffers(cfg))
    opt = orc.AdamOracle(full.critic_params(), lr=cfg["lr"])
    eopt = orc.AdamOracle([], lr=1e-4)
    la = [torch.tensor([np.log(cfg["init_alpha"])], dtype=torch.float32, requires_grad=True)]
    aug = orc.AugOracle("identity", cfg["B"])
    tds = []
    for upd in range(int(fx["n_updates"])):
        _, dicts = orc.critic_update(buf, full, target, opt, eopt, la, cfg["B"], cfg["gamma"], None, None,
                                     cfg["n"], None, None, False, aug, idx_list=[fx[f"u{upd}_idx0"]],
                                     eps_list=[torch.from_numpy(fx[f"u{upd}_eps0"])],
                                     subset_list=[[int(v) for v in fx[f"u{upd}_subset0"]]])
        tds.append(dicts[0]["td_target"].numpy())
        if int(fx[f"u{upd}_polyak"]):
            orc.soft_update(target.critic_params(), full.critic_params(), cfg["tau"])
    return np.stack(tds), [np.concatenate([c[k].detach().numpy().ravel() for k in orc.MLP_KEYS])
                           for c in full.critics[0]]


def test_shard_ownership_map():
    from super_sac_amd.parallel import Shard
    for n, w in ((10, 1), (10, 2), (10, 4), (10, 8), (16, 8), (3, 2)):
        owned = []
        for r in range(w):
            s = Shard(r, w, n)
            owned += list(range(s.lo, s.hi))
            assert s.n_local >= 1 and set(s.local_subset([s.lo, s.hi - 1])) == {0, s.n_local - 1}
        assert owned == list(range(n)), "every critic has exactly one owner"
    s = Shard(1, 2, 4)
    assert s.local_subset([0, 3]) == [1] and s.local_subset([0, 1]) == []
    with pytest.raises(AssertionError):
        Shard(0, 8, 4)


@pytest.mark.parametrize("world", [2, 4])   # 4 ranks over the fixture's ensemble: uneven shards, ranks that own no subset member
def test_sharded_critic_updates_equal_unsharded(tmp_path, world):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_sharded_critic_sequence, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    td_ref, params_ref = _unsharded_reference()
    for rank in range(world):
        got = np.load(tmp_path / f"rank{rank}.npz")
        assert np.allclose(got["td"], td_ref, atol=1e-6), "TD targets after the MIN all-reduce"
        want = np.concatenate(params_ref[int(got["lo"]):int(got["hi"])])
        assert np.max(np.abs(got["params"] - want)) < 1e-6, f"rank {rank}: owned critics diverged"


# ---------------------------------------------------------------------------------------------------------------
# SUNRISE member sharding (parallel.MemberShard): the members of the ensemble over the ranks
# ---------------------------------------------------------------------------------------------------------------
def test_member_shard_ownership_map():
    from super_sac_amd.parallel import MemberShard
    for e, w in ((5, 1), (5, 2), (5, 4), (5, 5), (3, 2), (8, 8)):
        owned = []
        for r in range(w):
            m = MemberShard(r, w, e)
            owned += list(range(m.lo, m.hi))
            assert m.n_local >= 1 and m.local(m.lo) == 0 and m.local(m.hi - 1) == m.n_local - 1
            assert m.local(m.hi % e) is None or w == 1
        assert owned == list(range(e)), "every member has exactly one owner"
    with pytest.raises(AssertionError):
        MemberShard(0, 4, 3)


def _member_sharded_sunrise_sequence(rank, world, port, out_dir):
    """the member-sharded critic update with the arithmetic done by the oracle: what is under test is the protocol --
    every rank gathers every member's batch, scores all of them with ITS members' target critics, the all-gather
    (parallel.all_gather_blocks over a real collective) completes the table, the weights of the owned members are formed
    from it, the loss is divided by the GLOBAL E * N"""
    sys.path.insert(0, HERE)
    import case_runner  # noqa: F401
    import ssac_oracle as orc
    import synth
    from super_sac_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = synth.CASES["sunrise"]
    fx = case_runner.load_fixture("sunrise")
    E, N, B = cfg["E"], cfg["N"], cfg["B"]
    ms = parallel.MemberShard(rank, world, E)
    full = case_runner._oracle_agent(cfg).requires_grad_(True)
    target = full.clone()
    buf = orc.ReplayOracle(cfg["cap"])
    buf.load_experience(*case_runner._buffers(cfg))
    mine = [p for i in range(ms.lo, ms.hi) for p in full.critics[i]]
    opt = orc.AdamOracle([p[k] for p in mine for k in orc.MLP_KEYS], lr=cfg["lr"])
    la = torch.tensor([np.log(cfg["init_alpha"])], dtype=torch.float32)
    tds, wts = [], []
    for upd in range(int(fx["n_updates"])):
        batches = [buf.gather(fx[f"u{upd}_idx{i}"]) for i in range(E)]   # same index draws on every rank
        table = torch.zeros(E, E, B)
        with torch.no_grad():
            for i, (o, a, r, o1, d) in enumerate(batches):
                for k in range(ms.lo, ms.hi):
                    table[i, k] = orc.ensemble_q(target.critics[k], o["obs"], a).view(-1)
        parallel.all_gather_blocks(table)   # the exchange step
        loss = 0.0
        for i in range(ms.lo, ms.hi):
            o, a, r, o1, d = batches[i]
            with torch.no_grad():
                w = torch.sigmoid(-table[i].std(0).view(B, 1) * cfg["temp"]) + 0.5
                a1, logp = orc.tanh_normal_sample(orc.mlp3(full.actors[i], o1["obs"])[0], cfg["lo"], cfg["hi"],
                                                  torch.from_numpy(fx[f"u{upd}_eps{i}"]))
                ids = [int(v) for v in fx[f"u{upd}_subset{i}"]]
                q1 = orc.ensemble_q(target.critics[i], o1["obs"], a1, subset_ids=ids)
                td = r + cfg["gamma"] * (1.0 - d) * (q1 - la.exp() * logp)
            tds.append(td.numpy()); wts.append(w.numpy())
            for p in full.critics[i]:
                loss = loss + (w * (td - orc.critic_q(p, o["obs"], a)) ** 2).mean()
        loss = loss / (E * N)   # GLOBAL ensemble in the denominator (learning.py:112)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if int(fx[f"u{upd}_polyak"]):
            for i in range(ms.lo, ms.hi):
                orc.soft_update([p[k] for p in target.critics[i] for k in orc.MLP_KEYS],
                                [p[k] for p in full.critics[i] for k in orc.MLP_KEYS], cfg["tau"])
    np.savez(os.path.join(out_dir, f"mrank{rank}.npz"), td=np.stack(tds), w=np.stack(wts), lo=ms.lo, hi=ms.hi,
             params=np.concatenate([p[k].detach().numpy().ravel() for p in mine for k in orc.MLP_KEYS]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_member_sharded_sunrise_updates_equal_reference_fixture(tmp_path, world):
    import case_runner
    import synth
    port = 29650 + (os.getpid() % 2000) + world
    mp.spawn(_member_sharded_sunrise_sequence, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    cfg = synth.CASES["sunrise"]
    fx = case_runner.load_fixture("sunrise")
    E, N = cfg["E"], cfg["N"]
    per_c = fx["final_critic"].size // E
    for rank in range(world):
        got = np.load(tmp_path / f"mrank{rank}.npz")
        lo, hi = int(got["lo"]), int(got["hi"])
        k = 0
        for upd in range(int(fx["n_updates"])):
            for i in range(lo, hi):
                assert np.allclose(got["td"][k], fx[f"u{upd}_td{i}"], atol=2e-5), (rank, upd, i)
                assert np.all(got["w"][k] >= 0.5) and np.all(got["w"][k] <= 1.0)
                k += 1
        want = fx["final_critic"][lo * per_c: hi * per_c]
        assert np.max(np.abs(got["params"] - want)) < 2e-6, f"rank {rank}: its members' critics diverged from the reference"


# ---------------------------------------------------------------------------------------------------------------
# the actor step's exchange steps on a critic-sharded job (SURVEY 8(e) "Collective -- actor step"; round 5: rank claim)
# ---------------------------------------------------------------------------------------------------------------
def _actor_routing_rank(rank, world, port, out_dir):
    """local arg-min -> MIN all-reduce -> claim (rank if the local minimum is the global one, else +inf) -> MIN all-reduce of
    the claims -> only the winner keeps its dQ/da row -> SUM all-reduce: what learning._online_actor_update issues around
    ssac_actor_route_local / _claim / _mask, with the kernels' arithmetic restated in torch and the REAL collectives"""
    sys.path.insert(0, HERE)
    from super_sac_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    N, B, A = 7, 200, 5                      # 7 critics over 3 ranks: shards of 3 / 2 / 2
    shard = parallel.Shard(rank, world, N)
    g = torch.Generator().manual_seed(77)    # every rank rebuilds the whole table, uses its own critics' rows
    q = torch.randn(N, B, generator=g)
    dxu = torch.randn(N, B, A, generator=g)
    q[3, :30] = q[0, :30] = q[:, :30].min(0).values - 1.0      # bit-equal minima on ranks 0 and 1
    q[6, 30:50] = q[4, 30:50] = q[:, 30:50].min(0).values - 1.0  # ... on ranks 1 and 2
    q[5, 50:60] = q[3, 50:60] = q[1, 50:60] = -40.0            # ... on all three
    ql, dl = q[shard.lo:shard.hi], dxu[shard.lo:shard.hi]
    qloc, am = ql.min(0)                                       # first local index on ties, as the kernel's scan
    dsel = dl[am, torch.arange(B)].clone()
    qglob = qloc.clone()
    parallel.all_reduce_min(qglob)
    claim = torch.where(qloc == qglob, torch.full((B,), float(rank)), torch.full((B,), float("inf")))
    parallel.all_reduce_min(claim)
    dsel[claim != float(rank)] = 0.0
    parallel.all_reduce_sum(dsel)
    am_glob = q.min(0).indices                                 # torch.min over ALL critics: the reference's routing
    ok = torch.equal(dsel, dxu[am_glob, torch.arange(B)]) and torch.equal(qglob, q.min(0).values)
    open(os.path.join(out_dir, f"route{rank}"), "w").write("ok" if ok else "WRONG")
    dist.destroy_process_group()


def test_sharded_actor_routing_with_ties_equals_torch_min(tmp_path):
    """three ranks, gloo: the routed action gradient of every row -- rows whose minimum is held bit-equal by two or three
    ranks included -- is the one torch.min over all critics routes (learning.py:402; before round 5 every tied rank kept
    its row and the SUM doubled the gradient)"""
    port = 29400 + (os.getpid() % 2000)
    mp.spawn(_actor_routing_rank, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert all((tmp_path / f"route{r}").read_text() == "ok" for r in range(3))
