"""GPU: the critic-sharded update on the real kernels.  Two ranks share the one GPU of the test box
(RCCL refuses two ranks on one device, so the collective runs over gloo with device tensors); every
rank replays the reference fixture with HALF of the critic ensemble and must reproduce the
reference's TD targets, its own critics' final parameters and the (replicated) actor."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


RETRIED = []   # (test id, message) of every spin-bound retry of this session: reported at the end of the run (conftest)


def _spawn(fn, args, nprocs, retries=1):
    """mp.spawn for ranks that SHARE the test box's one GPU.  Their one-shot exchange kernels spin on each other's flags
    while the device time-slices the processes' queues; once in a while (documented since round 3: about one run of the
    suite in six on a loaded box) a wait sits out the whole spin bound and the exchange reports it as designed -- error
    word, NaN-poisoned result, RuntimeError("... did not arrive within the spin bound").  That outcome says something
    about the box's scheduler, not about the protocol (one rank per GPU -- the product layout -- has no time-slicing), so
    THAT failure alone is retried (once; the stand-alone exchange test up to three times) on a fresh rendezvous port; every
    other failure propagates at once.
    A retry is never silent and never sees the failed attempt's files: it is recorded in RETRIED + a warning (pytest's
    summary shows it), and the attempt's output directory (the str argument that is an existing directory) is emptied
    first, so ok*/rec* files of the failed attempt cannot satisfy the caller's assertions."""
    import shutil
    import warnings
    port_arg = [i for i, a in enumerate(args) if isinstance(a, int) and 20000 <= a < 65000][-1]
    dir_args = [a for a in args if isinstance(a, str) and os.path.isdir(a)]
    for attempt in range(retries + 1):
        try:
            mp.spawn(fn, args=tuple(args), nprocs=nprocs, join=True)
            return
        except Exception as e:   # noqa: BLE001  (torch.multiprocessing.spawn.ProcessRaisedException)
            if attempt == retries or "within the spin bound" not in str(e):
                raise
            msg = (f"{fn.__name__}: exchange wait sat out the spin bound on the shared test GPU; retry {attempt + 1} "
                   f"of {retries}")
            RETRIED.append((os.environ.get("PYTEST_CURRENT_TEST", "?"), msg))
            warnings.warn(msg)
            for d in dir_args:
                for f in os.listdir(d):
                    fp = os.path.join(d, f)
                    shutil.rmtree(fp) if os.path.isdir(fp) else os.remove(fp)
            args = list(args)
            args[port_arg] += 97


def _rank_main(rank, world, port, out_dir, one_shot=False):
    sys.path.insert(0, HERE)
    import case_runner
    import synth
    import torch.distributed as dist
    from super_sac_amd import parallel
    import super_sac_amd as ssa
    if one_shot == 2:   # the exchange as a launch of its own (round 4's form) instead of the chained launch's tail workgroup
        ssa.learning_utils.FUSE_XCHG = False
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    if one_shot:
        assert parallel.enable_one_shot(torch.device("cuda:0")) is not None
    cfg = synth.CASES["redq_small"]
    shard = parallel.Shard(rank, world, cfg["N"])
    rec = case_runner.run_engine("redq_small", device="cuda:0", shard=shard)
    fx = case_runner.slice_fixture(case_runner.load_fixture("redq_small"), cfg, shard)
    worst = case_runner.compare(rec, fx, who=f"hip-sharded[rank {rank}]")
    assert not parallel.exchange_failed()
    np.savez(os.path.join(out_dir, f"ok{rank}.npz"), **{k: np.float64(v) for k, v in worst.items()})
    np.savez(os.path.join(out_dir, f"rec{rank}.npz"), **{k: np.asarray(v) for k, v in rec.items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("world,one_shot", [(2, False), (4, False), (4, True), (2, True)],
                         ids=["2-ranks-collective", "4-ranks-collective", "4-ranks-one-shot", "2-ranks-one-shot"])
def test_sharded_sequence_matches_reference(tmp_path, world, one_shot):
    """2 and 4 ranks on the one device (4: one critic per rank, two ranks per update own no member of the drawn
    subset), through the gloo collective and through the one-shot exchange -- which, round 5, runs in a TAIL WORKGROUP of
    the chained launch (behind its target-critic workgroups, which count themselves in) instead of as a launch of its own;
    both forms of it land on the fixture and on each other BIT FOR BIT (same protocol, same rank-ordered reduction)."""
    port = 29700 + (os.getpid() % 2000) + 7 * world + int(one_shot)
    _spawn(_rank_main, (world, port, str(tmp_path), one_shot), world)
    for rank in range(world):
        assert (tmp_path / f"ok{rank}.npz").exists()
    if one_shot:
        fused = [dict(np.load(tmp_path / f"rec{rank}.npz")) for rank in range(world)]
        sep_dir = tmp_path / "separate"
        sep_dir.mkdir()
        _spawn(_rank_main, (world, port + 1, str(sep_dir), 2), world)
        for rank in range(world):
            sep = dict(np.load(sep_dir / f"rec{rank}.npz"))
            assert sorted(sep) == sorted(fused[rank])
            for k in sep:
                assert np.array_equal(sep[k], fused[rank][k]), (rank, k)


# ---------------------------------------------------------------------------------------------------------------
# the one-shot exchange (csrc/ssac_xchg.hip): IPC-mapped receive buffers, one recordable launch per reduction
# ---------------------------------------------------------------------------------------------------------------
def _xchg_main(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    from super_sac_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    x = parallel.enable_one_shot(torch.device("cuda:0"), max_floats=9000)
    assert x is not None
    ok = True
    for it in range(40):
        for n, op in ((1024, 0), (8704, 1), (3, 0), (512, 1)):
            g = torch.Generator().manual_seed(1000 * it + n)
            parts = [torch.randn(n, generator=g) for _ in range(world)]   # every rank can rebuild every partial
            if op == 0:
                parts[it % world][::7] = float("inf")   # (+inf: "this rank owns no subset member")
            t = parts[rank].cuda()
            (parallel.all_reduce_min if op == 0 else parallel.all_reduce_sum)(t)
            want = parts[0].clone()
            for p in parts[1:]:
                want = torch.minimum(want, p) if op == 0 else want + p
            ok = ok and torch.equal(t.cpu(), want)   # rank-ordered reduction: identical bits everywhere
    # owners-only form (ssac_xchg_reduce_owned): a 2-slot id block per round names the ranks that send -- both slots
    # on one rank, one each, or (world > 2) on two ranks this rank is neither of; non-owners' partials are +inf
    for it in range(30):
        g = torch.Generator().manual_seed(5000 + it)
        owners = [int(v) for v in torch.randint(0, world, (2,), generator=g)]
        parts = [torch.randn(2, 300, generator=g) for _ in range(world)]
        for r in range(world):
            for j in range(2):
                if owners[j] != r:
                    parts[r][j] = float("inf")
        ids = torch.tensor([j if owners[j] == rank else -(owners[j] + 1) for j in range(2)], dtype=torch.int32).cuda()
        t = parts[rank].cuda()
        parallel.all_reduce_min_owned(t, ids, 2)
        want = parts[0].clone()
        for p_ in parts[1:]:
            want = torch.minimum(want, p_)
        ok = ok and torch.equal(t.cpu(), want) and bool(torch.isfinite(want).all())
    # ... and with every slot's values in TWO partial sums (column-split target critics: ssac_td_spec.n_parts): the
    # payload is the sum of a slot's parts, the reduced value comes back in part 0 with the other part zeroed
    for it in range(20):
        g = torch.Generator().manual_seed(9000 + it)
        owners = [int(v) for v in torch.randint(0, world, (2,), generator=g)]
        parts = [torch.randn(2, 2, 300, generator=g) for _ in range(world)]   # [slot][part][row]
        for r in range(world):
            for j in range(2):
                if owners[j] != r:
                    parts[r][j, 0] = float("inf")
                    parts[r][j, 1] = 0.0
        ids = torch.tensor([j if owners[j] == rank else -(owners[j] + 1) for j in range(2)], dtype=torch.int32).cuda()
        t = parts[rank].clone().cuda()
        parallel.all_reduce_min_owned(t, ids, 2, 2)
        want = (parts[0][:, 0] + parts[0][:, 1])
        for p_ in parts[1:]:
            want = torch.minimum(want, p_[:, 0] + p_[:, 1])
        got = t.cpu()
        ok = ok and torch.equal(got[:, 0], want) and bool((got[:, 1] == 0).all()) and bool(torch.isfinite(want).all())
    assert not x.failed() and ok
    open(os.path.join(out_dir, f"xok{rank}"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3])
def test_one_shot_exchange_ranks_on_one_device(tmp_path, world):
    # (the STAND-ALONE exchange kernels of ranks that share the box's one GPU: the test most exposed to the device's
    #  time-slicing of processes -- round 6 saw both parametrisations sit out the 60 s bound in one lease and none in the
    #  next five.  Up to three loud retries; the in-launch form's tests keep a single one.)
    port = 29900 + (os.getpid() % 2000) + world
    _spawn(_xchg_main, (world, port, str(tmp_path)), world, retries=3)
    assert all((tmp_path / f"xok{r}").exists() for r in range(world))


# ---------------------------------------------------------------------------------------------------------------
# slot reuse of the owners-only exchange: a rank that owns no subset member for many updates and whose host stalls
# ---------------------------------------------------------------------------------------------------------------
LAP_ROUNDS = 12      # consecutive exchanges owned by ranks 0 and 1 alone (X_SLOTS = 4: the ring is lapped twice over)


def _lap_main(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    import time
    import torch.distributed as dist
    from super_sac_amd import parallel
    from super_sac_amd._lib import check, lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    x = parallel.enable_one_shot(torch.device("cuda:0"), max_floats=2048)
    assert x is not None
    legacy_lib = not hasattr(lib, "ssac_xchg_reduce_owned")   # (only in a failing-first run against a pre-round-3 library)
    n = 2 * 512
    owners = (0, 1)   # slot 0 is rank 0's, slot 1 rank 1's -- in EVERY round: rank 2 never sends
    ids = torch.tensor([j if owners[j] == rank else -(owners[j] + 1) for j in range(2)], dtype=torch.int32).cuda()
    want, parts_dev = [], []
    for it in range(LAP_ROUNDS):
        g = torch.Generator().manual_seed(7000 + it)
        parts = [torch.randn(2, 512, generator=g) for _ in range(world)]
        for r in range(world):
            for j in range(2):
                if owners[j] != r:
                    parts[r][j] = float("inf")
        w = parts[0].clone()
        for p_ in parts[1:]:
            w = torch.minimum(w, p_)
        want.append(w)
        parts_dev.append(parts[rank])

    def run(delay_rank2):
        """LAP_ROUNDS exchanges issued back to back with NO host synchronisation in between (the training loop's
        regime); rank 2's host sleeps before its first launch and 5 ms between launches when asked to"""
        ts = [p.clone().cuda() for p in parts_dev]
        torch.cuda.synchronize()
        dist.barrier()
        for it in range(LAP_ROUNDS):
            if delay_rank2 and rank == 2:
                time.sleep(0.05 if it == 0 else 0.005)
            check(lib.ssac_xchg_reduce_owned(x.handle, ts[it].data_ptr(), n, ids.data_ptr(), 2, 1,
                                             torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        dist.barrier()
        return [t.cpu() for t in ts]

    lockstep = run(False)
    assert all(torch.equal(a, b) for a, b in zip(lockstep, want)) and not x.failed()
    delayed = run(True)
    verdict = {"rank": rank, "legacy_lib": legacy_lib,
               "delayed_equal": [bool(torch.equal(a, b)) for a, b in zip(delayed, lockstep)],
               "delayed_failed_flag": bool(x.failed())}
    # (the two protocol replays below need the LAB build, `./build.sh --lab` + SSAC_LAB_BUILD=1: the product library does not
    #  define ssac_xchg_test_mode at all, include/ssac_hip_test.h -- a production exchange cannot be switched to round 3's
    #  unsafe protocol; tools/gpu_suite_lab.sh is the lab leg that runs them)
    lab = hasattr(lib, "ssac_xchg_test_mode") and lib.ssac_xchg_test_mode(x.handle, 3) == 0
    verdict["lab_build"] = bool(lab)
    if lab:
        # (a) the round-3 protocol on today's kernel (no reuse wait, flag >= seq accepted): the hazard, demonstrated --
        # the delayed non-owner reduces LATER exchanges' payloads and nothing notices
        check(lib.ssac_xchg_test_mode(x.handle, 3))
        old = run(True)
        verdict["r3_protocol_equal"] = [bool(torch.equal(a, b)) for a, b in zip(old, lockstep)]
        verdict["r3_protocol_failed_flag"] = bool(x.failed())
        verdict["r3_protocol_finite"] = bool(all(torch.isfinite(t).all() for t in old))
        # (b) senders that skip the reuse wait against today's receivers: the lap is DETECTED -- poisoned result + error
        # word on the lapped rank.  (Last: the exchange of a rank that detected a lap stays dead.)
        check(lib.ssac_xchg_test_mode(x.handle, 1))
        det = run(True)
        verdict["detect_failed_flag"] = bool(x.failed())
        verdict["detect_nan_rounds"] = [bool(torch.isnan(t).all()) for t in det]
        verdict["detect_wrong_but_finite"] = [bool(torch.isfinite(t).all() and not torch.equal(t, w))
                                              for t, w in zip(det, lockstep)]
    import json
    json.dump(verdict, open(os.path.join(out_dir, f"lap{rank}.json"), "w"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_owners_only_exchange_survives_a_stalled_non_owner(tmp_path):
    """VERDICT round 3, weak #2 / ADVICE (high): with the owners-only exchange a rank that owns no member of the drawn
    subset wrote no flag and acknowledged nothing, so the senders could lap it (slot = seq % 4) and its poll `flag >= seq`
    then reduced a LATER update's target-Q.  Three ranks on the one device; ranks 0 and 1 own both subset slots for 12
    consecutive exchanges; rank 2's host sleeps 50 ms before its first launch and 5 ms between launches, nobody
    synchronises in between.  Today: bit-identical to the lock-step run on every rank (the senders wait for rank 2's
    acknowledgement after running 4 exchanges ahead).

    Failing-first evidence (this very test against round 3's library 21bde52, gpurun of 2026-10-03,
    profiles/r4_xchg_lap_evidence.md + profiles/r4_raw/xchg_lap_r3lib.log): `AssertionError: (2, {'delayed_equal': [False,
    False, False, False, False, False, ...], 'delayed_failed_flag': False, 'legacy_lib': True, 'rank': 2})` -- rank 2's
    results were finite, wrong and unflagged: silently wrong TD targets.  The same is reproduced in every CI run through
    `ssac_xchg_test_mode(3)` (round 3's protocol on today's kernel), and `ssac_xchg_test_mode(1)` shows that a lap,
    should one ever happen again, is detected by the receiver (flag > seq: NaN-poisoned result + error word) -- both in
    the LAB build only (`SSAC_LAB_BUILD=1 pytest ...` after `./build.sh --lab`); the product library refuses the switch."""
    import json
    world = 3
    port = 30300 + (os.getpid() % 2000)
    mp.spawn(_lap_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    v = [json.load(open(tmp_path / f"lap{r}.json")) for r in range(world)]
    for r in range(world):
        assert all(v[r]["delayed_equal"]), (r, v[r])
        assert not v[r]["delayed_failed_flag"], (r, v[r])
    if v[2]["legacy_lib"] or not v[2]["lab_build"]:
        return   # (product library: the replays of round 3's protocol are refused; profiles/r4_xchg_lap_evidence.md holds them)
    # round 3's protocol: the stalled rank is silently wrong in most rounds (only the last X_SLOTS survive in the ring)
    assert sum(not e for e in v[2]["r3_protocol_equal"]) >= LAP_ROUNDS - 4, v[2]
    assert v[2]["r3_protocol_finite"] and not v[2]["r3_protocol_failed_flag"], v[2]
    assert all(v[0]["r3_protocol_equal"]) and all(v[1]["r3_protocol_equal"])
    # detection: every lapped round of rank 2 is NaN and flagged, none is finite-but-wrong
    assert v[2]["detect_failed_flag"] and sum(v[2]["detect_nan_rounds"]) >= LAP_ROUNDS - 4, v[2]
    assert not any(v[2]["detect_wrong_but_finite"]), v[2]


# ---------------------------------------------------------------------------------------------------------------
# SUNRISE member sharding (parallel.MemberShard, SURVEY 8(e) "SUNRISE variant"): the ensemble MEMBERS over the ranks
# ---------------------------------------------------------------------------------------------------------------
def _member_main(rank, world, port, out_dir, name, one_shot):
    sys.path.insert(0, HERE)
    import case_runner
    import synth
    import torch.distributed as dist
    from super_sac_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    if one_shot:
        assert parallel.enable_one_shot(torch.device("cuda:0")) is not None
    cfg = synth.CASES[name]
    ms = parallel.MemberShard(rank, world, cfg["E"])
    rec = case_runner.run_engine(name, device="cuda:0", members=ms)
    fx = case_runner.slice_fixture_members(case_runner.load_fixture(name), cfg, ms)
    assert any(k.endswith(f"_td{ms.lo}") for k in fx) and not any(k.endswith(f"_td{(ms.hi) % cfg['E']}") for k in fx
                                                                  if not ms.owns(ms.hi % cfg["E"]))
    worst = case_runner.compare(rec, fx, who=f"hip-member-sharded[{name}, rank {rank}/{world}]")
    assert not parallel.exchange_failed()
    np.savez(os.path.join(out_dir, f"mok{rank}.npz"), **{k: np.float64(v) for k, v in worst.items()})
    dist.destroy_process_group()


def _member_stock_main(rank, world, port, out_dir, name):
    sys.path.insert(0, HERE)
    import case_runner
    import synth
    import torch.distributed as dist
    from super_sac_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ms = parallel.MemberShard(rank, world, synth.CASES[name]["E"])
    res = case_runner.run_engine_stock(name, device="cuda:0", members=ms)
    assert not parallel.exchange_failed()
    np.savez(os.path.join(out_dir, f"mstock{rank}.npz"), **res)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_member_sharded_stock_generators_match_the_unsharded_run(tmp_path):
    """Round-4 advisor (medium): with the STOCK generators a member-sharded rank seeded the in-kernel Philox stream of its
    chained actor update with the LOCAL member index -- members at the same local position on different ranks got identical
    noise, all of them a different one than in the unsharded run -- and consumed a device-generator draw for every member
    it does not own although the owner's launch draws nothing, so the ranks' generators drifted apart.  Three cycles of the
    `sunrise` case (E 3, members split 2 + 1) with NO injected draws, two ranks on the device against the unsharded engine
    in this process on the same seeds: actors, critics and temperatures of every member agree (the fixtures run through
    draw hooks and could not see this)."""
    import case_runner
    import synth
    name, world = "sunrise", 2
    ref = case_runner.run_engine_stock(name)
    port = 30900 + (os.getpid() % 2000)
    _spawn(_member_stock_main, (world, port, str(tmp_path), name), world)
    got = {}
    for r in range(world):
        got.update(dict(np.load(tmp_path / f"mstock{r}.npz")))
    E = synth.CASES[name]["E"]
    assert sorted(got) == sorted(ref) and len(got) == 3 * E
    for k in sorted(ref):
        err = float(np.max(np.abs(np.asarray(got[k], np.float64) - np.asarray(ref[k], np.float64))))
        assert err <= 2e-6, (k, err)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("name,world,one_shot", [("sunrise", 2, False), ("sunrise", 3, True), ("sunrise_discrete", 2, True),
                                                 ("sunrise_discrete", 3, False), ("softmax_weights", 2, True),
                                                 ("softmax_weights", 3, False), ("softmax_discrete", 2, True)],
                         ids=["sunrise-2-collective", "sunrise-3-one-shot", "discrete-2-one-shot", "discrete-3-collective",
                              "softmax-2-one-shot", "softmax-3-collective", "softmax-discrete-2-one-shot"])
def test_member_sharded_sunrise_matches_reference(tmp_path, name, world, one_shot):
    """E = 3 members over 2 ranks (2 + 1) and over 3 ranks (one each), the ranks sharing the one device: every rank makes
    every member's host draws, gathers all three batches, scores them with ITS members' target critics; the all-gather of
    the (batch x member x row) table -- through gloo and through the one-shot exchange kernel -- gives each rank the
    SUNRISE weights of its own members (learning_utils.py:372-382).  Each rank's TD targets, TD / temperature logs, critic,
    target, actor parameters, Adam moments and temperatures land on the REFERENCE fixture's slices for its members;
    sunrise_discrete also clips the critics' and actors' gradients by the norm over ALL members (one scalar all-reduce).
    Round 5: the "softmax" weights too (learning_utils.py:383-393: every member's ONLINE actor samples a' on every member's
    next states and its online critics score it) -- every rank makes all E x E policy draws in the reference's order, fills
    its members' rows of the table, one all-gather, B softmax_b(-std_k T) for its own members; continuous and discrete."""
    port = 31300 + (os.getpid() % 2000) + 11 * world + int(one_shot)
    _spawn(_member_main, (world, port, str(tmp_path), name, one_shot), world)
    assert all((tmp_path / f"mok{r}.npz").exists() for r in range(world))


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 5 at 4 and 8 ranks (the ranks sharing the one device): the value check bench.py --gpus N runs before
# it times, under pytest and at the HUMANOID shape (only builder-side bench logs, shape M)
# ---------------------------------------------------------------------------------------------------------------
def _config5_main(rank, world, port, out_dir, one_shot):
    sys.path.insert(0, os.path.dirname(HERE))
    import json
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.set_num_threads(max(1, 16 // world))   # (eight ranks initialising 16 orthogonal critics each: see bench.py)
    import bench
    import super_sac_amd as ssa
    from super_sac_amd import parallel
    bench.OBS, bench.ACT, bench.NCRIT = 376, 17, 16      # Humanoid-v4, N 16 (BASELINE config 5)
    bench.synth_data.__defaults__ = (376, 17)
    device = torch.device("cuda:0")
    if one_shot:
        assert parallel.enable_one_shot(device) is not None
    shard = parallel.Shard(rank, world, bench.NCRIT)
    step, _, _ = bench.build_engine(device, shard.n_local, shard)
    res = bench.sharded_value_check(step, ssa, device, shard, dist, n_updates=6)
    assert res is not None and not parallel.exchange_failed()
    json.dump(res, open(os.path.join(out_dir, f"c5_{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,one_shot", [(4, True), (8, False)], ids=["4-ranks-one-shot", "8-ranks-collective"])
def test_config5_humanoid_n16_value_check_at_4_and_8_ranks(tmp_path, world, one_shot):
    """obs 376 / act 17 / N 16 / B 512, 4 critics per rank (one-shot exchange) and 2 per rank (eight processes sharing the
    device go through the gloo collective: bench.py's own default there): 6 recorded updates per rank against the UNSHARDED
    engine run by rank 0 on the same seeds -- TD targets <= 2e-5, this rank's parameters / targets / moments <= 3e-5
    (bench.sharded_value_check raises otherwise).  A rank with 2 or 4 critics takes the latency form of the
    weight-gradient launch and the 16-row tile variants: the check is also those variants against the one-GPU kernels."""
    import json
    port = 32300 + (os.getpid() % 2000) + world
    _spawn(_config5_main, (world, port, str(tmp_path), one_shot), world)
    res = json.load(open(tmp_path / "c5_0.json"))
    assert res["updates"] == 6 and res["max_abs_diff"]["td"] <= 2e-5, res


def _humanoid_main(rank, world, port, out_dir):
    """BASELINE config 5's shape (obs 376 / act 17 / N 16 / B 512), critics sharded over two ranks that share this
    GPU: critic updates replayed from ONE launch list per rank (the exchange is a recorded launch), Polyak, actor and
    temperature updates -- against the UNSHARDED engine run by the same process, and the reference fixture."""
    sys.path.insert(0, HERE)
    import case_runner
    import synth
    import torch.distributed as dist
    from super_sac_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    name = "redq_S"
    cfg = synth.CASES[name]
    full = case_runner.run_engine(name, device="cuda:0")
    case_runner.compare(full, case_runner.load_fixture(name), who=f"hip[{name}] rank {rank} unsharded")
    assert parallel.enable_one_shot(torch.device("cuda:0")) is not None
    shard = parallel.Shard(rank, world, cfg["N"])
    import super_sac_amd as ssa
    old = ssa.learning.GRAPH_WARMUP
    ssa.learning.GRAPH_WARMUP = 0   # record on the first call: the 2 critic updates of the case replay a launch list
    try:
        rec = case_runner.run_engine(name, device="cuda:0", shard=shard)
    finally:
        ssa.learning.GRAPH_WARMUP = old
    assert not parallel._exchange.failed()
    # TD targets: the MIN over ranks of the per-shard target-critic outputs is the unsharded subset min -- bit for bit
    # on the first update; afterwards the shards' critics went through the 16-row tile variant (8 nets per rank fit one
    # round of 16-row workgroups, 16 nets do not) whose fp32 summation order differs in the last bits
    import re
    for key in full:
        if re.fullmatch(r"u\d+_td\d+", key):
            if key.startswith("u0_"):
                assert np.array_equal(rec[key], full[key]), key
            assert np.max(np.abs(rec[key] - full[key])) <= 2e-5, (key, float(np.max(np.abs(rec[key] - full[key]))))
    in_dim, H = cfg["obs"] + cfg["act"], cfg["hidden"]
    sizes = [H * in_dim, H, H * H, H, H, 1]
    perfp = sum(min(48, n) for n in sizes)
    for key in ("finalfp_critic", "finalfp_target_critic", "finalfp_critic_m", "finalfp_critic_v"):
        mine = full[key][shard.lo * perfp: shard.hi * perfp]
        tol = 1e-6 if not key.endswith("_v") else 1e-9
        assert np.max(np.abs(rec[key] - mine)) <= tol, (key, float(np.max(np.abs(rec[key] - mine))))
    # the actor step: MIN over all critics and SUM of the action gradient cross the ranks (rank-ordered sums)
    assert np.max(np.abs(rec["finalfp_actor"] - full["finalfp_actor"])) <= 2e-6
    assert np.max(np.abs(rec["final_log_alpha"] - full["final_log_alpha"])) <= 1e-6
    assert abs(rec["a0_log:losses/actor_pg_loss"] - full["a0_log:losses/actor_pg_loss"]) <= 1e-5
    open(os.path.join(out_dir, f"hok{rank}"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_humanoid_n16_sharded_two_ranks_one_shot_exchange(tmp_path):
    port = 30900 + (os.getpid() % 2000)
    _spawn(_humanoid_main, (2, port, str(tmp_path)), 2)
    assert all((tmp_path / f"hok{r}").exists() for r in range(2))


def _rccl_main(rank, world, port, out_dir):
    """ONE rank on the one GPU, backend "nccl" (= RCCL on ROCm): communicator set-up, ncclMin / ncclSum all-reduce of
    device tensors, and the sharded update sequence with its exchange steps going through that collective (the
    fallback path of parallel.all_reduce_*; FORCE_COLLECTIVE issues it although a group of one has nothing to add)."""
    sys.path.insert(0, HERE)
    import case_runner
    import synth
    import torch.distributed as dist
    from super_sac_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    assert dist.get_backend() == "nccl"
    parallel.FORCE_COLLECTIVE = True
    calls = {"n": 0}
    real = dist.all_reduce

    def counted(t, op=dist.ReduceOp.SUM, **kw):
        assert t.is_cuda
        calls["n"] += 1
        return real(t, op=op, **kw)
    dist.all_reduce = counted
    t = torch.tensor([1.0, float("inf"), -3.0], device="cuda:0")
    parallel.all_reduce_min(t)
    u = torch.arange(6, dtype=torch.float32, device="cuda:0")
    parallel.all_reduce_sum(u)
    torch.cuda.synchronize()
    assert calls["n"] == 2 and t.tolist() == [1.0, float("inf"), -3.0] and u.tolist() == [0, 1, 2, 3, 4, 5]
    cfg = synth.CASES["redq_small"]
    shard = parallel.Shard(0, 1, cfg["N"])
    rec = case_runner.run_engine("redq_small", device="cuda:0", shard=shard)
    fx = case_runner.slice_fixture(case_runner.load_fixture("redq_small"), cfg, shard)
    case_runner.compare(rec, fx, who="hip-sharded[1 rank, RCCL collective]")
    # 6 critic updates (one MIN each) + 2 actor updates (one MIN + one SUM each)
    assert calls["n"] >= 2 + 6 + 4, calls
    open(os.path.join(out_dir, "rccl_ok"), "w").write(str(calls["n"]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_collective_path_single_rank(tmp_path):
    port = 31900 + (os.getpid() % 2000)
    mp.spawn(_rccl_main, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "rccl_ok").exists()


def _late_main(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    import json
    import case_runner
    import synth
    import torch.distributed as dist
    from super_sac_amd import parallel
    from super_sac_amd._lib import lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    x = parallel.enable_one_shot(torch.device("cuda:0"))
    assert x is not None
    verdict = {"lab_build": hasattr(lib, "ssac_xchg_test_mode")}
    if verdict["lab_build"]:
        assert lib.ssac_xchg_test_mode(x.handle, 4) == 0   # every in-launch exchange of this rank: "the arrival wait gave up"
        cfg = synth.CASES["redq_small"]
        shard = parallel.Shard(rank, world, cfg["N"])
        raised, rec = None, None
        try:
            rec = case_runner.run_engine("redq_small", device="cuda:0", shard=shard)
        except RuntimeError as e:   # (the update functions' periodic check of the error word)
            raised = str(e)
        torch.cuda.synchronize()
        verdict["raised"] = raised
        verdict["failed_flag"] = bool(parallel.exchange_failed()) or raised is not None
        if rec is not None:
            tds = [v for k, v in rec.items() if "_td" in k]
            verdict["td_all_nan"] = bool(all(np.isnan(t).all() for t in tds)) if tds else None
    json.dump(verdict, open(os.path.join(out_dir, f"late{rank}.json"), "w"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_an_arrival_wait_that_gives_up_fails_the_exchange_instead_of_sending_incomplete_q(tmp_path):
    """Round-5 advisor (medium): when the tail workgroup's wait for the launch's target-critic workgroups timed out it only
    zeroed the spin limit -- with the peers' flags and acknowledgements present it then SENT the incomplete Qt, reduced it and
    raised nothing.  Now the wait's verdict goes into the exchange (xchg_body force_fail): nothing is sent, the result is NaN,
    the error word is raised.  Forced here through the LAB build's ssac_xchg_test_mode bit 2 on both ranks (the product
    library has no such switch: the assertions need `SSAC_LAB_BUILD=1`, tools/gpu_suite_lab.sh)."""
    import json
    port = 33500 + (os.getpid() % 2000)
    _spawn(_late_main, (2, port, str(tmp_path)), 2, retries=0)
    v = [json.load(open(tmp_path / f"late{r}.json")) for r in range(2)]
    if not v[0]["lab_build"]:
        pytest.skip("ssac_xchg_test_mode exists in the LAB build only (tools/gpu_suite_lab.sh runs this test)")
    for r in range(2):
        assert v[r]["failed_flag"], v[r]
        assert v[r]["raised"] is not None or v[r]["td_all_nan"], v[r]
