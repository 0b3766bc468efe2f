"""GPU: a SOAK of the recorded path (round-5 review, item 9).

5 000 critic updates (+ Polyak every 2nd) at the headline shape through bench.build_engine's own closure -- recorded launch
list, ONE C call per update (ssac_step_run), in-kernel Philox noise, deferred log finalisation, late-bound Polyak -- issued
free-running (the host reads logs only every 256 updates, so the input ring of 32 slots wraps ~150 times and the log ring of
512 blocks ~10 times), with ONE CHECKPOINT REWIND in the middle: save_training_state at update 2 500, 120 updates, load (the
same process: the device-side counters, the noise stream, the host generators and the ring position go BACK), the same 120
updates again -- bit-identical -- and on to 5 000.  Asserted: every log of every update finite (a hand-off wait that gave up
poisons its value with NaN, which reaches the loss); parameters, targets and moments finite at the end; the critic loss of
each of the first 200 updates within 1 % of the CPU oracle's (reference learning.py:18-141 restated, fed the same indices /
subsets / noise); the replayed stretch behind the rewind equal to its first pass bit for bit.
The longest sequence under test elsewhere is 14 updates."""
import numpy as np
import pytest
import torch

import ssac_oracle as orc
from test_hip_bench_bridge import _mlp_dict, philox_normal

pytestmark = pytest.mark.gpu

TOTAL, REWIND_AT, REPLAYED, ORACLE_UPDATES, READ_EVERY = 5000, 2500, 120, 200, 256


def _oracle_losses(init, seed, draws, idx, subset):
    import bench
    torch.manual_seed(0)
    buf = orc.ReplayOracle(bench.CAP)
    buf.load_experience(*bench.synth_data())
    oa = orc.AgentOracle(state_dim=bench.OBS, act_dim=bench.ACT, hidden=bench.HID, num_critics=bench.NCRIT,
                         ensemble_size=1, log_std_low=-5.0, log_std_high=2.0, seed=0)
    oa.actors[0], oa.critics[0] = init["actor"], init["critics"]
    oa.requires_grad_(True)
    ot = oa.clone()
    ot.critics[0] = init["target"]
    copt, eopt = orc.AdamOracle(oa.critic_params(), lr=bench.LR), orc.AdamOracle([], lr=1e-4)
    la = [torch.tensor([np.log(0.1)], dtype=torch.float32, requires_grad=True)]
    aug = orc.AugOracle("identity", bench.BATCH)
    out = []
    for u in range(len(idx)):
        eps = torch.from_numpy(philox_normal(seed, draws[u], bench.BATCH, bench.ACT))
        logs, _ = orc.critic_update(buf, oa, ot, copt, eopt, la, bench.BATCH, bench.GAMMA, None, None, bench.NSUB, None,
                                    None, False, aug, idx_list=[idx[u]], eps_list=[eps], subset_list=[subset[u]], grad_pick=0)
        out.append({k: float(v) for k, v in logs.items()})
        if u % bench.TARGET_DELAY == 0:
            orc.soft_update(ot.critic_params(), oa.critic_params(), bench.TAU)
    return out


@pytest.mark.timeout(900)
def test_five_thousand_recorded_updates_with_a_checkpoint_rewind(tmp_path):
    import bench
    import super_sac_amd as ssa
    from super_sac_amd import learning_utils as lu
    dev = torch.device("cuda")
    step, _, _ = bench.build_engine(dev, bench.NCRIT)
    ob = step.objects
    agent, target, buf, copt, la, state = (ob[k] for k in ("agent", "target", "buffer", "critic_optimizer", "log_alpha", "state"))
    assert ssa.rng.normal_is_stock() and ssa.learning.LAUNCH_MODE == "list" and ssa.learning.USE_GRAPHS
    init = dict(actor=_mlp_dict(agent.actors[0], "fc3"), critics=[_mlp_dict(n, "out") for n in agent.critics[0].nets],
                target=[_mlp_dict(n, "out") for n in target.critics[0].nets])
    ns = lu.noise_stream(agent, dev)
    seed = ns[0]
    draws, idx, subset = [], [], []
    losses = np.full(TOTAL, np.nan)
    held = []            # (update number, its lazy log dict) of the updates not read yet
    n_logs_read = 0

    def read_held():
        nonlocal n_logs_read
        for u, lg in held:
            vals = {k: float(v) for k, v in lg.items()}
            assert all(np.isfinite(v) for v in vals.values()), f"update {u}: non-finite log {vals}"
            losses[u] = vals["losses/critic_overall_loss"]
            n_logs_read += 1
        held.clear()

    def run(first, count, record=None):
        for u in range(first, first + count):
            if u < ORACLE_UPDATES:
                draws.append(ns[1])
            dicts = step()
            if u < ORACLE_UPDATES:
                idx.append(dicts[0]["priority_idxs"].copy())
                subset.append(list(dicts[0]["_subset"]))
            held.append((u, state["logs"]))
            if len(held) == READ_EVERY:      # (< the 512 blocks of the log ring: nothing unread is overwritten)
                read_held()
        read_held()
        if record is not None:
            torch.cuda.synchronize()
            ar, tar = agent.critics[0].arena(dev), target.critics[0].arena(dev)
            m, v = copt._ssac_adam.moments_for(("critic", 0), ar.params)
            record.update(params=ar.params.clone(), target=tar.params.clone(), m=m.clone(), v=v.clone(),
                          losses=losses[first:first + count].copy())

    run(0, REWIND_AT)
    k_at_save = state["k"]
    ssa.checkpoint.save_training_state(str(tmp_path), agent, target, {"critic": copt}, [la], buf)
    first_pass, second_pass = {}, {}
    run(REWIND_AT, REPLAYED, first_pass)
    # ---- the rewind, in the same process: every counter the recorded launches read on the device goes back
    ssa.checkpoint.load_training_state(str(tmp_path), agent, target, {"critic": copt}, [la], buf)
    state["k"] = k_at_save
    run(REWIND_AT, REPLAYED, second_pass)
    for key in ("params", "target", "m", "v"):
        assert torch.equal(first_pass[key], second_pass[key]), f"the stretch replayed behind the rewind differs in {key}"
    assert np.array_equal(first_pass["losses"], second_pass["losses"]), "replayed losses differ"
    run(REWIND_AT + REPLAYED, TOTAL - REWIND_AT - REPLAYED)
    torch.cuda.synchronize()

    # ---- what ran was the recorded fast path, the whole way
    gs = next(iter(agent.__dict__["_ssac_graphs"].values()))
    fast = next(iter(agent.__dict__["_ssac_fast"].values()))
    assert gs.path == "fast" and gs.in_kernel_noise and gs.deferred is not None and fast.late_arenas is not None
    assert fast.calls >= TOTAL + REPLAYED - 8, fast.calls
    assert n_logs_read == TOTAL + REPLAYED and np.all(np.isfinite(losses))
    ar, tar = agent.critics[0].arena(dev), target.critics[0].arena(dev)
    m, v = copt._ssac_adam.moments_for(("critic", 0), ar.params)
    for name, t in (("parameters", ar.params), ("targets", tar.params), ("Adam m", m), ("Adam v", v)):
        assert bool(torch.isfinite(t).all()), f"non-finite {name} after {TOTAL} updates"
    # (a learning run, not a frozen one: the loss moved, the targets trail the critics)
    assert abs(losses[-1] - losses[0]) > 1e-3 * abs(losses[0]) and not torch.equal(ar.params, tar.params)

    # ---- the first 200 updates against the CPU oracle: critic loss within 1 %
    torch.set_num_threads(min(16, torch.get_num_threads()))
    want = _oracle_losses(init, seed, draws, idx, subset)
    rel = [abs(losses[u] - want[u]["losses/critic_overall_loss"]) / max(abs(want[u]["losses/critic_overall_loss"]), 1e-6)
           for u in range(ORACLE_UPDATES)]
    print(f"soak: worst relative critic-loss deviation from the oracle over {ORACLE_UPDATES} updates {max(rel):.2e}; "
          f"loss {losses[0]:.4f} -> {losses[ORACLE_UPDATES - 1]:.4f} -> {losses[-1]:.4f}")
    assert max(rel) < 1e-2, f"update {int(np.argmax(rel))}: critic loss {max(rel):.3e} off the oracle's"
