"""Headline benchmark: gradient updates/sec of the REDQ critic update (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R] [--critics N] [--obs S] [--act A] [--batch B]

One "step" = one ``learning.critic_update`` call on a synthetic replay batch (obs 17, act 6, batch 512, N=10 critics,
n=2 target subset, hidden 256, fp32 -- the shape the metric is quoted on) followed, every ``target_delay``=2 updates,
by the Polyak update of the target critics, exactly as the reference's UTD loop does (main.py:379-414).  The replay
buffer (100k transitions) is resident in HBM before the timed region starts; the per-update host work that remains
(index draw from the torch CPU generator, REDQ subset draw, a 4 KB index upload) is part of the path and is inside the
timed region.  After W untimed warm-up steps, EXACTLY K steps are timed between barrier + synchronize brackets; this is
done R times (default 31: at the driver's K = 20 that is ~35 ms, past the ~14 ms the core clock takes to ramp after the idle
construction phase; every repeat's time and the shader clock beside it are listed in the line) and the MEDIAN repeat is
reported (``value`` = K / median seconds).

--critics / --obs / --act / --batch (defaults 10 / 17 / 6 / 512 = the headline) select another ensemble: `--gpus 8
--critics 16` is the configuration BASELINE.json's scaling target is quoted on (N = 16 critics over 8 GPUs), `--obs 376
--act 17 --critics 16` its Humanoid shape (config 5); the metric name and `config.workload` say what was run.

N > 1: one process per GPU.  Launched by ``torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE in the environment)
the process is one rank; from a bare shell (``python bench.py --gpus N``) the parent -- which never touches the GPU --
starts the N ranks itself as child processes and relays rank 0's line.  The critic ensemble is sharded across the
ranks (super_sac_amd.parallel); the one exchange per update (MIN of the per-shard subset min-Q) is a recorded launch
of the one-shot IPC exchange kernel (csrc/ssac_xchg.hip), RCCL being the fallback.  Total work is fixed: "strong".

Prints ONE JSON line (rank 0) with `roofline` for the dominant kernel, `cpu_baseline` (the CPU oracle -- a port of the
reference's update -- timed on this host's cores) and `secondary` rows (full REDQ environment step; bf16 config 2).
"""
import argparse
import copy
import json
import math
import os
import statistics
import subprocess
import sys
import time
from itertools import chain

# Host-side wait mode of the ROCm runtime, set BEFORE anything initialises it (torch is imported inside main()): with the
# default (interrupt-driven signal waits) the thread that blocks in torch.cuda.synchronize() sleeps and is woken by an
# interrupt -- the first update behind every synchronisation then costs 50 - 120 us of host time on a just-woken thread
# (HISTORY.md, round 3: tools/first_step_py.py), i.e. 2 - 3 us per update of a 20-step region.  Polling waits
# (HSA_ENABLE_INTERRUPT=0, a documented ROCr setting; INTEGRATION.md recommends it for training loops that synchronise every
# environment step) remove the sleep: measured 17.98 k -> 18.45 k updates/s in the driver's form, nothing at 2000 steps.
# `--interrupt-wait` (or the variable already set by the caller) keeps the runtime's default; the line says which it was.
# ONLY when this file is the program: a process that merely imports `bench` (the parity tests do, for build_engine) keeps
# its environment -- the variable would leak into the ranks the sharded tests spawn on one shared device.
if __name__ == "__main__" and "--interrupt-wait" not in sys.argv:
    os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

OBS, ACT, BATCH, NCRIT, NSUB, HID = 17, 6, 512, 10, 2, 256
ROWS, CAP = 100_000, 1_000_000
GAMMA, LR, TAU, TARGET_DELAY = 0.99, 3e-4, 0.005, 2
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# HBM bytes per launch of the chained kernel come from OFFLINE rocprofv3 --pmc passes of this command (FETCH_SIZE doubled per
# the guide's gfx950 note for 16-byte streaming reads + WRITE_SIZE; separate passes) -- they cannot be collected inside this
# run.  profiles/traffic.json records them together with a hash of the kernel sources they were taken on
# (tools/traffic_record.py writes it from the PMC summaries); a line printed from other sources carries `traffic: null` and
# says why, instead of a number that silently went stale (round-5 review, weak 11).
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")
TRAFFIC_SOURCES = ("super_sac_amd/csrc/ssac_fused.hip", "super_sac_amd/csrc/ssac_internal.h", "super_sac_amd/csrc/ssac_begin.h")


def kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    for rel in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def recorded_traffic():
    """(bytes per launch or None, provenance string)"""
    try:
        with open(TRAFFIC_FILE) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None, "no offline PMC record (profiles/traffic.json)"
    if rec.get("source_hash") != kernel_source_hash():
        return None, (f"stale: the kernel sources changed since the PMC passes of {rec.get('taken_at', '?')} "
                      f"({rec.get('bytes')} B per launch then)")
    return int(rec["bytes"]), (f"offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on these kernel sources "
                               f"({rec.get('taken_at', '?')}; {rec.get('evidence', 'profiles/')}): 2 x {rec.get('fetch_kib')} KiB "
                               f"fetched + {rec.get('write_kib')} KiB written")


_SCLK_NODE = {}


def sclk_mhz(index=0):
    """shader clock level of torch device `index` from sysfs (the starred level of pp_dpm_sclk of ITS PCI function -- a box
    lists every GPU of the node, visible to this process or not), or None: a few microseconds, no GPU work, no subprocess.
    Read beside every timed repeat.  (What it shows on these boxes is the DPM level, which sits at its cap whenever the
    device has work: the ramp after an idle stretch is visible in the first repeats' TIMES, not in this number.)"""
    try:
        node = _SCLK_NODE.get(index)
        if node is None:
            import torch
            p = torch.cuda.get_device_properties(index)
            bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            node = _SCLK_NODE[index] = f"/sys/bus/pci/devices/{bdf}/pp_dpm_sclk"
        with open(node) as f:
            for ln in f:
                if "*" in ln:
                    return int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
    except Exception:   # noqa: BLE001  (no sysfs node / not readable / no PCI ids: the line says null)
        pass
    return None


def synth_data(obs=OBS, act=ACT):
    import synth
    return synth.synth_transitions(ROWS, obs, act, seed=1)


def build_engine(device, n_local, shard=None, batch=None, precision="fp32", obs=None, act=None, ncrit=None):
    """(obs / act / ncrit / batch default to the module's OBS / ACT / NCRIT / BATCH, i.e. the headline unless main() was given
    --obs / --act / --critics / --batch)"""
    import numpy as np
    import torch
    import super_sac_amd as ssa
    torch.manual_seed(0)
    np.random.seed(0)
    import random
    random.seed(0)
    OBS_, ACT_, NCRIT_ = OBS if obs is None else obs, ACT if act is None else act, NCRIT if ncrit is None else ncrit
    batch = BATCH if batch is None else batch
    def make(n):
        return ssa.Agent(act_space_size=ACT_, encoder=ssa.nets.IdentityEncoder(OBS_),
                         actor_network_cls=ssa.nets.ContinuousStochasticActor,
                         critic_network_cls=ssa.nets.ContinuousCritic, discrete=False, ensemble_size=1,
                         num_critics=n, ucb_bonus=0.0, hidden_size=HID, auto_rescale_targets=False,
                         log_std_low=-5.0, log_std_high=2.0)
    agent = make(NCRIT_ if shard is not None else n_local)
    if shard is not None:
        # a rank of the sharded job holds critics [lo, hi) of THE SAME seeded ensemble the unsharded engine holds
        # (so a sharded run can be checked against it value by value), the replicated actor included
        full, agent = agent, make(n_local)
        agent.actors[0].load_state_dict(full.actors[0].state_dict())
        for j in range(n_local):
            agent.critics[0].nets[j].load_state_dict(full.critics[0].nets[shard.lo + j].state_dict())
        del full
    agent.to(device)
    agent.train()
    ssa.set_precision(agent, precision)
    target = copy.deepcopy(agent)
    buf = ssa.replay.ReplayBuffer(CAP, device=device)
    buf.load_experience(*synth_data(OBS_, ACT_))
    copt = torch.optim.Adam(chain(*(c.parameters() for c in agent.critics)), lr=LR, betas=(0.9, 0.999))
    aopt = torch.optim.Adam(chain(*(a.parameters() for a in agent.actors)), lr=LR, betas=(0.9, 0.999))
    eopt = torch.optim.Adam(agent.encoder.parameters(), lr=1e-4)
    la = torch.Tensor([math.log(0.1)]).to(device)
    la.requires_grad = True
    lopt = torch.optim.Adam([la], lr=1e-4, betas=(0.5, 0.999))
    aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.IdentityAug(batch)])
    if shard is not None:
        ssa.parallel.install(agent, target, shard)
    # every rank must draw the SAME replay indices / REDQ subsets (identically seeded generators instead of
    # a batch broadcast), whatever number of critics it has just initialised: re-seed behind the construction.  (Found
    # by sharded_value_check: with uneven shards -- 3 or 4 ranks over 10 critics -- the ranks' CPU generators had
    # advanced by different amounts and they sampled different batches.)
    torch.manual_seed(0)
    np.random.seed(0)
    random.seed(0)
    state = {"k": 0}

    def step():
        state["logs"], dicts = ssa.learning.critic_update(
            buffer=buf, agent=agent, target_agent=target, critic_optimizer=copt, encoder_optimizer=eopt,
            log_alphas=[la], batch_size=batch, gamma=GAMMA, critic_clip=None, encoder_clip=None,
            target_critic_ensemble_n=NSUB, weighted_bellman_temp=None, weight_type=None, pop=False,
            augmenter=aug, encoder_lambda=0, aug_mix=0.0, discrete=False, random_process=None,
            noise_clip=None, per=False, update_priorities=False, dr3_coeff=0.0)
        if state["k"] % TARGET_DELAY == 0:
            for ac, tc in zip(agent.critics, target.critics):
                ssa.learning_utils.soft_update(tc, ac, TAU)
        state["k"] += 1
        return dicts

    def env_step():
        """one environment step of redq.gin: UTD 20 critic updates (+ Polyak every 2nd), then one actor and one
        temperature update on the last critic batch (main.py:379-414, 489-542)"""
        for _ in range(20):
            dicts = step()
        ssa.learning.online_actor_update(buffer=buf, agent=agent, pop=False, actor_optimizer=aopt, log_alphas=[la],
                                         batch_size=batch, clip=None, random_process=None, noise_clip=None,
                                         augmenter=aug, aug_mix=0.0, premade_replay_dicts=dicts)
        ssa.learning.alpha_update(buffer=buf, agent=agent, optimizers=[lopt], batch_size=batch, log_alphas=[la],
                                  augmenter=aug, aug_mix=0.0, target_entropy=-float(ACT_), premade_replay_dicts=dicts,
                                  discrete=False)
    def actor_step(dicts):
        """the online actor update alone, on the batch of a critic update (premade_replay_dicts: main.py:489-511)"""
        return ssa.learning.online_actor_update(buffer=buf, agent=agent, pop=False, actor_optimizer=aopt, log_alphas=[la],
                                                batch_size=batch, clip=None, random_process=None, noise_clip=None,
                                                augmenter=aug, aug_mix=0.0, premade_replay_dicts=dicts)
    # (tests/test_hip_bench_bridge.py drives this very closure and compares it with the oracle)
    step.objects = dict(agent=agent, target=target, buffer=buf, critic_optimizer=copt, log_alpha=la, state=state,
                        actor_step=actor_step)
    return step, env_step, ssa


def cpu_baseline(budget_s=15.0):
    """The oracle's critic_update (+Polyak) on the host cores: same shape, same 100k-row buffer.  Rows: every thread
    count tried (1 thread and all cores included); `value` is the best of them."""
    import torch
    import ssac_oracle as orc
    torch.manual_seed(0)
    buf = orc.ReplayOracle(CAP)
    buf.load_experience(*synth_data())
    oa = orc.AgentOracle(state_dim=OBS, act_dim=ACT, hidden=HID, num_critics=NCRIT, ensemble_size=1,
                         log_std_low=-5.0, log_std_high=2.0, seed=0).requires_grad_(True)
    ot = oa.clone()
    copt = orc.AdamOracle(oa.critic_params(), lr=LR)
    eopt = orc.AdamOracle([], lr=1e-4)
    la = [torch.tensor([math.log(0.1)], requires_grad=True)]
    aug = orc.AugOracle("identity", BATCH)

    def one(k):
        orc.critic_update(buf, oa, ot, copt, eopt, la, BATCH, GAMMA, None, None, NSUB, None, None, False, aug)
        if k % TARGET_DELAY == 0:
            orc.soft_update(ot.critic_params(), oa.critic_params(), TAU)
    ncpu = os.cpu_count() or 1
    # thread counts: 1, 8, 16, 32 and every core -- but at most 64 threads: on the 256-core GPU boxes ONE update with every
    # core takes ~40 s (hundreds of threads spinning around 100-microsecond operators: 0.02 updates/s, measured in rounds 5
    # and 6, profiles/r6_raw/final/bench_steps20.json of the first lease), which alone blew the leg's 10 - 30 s bound
    top = min(ncpu, 64)
    counts = sorted({c for c in (1, 8, 16, 32) if c <= ncpu} | {top})
    rows = {}
    per = budget_s / len(counts)
    for th in counts:
        torch.set_num_threads(th)
        # (no separate warm-up call: on a many-core host ONE update with every core takes tens of seconds -- hundreds
        # of threads spinning around 100-microsecond operators -- so that row is a single timed update)
        n, t0 = 0, time.perf_counter()
        while n < 1 or time.perf_counter() - t0 < per:
            one(n)
            n += 1
        rows[th] = (n, time.perf_counter() - t0)
    torch.set_num_threads(min(16, ncpu))   # (back to the harness's own setting)
    best = max(rows, key=lambda th: rows[th][0] / rows[th][1])
    n, dt = rows[best]
    return {"value": round(n / dt, 2), "unit": "updates/s", "cores": best, "kind": "port",
            "host_cores": ncpu,
            "rows": {str(th): round(r[0] / r[1], 2) for th, r in rows.items()},
            "sample": f"{sum(r[0] for r in rows.values())} critic updates (+Polyak every 2nd) of the same workload in "
                      f"{sum(r[1] for r in rows.values()):.1f} s over thread counts {counts} "
                      f"(host has {ncpu} logical cores" + (f"; the all-cores row is capped at {top} threads: with {ncpu} threads "
                      f"one update takes tens of seconds, oversubscription" if top < ncpu else "") +
                      f"), torch {torch.__version__} CPU; value = best row ({best} threads)"}


def sharded_value_check(step, ssa, device, shard, dist, n_updates=12, must_pass=True):
    """`bench.py --gpus N` is a VALUE check before it is a timing: the first n_updates updates of the sharded engine
    (every rank, through the exchange) against the same updates of the unsharded engine run by rank 0 in this process
    on the same seeds -- this rank's critics, Polyak targets and Adam moments, plus every TD target.  Tolerances as
    in tests/test_hip_sharded.py: TD targets 2e-5 (a rank with few critics takes the 16-row tile variant, whose fp32
    sums differ in the last bits), parameters / targets / first moments 5e-6 * n_updates."""
    import random
    import numpy as np
    import torch

    def run(stp):
        tds = []
        for _ in range(n_updates):
            dicts = stp()
            tds.append(dicts[0]["td_target"].detach().float().cpu().numpy().copy())
        ob = stp.objects
        ar, tar = ob["agent"].critics[0].arena(device), ob["target"].critics[0].arena(device)
        m, _ = ob["critic_optimizer"]._ssac_adam.moments_for(("critic", 0), ar.params)
        torch.cuda.synchronize()
        n = ar.n_nets
        return tds, [t.view(n, -1).cpu().numpy().copy() for t in (ar.params, tar.params, m)]
    if dist is not None and dist.get_world_size() > 1:
        dist.barrier()   # ranks finish building seconds apart; the exchange kernel's spin is bounded
    worst = {"td": 0.0, "params": 0.0, "target": 0.0, "adam_m": 0.0}
    broke = None
    try:
        tds, mine = run(step)
        torch.cuda.synchronize()
        from super_sac_amd import parallel as _par
        _par.check_exchange()
    except RuntimeError as e:   # (an exchange that gave up on a peer: every rank still takes part in the verdict below)
        broke = str(e)
    if shard.rank == 0 and broke is None:
        # the unsharded engine on the same seeds (build_engine leaves every generator, the device's included, at seed 0)
        keep = (torch.get_rng_state(), random.getstate(), np.random.get_state(), torch.cuda.get_rng_state(device))
        ref_step, _, _ = build_engine(device, NCRIT, None)
        ref_tds, ref = run(ref_step)
        # ... and back to where the sharded job's generators stand (the other ranks did not make these extra draws)
        torch.set_rng_state(keep[0]); random.setstate(keep[1]); np.random.set_state(keep[2])
        torch.cuda.set_rng_state(keep[3], device)
        per_update = [float(np.max(np.abs(a - b))) for a, b in zip(tds, ref_tds)]
        worst["td"] = max(per_update)
        worst["td_per_update"] = [float(f"{v:.2g}") for v in per_update]
        for key, a, b in zip(("params", "target", "adam_m"), mine, ref):
            worst[key] = float(np.max(np.abs(a - b[shard.lo:shard.hi])))
        del ref_step
    per_update = worst.pop("td_per_update", None)
    ok = (broke is None and worst["td"] <= 2e-5 and np.isfinite(worst["td"])
          and max(worst["params"], worst["target"], worst["adam_m"]) <= 5e-6 * n_updates)
    if not must_pass and os.environ.get("SSAC_BENCH_TEST_FALLBACK") == "1":
        ok = False   # (exercises the fall-back from the one-shot exchange to the collective on a box where it works)
    if not ok:
        worst["td_per_update"] = per_update
        if broke:
            worst["error"] = broke
    flag = torch.tensor([1.0 if ok else 0.0], device=device)
    if dist is not None and dist.get_world_size() > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # (the process group's collective, never the one-shot kernel)
    if float(flag) != 1.0:
        if must_pass:
            raise RuntimeError(f"sharded run differs from the unsharded engine on the same seeds: {worst}")
        return None
    return {"updates": n_updates, "max_abs_diff": {k: float(f"{v:.3g}") for k, v in worst.items()},
            "against": "the unsharded engine, same seeds, run by rank 0 in the same process"}


def launch_ranks(args):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as fresh child processes (this parent has not
    touched the GPU and never does) and relay rank 0's line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--repeats", str(args.repeats), "--critics", str(args.critics),
               "--obs", str(args.obs), "--act", str(args.act), "--batch", str(args.batch)]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        if args.no_secondary:
            cmd.append("--no-secondary")
        if args.interrupt_wait:
            cmd.append("--interrupt-wait")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = max(rc, p.wait())
    sys.stdout.write(out.decode())
    sys.exit(rc)


def timed_repeats(fn, steps, repeats, dist, device, clocks=None):
    """clocks: a list that receives the shader clock (MHz, sysfs) read right after every repeat's closing synchronise"""
    import torch
    times = []
    for _ in range(repeats):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        times.append(dt)
        if clocks is not None:
            clocks.append(sclk_mhz(device.index or 0))
    return times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=31,
                    help="timed regions of EXACTLY --steps steps each (every one listed in the line); the median is the value")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--interrupt-wait", action="store_true",
                    help="keep the ROCm runtime's interrupt-driven host waits (default here: polling, HSA_ENABLE_INTERRUPT=0)")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--critics", type=int, default=NCRIT, help="ensemble size N (16: the scaling target's configuration)")
    ap.add_argument("--obs", type=int, default=OBS, help="observation size (376: Humanoid, BASELINE config 5)")
    ap.add_argument("--act", type=int, default=ACT, help="action size (17: Humanoid)")
    ap.add_argument("--batch", type=int, default=BATCH)
    args = ap.parse_args()
    headline = (args.critics, args.obs, args.act, args.batch) == (10, 17, 6, 512)
    globals().update(NCRIT=args.critics, OBS=args.obs, ACT=args.act, BATCH=args.batch)   # (read by every helper below)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)   # (does not return)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Host threads: torch's CPU pool defaults to every logical core (256 on the GPU boxes).  The only CPU work of this
    # process is construction (orthogonal initialisation = a QR per layer), where hundreds of threads spinning around
    # 100-us operators are slower than a few -- and N ranks each with a full-size pool oversubscribe the host N times:
    # eight ranks on a loaded box were still initialising weights after three minutes.  (The CPU-baseline leg sets its
    # own thread counts per row.)
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // max(1, world))))
    assert torch.cuda.is_available(), "bench.py measures the HIP path; it needs an MI355X"
    ndev = torch.cuda.device_count()
    shared_device = world > ndev          # (ranks sharing one device: the N>1 path on a 1-GPU box)
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    shard = None
    n_local = NCRIT
    exchange = "none"
    # SSAC_BENCH_FORCE_DIST=1: take the sharded path with a single rank too (the RCCL check a 1-GPU box can run)
    if world > 1 or os.environ.get("SSAC_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # (the forced single-rank run is started from a bare shell)
            for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k_, v_)
        # RCCL refuses two ranks on one device: such runs set the group up over gloo (the data path is the IPC kernel)
        backend = os.environ.get("SSAC_BENCH_BACKEND", "gloo" if shared_device else "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        from super_sac_amd import parallel
        parallel.FORCE_COLLECTIVE = world == 1
        shard = parallel.Shard(rank, world, NCRIT)
        n_local = shard.n_local
        exchange = f"torch.distributed all_reduce ({backend})"
        # (more than 4 ranks SHARING one device -- the functional check a 1-GPU box can run: eight processes whose
        # exchange kernels spin on each other's flags while the device time-slices their queues stalled for the whole
        # 10 s spin bound once in a while; the collective path has no spinning kernel.  One rank per GPU -- the real
        # layout -- keeps the one-shot exchange.)
        one_shot_default = "0" if (shared_device and world > 4) else "1"
        if world > 1 and os.environ.get("SSAC_BENCH_ONE_SHOT", one_shot_default) == "1":
            try:
                if parallel.enable_one_shot(device) is not None:
                    exchange = "one-shot IPC exchange kernel (csrc/ssac_xchg.hip), recorded in the launch list"
            except RuntimeError as e:   # peers not mappable: keep the collective
                exchange += f" [one-shot unavailable: {e}]"
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    step, env_step, ssa = build_engine(device, n_local, shard)
    value_check = None
    if shard is not None:
        from super_sac_amd import parallel
        one_shot_on = parallel._exchange is not None
        value_check = sharded_value_check(step, ssa, device, shard, dist, must_pass=not one_shot_on)
        if value_check is None:
            # The one-shot exchange produced values that differ from the unsharded engine (or gave up on a peer) on
            # this system -- peer-device mappings of the receive buffers have only ever run with the ranks on ONE device.
            # Every rank saw the same verdict: all of them fall back to the collective, record again, check again.
            parallel.disable_one_shot()
            exchange = f"torch.distributed all_reduce ({backend}) [one-shot exchange failed its value check on this system]"
            del step, env_step
            step, env_step, ssa = build_engine(device, n_local, shard)
            value_check = sharded_value_check(step, ssa, device, shard, dist)
    # Python's cyclic GC otherwise runs a full (generation-2) collection over the whole torch object graph every few
    # hundred updates -- a 40-80 ms pause, i.e. hundreds of updates: park the start-up objects in the permanent
    # generation (host runtime hygiene of a long-running training loop; nothing the update path allocates is cyclic)
    import gc
    gc.collect()
    gc.freeze()
    if dist is not None:
        dist.barrier()   # ranks finish building at different times: start the exchanging updates together
    sclk_before = sclk_mhz(local)
    for _ in range(args.warmup):
        step()

    # ---- timed region: EXACTLY --steps steps between barrier+sync brackets, `repeats` times; the median is reported.
    # It is the FIRST GPU work of the process behind construction and the --warmup steps (no hidden warm-up: the CPU
    # baseline, the roofline's event pass and every secondary row come after it).  The device has idled through the
    # construction (seconds of host work), so the first ~14 ms of it run inside the core clock's ramp: every repeat's time and
    # the shader clock read beside it are in the line, as are the first and the fastest repeat.
    sclk = []
    times = timed_repeats(step, args.steps, args.repeats, dist, device, clocks=sclk)
    dt = statistics.median(times)
    if world > 1:
        from super_sac_amd import parallel
        if parallel.exchange_failed():   # a peer's flag never arrived: the numbers above would be of a broken run
            raise RuntimeError("one-shot exchange: a peer's flag did not arrive within the spin limit")

    # ---- roofline of the dominant kernel: the chained launch (ensemble-Q forward + TD-independent backward of all local
    # critics beside the target chains).  Algorithmic FLOPs per launch: 2*B*N*(in*H + H*H + H) for the
    # critics' forward, 2*B*N*(H + H*H) for their backward-data, the target critics of the subset and the actor once
    # (round 4: the launch's producer / consumer form runs the actor once per tile -- executed = algorithmic).  The timed
    # region re-issues the
    # update from ONE recorded launch list, inside which a single kernel is not bracketed by events; so the same
    # update is run again right here with plain launches and the launch is bracketed by HIP events recorded on the
    # stream it is launched on (same shapes, buffers, binary), 8 back-to-back issues per event pair.
    graphs_were_on = ssa.learning.USE_GRAPHS
    ssa.learning.USE_GRAPHS = False
    ssa.engine.PROFILE["tag"] = ("critic_fwd", "critic_bwd", "critic_fused", "dual_fwd", "dual_bwd", "chain")
    ssa.engine.PROFILE["events"] = []
    ssa.engine.PROFILE["reps"] = 8
    for _ in range(min(args.steps, 300)):
        step()
    torch.cuda.synchronize()
    ssa.engine.PROFILE["tag"] = None
    ssa.engine.PROFILE["reps"] = 1
    ssa.learning.USE_GRAPHS = graphs_were_on
    by_tag = {}
    for a, b, tag, reps in ssa.engine.PROFILE["events"]:
        by_tag.setdefault(tag, []).append(a.elapsed_time(b) / reps)
    IN = OBS + ACT
    f_fwd = 2.0 * BATCH * n_local * (IN * HID + HID * HID + HID)
    f_bwd = 2.0 * BATCH * n_local * (HID + HID * HID)
    f_actor = 2.0 * BATCH * (OBS * HID + HID * HID + HID * 2 * ACT)
    f_tgt = 2.0 * BATCH * NSUB * (IN * HID + HID * HID + HID)
    if "chain" in by_tag:
        ms = by_tag["chain"]
        flops, kname = f_fwd + f_bwd + f_tgt + f_actor, (
            "fused_chain_pc_kernel: ensemble-Q forward (fc1+fc2+head, h1/h2/q saved) AND the TD-independent half of the "
            "backward pass (head backward + fc2 backward-data) of all local critics as 32-row workgroups, beside the "
            "target chains as producer / consumer workgroups of 16 rows (the actor forward + tanh-normal sample once per "
            "tile hands a' to the tile's target critics of the REDQ subset, which have run fc1 on the state columns "
            "meanwhile); every workgroup gathers its own replay rows; ONE launch per update")
    elif "dual_fwd" in by_tag:
        ms = by_tag["dual_fwd"]
        flops, kname = f_fwd + f_actor, "fused_dual_kernel: ensemble-Q forward + actor forward/sample, ONE launch"
    elif "critic_fwd" in by_tag:
        ms = by_tag["critic_fwd"]
        flops, kname = f_fwd, "fused_mlp_kernel<plain>: ensemble-Q forward (fc1+fc2+head) of all local critics"
    else:
        ms = by_tag["critic_fused"]
        flops, kname = f_fwd + f_bwd, "fused_mlp_kernel<critic>: forward + loss gradient + backward-data, ONE launch"
    avg_ms = sum(ms) / len(ms)
    achieved = flops / (avg_ms * 1e-3) / 1e12
    single = world == 1 and n_local == NCRIT and "chain" in by_tag and headline
    roofline = {"kernel": kname, "bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                "avg_launch_us": round(avg_ms * 1e3, 3), "launches_timed": len(ms),
                "timing": "HIP events on the launch stream around 8 back-to-back issues of the (idempotent) launch, "
                          "in an eager pass right after the timed (replayed) region",
                "flops_per_launch": flops,
                "traffic": None, "traffic_source": None}
    if single:
        roofline["traffic"], roofline["traffic_source"] = recorded_traffic()

    if rank == 0:
        out = {"metric": f"gradient updates/sec (REDQ N={NCRIT}, batch {BATCH})", "value": round(args.steps / dt, 2),
               "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 5), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "repeats": args.repeats, "repeat_ms_per_step": [round(1e3 * t / args.steps, 5) for t in times],
               "ms_per_step_first": round(1e3 * times[0] / args.steps, 5),
               "ms_per_step_min": round(1e3 * min(times) / args.steps, 5),
               "sclk_mhz": {"before_warmup": sclk_before, "after_each_repeat": sclk,
                            "source": "sysfs pp_dpm_sclk (current DPM level) of rank 0's device, by PCI address"},
               "config": {"workload": f"REDQ critic_update + Polyak/2: obs {OBS}, act {ACT}, batch {BATCH}, "
                                      f"N={NCRIT} critics (n=2 target subset), hidden 256, replay 100k rows in HBM",
                          "global_batch": BATCH, "num_critics": NCRIT,
                          "launch": "recorded launch list, one C call per update (ssac_step_run)"
                                    if graphs_were_on else "plain launches",
                          "host_wait": ("polling (HSA_ENABLE_INTERRUPT=0)" if os.environ.get("HSA_ENABLE_INTERRUPT") == "0"
                                        else "interrupt (ROCm default)"),
                          "parallelism": "single GPU" if world == 1 else
                                         f"critic-ensemble sharded x{world}" +
                                         (f" ({world} ranks sharing {ndev} device(s))" if shared_device else ""),
                          "exchange": exchange},
               "roofline": roofline}
        if value_check is not None:
            out["sharded_value_check"] = value_check
    secondary = {}
    if not args.no_secondary and world == 1:
        try:   # (a failing secondary row must not cost the headline line)
            # ---- full REDQ environment step (SURVEY 8(d) secondary unit): 20 critic updates + 10 Polyak + actor + alpha
            for _ in range(3):
                env_step()
            ts = timed_repeats(env_step, 30, 3, None, device)
            te = statistics.median(ts) / 30
            secondary["full_redq_step_fp32"] = {
                "workload": "redq.gin environment step at the headline shape: 20 critic updates + 10 Polyak + 1 actor + "
                            "1 temperature update (B 512, N 10)",
                "ms_per_env_step": round(te * 1e3, 4), "critic_updates_per_s": round(20 / te, 1),
                "env_steps_per_s": round(1 / te, 1)}
            del step, env_step
            # ---- latency of the acting path (what an environment step pays before the updates)
            secondary["acting"] = acting_rows(device)
            # ---- the online actor update alone (UTD-1 configurations: it is ~60 % of an environment step's device time)
            secondary["actor_update"] = actor_update_rows(device)
            # ---- a whole UTD-1 environment step (configs 1, 3, 4, 5 run UTD 1): acting + the three updates in sequence
            secondary["utd1_env_step"] = utd1_rows(device)
            # ---- the N = 1 anchors of the scaling target's configurations (BASELINE.json: ">= 3.5x at 8 vs 1 GPU for N = 16";
            #      `bench.py --gpus 8 --critics 16 [--obs 376 --act 17]` measures the other end when an 8-GPU node runs it)
            secondary["scaling_anchors_n16_1gpu"] = n16_rows(device)
            # ---- large-batch sweep of the fp32 path's two launches (the asymptotic fraction of the matrix peak)
            secondary["fp32_sweep"] = fp32_sweep(ssa, device)
            # ---- BASELINE config 2: REDQ N=10 UTD=20 batch 256 in the bf16-operand mode; the ensemble-Q kernel's HBM fraction
            step_b, env_b, _ = build_engine(device, NCRIT, None, batch=256, precision="bf16")
            for _ in range(60):
                step_b()
            tb = statistics.median(timed_repeats(step_b, 1000, 3, None, device)) / 1000
            for _ in range(3):
                env_b()
            teb = statistics.median(timed_repeats(env_b, 30, 3, None, device)) / 30
            secondary["config2_bf16"] = dict(bf16_rows(ssa, device), **{
                "workload": "REDQ N=10 UTD=20 batch 256, bf16 operands / fp32 accumulate + fp32 masters (no reference "
                            "counterpart; parity: tests/test_hip_bf16.py)",
                "critic_updates_per_s": round(1 / tb, 1), "us_per_critic_update": round(tb * 1e6, 2),
                "ms_per_env_step": round(teb * 1e3, 4), "env_steps_per_s": round(1 / teb, 1)})
            # ---- BASELINE configs 3 and 4: the pixel configurations (one critic update incl. the encoder's backward pass)
            del step_b, env_b
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
            import bench_pixels
            for which, label in (("dmc", "config3_dmc_pixels"), ("atari", "config4_atari_pixels")):
                pstep, pB = bench_pixels.build(which, device)
                for _ in range(8):   # (also past the clock ramp of a box that has just started)
                    pstep()
                tp = statistics.median(timed_repeats(pstep, 20, 3, None, device)) / 20
                secondary[label] = {
                    "workload": ("DrQv2 on 9x84x84 uint8 observations: shift augmentation, BigPixelEncoder, 2 critics of "
                                 "hidden 1024, B 512" if which == "dmc" else
                                 "SAC-Discrete on 4x84x84 uint8 observations: shift augmentation, SmallPixelEncoder, 2 "
                                 "critics of hidden 256, B 1024, gradient clip 40") +
                                "; one critic update (encoder forward x2, backward, Adam, Polyak)",
                    "ms_per_critic_update": round(tp * 1e3, 3), "critic_updates_per_s": round(1 / tp, 1),
                    "frames_per_s": round(pB / tp, 0)}
                del pstep
                torch.cuda.empty_cache()
        except Exception as e:   # noqa: BLE001
            secondary["error"] = f"{type(e).__name__}: {e}"
    if rank == 0:
        if secondary:
            out["secondary"] = secondary
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:   # noqa: BLE001  (the GPU numbers above are still worth a line)
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def acting_rows(device):
    """SURVEY 8(f) rank 2, the acting path (Agent.sample_action / Agent.forward, agent.py:204-315) as the collection loop calls
    it: numpy observation in, numpy action out (one H2D, the launches, one D2H), wall-clock per call, median of 300 calls
    after 30 warm-up calls.  Shapes: the headline REDQ agent (obs 17 / act 6, one actor), a SUNRISE agent (5 members x 2
    critics, ucb_bonus 5: candidates of every actor, ensemble-Q of every member on the stacked candidates, mean + bonus * std,
    arg-max), the Atari agent (4 x 84 x 84 uint8 frames through the SmallPixelEncoder, categorical sample) and the DMC agent
    (9 x 84 x 84 frames through the BigPixelEncoder, hidden-1024 deterministic actor)."""
    import numpy as np
    import torch
    import super_sac_amd as ssa
    rows = {}
    rs = np.random.RandomState(0)

    def agent_of(kind):
        if kind == "dmc":   # BASELINE config 3: 9 x 84 x 84 frames, BigPixelEncoder, deterministic actor 50 -> 1024 -> 1024 -> 6
            conv = ssa.nets.BigPixelEncoder((9, 84, 84), 50)
            ag = ssa.Agent(act_space_size=6, encoder=ssa.nets.PixelEncoder(conv), actor_network_cls=ssa.nets.ContinuousDeterministicActor,
                           critic_network_cls=ssa.nets.ContinuousCritic, discrete=False, ensemble_size=1, num_critics=2,
                           hidden_size=1024, auto_rescale_targets=False)
            obs = lambda n: {"obs": rs.randint(0, 256, (n, 9, 84, 84) if n > 1 else (9, 84, 84)).astype(np.uint8)}
        elif kind == "atari":
            conv = ssa.nets.SmallPixelEncoder((4, 84, 84), 128)
            ag = ssa.Agent(act_space_size=4, encoder=ssa.nets.PixelEncoder(conv), actor_network_cls=ssa.nets.DiscreteActor,
                           critic_network_cls=ssa.nets.DiscreteCritic, discrete=True, ensemble_size=1, num_critics=2,
                           hidden_size=256, auto_rescale_targets=False)
            obs = lambda n: {"obs": rs.randint(0, 256, (n, 4, 84, 84) if n > 1 else (4, 84, 84)).astype(np.uint8)}
        else:
            E, N, ucb = (5, 2, 5.0) if kind == "sunrise" else (1, 10, 0.0)
            ag = ssa.Agent(act_space_size=6, encoder=ssa.nets.IdentityEncoder(17),
                           actor_network_cls=ssa.nets.ContinuousStochasticActor, critic_network_cls=ssa.nets.ContinuousCritic,
                           discrete=False, ensemble_size=E, num_critics=N, ucb_bonus=ucb, hidden_size=HID,
                           auto_rescale_targets=False, log_std_low=-5.0, log_std_high=2.0)
            obs = lambda n: {"obs": rs.standard_normal((n, 17) if n > 1 else (17,)).astype(np.float32)}
        ag.to(device)
        ag.eval()
        return ag, obs

    from super_sac_amd import acting

    def timed(fn, o, n, calls=300):
        for _ in range(30):
            fn(o, num_envs=n)
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            fn(o, num_envs=n)
            ts.append(time.perf_counter() - t0)
        return ts

    for kind in ("redq_M", "sunrise", "atari", "dmc"):
        ag, obs = agent_of(kind)
        for n in (1, 16):
            o = obs(n)
            for fn_name in ("sample_action", "forward"):
                ts = timed(getattr(ag, fn_name), o, n)
                rows[f"{kind}.{fn_name}.envs{n}"] = {"us_per_call_median": round(statistics.median(ts) * 1e6, 1),
                                                     "us_per_call_p90": round(sorted(ts)[int(0.9 * len(ts))] * 1e6, 1)}
        # the same agent through agent.py's general (eager) path -- what the one-call path replaces -- one row per agent
        acting.ENABLED = False
        try:
            ts = timed(ag.sample_action, obs(1), 1, calls=150)
        finally:
            acting.ENABLED = True
        rows[f"{kind}.sample_action.envs1"]["general_path_us_median"] = round(statistics.median(ts) * 1e6, 1)
        del ag
    return {"what": "Agent.sample_action / Agent.forward, numpy observation in -> numpy action out, wall clock per call, 300 calls "
                    "after 30 warm-up calls: ONE C call per step (super_sac_amd/acting.py: recorded launch list, observation over the "
                    "BAR, action through pinned memory); general_path_us_median = agent.py's eager path on the same agent",
            "rows": rows}


def actor_update_rows(device):
    """the online actor update (learning.py:344-421) alone -- what a UTD-1 configuration pays per environment step beside
    its critic update -- at the headline shape and at the Humanoid shape (BASELINE config 5), recorded launch list, on the
    fixed batch buffers of a critic update"""
    rows = {}
    for label, obs, act, ncrit in (("M_obs17_act6_N10", 17, 6, 10), ("humanoid_obs376_act17_N16", 376, 17, 16)):
        st, _, _ = build_engine(device, ncrit, None, batch=512, obs=obs, act=act, ncrit=ncrit)
        for _ in range(5):
            dicts = st()
        actor = st.objects["actor_step"]
        for _ in range(30):
            actor(dicts)
        t = statistics.median(timed_repeats(lambda: actor(dicts), 500, 3, None, device)) / 500
        rows[label] = {"batch": 512, "us_per_actor_update": round(t * 1e6, 2)}
        del st, actor, dicts
    return rows


def utd1_rows(device):
    """A UTD-1 environment step on the device path (BASELINE configs 1, 3, 4, 5 run UTD 1): Agent.sample_action on a numpy
    observation, critic_update (+ Polyak / 2), online_actor_update, alpha_update -- strictly in sequence, as main.super_sac runs
    them (the action of step t + 1 needs the actor of step t); wall clock per step"""
    import numpy as np
    import torch
    rows = {}
    for label, obs, act, ncrit, batch in (("sac_obs3_act1_N2_B256", 3, 1, 2, 256), ("M_obs17_act6_N10_B512", 17, 6, 10, 512)):
        st, _, ssa = build_engine(device, ncrit, None, batch=batch, obs=obs, act=act, ncrit=ncrit)
        ob = st.objects
        agent, la, buf, actor = ob["agent"], ob["log_alpha"], ob["buffer"], ob["actor_step"]
        lopt = torch.optim.Adam([la], lr=1e-4, betas=(0.5, 0.999))
        aug = ssa.augmentations.AugmentationSequence([ssa.augmentations.IdentityAug(batch)])
        o = {"obs": np.random.RandomState(0).standard_normal(obs).astype(np.float32)}

        def one():
            agent.sample_action(o)
            dicts = st()
            actor(dicts)
            ssa.learning.alpha_update(buffer=buf, agent=agent, optimizers=[lopt], batch_size=batch, log_alphas=[la],
                                      augmenter=aug, aug_mix=0.0, target_entropy=-float(act), premade_replay_dicts=dicts,
                                      discrete=False)
        for _ in range(40):
            one()
        t = statistics.median(timed_repeats(one, 300, 3, None, device)) / 300
        rows[label] = {"us_per_env_step": round(t * 1e6, 1), "env_steps_per_s": round(1 / t, 0)}
        del st, actor, one
    return {"what": "UTD-1 environment step: sample_action (numpy in / out) + critic_update (+ Polyak / 2) + online_actor_update + "
                    "alpha_update, in sequence", "rows": rows}


def n16_rows(device):
    """critic_update (+ Polyak / 2) with ALL 16 critics on one GPU, recorded launch list as the headline: the metric shape and
    the Humanoid shape of BASELINE config 5"""
    rows = {}
    for label, obs, act in (("M_obs17_act6", 17, 6), ("humanoid_obs376_act17", 376, 17)):
        st, _, _ = build_engine(device, 16, None, batch=512, obs=obs, act=act, ncrit=16)
        for _ in range(60):
            st()
        t = statistics.median(timed_repeats(st, 1000, 3, None, device)) / 1000
        rows[label] = {"num_critics": 16, "batch": 512, "us_per_critic_update": round(t * 1e6, 2),
                       "critic_updates_per_s": round(1 / t, 1)}
        del st
    return rows


def fp32_sweep(ssa, device):
    """the fp32 path's two launches at B = 4096 / 16384 / 65536 (N = 10, obs 17 / act 6, n = 2): the chained
    launch (`ssac_chain_update`, producer / consumer form: the headline's kernel, its rows from a batch buffer instead of the
    replay gather) and the merged weight-gradient launch (`ssac_mlp_wgrad_all_scaled`, gradient-store epilogue: what the
    engine issues above 4096 rows, where the loss fold's LDS table no longer fits), HIP events around back-to-back launches
    on the launch stream, against the fp32 matrix peak.  FLOPs as in `roofline`: chained = critics' forward + backward-data +
    the subset's target critics + the actor once; weight gradient = 2 B N (in H + H H + H)."""
    import ctypes as C
    import torch
    from super_sac_amd import engine
    from super_sac_amd._lib import check, lib
    S, A, N, H = 17, 6, 10, HID
    IN = S + A
    out = {"peak_TFLOPs": FP32_MFMA_PEAK_TFLOPS, "rows": {}}
    torch.manual_seed(2)
    aa, ca, ta = engine.MlpArena(1, S, H, 2 * A, device), engine.MlpArena(N, IN, H, 1, device), engine.MlpArena(N, IN, H, 1, device)
    for ar in (aa, ca, ta):
        ar.params.copy_(torch.randn_like(ar.params) * 0.05)
    ids = torch.tensor([N - 1, 0], dtype=torch.int32, device=device)
    st = engine.stream()
    for B in (4096, 16384, 65536):
        x1, xc = torch.randn(B, IN, device=device), torch.randn(B, IN, device=device)
        eps, lp = torch.randn(B, A, device=device), torch.zeros(B, device=device)
        h1 = torch.empty(N, B, H, device=device); h2 = torch.empty_like(h1); dz2 = torch.empty_like(h1); dz1 = torch.empty_like(h1)
        q, qt = torch.empty(N, B, 1, device=device), torch.empty(2, B, 1, device=device)
        ho = torch.zeros(B * A, dtype=torch.int64, device=device)
        grads = torch.zeros_like(ca.params)
        ss = torch.zeros(N * engine.wgrad_tiles_total(ca), device=device)
        dq = torch.randn(N, B, 1, device=device)

        def chain_launch():
            check(lib.ssac_chain_update(
                C.byref(aa.desc()), x1.data_ptr(), IN, B, eps.data_ptr(), -5.0, 2.0, x1.data_ptr(), IN, S, lp.data_ptr(), 0,
                C.byref(ta.desc()), ids.data_ptr(), 2, qt.data_ptr(), C.byref(ca.desc()), xc.data_ptr(), IN, h1.data_ptr(),
                h2.data_ptr(), q.data_ptr(), dz2.data_ptr(), dz1.data_ptr(), 0, 0, 0, ho.data_ptr(), 1, 0, st))

        def wgrad_launch():
            engine.weight_grads(ca, xc, IN, 0, h1, h2, dq, dz2, dz1, B, grads=grads, sumsq=ss, rowscale=dq)
        f_chain = (2.0 * B * N * (IN * H + H * H + H) + 2.0 * B * N * (H + H * H) + 2.0 * B * NSUB * (IN * H + H * H + H) +
                   2.0 * B * (S * H + H * H + H * 2 * A))
        f_wg = 2.0 * B * N * (IN * H + H * H + H)
        row = {}
        for tag, fn, fl in (("chain", chain_launch, f_chain), ("wgrad", wgrad_launch, f_wg)):
            for _ in range(3):
                fn()
            reps = 20 if B <= 16384 else 6
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            row[tag] = {"us_per_launch": round(us, 2), "TFLOPs": round(fl / us / 1e6, 1),
                        "frac": round(fl / us / 1e6 / FP32_MFMA_PEAK_TFLOPS, 3)}
        out["rows"][f"B{B}"] = row
        del x1, xc, eps, lp, h1, h2, dz2, dz1, q, qt, ho, grads, ss, dq
        torch.cuda.empty_cache()
    return out


def bf16_rows(ssa, device):
    """the ensemble-Q kernel alone in bf16 (forward of N=10 critics): algorithmic bytes = bf16 weights +
    fp32 batch in + fp32 Q out; reported against the HBM roof at the config-2 batch and at large batches"""
    import ctypes as C
    import torch
    from super_sac_amd import engine
    from super_sac_amd._lib import check, lib
    N, IN = NCRIT, OBS + ACT
    ar = engine.MlpArena(N, IN, HID, 1, device)
    torch.manual_seed(1)
    ar.params.copy_(torch.randn_like(ar.params) * 0.05)
    ar.enable_bf16()
    rows = {}
    for B in (256, 4096, 16384, 65536, 262144):
        x = torch.randn(B, IN, device=device)
        y = torch.empty(N, B, 1, device=device)

        def run():
            check(lib.ssac_bf16_mlp3_fwd(C.byref(ar.desc()), ar.shadow.data_ptr(), 0, N, x.data_ptr(), IN, B,
                                         y.data_ptr(), engine.stream()))
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50 if B <= 4096 else 10
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        nbytes = N * (IN * HID + HID * HID + HID) * 2 + N * (2 * HID + 1) * 4 + B * IN * 4 + N * B * 4
        flops = 2.0 * B * N * (IN * HID + HID * HID + HID)
        rows[f"B{B}"] = {"us_per_launch": round(us, 2), "algorithmic_bytes": nbytes,
                         "hbm_GBs": round(nbytes / us / 1e3, 1), "hbm_frac": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4),
                         "TFLOPs": round(flops / us / 1e6, 1), "mfma_frac": round(flops / us / 1e6 / 2500.0, 3)}
    return {"ensemble_q_kernel_bf16": {"kernel": "ssac_bf16_mlp3_fwd: forward of N=10 critics (23->256->256->1), bf16 shadow "
                                                 "weights (B 256 / 4096: one workgroup per 32-row tile; B 65536: the register-"
                                                 "chained persistent kernel, weights in LDS, no activation leaves the registers); "
                                                 "HIP events around back-to-back launches; bf16 MFMA peak 2.5 PFLOP/s",
                                       "hbm_peak_GBs": HBM_PEAK_GBS, "rows": rows}}


if __name__ == "__main__":
    main()
